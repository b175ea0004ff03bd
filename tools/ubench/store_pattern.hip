// store_pattern.hip -- micro-benchmark behind DESIGN.md 4.2 ("transpose-free epilogue"): what does the epilogue's 1-KiB store
// (one 16-channel plane row: 32 pixels x 32 B per wave-instruction, buffer_store / global_store b128) cost when the 16-byte
// pieces of the lanes are laid out
//   mode 0  dense per quarter-wave : lane l -> piece l                 (lanes 2p, 2p+1 = the two halves of pixel p: today, after
//                                                                        the transpose through LDS)
//   mode 1  interleaved            : lane l -> piece 2*(l & 31) + (l >> 5)  (lane = (pixel, half = l >> 5): what the MFMA accumulator
//                                                                        layout gives without a transpose -- every quarter-wave
//                                                                        writes 16 pieces at a 32-byte stride)
//   mode 2  two planes             : lanes 0-31 -> plane A, pixel l, half h ; lanes 32-63 -> plane B, pixel l, half h, two
//                                    instructions (h = 0, 1) per row        (hi selects the plane)
// every CU streams its own region; 4 or 8 waves per workgroup store, like the kernel's MFMA waves.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(char* dst, long long span, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char* base = dst + (long long)blockIdx.x * span;
    const long long rows = span / 2048; // a "row" = 2 planes x 1 KiB (the two output planes of a 32-cout conv)
    const long long plane_stride = span / 2;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    long long r = wave;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            char* rowp = base + (r % (rows)) * 1024;
            if (MODE == 0)
            {
                *reinterpret_cast<u32x4*>(rowp + lane * 16) = v;
                *reinterpret_cast<u32x4*>(rowp + plane_stride + lane * 16) = v;
            }
            else if (MODE == 1)
            {
                const int off = ((lane & 31) * 2 + (lane >> 5)) * 16;
                *reinterpret_cast<u32x4*>(rowp + off) = v;
                *reinterpret_cast<u32x4*>(rowp + plane_stride + off) = v;
            }
            else
            {
                char* p = rowp + (lane >> 5) * plane_stride + (lane & 31) * 32;
                *reinterpret_cast<u32x4*>(p) = v;
                *reinterpret_cast<u32x4*>(p + 16) = v;
            }
            v[2] += 1;
            r += nw;
        }
    }
}

template <int MODE>
static double run(int waves, long long span, char* dst)
{
    int dev = 0, ncu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int iters = 4000;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(waves * 64), 0, 0, dst, span, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(waves * 64), 0, 0, dst, span, iters);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = double(ncu) * waves * iters * 4 * 2048.0;
    return bytes / (ms * 1e-3) / 1e12;
}

int main()
{
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    for (long long span : {1ll << 20, 64ll << 20})
    { // 1 MiB per CU: the stores stay in the L2s / Infinity Cache (256 MB in all); 64 MiB per CU: 16 GB, they stream to HBM
        char* dst = nullptr;
        CK(hipMalloc(&dst, size_t(ncu) * span));
        CK(hipMemset(dst, 0, size_t(ncu) * span));
        for (int waves : {4, 8})
        {
            const double d = run<0>(waves, span, dst), i = run<1>(waves, span, dst), p = run<2>(waves, span, dst);
            printf("region %3lld MiB/CU, %d store waves/CU: dense-per-quarter %.2f TB/s | (pixel, half = hi) interleaved %.2f TB/s (x%.2f) | hi = plane, 2 x b128 per lane %.2f TB/s (x%.2f)\n",
                   span >> 20, waves, d, i, i / d, p, p / d);
        }
        CK(hipFree(dst));
    }
    return 0;
}
