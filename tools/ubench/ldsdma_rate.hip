// ldsdma_rate.hip -- micro-benchmark behind DESIGN.md 4.1: how many bytes per second can ONE CU pull through its vector-memory
// path -- LDS-DMA (global_load_lds_dwordx4), plain global_load_dwordx4 into VGPRs, and buffer stores -- when every CU of the
// chip does the same, from an L2-resident source and from HBM?   hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode 0: LDS-DMA 16 B/lane   1: global_load_dwordx4 -> VGPR   2: global_store_dwordx4   3: LDS-DMA + stores 3:1
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, char* dst, long long span, int iters, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + (long long)blockIdx.x * span;
    char* lds = smem + wave * 8192;
    u32x4 acc = {0, 0, 0, 0};
    const long long pieces = span / 1024;
    long long p = wave;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int j = 0; j < 8; j++)
        {
            const char* g = base + p * 1024 + lane * 16;
            if (MODE == 0 || (MODE == 3 && j != 3 && j != 7)) __builtin_amdgcn_global_load_lds(GLBP(g), LDSP(lds + j * 1024), 16, 0, 0);
            else if (MODE == 1)
            {
                const u32x4 v = *reinterpret_cast<const u32x4*>(g);
                acc += v;
            }
            else *reinterpret_cast<u32x4*>(dst + (long long)blockIdx.x * span + p * 1024 + lane * 16) = acc;
            p += nw;
            if (p >= pieces) p -= pieces;
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 1 && acc[0] == 0x12345678u) sink[0] = acc[1];
    if (MODE != 1 && lane == 0 && iters < 0) sink[1] = reinterpret_cast<unsigned*>(smem)[0];
}

template <int MODE>
static void run(const char* name, int waves, long long span, const char* src, char* dst, unsigned* sink)
{
    int dev = 0, ncu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int iters = 2000;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(waves * 64), 65536, 0, src, dst, span, 50, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(waves * 64), 65536, 0, src, dst, span, iters, sink);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = double(ncu) * waves * iters * 8 * 1024;
    printf("%-34s waves/CU %d  span/CU %6lld KiB: %7.1f GB/s per CU, %6.2f TB/s chip (%.2f ms)\n", name, waves, span / 1024, bytes / ncu / ms / 1e6,
           bytes / ms / 1e9, ms);
}

int main()
{
    const long long big = 8ll << 20; // per CU: 2 GiB total, far beyond L2 + Infinity Cache
    char *src, *dst;
    unsigned* sink;
    CK(hipMalloc(&src, 256 * big));
    CK(hipMalloc(&dst, 256 * big));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, 256 * big));
    for (int waves : {1, 2, 4, 8})
    {
        run<0>("LDS-DMA dwordx4, L2-resident", waves, 64 << 10, src, dst, sink);
        run<0>("LDS-DMA dwordx4, HBM stream", waves, big, src, dst, sink);
        run<1>("global_load_dwordx4, L2-resident", waves, 64 << 10, src, dst, sink);
        run<1>("global_load_dwordx4, HBM stream", waves, big, src, dst, sink);
        run<2>("global_store_dwordx4, 64 KiB", waves, 64 << 10, src, dst, sink);
        run<2>("global_store_dwordx4, HBM stream", waves, big, src, dst, sink);
        run<3>("LDS-DMA 6 : store 2, L2-resident", waves, 64 << 10, src, dst, sink);
        run<0>("LDS-DMA dwordx4, MALL-resident", waves, 384 << 10, src, dst, sink); // 96 MiB in total: beyond L2, inside the Infinity Cache
        run<3>("LDS-DMA 6 : store 2, MALL-resident", waves, 256 << 10, src, dst, sink);
        run<3>("LDS-DMA 6 : store 2, HBM stream", waves, big, src, dst, sink);
    }
    return 0;
}
