// What does a chain of dependent, nearly empty launches cost with the conv kernels' footprint (200 workgroups x 512 threads, 160 KB of
// dynamic LDS = one workgroup per CU)?  The floor of a small image's 352 launches (tools/ubench: hipcc --offload-arch=gfx950 -O3 launch_chain.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void touch(float* p, int n)
{
    extern __shared__ char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    __syncthreads();
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + smem[0];
}
int main()
{
    float* d;
    hipMalloc(&d, 1 << 22);
    hipMemset(d, 0, 1 << 22);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int lds : {0, 160 * 1024})
        for (int grid : {18, 200, 256})
        {
            for (int i = 0; i < 352; i++) hipLaunchKernelGGL(touch, dim3(grid), dim3(512), lds, st, d, grid * 512);
            hipStreamSynchronize(st);
            hipEventRecord(a, st);
            for (int r = 0; r < 10; r++)
                for (int i = 0; i < 352; i++) hipLaunchKernelGGL(touch, dim3(grid), dim3(512), lds, st, d, grid * 512);
            hipEventRecord(b, st);
            hipStreamSynchronize(st);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("grid %3d x 512 threads, %3d KB LDS: %.2f us per dependent launch (352 launches = %.2f ms)\n", grid, lds >> 10, ms * 1e3 / 3520, ms / 10);
        }
    return 0;
}
