// Launch cost of a cooperative kernel with one grid-wide barrier against two plain launches
// (developer probe; build: hipcc --offload-arch=gfx950 -O2 tools/coop_probe.hip -o gpurun_out/coop_probe).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void plain_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.f;
}
__global__ void coop_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    cg::this_grid().sync();
    if (i < n) p[i] = v + 2.f;
}

int main() {
    const int blocks = 1250, threads = 256, n = blocks * threads;
    float* d;
    hipMalloc(&d, n * sizeof(float));
    hipMemset(d, 0, n * sizeof(float));
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, s);
        for (int k = 0; k < 200; ++k) {
            hipLaunchKernelGGL(plain_kernel, dim3(blocks), dim3(threads), 0, s, d, n);
            hipLaunchKernelGGL(plain_kernel, dim3(blocks), dim3(threads), 0, s, d, n);
        }
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("two plain launches: %.2f us per pair\n", ms * 1000.f / 200);
        hipEventRecord(e0, s);
        int nn = n;
        void* args[] = {&d, &nn};
        hipError_t err = hipSuccess;
        for (int k = 0; k < 200 && err == hipSuccess; ++k)
            err = hipLaunchCooperativeKernel((const void*)coop_kernel, dim3(blocks), dim3(threads), args, 0, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("one cooperative launch with a grid barrier: %.2f us (%s)\n", ms * 1000.f / 200, hipGetErrorString(err));
        // mixed: plain kernel between cooperative ones (the library's stream carries both kinds)
        hipEventRecord(e0, s);
        for (int k = 0; k < 200 && err == hipSuccess; ++k) {
            err = hipLaunchCooperativeKernel((const void*)coop_kernel, dim3(blocks), dim3(threads), args, 0, s);
            hipLaunchKernelGGL(plain_kernel, dim3(blocks), dim3(threads), 0, s, d, n);
        }
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("cooperative + plain alternating: %.2f us per pair (%s)\n", ms * 1000.f / 200, hipGetErrorString(err));
    }
    float h[4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("check %.0f\n", h[0]);
    return 0;
}
