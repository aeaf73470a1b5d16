// Per-launch cost of dependent kernels on one stream: empty kernels of different grid / block shapes, and a kernel with one
// dependent global round trip, timed over 2000 back-to-back launches.   hipcc --offload-arch=gfx950 -O3 launchfloor.hip -o launchfloor
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_empty() {}
__global__ void k_rt1(const int* p, int* out) { if (p[threadIdx.x & 31] == 12345) out[0] = 1; }
__global__ void k_rt2(const int* p, int* out) { int a = p[threadIdx.x & 31]; if (p[(a & 31) + 32] == 12345) out[0] = 1; }

template <typename F> static float timeit(F f, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / n;
}

int main() {
    int *p, *out; hipMalloc(&p, 4096); hipMalloc(&out, 64); hipMemset(p, 0, 4096);
    const int grids[] = {1, 256, 1024, 4096, 16384};
    const int blocks[] = {64, 256, 512};
    for (int b : blocks)
        for (int g : grids)
            printf("empty   grid %6d x %3d threads: %6.2f us per launch\n", g, b, timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(b), 0, 0); }, 2000));
    for (int g : {256, 1024, 4096}) {
        printf("1 round trip, grid %5d x 256: %6.2f us\n", g, timeit([&] { hipLaunchKernelGGL(k_rt1, dim3(g), dim3(256), 0, 0, p, out); }, 2000));
        printf("2 round trips, grid %5d x 256: %6.2f us\n", g, timeit([&] { hipLaunchKernelGGL(k_rt2, dim3(g), dim3(256), 0, 0, p, out); }, 2000));
    }
    // with 64 KB of dynamic LDS (limits residency like the cross / xa kernels)
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int g : {256, 1024, 4096})
        printf("empty + 64 KB LDS, grid %5d x 256: %6.2f us\n", g, timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 65536, 0); }, 2000));
    return 0;
}
