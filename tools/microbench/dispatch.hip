// Workgroup dispatch rate probe (design probe, not product code): kernels whose blocks exit at once, or after one dependent global
// load, for grids of 256 .. 65536 blocks of 64 .. 1024 threads.  Question: what does LAUNCHING a block cost, i.e. how long is the
// ramp of a kernel that needs 2000-8000 waves resident before it streams at full rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(const int* p, int* out) { if (p == (const int*)0x1) out[0] = 1; }
__global__ void k_load(const int* p, int* out) { const int v = p[(blockIdx.x * 64) & 1023]; if (v == 0x12345678) out[0] = v; }
template <int R> __global__ void __launch_bounds__(256) k_regs(const int* p, int* out) {       // a block that holds R VGPRs per lane
    int v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = p[(threadIdx.x + i) & 1023];
    int s = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) s ^= v[i];
    if (s == 0x12345678) out[0] = s;
}

template <typename F> static float timeit(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    int *p, *out; CK(hipMalloc(&p, 4096)); CK(hipMalloc(&out, 64)); CK(hipMemset(p, 0, 4096));
    printf("%-10s %8s %8s %12s %14s\n", "kernel", "blocks", "threads", "us/launch", "blocks/us");
    for (int thr : {64, 256, 512, 1024})
        for (int nb : {256, 1024, 4096, 16384, 65536}) {
            const float t0 = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(nb), dim3(thr), 0, 0, p, out); });
            const float t1 = timeit([&] { hipLaunchKernelGGL(k_load, dim3(nb), dim3(thr), 0, 0, p, out); });
            printf("%-10s %8d %8d %12.1f %14.1f\n", "empty", nb, thr, t0, nb / t0);
            printf("%-10s %8d %8d %12.1f %14.1f\n", "one load", nb, thr, t1, nb / t1);
        }
    for (int nb : {512, 2048, 8192}) {
        const float a = timeit([&] { hipLaunchKernelGGL(k_regs<32>, dim3(nb), dim3(256), 0, 0, p, out); });
        const float b = timeit([&] { hipLaunchKernelGGL(k_regs<96>, dim3(nb), dim3(256), 0, 0, p, out); });
        printf("%-10s %8d %8d %12.1f %14.1f\n", "32 regs", nb, 256, a, nb / a);
        printf("%-10s %8d %8d %12.1f %14.1f\n", "96 regs", nb, 256, b, nb / b);
    }
    return 0;
}
