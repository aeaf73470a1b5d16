// Column-stripe read probe (design probe, not product code): a [T x C] bf16 matrix read the way the
// weight-gradient kernels read their input -- every block owns a stripe of W bytes of every row of a token run.
// Question: how much HBM bandwidth does a narrow stripe (128 B = one 64-column MFMA tile) cost against wider ones?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// block: THREADS threads; W bytes per row -> W/16 threads per row, THREADS/(W/16) rows per step, U steps in flight
template <int W, int U, int THREADS>
__global__ __launch_bounds__(THREADS) void stripe_read(const char* __restrict__ x, int T, int rowbytes, int rows_per_block, unsigned* out) {
    constexpr int TPR = W / 16, RPS = THREADS / TPR;
    const int stripe = blockIdx.x, run = blockIdx.y;
    const int r0 = run * rows_per_block, r1 = min(T, r0 + rows_per_block);
    const int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
    const char* p = x + (size_t)stripe * W + tc * 16;
    unsigned acc = 0;
    for (int r = r0 + tr; r < r1; r += U * RPS) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const u32x4*)(p + (size_t)min(r + u * RPS, r1 - 1) * rowbytes);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int W, int U, int THREADS>
static void run(const char* xa, const char* xb, int T, int C, int blocks_target, unsigned* out, const char* tag) {
    const int rowbytes = C * 2, ns = rowbytes / W;
    int nb = blocks_target / ns; if (nb < 1) nb = 1;
    const int rpb = (T + nb - 1) / nb;
    nb = (T + rpb - 1) / rpb;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f; const int reps = 10;
    for (int it = 0; it < reps + 2; ++it) {
        const char* x = (it & 1) ? xb : xa;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stripe_read<W, U, THREADS>), dim3(ns, nb), dim3(THREADS), 0, 0, x, T, rowbytes, rpb, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (double)T * rowbytes;
    printf("%-28s T=%6d C=%5d W=%4d U=%d thr=%4d grid=%4dx%-3d  avg %7.1f us  best %7.1f us  %6.2f TB/s (best)\n", tag, T, C, W, U, THREADS, ns, nb,
           sum / reps * 1e3, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t cap = (size_t)1 << 31;
    char *xa, *xb; unsigned* out;
    CK(hipMalloc(&xa, cap)); CK(hipMalloc(&xb, cap)); CK(hipMalloc(&out, 64));
    CK(hipMemset(xa, 1, cap)); CK(hipMemset(xb, 2, cap));
    for (int T : {8192, 65536}) {
        for (int C : {5120, 4096}) {
            for (int bt : {256, 512, 1024, 2048}) {
                printf("-- blocks target %d\n", bt);
                run<128, 4, 256>(xa, xb, T, C, bt, out, "stripe 128 B");
                run<128, 8, 256>(xa, xb, T, C, bt, out, "stripe 128 B");
                run<256, 4, 256>(xa, xb, T, C, bt, out, "stripe 256 B");
                run<256, 8, 256>(xa, xb, T, C, bt, out, "stripe 256 B");
                run<512, 4, 256>(xa, xb, T, C, bt, out, "stripe 512 B");
                run<512, 8, 256>(xa, xb, T, C, bt, out, "stripe 512 B");
                run<1024, 4, 256>(xa, xb, T, C, bt, out, "stripe 1024 B");
                run<1024, 8, 256>(xa, xb, T, C, bt, out, "stripe 1024 B");
                run<2048, 8, 256>(xa, xb, T, C, bt, out, "stripe 2048 B");
            }
        }
    }
    return 0;
}
