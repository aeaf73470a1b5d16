// Load-shape microbenchmark (design probe, not product code).
// Question: can MFMA-fragment-shaped global loads (16 rows x 64 B per wave
// instruction) stream a [T x K] bf16 matrix from HBM as fast as lane-linear
// 16 B/lane loads?  Decides whether the MokA down-projection may feed x to the
// MFMA straight from VGPRs or has to stage it through LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// (1) lane-linear: wave reads 1 KiB contiguous per instruction.
__global__ __launch_bounds__(256) void read_linear(const u32x4* __restrict__ x, size_t n16, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// (2) fragment 16x64B: wave owns 16 rows; lane (i = l&15, g = l>>4) reads 16 B at
// row i, byte 16*g + 64*step.  K elements per row = Kdim (bf16).
template <int UNROLL>
__global__ __launch_bounds__(256) void read_frag16(const char* __restrict__ x, int T, int Kdim, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const size_t rowbytes = (size_t)Kdim * 2;
    unsigned acc = 0;
    for (int tile = wave; tile < T / 16; tile += nwaves) {
        const char* p = x + (size_t)(tile * 16 + i) * rowbytes + 16 * g;
        for (size_t s = 0; s < rowbytes; s += 64 * UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = *(const u32x4*)(p + s + 64 * u);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// (3) fragment + real MFMA: h[16x16] += x_frag * A_frag (A held in registers, one k-slice per wave
// would be unrealistic; here A frag is a constant to isolate the x stream + MFMA issue).
template <int UNROLL>
__global__ __launch_bounds__(256) void read_frag16_mfma(const char* __restrict__ x, int T, int Kdim, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const size_t rowbytes = (size_t)Kdim * 2;
    bf16x8 afrag;
#pragma unroll
    for (int j = 0; j < 8; ++j) afrag[j] = (short)(0x3f80 + lane + j);
    f32x4 acc = {0, 0, 0, 0};
    for (int tile = wave; tile < T / 16; tile += nwaves) {
        const char* p = x + (size_t)(tile * 16 + i) * rowbytes + 16 * g;
        for (size_t s = 0; s < rowbytes; s += 64 * UNROLL) {
            bf16x8 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = *(const bf16x8*)(p + s + 64 * u);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[u], afrag, acc, 0, 0, 0);
        }
    }
    if (acc[0] == 1234.5f) out[0] = acc[0] + acc[1] + acc[2] + acc[3];
}

// (4) read-modify-write linear (y += c), 16 B per lane.
__global__ __launch_bounds__(256) void rmw_linear(u32x4* __restrict__ y, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = y[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u].x += 1; v[u].y += 1; v[u].z += 1; v[u].w += 1; y[i + u * stride] = v[u]; }
    }
}

// (5) read-modify-write fragment shaped: 16 rows x 64 B per instruction.
template <int UNROLL>
__global__ __launch_bounds__(256) void rmw_frag16(char* __restrict__ y, int T, int Ndim) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const size_t rowbytes = (size_t)Ndim * 2;
    // work item = (16-row tile, 64*UNROLL-byte column chunk)
    const int chunks = rowbytes / (64 * UNROLL);
    const long nitems = (long)(T / 16) * chunks;
    for (long it = wave; it < nitems; it += nwaves) {
        const int tile = it / chunks, ch = it % chunks;
        char* p = y + (size_t)(tile * 16 + i) * rowbytes + 16 * g + (size_t)ch * 64 * UNROLL;
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = *(const u32x4*)(p + 64 * u);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { v[u].x += 1; v[u].y += 1; v[u].z += 1; v[u].w += 1; *(u32x4*)(p + 64 * u) = v[u]; }
    }
}

// (6) XCD-local L2 atomics probe: every workgroup adds 1.0f into buf[xcc_id*N + j] with
// workgroup-scope (L2-resident, no sc1) atomics.  Sum over XCD copies must equal #blocks.
__global__ __launch_bounds__(256) void xcd_atomic(float* buf, int N, int* xcc_hist) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[xcc], 1);
    float* dst = buf + (size_t)xcc * N;
    for (int j = threadIdx.x; j < N; j += blockDim.x)
        __hip_atomic_fetch_add(&dst[j], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void agent_atomic(float* buf, int N) {
    for (int j = threadIdx.x; j < N; j += blockDim.x)
        __hip_atomic_fetch_add(&buf[j], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class F>
static float time_ms(F f, int iters = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 32768, K = 4096;   // default 268 MB: past the 256 MiB Infinity Cache
    const bool quick = argc > 2;
    const size_t bytes = (size_t)T * K * 2;
    const int NB = 6; char* xb[NB]; for (int b = 0; b < NB; ++b) { CK(hipMalloc(&xb[b], bytes)); CK(hipMemset(xb[b], 1, bytes)); }
    int rot = 0; char* x = xb[0];
#define ROT (x = xb[(rot++) % NB])
    unsigned* out; CK(hipMalloc(&out, 64));
    float* fout = (float*)out;
    printf("T=%d K=%d bytes=%.1f MB\n", T, K, bytes / 1e6);
    for (int grid : {512, 1024, 2048, 4096}) {
        float ms = time_ms([&] { ROT; hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, 0, (const u32x4*)x, bytes / 16, out); });
        printf("read_linear        grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
    }
    for (int grid : {256, 512, 1024, 2048}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(read_frag16<8>, dim3(grid), dim3(256), 0, 0, x, T, K, out); });
        printf("read_frag16<8>     grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(read_frag16<16>, dim3(grid), dim3(256), 0, 0, x, T, K, out); });
        printf("read_frag16<16>    grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = time_ms([&] { ROT; hipLaunchKernelGGL(read_frag16_mfma<8>, dim3(grid), dim3(256), 0, 0, x, T, K, fout); });
        printf("read_frag16_mfma<8> grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(read_frag16_mfma<16>, dim3(grid), dim3(256), 0, 0, x, T, K, fout); });
        printf("read_frag16_mfma<16> grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
    }
    for (int grid : {1024, 2048, 4096}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(rmw_linear, dim3(grid), dim3(256), 0, 0, (u32x4*)x, bytes / 16); });
        printf("rmw_linear         grid=%5d  %.3f ms  %.0f GB/s (r+w)\n", grid, ms, 2 * bytes / ms / 1e6);
        ms = time_ms([&] { ROT; hipLaunchKernelGGL(rmw_frag16<4>, dim3(grid), dim3(256), 0, 0, x, T, K); });
        printf("rmw_frag16<4>      grid=%5d  %.3f ms  %.0f GB/s (r+w)\n", grid, ms, 2 * bytes / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(rmw_frag16<8>, dim3(grid), dim3(256), 0, 0, x, T, K); });
        printf("rmw_frag16<8>      grid=%5d  %.3f ms  %.0f GB/s (r+w)\n", grid, ms, 2 * bytes / ms / 1e6);
    }
    // XCD-local atomics probe
    if (!quick) {
        const int N = 65536, NB = 1024;
        float* buf; int* hist; CK(hipMalloc(&buf, 8 * N * 4)); CK(hipMalloc(&hist, 32));
        CK(hipMemset(buf, 0, 8 * N * 4)); CK(hipMemset(hist, 0, 32));
        hipLaunchKernelGGL(xcd_atomic, dim3(NB), dim3(256), 0, 0, buf, N, hist);
        CK(hipDeviceSynchronize());
        std::vector<float> h(8 * N); int hh[8];
        CK(hipMemcpy(h.data(), buf, 8 * N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int j = 0; j < N; ++j) { float s = 0; for (int c = 0; c < 8; ++c) s += h[c * N + j]; if (s != (float)NB) ++bad; }
        long badper = 0;
        for (int c = 0; c < 8; ++c) for (int j = 0; j < N; ++j) if (h[c * N + j] != (float)hh[c]) ++badper;
        printf("xcd_atomic: hist=%d %d %d %d %d %d %d %d  bad_sum=%ld bad_per_xcd=%ld\n", hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7], bad, badper);
        float ms = time_ms([&] { hipLaunchKernelGGL(xcd_atomic, dim3(NB), dim3(256), 0, 0, buf, N, hist); }, 10);
        printf("xcd_atomic   (wg scope, per-XCD buf) %d blocks x %d floats: %.3f ms  %.1f G atomics/s\n", NB, N, ms, (double)NB * N / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(agent_atomic, dim3(NB), dim3(256), 0, 0, buf, N); }, 10);
        printf("agent_atomic (agent scope, one buf)  %d blocks x %d floats: %.3f ms  %.1f G atomics/s\n", NB, N, ms, (double)NB * N / ms / 1e6);
    }
    return 0;
}
