// Streaming-ceiling microbenchmark (design probe, not product code).
// Round 2 re-do of profiles/r01_microbench_stream_ceiling_by_size.txt at working sets far above the
// 256 MiB Infinity Cache: 2 GiB buffers, two of them alternated, so that nothing a
// launch reads was left on die by the previous one.  Variants: lane-linear 16 B/lane reads with
// U loads in flight per lane, the MFMA-fragment shape (16 rows x 64 B per wave instruction), non-temporal
// loads, float4 copy (the guide's 6.29 TB/s figure), and read-modify-write in place.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_linear(const u32x4* __restrict__ x, size_t n16, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// block-contiguous: every block owns one contiguous span (n16 / grid), walks it with U x 4 KiB in flight.
template <int U>
__global__ __launch_bounds__(256) void read_blockspan(const u32x4* __restrict__ x, size_t n16, unsigned* out) {
    const size_t span = n16 / gridDim.x;
    const u32x4* p = x + span * blockIdx.x + threadIdx.x;
    unsigned acc = 0;
    for (size_t i = 0; i + (U - 1) * 256 < span; i += U * 256) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// fragment shape over a [T x K] bf16 matrix: block = 8 waves on [32*NG tokens x 512 columns] like the product kernels.
template <int U>
__global__ __launch_bounds__(512) void read_frag(const char* __restrict__ x, int T, int Kdim, int NG, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ks = Kdim / 512;
    const int kslice = blockIdx.x % ks, trun = blockIdx.x / ks;
    const size_t rowbytes = (size_t)Kdim * 2;
    const char* p = x + (size_t)(trun * NG * 32 + i) * rowbytes + (size_t)kslice * 1024 + wave * 128 + 16 * g;
    unsigned acc = 0;
    for (int grp = 0; grp < NG; grp += U / 4) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)   // 4 loads = one 32-token group of this wave's 64 columns
            v[u] = *(const u32x4*)(p + (size_t)((grp + u / 4) * 32 + (u & 1) * 16) * rowbytes + ((u >> 1) & 1) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void copy_linear(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = v[u];
    }
}

template <int U>
__global__ __launch_bounds__(256) void rmw_linear(u32x4* __restrict__ y, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = y[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u].x += 1; v[u].w += 1; y[i + u * stride] = v[u]; }
    }
}

// reads three quarters of a buffer, writes the sum to the fourth quarter of another one
__global__ __launch_bounds__(256) void mix31(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t nq) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + stride < nq; i += 2 * stride) {
        u32x4 a0 = x[i], b0 = x[nq + i], c0 = x[2 * nq + i], a1 = x[i + stride], b1 = x[nq + i + stride], c1 = x[2 * nq + i + stride];
        a0.x += b0.x + c0.x; a0.w += b0.w ^ c0.w; a1.x += b1.x + c1.x; a1.w += b1.w ^ c1.w;
        y[3 * nq + i] = a0; y[3 * nq + i + stride] = a1;
    }
}

static hipEvent_t e0, e1;
template <class F> static float timeit(F f, int reps) {
    f(0); f(1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) f(r & 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const size_t bytes = (size_t)2 << 30;           // 2 GiB per buffer
    const size_t n16 = bytes / 16;
    char* buf[2]; unsigned* out;
    CK(hipMalloc(&buf[0], bytes)); CK(hipMalloc(&buf[1], bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf[0], 1, bytes)); CK(hipMemset(buf[1], 2, bytes));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
    printf("working set 2 x %.2f GiB, alternated (>> 256 MiB Infinity Cache); GB/s = bytes moved / time\n", bytes / 1073741824.0);
#define RUN(name, bytes_moved, launch) do { float ms = timeit([&](int b) { launch; }, reps); \
        printf("%-34s grid=%6d  %8.3f ms  %7.0f GB/s\n", name, grid, ms, (bytes_moved) / ms * 1e-6); } while (0)
    const bool quick = getenv("CEIL_QUICK") != nullptr;
    for (int grid : {512, 1024, 2048, 4096, 8192, 16384}) {
        if (quick) break;
        RUN("read_linear U=4", bytes, (read_linear<4, false><<<grid, 256>>>((const u32x4*)buf[b], n16, out)));
        RUN("read_linear U=8", bytes, (read_linear<8, false><<<grid, 256>>>((const u32x4*)buf[b], n16, out)));
        RUN("read_linear U=16", bytes, (read_linear<16, false><<<grid, 256>>>((const u32x4*)buf[b], n16, out)));
        RUN("read_linear U=8 nt", bytes, (read_linear<8, true><<<grid, 256>>>((const u32x4*)buf[b], n16, out)));
        RUN("read_blockspan U=8", bytes, (read_blockspan<8><<<grid, 256>>>((const u32x4*)buf[b], n16, out)));
        RUN("copy_linear U=4 (r+w)", 2 * bytes, (copy_linear<4><<<grid, 256>>>((const u32x4*)buf[b], (u32x4*)buf[1 - b], n16)));
        RUN("rmw_linear U=4 (r+w)", 2 * bytes, (rmw_linear<4><<<grid, 256>>>((u32x4*)buf[b], n16)));
        RUN("rmw_linear U=8 (r+w)", 2 * bytes, (rmw_linear<8><<<grid, 256>>>((u32x4*)buf[b], n16)));
    }
    // fragment shape on [T x K] bf16 views of the same buffers
    for (int Kdim : {4096, 11008}) {
        if (quick) break;
        for (int NG : {4, 8, 16}) {
            const int T = (int)(bytes / ((size_t)Kdim * 2)) / (NG * 32) * (NG * 32);
            const int grid = T / (NG * 32) * (Kdim / 512);
            const size_t moved = (size_t)T * (Kdim / 512) * 1024;
            char nm[64];
            snprintf(nm, sizeof nm, "read_frag U=8  K=%d NG=%d", Kdim, NG);
            RUN(nm, moved, (read_frag<8><<<grid, 512>>>(buf[b], T, Kdim, NG, out)));
            if (NG >= 4) { snprintf(nm, sizeof nm, "read_frag U=16 K=%d NG=%d", Kdim, NG);
                RUN(nm, moved, (read_frag<16><<<grid, 512>>>(buf[b], T, Kdim, NG, out))); }
        }
    }
    // the product's sizes: one launch's worth (67 MB / 180 MB), 16 distinct regions cycled (cold each time)
    for (size_t sz : {(size_t)67108864, (size_t)180355072}) {
        for (int grid : {512, 1024, 2048}) {
            int k = 0;
            float ms = timeit([&](int b) { size_t off = ((size_t)(k++ % 8) * ((size_t)256 << 20));
                read_linear<8, false><<<grid, 256>>>((const u32x4*)(buf[b] + off), sz / 16, out); }, 16);
            printf("read_linear U=8 one launch of %.0f MB grid=%5d  %8.2f us  %7.0f GB/s (incl. launch boundary)\n", sz * 1e-6, grid, ms * 1e3, sz / ms * 1e-6);
        }
    }
    // the product's launch shape at the product's sizes: [T x K] bf16, block = 8 waves on [NG*32 tokens x 512 columns], cold regions
    for (int Kdim : {4096, 11008}) {
        const int T = 8192;
        const size_t sz = (size_t)T * Kdim * 2;
        for (int NG : {2, 4, 8, 16}) {
            const int grid = T / (NG * 32) * ((Kdim + 511) / 512);
            int k = 0;
            float ms = timeit([&](int b) { size_t off = ((size_t)(k++ % 8) * ((size_t)256 << 20));
                read_frag<8><<<grid, 512>>>(buf[b] + off, T, Kdim, NG, out); }, 16);
            printf("read_frag U=8 one launch T=8192 K=%5d NG=%2d grid=%5d  %8.2f us  %7.0f GB/s (incl. launch boundary)\n", Kdim, NG, grid, ms * 1e3, sz / ms * 1e-6);
            if (NG >= 4) {
                k = 0;
                ms = timeit([&](int b) { size_t off = ((size_t)(k++ % 8) * ((size_t)256 << 20));
                    read_frag<16><<<grid, 512>>>(buf[b] + off, T, Kdim, NG, out); }, 16);
                printf("read_frag U=16 one launch T=8192 K=%5d NG=%2d grid=%5d  %8.2f us  %7.0f GB/s\n", Kdim, NG, grid, ms * 1e3, sz / ms * 1e-6);
            }
        }
        for (int grid : {256, 512, 1024, 2048}) {
            int k = 0;
            float ms = timeit([&](int b) { size_t off = ((size_t)(k++ % 8) * ((size_t)256 << 20));
                read_blockspan<8><<<grid, 256>>>((const u32x4*)(buf[b] + off), sz / 16, out); }, 16);
            printf("read_blockspan U=8 one launch of %.0f MB grid=%5d  %8.2f us  %7.0f GB/s\n", sz * 1e-6, grid, ms * 1e3, sz / ms * 1e-6);
            k = 0;
            ms = timeit([&](int b) { size_t off = ((size_t)(k++ % 8) * ((size_t)256 << 20));
                read_linear<4, false><<<grid, 256>>>((const u32x4*)(buf[b] + off), sz / 16, out); }, 16);
            printf("read_linear U=4 one launch of %.0f MB grid=%5d  %8.2f us  %7.0f GB/s\n", sz * 1e-6, grid, ms * 1e3, sz / ms * 1e-6);
        }
    }
    // what a read-only kernel costs BEHIND a read-modify-write kernel (the product's sequence: y += .. of one unit, then the x read of
    // the next): dirty lines of the RMW kernel sit in the 256 MiB Infinity Cache and are written back while the read kernel allocates
    {
        const size_t rd = 67108864, rmw = (size_t)402653184;   // 67 MB read, 402 MB read-modify-write (q+k+v outputs)
        auto seq = [&](bool with_rmw, bool with_read) {
            int k = 0;
            return timeit([&](int b) {
                const size_t off = ((size_t)(k++ % 4) * ((size_t)512 << 20));
                if (with_rmw) rmw_linear<4><<<2048, 256>>>((u32x4*)(buf[b] + off), rmw / 16);
                if (with_read) read_frag<8><<<512, 512>>>(buf[1 - b] + off, 8192, 4096, 4, out);
            }, 12);
        };
        const float t_rmw = seq(true, false), t_rd = seq(false, true), t_both = seq(true, true);
        printf("sequence probe: rmw 402 MB alone %.2f us, read 67 MB alone %.2f us, rmw + read %.2f us -> the read behind the rmw costs %.2f us\n",
               t_rmw * 1e3, t_rd * 1e3, t_both * 1e3, (t_both - t_rmw) * 1e3);
        (void)rd;
    }
    // mixed traffic ceiling: 3 parts read, 1 part written (the step's mix: x / gy / base values read, y / dx written)
    for (int grid : {1024, 2048, 4096}) {
        float ms = timeit([&](int b) { mix31<<<grid, 256>>>((const u32x4*)buf[b], (u32x4*)buf[1 - b], n16 / 4); }, reps);
        printf("mix 3 reads : 1 write                grid=%6d  %8.3f ms  %7.0f GB/s (r+w)\n", grid, ms, (double)bytes / ms * 1e-6);
    }
    return 0;
}
