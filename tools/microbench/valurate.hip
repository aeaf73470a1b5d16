// VALU issue-rate probe (design probe, not product code): cycles per wave instruction and SIMD for the integer ops a counter-based
// dropout mask can be built from.  8 independent chains per lane, 8 waves per SIMD, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void rate(unsigned* out, unsigned k, int iters) {
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 2654435761u + j * 40503u + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) v[j] = v[j] * k;                                                   // v_mul_lo_u32
            else if (OP == 1) v[j] ^= v[j] >> 15;                                           // shift + xor (2 instructions)
            else if (OP == 2) v[j] = ((v[j] & 0xffffffu) * (k & 0xffffffu));                     // v_mul_u32_u24
            else if (OP == 3) { union { unsigned u; u16x2 p; } a, b; a.u = v[j]; b.u = k; a.p = a.p * b.p; v[j] = a.u; }   // v_pk_mul_lo_u16
            else if (OP == 4) { union { unsigned u; s16x2 p; } a, b; a.u = v[j]; b.u = k; a.p = __builtin_elementwise_sub_sat(b.p, a.p); v[j] = a.u; }   // v_pk_sub_i16 clamp
            else if (OP == 5) { union { unsigned u; s16x2 p; } a; a.u = v[j]; a.p = a.p >> 15; v[j] = a.u + k; }   // v_pk_ashrrev_i16 (+ add)
            else if (OP == 6) v[j] = __builtin_amdgcn_perm(v[j], k, 0x00010203u + v[j]);    // v_perm_b32 (+ add)
            else if (OP == 7) v[j] = __builtin_amdgcn_alignbit(v[j], v[j], 13) + k;         // rotate + add (2)
            else if (OP == 8) v[j] = ((v[j] & 0xffffffu) * (k & 0xffffffu)) + (v[j] >> 9);       // mad_u32_u24-able
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
    if (acc == 0x12345u) out[0] = acc;
}

template <int OP>
static void run(const char* name, int instr_per_op, unsigned* out) {
    const int iters = 2048, blocks = 256 * 4;            // 4 blocks of 8 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate<OP>), dim3(blocks), dim3(512), 0, 0, out, 0x9E3779B1u, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<OP>), dim3(blocks), dim3(512), 0, 0, out, 0x9E3779B1u, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_ops_per_simd = (double)iters * 8 * 8;        // 8 chains x 8 waves per SIMD
    printf("%-44s %8.1f us   %6.2f ns per wave-op and SIMD  (= %5.2f cycles at 2.4 GHz; %d instruction%s per op)\n", name, ms * 1e3,
           ms * 1e6 / wave_ops_per_simd, ms * 1e6 / wave_ops_per_simd * 2.4, instr_per_op, instr_per_op > 1 ? "s" : "");
}

int main() {
    unsigned* out; CK(hipMalloc(&out, 64));
    run<0>("v_mul_lo_u32", 1, out);
    run<1>("v_lshrrev + v_xor", 2, out);
    run<2>("v_mul_u32_u24", 1, out);
    run<3>("v_pk_mul_lo_u16", 1, out);
    run<4>("v_pk_sub_i16 clamp", 1, out);
    run<5>("v_pk_ashrrev_i16 + v_add", 2, out);
    run<6>("v_perm_b32 + v_add", 2, out);
    run<7>("v_alignbit + v_add", 2, out);
    run<8>("v_mul_u32_u24 + v_lshrrev + v_add", 3, out);
    return 0;
}
