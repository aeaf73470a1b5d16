// Down-projection candidate "xs": the x stream goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1 KB row segment per wave
// instruction, row-contiguous -- the access shape that streamed best in the timelines), NS tiles of 16 tokens x 512 columns deep, with no
// VGPRs tied up by data in flight; the waves read their MFMA fragments out of the ring.  (Design probe; included by passlab.hip after
// the library source.)
#pragma once

// (the kernel itself became the library's moka_xs_kernel; this header keeps the launcher that sweeps ring depth / tiles per workgroup)
template <int G, int NS>
static void lab_launch_xs(const XaArgs& a, int tpb) {
    const int ncb = (a.C + 511) / 512, ntile = a.T / 16, ntb = (ntile + tpb - 1) / tpb;
    const size_t lds = (size_t)NS * 16 * 1040 + (size_t)2 * 8 * G * 256 * 4 + (size_t)tpb * 16;
    ensure_lds((const void*)moka_xs_kernel<G, NS>, lds);
    hipLaunchKernelGGL((moka_xs_kernel<G, NS>), dim3(ncb, ntb), dim3(512), lds, 0, a, tpb);
}
static void lab_down_fwd_xs(const void* x, const void* const* A, const uint8_t* tok_mod, float* const* part, int T, int d_in, int r, int M, int G,
                            float s_in, float dropout_p, const unsigned long long* seeds, int ns, int tpb) {
    XaArgs xa; memset(&xa, 0, sizeof(xa));
    xa.x = (const unsigned char*)x; xa.tok_mod = tok_mod; xa.T = T; xa.C = d_in; xa.r = r; xa.M = M;
    float inv_keep = 1.f;
    for (int g = 0; g < G; ++g) {
        make_drop("lab", dropout_p, seeds ? seeds[g] : 0ull, &xa.drop[g]); inv_keep = xa.drop[g].inv_keep;
        xa.part[g] = part[g];
        for (int m = 0; m < M; ++m) xa.A[g][m] = (const unsigned char*)A[g * M + m];
    }
    for (int m = 0; m < M; ++m) xa.s_mod[m] = s_in * inv_keep;
    if (G == 1) { if (ns == 2) lab_launch_xs<1, 2>(xa, tpb); else if (ns == 3) lab_launch_xs<1, 3>(xa, tpb); else lab_launch_xs<1, 4>(xa, tpb); }
    else if (G == 2) { if (ns == 2) lab_launch_xs<2, 2>(xa, tpb); else lab_launch_xs<2, 3>(xa, tpb); }
    else { if (ns == 2) lab_launch_xs<3, 2>(xa, tpb); else lab_launch_xs<3, 3>(xa, tpb); }
}
