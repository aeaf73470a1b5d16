import os as _os
_os.environ.setdefault("MOKA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "moka_amd", "libmoka_hip_diag.so"))   # moka_tune: diagnostics build only
#!/usr/bin/env python3
"""Time every C entry point at the bench shape (T = 8192, Llama-2-7B widths, r = 16, M = 3) under
different diagnostic tuning settings (moka_tune).  Run on the GPU box; prints a table.
env: B, S, R, DROP, WIDTHS="5120x5120,5120x13824,13824x5120", ONLY=<entry point substring>, SWEEP=0 (defaults only), MOKA_HIP_LIB=<library to time>."""
import math
import os
import sys
from ctypes import byref, c_float, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from moka_amd import _lib  # noqa: E402
from moka_amd.routing import MokaRouting  # noqa: E402
import bench  # noqa: E402


DROP = 0.0


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, S, r, M = int(os.environ.get("B", 4)), int(os.environ.get("S", 2048)), int(os.environ.get("R", 16)), 3
    global DROP
    DROP = float(os.environ.get("DROP", 0.0))
    T = B * S
    tok, q = bench.synthetic_layout(S)
    if os.environ.get("ALLTEXT"):
        tok = torch.zeros_like(tok)
    if os.environ.get("ALLIMG"):
        tok = torch.ones_like(tok)
    if os.environ.get("ALIGNED"):       # same proportions, every span boundary on a multiple of 32
        tok = torch.zeros_like(tok)
        tok[32:288] = 1
        tok[320:448] = 2
    masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)]
    masks.append(q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev))
    rt = MokaRouting.from_avt_masks(masks)
    bf, f32 = torch.bfloat16, torch.float32
    RP, Tp = _lib.rank_pad(r), _lib.tok_pad(T)
    NBUF = int(os.environ.get("NBUF", 6))      # rotate over distinct buffers so nothing stays in the 256 MiB Infinity Cache

    def shapes(d_in, d_out):
        xs = [torch.randn(T, d_in, device=dev, dtype=bf) for _ in range(NBUF)]
        ys = [torch.randn(T, d_out, device=dev, dtype=bf) for _ in range(NBUF)]
        dxs = [torch.randn(T, d_in, device=dev, dtype=bf) for _ in range(NBUF)]
        A = [torch.randn(r, d_in, device=dev, dtype=bf) * 0.01 for _ in range(M)]
        Bw = torch.randn(d_out, r, device=dev, dtype=bf) * 0.02
        part = torch.empty(24, T, RP, dtype=f32, device=dev)
        h = torch.empty(T, RP, dtype=f32, device=dev)
        hp_tok = torch.empty(Tp, 2 * RP, dtype=bf, device=dev)
        hp_kmj = torch.empty(2, RP, Tp, dtype=bf, device=dev)
        BwT = torch.empty(RP, d_out, dtype=bf, device=dev)
        AT = torch.empty(M, d_in, RP, dtype=bf, device=dev)
        dh_tok = torch.empty(Tp, 2 * RP, dtype=bf, device=dev)
        dh_kmj = torch.empty(M, 2, RP, Tp, dtype=bf, device=dev)
        dA = [torch.zeros(r, d_in, dtype=f32, device=dev) for _ in range(M)]
        dB = torch.zeros(d_out, r, dtype=f32, device=dev)
        return dict(xs=xs, ys=ys, dxs=dxs, A=A, Bw=Bw, part=part, h=h, hp_tok=hp_tok, hp_kmj=hp_kmj, BwT=BwT, AT=AT,
                    dh_tok=dh_tok, dh_kmj=dh_kmj, dA=dA, dB=dB, d_in=d_in, d_out=d_out)

    def calls(w):
        d_in, d_out = w["d_in"], w["d_out"]
        Ap = (c_void_p * M)(*[a.data_ptr() for a in w["A"]])
        dAp = (c_void_p * M)(*[a.data_ptr() for a in w["dA"]])
        so = (c_float * M)(1.0, 1.0, 1.0)
        tm = rt.tok_mod.data_ptr()
        w["_keep"] = (Ap, dAp, so)
        c = 1 / math.sqrt(r)
        sp = lambda: c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
        return {
            "down_fwd": lambda i: lib.moka_down_fwd(w["xs"][i % NBUF].data_ptr(), Ap, tm, w["part"].data_ptr(), T, d_in, r, M, 1.0, DROP, 1234, 0, None, sp()),
            "cross_fwd": lambda i: lib.moka_cross_fwd(w["part"].data_ptr(), _lib.ksplit(T, d_in, r), byref(rt.struct), so, w["Bw"].data_ptr(), d_out, Ap, d_in,
                                                      w["h"].data_ptr(), None, w["hp_tok"].data_ptr(), w["hp_kmj"].data_ptr(), w["BwT"].data_ptr(), w["AT"].data_ptr(), r, 1.0, c, sp()),
            "up_fwd": lambda i: lib.moka_up_fwd(w["hp_tok"].data_ptr(), w["Bw"].data_ptr(), tm, w["ys"][i % NBUF].data_ptr(), T, r, d_out, 0, sp()),
            "up_bwd(g only)": lambda i: lib.moka_up_bwd(w["ys"][i % NBUF].data_ptr(), w["hp_kmj"].data_ptr(), w["BwT"].data_ptr(), tm, so, w["part"].data_ptr(), None, T, r, d_out, M, 0, None, sp()),
            "up_bwd(dB only)": lambda i: lib.moka_up_bwd(w["ys"][i % NBUF].data_ptr(), w["hp_kmj"].data_ptr(), None, tm, so, None, w["dB"].data_ptr(), T, r, d_out, M, 0, None, sp()),
            "up_bwd(g+dB)": lambda i: lib.moka_up_bwd(w["ys"][i % NBUF].data_ptr(), w["hp_kmj"].data_ptr(), w["BwT"].data_ptr(), tm, so, w["part"].data_ptr(), w["dB"].data_ptr(), T, r, d_out, M, 0, None, sp()),
            "cross_bwd": lambda i: lib.moka_cross_bwd(w["part"].data_ptr(), _lib.ksplit_bwd(T, d_out, r), w["h"].data_ptr(), byref(rt.struct), 1.0, None,
                                                      w["dh_tok"].data_ptr(), w["dh_kmj"].data_ptr(), rt.cross_ws(r).data_ptr(), r, 1.0, c, sp()),
            "down_bwd(dA only)": lambda i: lib.moka_down_bwd(w["dh_tok"].data_ptr(), w["dh_kmj"].data_ptr(), w["xs"][i % NBUF].data_ptr(), w["AT"].data_ptr(), tm, dAp, None, T, d_in, r, M, DROP, 1234, 0, None, sp()),
            "down_bwd(dx only)": lambda i: lib.moka_down_bwd(w["dh_tok"].data_ptr(), w["dh_kmj"].data_ptr(), w["xs"][i % NBUF].data_ptr(), w["AT"].data_ptr(), tm, None, w["dxs"][i % NBUF].data_ptr(), T, d_in, r, M, DROP, 1234, 0, None, sp()),
        }

    def timeit(fn, iters=24):
        for i in range(4):
            assert fn(i) == 0, lib.moka_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    E = 2
    only = os.environ.get("ONLY")
    widths = [tuple(int(v) for v in w.split("x")) for w in os.environ.get("WIDTHS", "4096x4096,4096x11008,11008x4096").split(",")]
    for (d_in, d_out) in widths:
        w = shapes(d_in, d_out)
        cs = calls(w)
        # make the rank-space inputs valid once
        for n in ("down_fwd", "cross_fwd", "up_fwd", "up_bwd(g+dB)", "cross_bwd"):
            assert cs[n](0) == 0, lib.moka_last_error()
        torch.cuda.synchronize()
        algo = {"down_fwd": E * T * d_in, "up_fwd": 2 * E * T * d_out, "up_bwd(g only)": E * T * d_out, "up_bwd(dB only)": E * T * d_out, "up_bwd(g+dB)": E * T * d_out,
                "down_bwd(dA only)": E * T * d_in, "down_bwd(dx only)": 2 * E * T * d_in, "cross_fwd": 0, "cross_bwd": 0}
        sweeps = {
            "down_fwd": [("xa_ng", v) for v in (2, 4, 8)],
            "up_fwd": [("expand_depth", 3)] + [("expand_bpc", v) for v in (2, 4, 8, 12, 16)],
            "down_bwd(dx only)": [("expand_depth", 2)] + [("expand_bpc", v) for v in (2, 4, 8)],
            "up_bwd(g+dB)": [("gy_ng", v) for v in (4, 8, 16)],
            "up_bwd(g only)": [("gy_ng", v) for v in (4, 8, 16)],
            "down_bwd(dA only)": [("wgrad_bpc", v) for v in (1, 2, 3, 4, 6, 8)],
        }
        print(f"\n=== {d_in} -> {d_out}  (T={T}) ===")
        for name, fn in cs.items():
            if only and only not in name:
                continue
            base = timeit(fn)
            gb = algo[name] / (base * 1e-6) / 1e9 if algo[name] else 0
            print(f"{name:20s} default            {base:8.1f} us  {gb:7.0f} GB/s algorithmic")
            for key, val in ([] if os.environ.get("SWEEP") == "0" else sweeps.get(name, [])):
                lib.moka_tune(key.encode(), val)
                t = timeit(fn)
                lib.moka_tune(key.encode(), 0)
                gb = algo[name] / (t * 1e-6) / 1e9 if algo[name] else 0
                print(f"{'':20s} {key:12s}={val:<4d} {t:8.1f} us  {gb:7.0f} GB/s")
        del w, cs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
