import os as _os
_os.environ.setdefault("MOKA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "moka_amd", "libmoka_hip_diag.so"))   # moka_tune: diagnostics build only
#!/usr/bin/env python3
"""Time the grouped C entry points (q/k/v: G = 3 at 4096 -> 4096; gate/up: G = 2 at 4096 -> 11008) at the
bench shape, each half separately.  Run on the GPU box; prints a table.  Env: B (sequences), DROP."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from moka_amd import _lib  # noqa: E402
from moka_amd import functional as F  # noqa: E402
from moka_amd.routing import MokaRouting  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, S, r, M = int(os.environ.get("B", 4)), 2048, 16, 3
    DROP = float(os.environ.get("DROP", 0.0))
    T = B * S
    tok, q = bench.synthetic_layout(S)
    masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)]
    masks.append(q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev))
    rt = MokaRouting.from_avt_masks(masks)
    bf, f32 = torch.bfloat16, torch.float32
    NBUF = 6
    c = 1 / math.sqrt(r)
    only = os.environ.get("ONLY")

    def timeit(fn, iters=24):
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    E = 2
    for d_in, d_outs in [(4096, (4096, 4096, 4096)), (4096, (11008, 11008)), (4096, (4096,))]:
        G = len(d_outs)
        xs = [torch.randn(T, d_in, device=dev, dtype=bf) for _ in range(NBUF)]
        dxs = [torch.randn(T, d_in, device=dev, dtype=bf) for _ in range(NBUF)]
        ys = [[torch.randn(T, do, device=dev, dtype=bf) for do in d_outs] for _ in range(max(2, NBUF // G))]
        As = [[torch.randn(r, d_in, device=dev, dtype=bf) * 0.01 for _ in range(M)] for _ in range(G)]
        Bws = [torch.randn(do, r, device=dev, dtype=bf) * 0.02 for do in d_outs]
        dA = [[torch.zeros(r, d_in, dtype=f32, device=dev) for _ in range(M)] for _ in range(G)]
        dB = [torch.zeros(do, r, dtype=f32, device=dev) for do in d_outs]
        seeds = [11 + g for g in range(G)]
        parts = F.down_fwd_group(xs[0], As, rt, r, 1.0, DROP, seeds)
        sts = F.cross_fwd_group(parts, rt, r, [1.0] * M, 1.0, c, Bws, As)
        g_parts = F.up_bwd_group(ys[0], [s.hp_kmj for s in sts], [s.BwT for s in sts], rt, r, [1.0] * M, dB)
        bsts = F.cross_bwd_group(g_parts, [s.h for s in sts], rt, r, 1.0, 1.0, c)
        torch.cuda.synchronize()
        ny = len(ys)
        calls = {
            "down_fwd": (lambda i: F.down_fwd_group(xs[i % NBUF], As, rt, r, 1.0, DROP, seeds), E * T * d_in * G),
            "cross_fwd": (lambda i: F.cross_fwd_group(parts, rt, r, [1.0] * M, 1.0, c, Bws, As), 0),
            "up_fwd": (lambda i: F.up_fwd_group_(ys[i % ny], [s.hp_tok for s in sts], Bws, rt, r), 2 * E * T * sum(d_outs)),
            "up_bwd(g only)": (lambda i: F.up_bwd_group(ys[i % ny], [s.hp_kmj for s in sts], [s.BwT for s in sts], rt, r, [1.0] * M, None), E * T * sum(d_outs)),
            "up_bwd(g+dB)": (lambda i: F.up_bwd_group(ys[i % ny], [s.hp_kmj for s in sts], [s.BwT for s in sts], rt, r, [1.0] * M, dB), E * T * sum(d_outs)),
            "cross_bwd": (lambda i: F.cross_bwd_group(g_parts, [s.h for s in sts], rt, r, 1.0, 1.0, c), 0),
            "down_bwd(dA only)": (lambda i: F.down_bwd_group_(bsts, xs[i % NBUF], None, rt, r, dA, None, DROP, seeds), E * T * d_in * G),
            "down_bwd(dx only)": (lambda i: F.down_bwd_group_(bsts, xs[i % NBUF], [s.AT for s in sts], rt, r, None, dxs[i % NBUF], DROP, seeds), 2 * E * T * d_in * G),
        }
        sweeps = {"up_fwd": [("expand_bpc", v) for v in (2, 4, 8)], "down_bwd(dx only)": [("expand_depth", 2)] + [("expand_bpc", v) for v in (2, 4, 8)],
                  "down_bwd(dA only)": [("wgrad_bpc", v) for v in (1, 2)], "down_fwd": [("xa_ng", v) for v in (2, 4, 8)],
                  "up_bwd(g+dB)": [("gy_ng", v) for v in (4, 8, 16)], "up_bwd(g only)": [("gy_ng", v) for v in (4, 8, 16)]}
        print(f"\n=== G={G}: {d_in} -> {'/'.join(map(str, d_outs))}  (T={T}, dropout {DROP}) ===")
        for name, (fn, nb) in calls.items():
            if only and only not in name:
                continue
            t = timeit(fn)
            print(f"{name:20s} default            {t:8.1f} us  {nb / (t * 1e-6) / 1e9:7.0f} GB/s algorithmic (per-projection bytes)")
            for key, val in sweeps.get(name, []):
                lib.moka_tune(key.encode(), val)
                t = timeit(fn)
                lib.moka_tune(key.encode(), 0)
                print(f"{'':20s} {key:12s}={val:<4d} {t:8.1f} us  {nb / (t * 1e-6) / 1e9:7.0f} GB/s")
        del xs, dxs, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
