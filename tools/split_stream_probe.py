"""Design probe (not product code): the adapter fwd+bwd of one micro-batch as ONE chain of launches against the same tokens split
into two half-batches whose chains run on two streams of one hipGraph.  Question: do the per-launch fixed costs (boundaries, ramps,
latency-bound rank-space kernels) of one chain hide behind the streaming kernels of the other?

    python tools/split_stream_probe.py [--batch 4]
"""
import argparse
import os
import sys
import time
from ctypes import c_void_p

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--parts", type=int, default=2)
    a = ap.parse_args()
    from moka_amd import _lib
    from moka_amd.parallel import FlatGradBucket
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    def mk(batch):
        args = argparse.Namespace(variant="avt", batch=batch, seq=2048, rank=16, model="7b", layers=32, distinct=4, dropout=0.05,
                                  no_group=False)
        return bench.build_workload(args, dev, lib, lambda n, ends: FlatGradBucket(n, ends, dev, n_buckets=8)), args

    def graph_of(wls, streams_n):
        side = torch.cuda.Stream(device=dev)
        extra = [torch.cuda.Stream(device=dev) for _ in range(streams_n - 1)]
        with torch.cuda.stream(side):
            for wl in wls:
                sp = c_void_p(side.cuda_stream)
                bench.run_forward(lib, wl, sp)
                bench.run_backward(lib, wl, sp, 32)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            cur = torch.cuda.current_stream()
            if streams_n == 1:
                for wl in wls:
                    sp = c_void_p(cur.cuda_stream)
                    bench.run_forward(lib, wl, sp)
                    bench.run_backward(lib, wl, sp, 32)
            else:
                for st in extra:
                    st.wait_stream(cur)
                for wl, st in zip(wls, [cur] + extra):
                    with torch.cuda.stream(st):
                        sp = c_void_p(st.cuda_stream)
                        bench.run_forward(lib, wl, sp)
                        bench.run_backward(lib, wl, sp, 32)
                for st in extra:
                    cur.wait_stream(st)
        torch.cuda.synchronize()
        return g

    def timed(g, n):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    full, _ = mk(a.batch)
    t_full = timed(graph_of([full], 1), a.steps)
    del full
    torch.cuda.empty_cache()
    hs = [mk(a.batch // a.parts)[0] for _ in range(a.parts)]
    t_serial = timed(graph_of(hs, 1), a.steps)
    t_split = timed(graph_of(hs, a.parts), a.steps)
    print(f"one chain, batch {a.batch}: {t_full:.2f} ms   {a.parts} part-batch chains back to back: {t_serial:.2f} ms   "
          f"on {a.parts} streams: {t_split:.2f} ms   (adapter fwd+bwd only, no optimizer)")


if __name__ == "__main__":
    main()
