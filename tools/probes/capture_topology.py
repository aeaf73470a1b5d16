"""Which multi-stream topologies does hipStreamEndCapture (ROCm 7.2, torch 2.10) survive?  Each variant in its own process."""
import subprocess
import sys

import torch

VARIANTS = ["two_chains", "chain_side_origin", "chain_side_branch", "chain_side_branch_prejoined", "two_chains_sides", "two_chains_sides_opt",
            "branch_side_no_event_back", "branch_side_event_back_once", "hub"]


def run(v):
    dev = torch.device("cuda:0")
    x = [torch.zeros(1 << 20, device=dev) for _ in range(8)]
    origin = torch.cuda.Stream(device=dev)
    branch = torch.cuda.Stream(device=dev)
    s0, s1, opt = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    L = 4

    def work(st, i):
        with torch.cuda.stream(st):
            x[i].add_(1.0)

    def chain(main, side, i, sink=None, back=True):
        done = {}
        for l in range(L - 1, -1, -1):
            if side is not None and back and (l + 2) in done:
                main.wait_event(done.pop(l + 2))
            work(main, i)
            if side is not None:
                side.wait_stream(main)
                work(side, i + 1)
                ev = torch.cuda.Event()
                ev.record(side)
                done[l] = ev
                if sink is not None:
                    sink[l] = ev
        if side is not None:
            main.wait_stream(side)

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=origin):
        cur = torch.cuda.current_stream()
        if v == "two_chains":
            branch.wait_stream(cur)
            chain(cur, None, 0)
            chain(branch, None, 2)
            cur.wait_stream(branch)
        elif v == "chain_side_origin":
            chain(cur, s0, 0)
        elif v == "chain_side_branch":
            branch.wait_stream(cur)
            chain(branch, s1, 2)
            cur.wait_stream(branch)
        elif v == "chain_side_branch_prejoined":
            branch.wait_stream(cur)
            s1.wait_stream(cur)
            chain(branch, s1, 2)
            cur.wait_stream(branch)
            cur.wait_stream(s1)
        elif v == "branch_side_no_event_back":
            branch.wait_stream(cur)
            s1.wait_stream(cur)
            chain(branch, s1, 2, back=False)
            cur.wait_stream(branch)
            cur.wait_stream(s1)
        elif v == "branch_side_event_back_once":
            branch.wait_stream(cur)
            s1.wait_stream(cur)
            work(branch, 2)
            s1.wait_stream(branch)
            work(s1, 3)
            branch.wait_stream(s1)
            work(branch, 2)
            cur.wait_stream(branch)
            cur.wait_stream(s1)
        elif v == "hub":
            # what bench.py --chains does: the chains on two forks, everything off the chains on the ORIGIN stream (every edge touches the origin)
            b2 = torch.cuda.Stream(device=dev)
            for st in (branch, b2):
                st.wait_stream(cur)
            da, db = {}, {}
            for l in range(L - 1, -1, -1):
                for main, i, done in ((branch, 0, da), (b2, 2, db)):
                    if (l + 2) in done:
                        main.wait_event(done.pop(l + 2))
                    work(main, i)
                    cur.wait_stream(main)
                    work(cur, i + 1)
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    done[l] = ev
                if l % 2 == 0:
                    work(cur, 5)
            for st in (branch, b2):
                cur.wait_stream(st)
        elif v in ("two_chains_sides", "two_chains_sides_opt"):
            for st in (branch, s1, opt):
                st.wait_stream(cur)
            a, b = {}, {}
            chain(cur, s0, 0, a)
            chain(branch, s1, 2, b)
            cur.wait_stream(branch)
            if v.endswith("opt"):
                for l in (2, 0):
                    opt.wait_event(a[l])
                    opt.wait_event(b[l])
                    work(opt, 5)
            for st in (branch, s1, opt):
                cur.wait_stream(st)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(v, "ok", [float(t[0]) for t in x[:6]], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in VARIANTS:
            rc = subprocess.call([sys.executable, "-X", "faulthandler", __file__, v], stderr=subprocess.DEVNULL)
            if rc:
                print(v, "FAILED rc", rc, flush=True)
