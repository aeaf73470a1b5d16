#!/usr/bin/env python3
"""Randomised stage checks of the adapter path against the fp64 oracle (not part of the suite: a stress run for new kernels).

    python tools/stress_cross.py [n_cases] [seed]

Random variant / batch / sequence length (including lengths that are no multiple of 4, 16 or 32) / rank / widths / span layout
(missing modalities, padding, long and short question spans) through tests/test_gpu_parity.py::_stage_check."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cases as C                                   # noqa: E402
from tests.test_gpu_parity import _stage_check                    # noqa: E402


def main(n=40, seed=0):
    rnd = random.Random(seed)
    bad = 0
    for k in range(n):
        variant = rnd.choice(["avt", "vt"])
        r = rnd.choice([4, 8, 16, 16, 16, 24, 32, 48, 64])
        B = rnd.choice([1, 2, 3])
        d_in = 32 * rnd.randint(2, 40)
        d_out = 32 * rnd.randint(2, 40)
        lays, S = [], None
        for b in range(B):
            lay = []
            if rnd.random() < 0.4:
                lay.append(("p", rnd.randint(1, 40)))
            lay.append(("t", rnd.randint(1, 30)))
            if rnd.random() < 0.85:
                lay.append(("v", rnd.randint(1, 150)))
            lay.append(("t", rnd.randint(1, 20)))
            if variant == "avt" and rnd.random() < 0.7:
                lay.append(("a", rnd.randint(1, 90)))
            nq = rnd.choice([1, 2, 7, 16, 33, 64, 65, 100, 130, 200])
            lay.append(("q", nq))
            lay.append(("t", rnd.randint(1, 60)))
            lays.append(lay)
        S = max(sum(nn for _, nn in lay) for lay in lays)
        for lay in lays:                                           # pad to the common length (AVT pads left, VT right)
            short = S - sum(nn for _, nn in lay)
            if short:
                if variant == "avt":
                    lay.insert(0, ("p", short))
                else:
                    lay.append(("p", short))
        name = f"stress_{seed}_{k}"
        C._CASES[name] = dict(variant=variant, B=B, S=S, d_in=d_in, d_out=d_out, r=r, alpha=16.0,
                              w=1.0 if variant == "avt" else 0.05, layouts=lays, seed=1000 + 17 * k + seed, big=True)
        try:
            _stage_check(C.make_case_data(name))
        except Exception as exc:                                   # keep going: report every failing configuration
            bad += 1
            print(f"FAIL {name}: {variant} B={B} S={S} r={r} {d_in}->{d_out} layouts={lays}: {exc!r}"[:600])
    print(f"stress: {n - bad} of {n} random cases passed")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
