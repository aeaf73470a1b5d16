"""The schedules bench.py captures (part-batch chains on forked streams, deferred dA_m on the hub, per-bucket graphs) compute what
the same launches compute live on one stream: `bench.py --verify-graph` (graph replay vs live, from the same activation state)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layers", "5", "--seq", "512", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-traffic", "--no-optimizer", "--verify-graph", *extra],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [
    (),                                                   # the default: 2 chains, hub-shaped one-graph step, dA_m one launch per layer
    ("--defer-da", "side"),
    ("--defer-da", "layer"),
    ("--defer-da", "bucket"),
    ("--defer-da", "off"),
    ("--chains", "4"),
    ("--chains", "1"),                                    # round 4's shape with the chain-first capture order
    ("--chains", "1", "--graph-topology", "hub"),
    ("--graph", "bwd"),                                   # what N > 1 replays: forward graph + one graph per gradient bucket, 2 chains
    ("--graph", "bwd", "--chains", "1"),
    ("--rank", "64", "--batch", "2", "--chains", "2"),    # dB off the chain too (a pass of its own at this rank)
    ("--variant", "vt"),
], ids=lambda e: " ".join(e) or "default")
def test_captured_schedule_equals_live_launches(extra):
    out = _bench(*extra)
    chk = out["graph_check"]
    assert chk is not None and chk["tensors_compared"] > 0
    assert chk["activations_bit_identical"], (extra, chk)
    assert chk["grad_nonzero_frac"] > 0.9 and chk["grad_max_abs"] > 0
    assert chk["grad_max_abs_diff_over_max"] <= 2e-5, (extra, chk)        # fp32 atomics in a different order
    if "--chains" not in extra and "--rank" not in extra:
        assert out["chains"] == 2


def test_default_line_says_how_it_ran():
    out = _bench()
    assert out["graph"] == "all" and out["graph_topology"] == "hub" and out["chains"] == 2 and out["defer_dA"] in ("layer", "side", "unit")
    assert "attach" in out["schedule"] and "--e2e" in out["schedule"]      # whose schedule the headline is (and which flag measures the autograd path)


def test_capture_failure_falls_back_to_live_launches_of_every_chain():
    """hipGraph capture is an optimisation: when it fails the step is launched live -- with two chains, both part-batches back to back on the
    one stream (it used to run only the first chain's launches; then it exited).  MOKA_BENCH_FAIL_CAPTURE=1 is the test hook."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    outs = {}
    for name, extra_env, extra in (("graph", {}, ()), ("fallback", {"MOKA_BENCH_FAIL_CAPTURE": "1"}, ()), ("off", {}, ("--graph", "off", "--chains", "2"))):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layers", "4", "--seq", "512", "--steps", "3", "--warmup", "1",
                              "--no-cpu-baseline", "--no-traffic", *extra], capture_output=True, text=True, timeout=900, env={**env, **extra_env}, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        outs[name] = json.loads(lines[0])
        if name == "fallback":
            assert "launching live" in res.stderr and "2 chains back to back" in res.stderr
    assert outs["graph"]["chains"] == outs["fallback"]["chains"] == outs["off"]["chains"] == 2
    # the same launches, the same bytes: the per-pass entry-point table (taken from live passes in every mode) covers both chains
    for k in ("moka_up_fwd", "moka_down_bwd"):
        a, b = outs["graph"]["entry_point_ms_per_pass"][k], outs["fallback"]["entry_point_ms_per_pass"][k]
        assert a > 0 and b > 0 and 0.5 < a / b < 2.0
    assert outs["fallback"]["value"] > 0 and outs["off"]["value"] > 0 and outs["graph"]["value"] >= 0.8 * outs["fallback"]["value"]


@pytest.mark.parametrize("extra", [(), ("--chains", "1"), ("--defer-da", "layer")], ids=lambda e: " ".join(e) or "default")
def test_default_schedule_with_optimizer_equals_live_steps(extra):
    """the headline schedule WITH its optimizer -- two chains sharing gradient accumulators, AdamW slices and weight-shadow
    rewrites on the hub behind per-branch events -- was never compared against live launches.  `bench.py --verify-graph` (without
    --no-optimizer): 3 steps as the captured graph and 3 steps live from identical master / moments / activations."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layers", "5", "--seq", "512", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-traffic", "--verify-graph", *extra], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    chk = out["graph_check"]
    assert chk["steps"] == 3 and chk["grad_left_zero"], chk
    assert chk["master_rel_diff"] <= 1e-5, chk                     # (fp32 atomics in a different order: last bits of the gradients)
    assert chk["work_max_abs_diff"] <= 1e-3 and chk["shadow_max_abs_diff"] <= 1e-3, chk       # bf16 copies: an ulp where the master's last bits differ
    assert chk["activations_max_rel_diff"] <= 1e-3, chk
