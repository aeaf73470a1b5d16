"""bench.py helpers that carry the measurement contract -- no GPU needed."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def test_algorithmic_bytes_match_survey_totals():
    fwd, bwd = bench.algorithmic_bytes_per_token(bench.MODELS["7b"], 16, 32)
    # 7B r=16: fwd 7.731 + bwd 9.573 = 17.305 MB/token
    assert abs(fwd / 1e6 - 7.731) < 2e-3 and abs(bwd / 1e6 - 9.573) < 2e-3
    assert abs((fwd + bwd) / 1e6 - 17.305) < 3e-3
    # 13B r=64: 27.21 MB/token; 70B r=16 (grouped-query k / v: 8192 -> 1024): 90.20 MB/token
    f13, b13 = bench.algorithmic_bytes_per_token(bench.MODELS["13b"], 64, 40)
    assert abs((f13 + b13) / 1e6 - 27.21) < 2e-2
    f70, b70 = bench.algorithmic_bytes_per_token(bench.MODELS["70b"], 16, 80)
    assert abs((f70 + b70) / 1e6 - 90.20) < 5e-2


def test_synthetic_layout_is_the_survey_layout():
    from oracle import cases as C
    for S in (2048, 4096, 512):
        tok, q = bench.synthetic_layout(S)
        tok2, q2 = C.build_layout(C.synthetic_sequence_layout(S), S)
        assert torch.equal(tok, tok2) and torch.equal(q, q2)
    tok, q = bench.synthetic_layout(2048)
    assert int((tok == 1).sum()) == 256 and int((tok == 2).sum()) == 128 and int(q.sum()) == 64
    assert bool((tok[q] == 0).all())                      # question tokens are text tokens, contiguous
    idx = torch.where(q)[0]
    assert int(idx[-1] - idx[0]) == 63


def test_projection_units_follow_the_decoder():
    # q/k/v share the hidden states, gate/up the post-attention norm output; o and down stand alone
    srcs = [p[3] for p in bench.PROJS]
    assert srcs == ["hid", "hid", "hid", "attn", "hid2", "hid2", "act"]
    assert [p[0] for p in bench.PROJS] == ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]


def test_usable_cpus_is_positive():
    assert bench.usable_cpus() >= 1


def test_the_pmc_traffic_summary_of_this_round_is_committed():
    """roofline.traffic of the default bench line is read from this file (bench.PMC_TRAFFIC_FILE, written by tools/pmc_traffic.sh on the
    GPU box); without it the line carries traffic = null.  The summary is stamped with the sha256 of the kernel source it was measured on:
    a summary of other kernels is refused (traffic = null + a warning), never silently reused."""
    import json
    assert os.path.exists(bench.PMC_TRAFFIC_FILE), bench.PMC_TRAFFIC_FILE
    d = json.load(open(bench.PMC_TRAFFIC_FILE))
    # measured on the launches the default schedule makes: part-batches of 2 sequences (4096 tokens)
    assert d["tokens"] == 4096 and d["traffic_bytes_per_layer"] > 0.65e9          # >= the algorithmic 696.3 MB of the four y launches
    assert len(d["kernel_source_sha256"]) == 64 and len(d["library_sha256"]) == 64
    t, src = bench.pmc_traffic_per_launch(8192, 4)
    if d["kernel_source_sha256"] == bench.kernel_source_sha256():
        assert src == "profiles/" + os.path.basename(bench.PMC_TRAFFIC_FILE) and 3.4e8 < t < 4.6e8
    else:
        assert t is None and src.startswith("stale: ")


def test_a_traffic_summary_of_other_kernels_is_refused(monkeypatch, tmp_path, capsys):
    import json
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps({"tokens": 4096, "traffic_bytes_per_layer": 8e8, "kernel_source_sha256": "0" * 64, "library_sha256": "0" * 64}))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(p))
    t, src = bench.pmc_traffic_per_launch(8192, 4)
    assert t is None and src.startswith("stale: ")
    assert "other kernels" in capsys.readouterr().err
    good = {"tokens": 4096, "traffic_bytes_per_layer": 8e8, "kernel_source_sha256": bench.kernel_source_sha256(), "library_sha256": "0" * 64}
    p.write_text(json.dumps(good))
    t, src = bench.pmc_traffic_per_launch(8192, 4)
    assert t == 8e8 * 2 / 4


def test_a_missing_traffic_summary_does_not_kill_the_line(monkeypatch, capsys):
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", os.path.join(ROOT, "profiles", "no_such_file.json"))
    t, src = bench.pmc_traffic_per_launch(8192, 4)
    assert t is None and src.startswith("missing: ")
    assert "is missing" in capsys.readouterr().err
