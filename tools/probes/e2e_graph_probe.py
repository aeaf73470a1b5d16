#!/usr/bin/env python3
"""The trainer path of ONE configuration, live or as a captured GraphedTrainStep, for a kernel trace:
    rocprofv3 --kernel-trace --stats -d /tmp/p -o r -- python tools/probes/e2e_graph_probe.py {live|graph} [layers] [defer_dA 0|1] [chains]
(round 6: the captured whole-stack step ran SLOWER than the live one; which kernels, or which gaps?)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

from bench import synthetic_layout  # noqa: E402
from moka_amd.decoder import LlamaDims, MokaLlamaStack  # noqa: E402
from moka_amd.parallel import attach  # noqa: E402
from moka_amd.peft_hyper import Linear  # noqa: E402
from moka_amd.routing import MokaRouting  # noqa: E402
from moka_amd.schedule import GraphedTrainStep  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "live"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
defer = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
chains = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B, S, r = 4, 2048, 16
dev = torch.device("cuda:0")
bf = torch.bfloat16
dims = LlamaDims()
tok, q = synthetic_layout(S)
masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)] + [q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev)]
batch = {"h": torch.randn(B, S, dims.hidden, device=dev, dtype=bf), "gout": torch.randn(B, S, dims.hidden, device=dev, dtype=bf),
         "m_t": masks[0], "m_v": masks[1], "m_a": masks[2], "m_q": masks[3]}


def make(d_in, d_out):
    m = Linear(d_in, d_out, r=(r, r, r), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.05, loramethod="train", bias=False)
    torch.nn.init.normal_(m.weight, std=0.02)
    torch.nn.init.normal_(m.lora_B0.weight, std=0.02)
    return m


old = torch.get_default_dtype()
torch.set_default_dtype(bf)
with torch.device(dev):
    st = MokaLlamaStack(dims, L, make)
torch.set_default_dtype(old)
st.train()
for n, p_ in st.named_parameters():
    p_.requires_grad = "lora_" in n
dp = attach(st, n_buckets=min(8, L), lr=1e-4, defer_dA=defer)


def f(part):
    x = part["h"].detach().requires_grad_(True)
    out, _ = st(x, [part["m_t"], part["m_v"], part["m_a"], part["m_q"]])
    return (out.float() * part["gout"].float()).sum() / out.shape[0]


if mode == "graph":
    gs = GraphedTrainStep(dp, f, batch, chains=chains, routing_fn=lambda p_: MokaRouting.from_avt_masks([p_["m_t"], p_["m_v"], p_["m_a"], p_["m_q"]]))
    step = lambda: gs(None)          # noqa: E731
else:
    def step():
        f(batch).backward()
        dp.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
print("%s L=%d defer_dA=%s chains=%d: %.2f ms per step" % (mode, L, defer, chains, (time.perf_counter() - t0) * 1e3 / n))
