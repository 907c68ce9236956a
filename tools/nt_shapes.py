#!/usr/bin/env python3
"""Per-shape time of every svit_gemm_nt call in one eager training step (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import config, hip
from svit_amd.model import build_model

cfg = config.ssv2_cfg(num_frames=16, crop=224, num_gpus=1)
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0)
model.train()
x = torch.randn(8, 3, 16, 224, 224, device="cuda")
y = torch.randint(0, 174, (8,), device="cuda")


def step():
    logits, _ = model([x], {})
    torch.nn.functional.cross_entropy(logits, y).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
hip.start_trace()
step()
torch.cuda.synchronize()
tr = hip.stop_trace()
agg = {}
for name, e0, e1, meta in tr:
    if name != "svit_gemm_nt":
        continue
    key = tuple(meta[1:5])
    a = agg.setdefault(key, [0.0, 0])
    a[0] += e0.elapsed_time(e1) * 1e3
    a[1] += 1
tot = sum(a[0] for a in agg.values())
print("total %.1f us" % tot)
EPI = {0: "bf16", 1: "gelu", 2: "resid", 3: "f32", 4: "dgelu"}
for key, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    M, N, K, e = key
    fl = 2.0 * M * N * K * n
    print("M %6d N %5d K %5d %-5s calls %2d  %7.1f us  avg %6.1f  %5.0f TF  %4.1f%%" %
          (M, N, K, EPI[e], n, us, us / n, fl / us / 1e6, 100 * us / tot))
