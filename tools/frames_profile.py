#!/usr/bin/env python3
"""Kernel breakdown of the no-grad single-frame pass (B*T = 128 images, T' = 1): HIP events."""
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
xf = x.transpose(1, 2).flatten(0, 1).unsqueeze(2).contiguous()
with torch.no_grad():
    for _ in range(2):
        model([xf], {})
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        model([xf], {})
    e1.record(); torch.cuda.synchronize()
    print("frames pass: %.2f ms" % (e0.elapsed_time(e1) / 5))
    hip.start_trace()
    model([xf], {})
    torch.cuda.synchronize()
    tr = hip.stop_trace()
agg = {}
for name, a, b, meta in tr:
    if name.startswith("mark:"):
        continue
    d = agg.setdefault(name, [0.0, 0]); d[0] += a.elapsed_time(b); d[1] += 1
tot = sum(v[0] for v in agg.values())
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-26s %7.3f ms %4d calls %5.1f%%" % (k, ms, n, 100 * ms / tot))
print("traced kernel total %.2f ms" % tot)
