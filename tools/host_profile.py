#!/usr/bin/env python3
"""Host-side cost of one training step: enqueue time (before the final synchronize) vs wall time,
plus a cProfile of the enqueue path.  python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import config, optim
from svit_amd.model import build_model

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = config.ssv2_cfg(num_frames=16, crop=224, num_gpus=1)
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0)
model.train()
opt = optim.construct_optimizer(model, cfg)
x = torch.randn(8, 3, 16, 224, 224, device="cuda")
y = torch.randint(0, 174, (8,), device="cuda")


def step():
    logits, _ = model([x], {})
    loss = torch.nn.functional.cross_entropy(logits, y)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
