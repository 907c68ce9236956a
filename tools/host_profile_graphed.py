#!/usr/bin/env python3
"""Host-side enqueue time vs wall time of the GRAPHED training step (bench.py's path): if enqueue ~ wall
the host is the bottleneck and the device idles at step boundaries.  python tools/host_profile_graphed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import config, optim
from svit_amd.model import build_model
from svit_amd.graph import GraphedTrainStep
cfg = config.ssv2_cfg(num_frames=16, crop=224, num_gpus=1)
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0)
model.train()
opt = optim.construct_optimizer(model, cfg)
x = torch.randn(8, 3, 16, 224, 224, device="cuda")
y = torch.randint(0, 174, (8,), device="cuda")
ce = lambda p, e, l: torch.nn.functional.cross_entropy(p, l)
g = GraphedTrainStep(model, ce, [x], y)
for _ in range(5):
    g([x], y); opt.step()
torch.cuda.synchronize()
parts = {"graph": 0.0, "opt": 0.0}
N = 30
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); g([x], y); b = time.perf_counter(); opt.step(); c = time.perf_counter()
    parts["graph"] += b - a; parts["opt"] += c - b
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graphed: enqueue %.3f ms/step (graph replay call %.3f, optimizer %.3f), wall %.3f ms/step, segments %d"
      % ((t1 - t0) / N * 1e3, parts["graph"] / N * 1e3, parts["opt"] / N * 1e3, (t2 - t0) / N * 1e3, len(g.segments)))
def wall(fn, n=N):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
sx, sy = g.x, g.labels
print("variants, wall ms/step: as is %.3f | static inputs (no copies) %.3f | graph only %.3f | graph only, static inputs %.3f | "
      "optimizer only %.3f" % (wall(lambda: (g([x], y), opt.step())), wall(lambda: (g([sx], sy), opt.step())),
                               wall(lambda: g([x], y)), wall(lambda: g([sx], sy)), wall(lambda: opt.step())))
# with a device sync after every step (no run-ahead): what one step costs end to end
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    g([x], y); opt.step(); torch.cuda.synchronize()
print("synchronised every step: %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
