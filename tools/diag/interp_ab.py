"""Batched rel-pos table interpolation (one launch per forward pass) against one launch per block, alternating on one box:
the 3 x 312^2 test path (eager) -- python tools/diag/interp_ab.py   (GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svit_amd import config, evaluate
from svit_amd.model import build_model

cfg = config.ssv2_cfg(num_frames=16, crop=312)
torch.manual_seed(0)
model = build_model(cfg).eval()
eng = model.engine if hasattr(model, "engine") else model.core.engine
crops, _ = evaluate.unique_views(cfg)
wide = torch.randn(2, 3, 16, 312, 416, device="cuda")
clips = evaluate.spatial_crops(wide, 312, crops)


def run(n):
    for _ in range(n):
        with torch.no_grad():
            model([clips], {})


for mode in (True, False, True, False):
    eng.batch_interp = mode
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize()
    print("batch_interp=%s  %.3f ms per batch of %d clips" % (mode, (time.perf_counter() - t0) / 40 * 1e3, clips.shape[0]), flush=True)
