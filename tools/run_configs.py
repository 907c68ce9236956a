#!/usr/bin/env python3
"""One training step (fwd + CE + bwd + optimizer) at the other BASELINE.json shapes, timed:
   python tools/run_configs.py   -> 32x224^2 (long clip) and 16x312^2 (test-time crop)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import config, optim
from svit_amd.model import build_model

for frames, crop, B in ((32, 224, 4), (16, 312, 4), (8, 224, 8)):
    cfg = config.ssv2_cfg(num_frames=frames, crop=crop, num_gpus=1)
    torch.manual_seed(0)
    model = build_model(cfg, gpu_id=0)
    model.train()
    opt = optim.construct_optimizer(model, cfg)
    x = torch.randn(B, 3, frames, crop, crop, device="cuda")
    y = torch.randint(0, 174, (B,), device="cuda")

    def step():
        logits, _ = model([x], {})
        loss = torch.nn.functional.cross_entropy(logits, y)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    g = model.flat.grad
    print("%dx%d^2 B=%d: loss %.4f finite=%s grad-norm %.3f  %.1f ms/step  %.1f clips/s" %
          (frames, crop, B, float(loss), bool(torch.isfinite(g).all()), float(g.norm()), dt * 1e3, B / dt))
    del model, opt
    torch.cuda.empty_cache()
