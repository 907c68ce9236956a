"""Throughput of the multi-view test path (BASELINE config C5) on one MI355X:
eval-mode forward of the unique spatial crops + device-side ensemble, no host sync in the loop.

    python tools/bench_eval.py --crop 224 --videos 4     # ssv2.yaml test setting (3 x 224^2)
    python tools/bench_eval.py --crop 312 --videos 2     # BASELINE C5 shape (3 x 312^2)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=224)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--videos", type=int, default=4, help="videos per batch (x3 crops)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nograd-scratch", type=int, default=1, help="0: no-grad pooling on the streaming kernels (A/B)")
    args = ap.parse_args()
    from svit_amd import config, evaluate, ops
    ops.NOGRAD_SCRATCH = bool(args.nograd_scratch)
    from svit_amd.model import build_model
    cfg = config.ssv2_cfg(num_frames=args.frames, crop=args.crop)
    torch.manual_seed(0)
    model = build_model(cfg).eval()
    crops, repeat = evaluate.unique_views(cfg)
    V = args.videos * (args.iters + args.warmup)
    width = int(round(args.crop * 4 / 3))                       # 4:3 source, short side = crop
    wide = torch.randn(args.videos, 3, args.frames, args.crop, width, device="cuda")
    labels_v = torch.randint(0, cfg.MODEL.NUM_CLASSES, (V,), device="cuda")
    meter = evaluate.TestMeter(V, crops, cfg.MODEL.NUM_CLASSES, args.iters)

    def one(it):
        vids = torch.arange(it * args.videos, (it + 1) * args.videos, device="cuda")
        ids = (vids[:, None] * crops + torch.arange(crops, device="cuda")[None]).flatten()
        clips = evaluate.spatial_crops(wide, args.crop, crops)
        with torch.no_grad():
            probs, _ = model([clips], {})
        meter.update_stats(probs, labels_v[ids // crops], ids, repeat=repeat)

    for it in range(args.warmup):
        one(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.iters):
        one(it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = meter.finalize_metrics((1, 5))
    print(json.dumps({
        "metric": "videos/sec, %d-crop test ensemble, SViT %dx%d^2 bf16 eval" % (crops, args.frames, args.crop),
        "value": round(args.videos * args.iters / dt, 2), "unit": "videos/s",
        "clips_per_s": round(args.videos * crops * args.iters / dt, 2),
        "ms_per_batch": round(dt / args.iters * 1e3, 3), "videos_per_batch": args.videos,
        "listed_clips_per_video": crops * repeat, "computed_clips_per_video": crops,
        "clip_count_per_video": int(meter.clip_count[0]), "stats": stats}))


if __name__ == "__main__":
    main()
