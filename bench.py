#!/usr/bin/env python3
"""SViT 16x224^2 bf16 training-step benchmark on MI355X (BASELINE.json metric: clips/sec,
fwd+bwd, B=8 clips per GPU, synthetic SSv2-shaped input + 4 object tokens per frame).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward (DropPath + dropout on, as in training) + CE loss + backward through every
HIP kernel + (N>1: RCCL gradient all-reduce overlapped with backward) + global-norm clip +
AdamW.  Prints ONE JSON line on rank 0.  `--frames-pass` adds the reference's as-released
no-grad B*T single-frame pass (tools/train_net.py:105-110); it is off for the headline metric.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
FWD_GFLOP_PER_CLIP = 138.16    # SURVEY.md 8(d), 16x224^2, 2*MAC, reference flop counter
STEP_GFLOP_PER_CLIP = 412.4    # fwd+bwd: 3x GEMM/bmm/depthwise + 2x patch-embed
# BASELINE.md section 2, per config: (frames, crop) -> fwd+bwd GFLOP per clip (the denominator of step_mfma_frac);
# a shape that is not in the table gets no whole-step fraction (None) rather than the 16x224^2 constant
STEP_GFLOP_BY_SHAPE = {(8, 224): 181.0, (16, 224): STEP_GFLOP_PER_CLIP, (32, 224): 1024.0}


def synth_batch(cfg, batch, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, 3, cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE,
                    cfg.DATA.TRAIN_CROP_SIZE, generator=g)
    y = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), generator=g)
    return x.to(device), y.to(device)


def synth_image_batch(cfg, batch, device, seed):
    """an image rank's batch: stills [B,3,1,S,S]; cxcywh boxes with centres
    U[.25,.75], sizes U[.1,.4], ~10 % empty rows; contact states in {-1, 0, 3}."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    s = cfg.DATA.TRAIN_CROP_SIZE
    x = torch.randn(batch, 3, 1, s, s, generator=g)
    box = torch.cat((torch.rand(batch, 1, 4, 2, generator=g) * 0.5 + 0.25,
                     torch.rand(batch, 1, 4, 2, generator=g) * 0.3 + 0.1), -1)
    box[torch.rand(batch, 1, 4, generator=g) < 0.1] = 0.0
    contact = torch.tensor([-1, 0, 3])[torch.randint(0, 3, (batch, 2), generator=g)]
    return x.to(device), {"haog_bboxes": box.to(device), "contact_state": contact.to(device)}


def usable_cores():
    """CPU share of this process: affinity mask capped by the cgroup cpu quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_leg(R, frames, crop, autocast, timed_steps):
    """1 warm-up + `timed_steps` timed fwd+bwd steps of the oracle at B=1 -> (best s, median s)."""
    spec = R.make_spec(num_frames=frames, crop=crop)
    torch.manual_seed(0)
    p = {k: (torch.randn(s) * 0.02).requires_grad_(True) for k, s in R.param_shapes(spec).items()}
    for k in p:
        if k.endswith("norm.weight") or ".norm" in k and k.endswith("weight"):
            p[k].data.fill_(1.0)
    x = torch.randn(1, 3, frames, crop, crop)
    y = torch.randint(0, 174, (1,))
    times = []
    for i in range(1 + timed_steps):
        t0 = time.perf_counter()
        ds = R.sample_drop_scales(spec, 1)
        keep = (torch.rand(1, 1 + frames * 4, spec.final_dim) > 0.5).float() * 2.0
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            logits, _ = R.forward(p, spec, x, training=True, drop_scales=ds, dropout_keep=keep)
            loss = R.video_loss(logits.float(), y)
        loss.backward()
        for v in p.values():
            v.grad = None
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])
    return t[0], t[len(t) // 2]


def cpu_baseline(frames, crop, timed_steps=3):
    """SURVEY.md 8(d) protocol: the fp32 CPU oracle (oracle/svit_ref.py, pinned against the
    reference) fwd+bwd at B=1, DropPath / dropout on, 1 warm-up + 3 timed steps, best and median,
    for the workload shape (fp32 and bf16-autocast legs) and for C1 (8x224^2 fp32).  `value` is
    the fp32 leg of the workload shape; about 20-30 s of CPU work in all."""
    from oracle import svit_ref as R
    threads = min(usable_cores(), 64)
    torch.set_num_threads(threads)
    legs = {}
    for name, (f, c, ac) in (("%dx%d_fp32" % (frames, crop), (frames, crop, False)),
                             ("%dx%d_bf16_autocast" % (frames, crop), (frames, crop, True)),
                             ("8x224_fp32", (8, 224, False))):
        if name in legs:
            continue
        best, med = _cpu_leg(R, f, c, ac, timed_steps)
        legs[name] = {"best_clips_per_s": round(1.0 / best, 4), "median_clips_per_s": round(1.0 / med, 4),
                      "best_s": round(best, 3), "median_s": round(med, 3)}
    head = legs["%dx%d_fp32" % (frames, crop)]
    return {"value": head["best_clips_per_s"], "unit": "clips/s", "cores": threads, "kind": "port",
            "cpu": cpu_model_name(), "median": head["median_clips_per_s"], "legs": legs,
            "sample": "fp32 CPU oracle (oracle/svit_ref.py), %dx%d^2, B=1, DropPath/dropout on, "
                      "1 warm-up + %d timed fwd+bwd steps, best step %.2f s (median %.2f s); legs: "
                      "same shape under CPU bf16 autocast, and C1 8x224^2 fp32"
                      % (frames, crop, timed_steps, head["best_s"], head["median_s"])}


def aten_gpu_baseline(frames, crop, batch, timed_steps=3):
    """Context only, never credit: the SAME oracle (oracle/svit_ref.py: plain
    ATen ops -- hipBLASLt GEMMs, MIOpen depthwise conv, materialised attention matrices) on THIS GPU
    under torch.autocast("cuda", bfloat16), fwd + CE + bwd at the bench batch, DropPath / dropout on,
    1 warm-up + 3 timed steps.  Says what stock PyTorch-ROCm does with the same math on the same chip."""
    from oracle import svit_ref as R
    dev = torch.device("cuda")
    spec = R.make_spec(num_frames=frames, crop=crop)
    torch.manual_seed(0)
    p = {k: (torch.randn(s, device=dev) * 0.02).requires_grad_(True) for k, s in R.param_shapes(spec).items()}
    for k in p:
        if k.endswith("norm.weight") or ".norm" in k and k.endswith("weight"):
            p[k].data.fill_(1.0)
    x = torch.randn(batch, 3, frames, crop, crop, device=dev)
    y = torch.randint(0, 174, (batch,), device=dev)
    times = []
    for i in range(1 + timed_steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds = [None if d is None else tuple(t.to(dev) for t in d) for d in R.sample_drop_scales(spec, batch)]
        keep = (torch.rand(batch, 1 + frames * 4, spec.final_dim, device=dev) > 0.5).float() * 2.0
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits, _ = R.forward(p, spec, x, training=True, drop_scales=ds, dropout_keep=keep)
            loss = R.video_loss(logits.float(), y)
        loss.backward()
        for v in p.values():
            v.grad = None
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])
    peak_gb = torch.cuda.max_memory_allocated() / 1e9
    return {"value": round(batch / t[0], 2), "unit": "clips/s", "median": round(batch / t[len(t) // 2], 2),
            "ms_per_step": round(t[0] * 1e3, 2), "batch": batch, "kind": "stock ATen ops on the same GPU, bf16 autocast",
            "note": "fwd + CE + bwd, no optimizer step, eager; context only (oracle/svit_ref.py on cuda)",
            "peak_mem_gb": round(peak_gb, 1)}


def kernel_report(trace, batch):
    """Aggregate the HIP-event trace of one profiled step per C-ABI entry point."""
    torch.cuda.synchronize()
    agg = {}
    phases, last = [], None
    for name, e0, e1, meta in trace:
        if name.startswith("mark:"):
            if last is not None:
                phases.append((name[5:], round(last.elapsed_time(e0), 3)))
            last = e0
            continue
        if last is None:
            last = e0
        ms = e0.elapsed_time(e1)
        a = agg.setdefault(name, {"ms": 0.0, "calls": 0, "flop": 0.0, "bytes": 0.0})
        a["ms"] += ms
        a["calls"] += 1
        if meta and meta[0] == "mnk":
            a["flop"] += 2.0 * meta[1] * meta[2] * meta[3]
            if len(meta) > 5:
                a["bytes"] += meta[5]
        elif meta and meta[0] == "flop":
            a["flop"] += meta[1]
        elif meta and meta[0] == "attn":
            _, B, h, Nq, Nk, DA = meta[:6]
            alg = 2.0 * B * h * Nq * Nk * (96 + 96)        # QK^T + AV at head_dim 96
            a["flop"] += alg * (2.0 if name.endswith("bwd") else 1.0)
            if len(meta) > 6:                              # rel-pos dq = D . R^T folded into the dq kernel (a GEMM before)
                a["flop"] += meta[6]
    total = sum(a["ms"] for a in agg.values())
    rows = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        r = {"kernel": name, "ms": round(a["ms"], 3), "calls": a["calls"],
             "share": round(a["ms"] / total, 4)}
        if a["flop"] > 0:
            r["tflops"] = round(a["flop"] / (a["ms"] * 1e-3) / 1e12, 2)
        if a["bytes"] > 0:
            r["gbs"] = round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)
        rows.append(r)
    kernel_report.phases = phases
    return rows, total


def self_launch(n):
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env, as
    torch.distributed.run would), wait for all of them, return the worst exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            failed = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if failed:
                rc = failed[0]   # the rank that exited non-zero ON ITS OWN; its peers would wait in a collective forever
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()         # (a killed peer's -9 must not mask the real failure recorded above)
                p.wait()
            elif rc == 0 and p.returncode:
                rc = p.returncode
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--crop", type=int, default=224)
    ap.add_argument("--frames-pass", action="store_true")
    ap.add_argument("--image-ranks", type=int, default=0,
                    help="the published recipe's heterogeneous layout: the LAST k ranks train still "
                         "images with the HAOG losses (IMAGE_TRAIN.GPU_IDS; SURVEY 8(f) rank 2)")
    ap.add_argument("--image-batch", type=int, default=63, help="images per image rank")
    ap.add_argument("--u8", action="store_true",
                    help="feed decoded uint8 frames [B,T,S,S,3]; normalisation fused into the patch "
                         "embedding (svit_amd/input.py; SURVEY 8(f) rank 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-trace", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel from Python instead of replaying the HIP graphs")
    ap.add_argument("--nt-cfg", type=int, default=None,
                    help="A/B knob for measurements: svit_debug_set(1, N) -- 8 = the NT GEMM heuristic without the ring kernels")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  N fresh child processes, one rank
        # per GPU, started BEFORE this process makes any GPU call (it never does); rank 0 prints
        # the JSON line on the inherited stdout.
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal knobs (not used by the driver): SVIT_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    # SVIT_BENCH_BACKEND=gloo replaces RCCL, so the N > 1 code path can be run on a one-GPU box
    share = os.environ.get("SVIT_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("SVIT_BENCH_BACKEND", "nccl")
    # SVIT_BENCH_FORCE_DP=1 with --gpus 1: a one-rank RCCL process group and the data-parallel wrapper with
    # force_collectives -- the production exchange (bucketed async all-reduce(AVG) between the backward's graph segments)
    # executed on a one-GPU box; the line then reports backend "rccl", ranks_seen 1, "forced_dp": true
    force_dp = os.environ.get("SVIT_BENCH_FORCE_DP") == "1" and world == 1
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from svit_amd import config, optim
    from svit_amd.model import build_model
    if args.nt_cfg is not None:
        import ctypes
        from svit_amd import hip as _hip
        _lib = _hip.load()
        _lib.svit_debug_set.restype, _lib.svit_debug_set.argtypes = ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]
        _lib.svit_debug_set(1, args.nt_cfg)
    cfg = config.ssv2_cfg(num_frames=args.frames, crop=args.crop, num_gpus=world)
    if not 0 <= args.image_ranks <= world:
        raise SystemExit("--image-ranks must be between 0 and --gpus")
    cfg.IMAGE_TRAIN.GPU_IDS = list(range(world - args.image_ranks, world))
    from svit_amd.dp import rank_role
    is_image = rank_role(cfg, local_rank).is_image
    torch.manual_seed(cfg.RNG_SEED)
    if (share and world > 1) or force_dp:        # build_model asserts NUM_GPUS <= visible devices (build.py:28-35)
        from svit_amd.dp import DataParallel
        from svit_amd.model import MODEL_REGISTRY
        model = DataParallel(MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg).cuda(dev_index), force_collectives=force_dp)
    else:
        model = build_model(cfg, gpu_id=dev_index)
    model.train()
    opt = optim.construct_optimizer(model, cfg)
    if is_image:
        from svit_amd import losses
        x, y = synth_image_batch(cfg, args.image_batch, dev, seed=cfg.RNG_SEED + rank)
        haog = losses.VideoImageLoss(cfg, is_video_rank=False)

        def ce(preds, extra, labels):            # this rank's loss: sum_k lambda_k * HAOG loss_k
            return haog.total(haog(preds, extra, None, labels))
    else:
        x, y = synth_batch(cfg, args.batch, dev, seed=cfg.RNG_SEED + rank)

        from svit_amd import losses

        def ce(preds, extra, labels):            # VideoImageLoss's loss_ce on a video rank (one launch each way since round 6)
            return losses.cross_entropy(preds, labels)
    if args.u8:
        if is_image or args.frames_pass:
            raise SystemExit("--u8 is wired for the clip step only")
        from svit_amd.input import U8Clips
        g = torch.Generator(device="cpu").manual_seed(cfg.RNG_SEED + rank)
        x = U8Clips(torch.randint(0, 256, (args.batch, args.frames, args.crop, args.crop, 3),
                                  generator=g, dtype=torch.uint8).to(dev), args.crop,
                    mean=cfg.DATA.MEAN, std=cfg.DATA.STD)
    core = model.module if hasattr(model, "module") else model
    frames_pass = args.frames_pass and not is_image    # stills have no frames pass (train_net.py:105)

    graphed, launch_note = None, "eager"
    if not args.eager:
        from svit_amd.graph import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(model, ce, [x], y, frames_pass=frames_pass)
            launch_note = "hip-graph replay"
            if not args.u8:
                # the synthetic batch lives in the step's own input buffers (where a loader's host-to-device copy
                # would land): no device-to-device copy inside the timed region
                x, y = graphed.static_inputs[0], graphed.static_labels
        except Exception as exc:            # never lose a measurement to a capture problem
            print("graph capture failed (%s: %s); falling back to eager launches"
                  % (type(exc).__name__, exc), file=sys.stderr)
            torch.cuda.synchronize()
            launch_note = "eager (graph capture failed: %s)" % type(exc).__name__

    def step(it, eager=False):
        optim.set_lr(opt, optim.get_lr_at_epoch(cfg, it / 1000.0))
        if graphed is not None and not eager:
            loss, _ = graphed([x], y)       # fwd + CE + bwd (+ all-reduce launches) replayed
            opt.step()
            return loss
        logits, extra = model([x], {})
        if frames_pass:
            with torch.no_grad():
                model([x.transpose(1, 2).flatten(0, 1).unsqueeze(2)], {})
        loss = ce(logits, extra, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for it in range(args.warmup):
        loss = step(it)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = step(args.warmup + it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    loss_val = float(loss)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    ranks_seen = dist.get_world_size() if (world > 1 or force_dp) else 1
    n_vid = world - args.image_ranks
    clips_per_s = args.batch * n_vid * args.steps / dt

    out = {
        "metric": "clips/sec (fwd+bwd) SViT %dx%d^2 bf16" % (args.frames, args.crop),
        "value": round(clips_per_s, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "SViT %dx%d^2, %d clips/GPU, 4 object tokens/frame, fwd+CE+bwd+"
                               "clip+AdamW%s" % (args.frames, args.crop, args.batch,
                                                 " + no-grad frames pass" if args.frames_pass else ""),
                   "global_batch": args.batch * world, "seq_len": None,
                   "parallelism": "dp%d" % world, "ranks_seen": ranks_seen,
                   "backend": ("rccl" if backend == "nccl" else backend) if (world > 1 or force_dp) else None,
                   "forced_dp": force_dp,
                   "launch": launch_note, "input": "uint8 frames" if args.u8 else
                   ("fp32 clips, resident in the replayed step's input buffers" if graphed is not None else "fp32 clips"),
                   "hip_library": os.path.relpath(__import__("svit_amd.hip", fromlist=["LIB_PATH"]).LIB_PATH, ROOT)},
        "loss": round(loss_val, 4),
        "step_mfma_frac": None,
    }
    step_gflop = STEP_GFLOP_BY_SHAPE.get((cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE))
    if step_gflop is not None:      # clips/s per video rank x FLOPs per clip / peak
        out["step_mfma_frac"] = round(clips_per_s / max(1, n_vid) * step_gflop * 1e9 / (MFMA_PEAK_TFLOPS * 1e12), 4)
        out["step_gflop_per_clip"] = step_gflop
    if args.image_ranks:
        # heterogeneous layout: `value` counts the clips of the video ranks only
        out["config"]["workload"] += "; %d image rank(s) x %d stills with HAOG losses" % (
            args.image_ranks, args.image_batch)
        out["config"]["global_batch"] = args.batch * n_vid
        out["config"]["parallelism"] = "dp%d (%d video + %d image ranks)" % (world, n_vid, args.image_ranks)
        out["images_per_s"] = round(args.image_batch * args.image_ranks * args.steps / dt, 2)
        if n_vid == 0:     # image ranks only: a side measurement (load balance of the recipe)
            out.update({"metric": "images/sec (fwd+bwd) SViT image rank 1x%d^2 bf16" % args.crop,
                        "value": out["images_per_s"], "unit": "images/s"})
        args.no_kernel_trace = args.no_kernel_trace or is_image or world > 1
        args.no_cpu_baseline = True
    if rank == 0 and not args.no_kernel_trace:
        from svit_amd import hip
        # per-kernel HIP events need eager launches.  An event pair brackets the launch's dispatch as well as its
        # execution: against the rocprofv3 kernel trace of the same command every launch reads ~2 us long (sum of the
        # event durations 13.6 ms for a 12.8 ms step; queueing the traced step behind a 120-ms GPU-side spin, so that the
        # host cannot starve the device, changed nothing: profiles/r04 notes in DESIGN.md 5d) -- the fractions below are
        # therefore slightly LOW; profiles/*_bench_kernel_stats.csv holds the trace's durations
        hip.start_trace()
        step(args.warmup + args.steps, eager=True)
        rows, total = kernel_report(hip.stop_trace(), args.batch)
        out["kernel_timing"] = "HIP events around every launch of one eager step (dispatch included: ~2 us per launch above the rocprofv3 trace)"
        out["kernels"] = rows[:12]
        # NOT a breakdown of ms_per_step: wall spans between host-side marks of the ONE EAGER step traced above (host launch
        # gaps and the per-launch event brackets included) -- they sum to ~1.7x the replayed step
        out["eager_trace_phase_spans_ms"] = dict(kernel_report.phases)
        out["kernel_ms_total"] = round(total, 3)
        top = rows[0]
        # the north-star kernel is the fused attention; report the dominant kernel's roofline and
        # always the attention forward's and backward's
        attn = next(r for r in rows if r["kernel"] == "svit_attn_fwd")
        attn_b = next((r for r in rows if r["kernel"] == "svit_attn_bwd"), None)
        dom = top if "tflops" in top else attn
        frac_mfma = dom["tflops"] / MFMA_PEAK_TFLOPS
        frac_hbm = dom.get("gbs", 0.0) / HBM_PEAK_GBS
        # the Linear GEMMs are priced against the dense bf16 MFMA roof (frac = frac_mfma);
        # frac_hbm beside it says how far the same launches are from the HBM roof (most K <= 384 shapes
        # sit under the ridge).  The peak is the contract's 2.5 PFLOP/s (2.4 GHz); under this load the
        # shader clock holds ~1.9 GHz, i.e. a kernel that kept the matrix pipe 100 % busy would read 0.79.
        # which roof bounds the dominant kernel: its arithmetic intensity (algorithmic flop / algorithmic byte, summed over
        # its launches) against the ridge 2.5 PFLOP/s / 8 TB/s = 312 flop/B; both fractions are always reported
        intensity = dom["tflops"] * 1e3 / dom["gbs"] if dom.get("gbs") else float("inf")
        ridge = MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS
        if intensity >= ridge:
            out["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"],
                               "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(frac_mfma, 4)}
        else:
            out["roofline"] = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["gbs"],
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac_hbm, 4)}
        out["roofline"].update({"intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                                "frac_mfma": round(frac_mfma, 4), "frac_hbm": round(frac_hbm, 4),
                                "traffic": None, "avg_launch_ms": round(dom["ms"] / dom["calls"], 4),
                                "algorithmic": "2*M*N*K flop and operand+output+epilogue-slab bytes "
                                               "per call, summed over the step's %d calls" % dom["calls"]})
        # memory-side bytes per launch: PMC counters cannot be read from inside this process, so
        # the committed rocprofv3 --pmc summary is quoted together with the git head it was taken at
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            t = pmc["bytes_per_launch"][dom["kernel"]]
            out["roofline"]["traffic"] = t["fetch"] + t["write"]
            out["roofline"]["traffic_unit"] = "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE)"
            out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json"
            out["roofline"]["traffic_head"] = pmc.get("git_head")
        except (OSError, KeyError, ValueError):
            pass
        out["roofline_attn_fwd"] = {"bound": "mfma", "achieved": attn["tflops"],
                                    "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(attn["tflops"] / MFMA_PEAK_TFLOPS, 4),
                                    "avg_launch_ms": round(attn["ms"] / attn["calls"], 4)}
        if attn_b is not None:
            out["roofline_attn_bwd"] = {"bound": "mfma", "achieved": attn_b["tflops"],
                                        "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": round(attn_b["tflops"] / MFMA_PEAK_TFLOPS, 4),
                                        "avg_launch_ms": round(attn_b["ms"] / attn_b["calls"], 4)}
    elif world > 1 and not args.no_kernel_trace:
        step(args.warmup + args.steps, eager=True)  # keep ranks in lock-step with rank 0's traced step
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.frames, args.crop)
        del graphed, model, opt
        torch.cuda.empty_cache()
        try:
            out["aten_gpu_baseline"] = aten_gpu_baseline(args.frames, args.crop, args.batch)
        except Exception as exc:        # context only: never lose the line to it
            out["aten_gpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
