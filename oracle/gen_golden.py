#!/usr/bin/env python3
"""Generate tests/golden/* from the UNMODIFIED reference (this container only).

TEST INFRASTRUCTURE ONLY.  Run:   python oracle/gen_golden.py [--out tests/golden]

What it does, per case:
  1. builds the reference model (`slowfast.models.build_model`, imported from
     /root/reference through oracle/ref_shim.py) on CPU, fp32;
  2. loads the closed-form weights of oracle/procedural.py (`load_state_dict`, strict);
  3. runs the reference forward (+ loss + backward) on closed-form inputs;
  4. runs the build's CPU restatement (oracle/svit_ref.py) on the same tensors and records
     the max-abs disagreement (the restatement is *pinned* by this);
  5. writes small fixtures: full tensors where small, digests (oracle/procedural.digest)
     where large.
Only arrays / numbers produced by running the reference are written -- no reference source.
"""
import argparse
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import procedural as P
import ref_shim
import svit_ref as R

torch.set_num_threads(8)
SMALL = 1024  # tensors up to this many elements are stored in full
WHOLE_REL_POS = 11000  # ... and the rel-pos table gradients whole (<= 111 x 96 each): the tensors that are
                       # always the worst of a bf16 step, so they are judged on every element, not on a sample


def build_reference(num_frames, crop, drop_path=0.0, dropout=0.0, extra=()):
    cfg = ref_shim.reference_cfg(num_frames, crop, overrides=(
        "MVIT.DROPPATH_RATE", drop_path, "MODEL.DROPOUT_RATE", dropout) + tuple(extra))
    from slowfast.models import build_model
    model = build_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = P.state_dict(shapes)
    model.load_state_dict(sd, strict=True)
    return cfg, model, shapes, sd


def maxabs(a, b):
    if a is None:  # the restatement leaves untouched params without a grad; the reference's
        a = torch.zeros_like(b)  # 0*sum(params) touch gives them exact zeros
    return float((a.detach().double() - b.detach().double()).abs().max())


def relerr(a, b, floor=1e-6):
    """max-abs error relative to the tensor's scale; `floor` keeps mathematically-zero grads
    (e.g. attn.norm_k.bias: softmax is invariant to a per-row constant) from reading as 100%."""
    return maxabs(a, b) / max(float(b.detach().abs().max()), floor)


def store(arrays, digests, key, t, sample=False):
    t = t.detach()
    if t.numel() <= SMALL or key in ("logits", "eval_probs") or \
            ("rel_pos_" in key and "grad:" in key and t.numel() <= WHOLE_REL_POS):
        arrays[key] = t.to(torch.float32).numpy()
    elif sample:        # strided sample of a large tensor (P.sample_of rebuilds the same indices)
        arrays["sample:" + key] = P.sample_of(t).to(torch.float32).numpy()
    digests[key] = P.digest(t)


def run_model_case(name, num_frames, crop, batch, out_dir, manifest, backward=True,
                   frames_path=False, eval_too=False, drop=False):
    torch.manual_seed(0)
    dp, do = (0.4, 0.5) if drop else (0.0, 0.0)
    cfg, model, shapes, sd = build_reference(num_frames, crop, dp, do)
    spec = R.make_spec(num_frames=num_frames, crop=crop, drop_path_rate=dp, dropout_rate=do)
    assert R.param_shapes(spec) == shapes, "state_dict layout mismatch vs reference"
    x = P.frames(batch, 1 if frames_path else num_frames, crop)
    if frames_path:
        x = x  # [B,3,1,S,S]: the T=1 path of a num_frames model (train_net.py:105-110)
    y = P.labels(batch)
    arrays, digests, agree = {}, {}, {}

    # ---- reference, train mode -----------------------------------------------------
    model.train()
    taps_ref = {}
    hooks = []
    for i, blk in enumerate(model.blocks):
        hooks.append(blk.register_forward_hook(
            lambda m, inp, out, i=i: taps_ref.__setitem__("block%d" % i, out[0].detach())))
    masks = {}
    if drop:
        from slowfast.models.common import DropPath
        for i, blk in enumerate(model.blocks):
            if isinstance(blk.drop_path, DropPath):
                def grab(m, inp, out, i=i):
                    xin, xout = inp[0].detach(), out.detach()
                    idx = xin.flatten(1).abs().argmax(dim=1)
                    num = xout.flatten(1).gather(1, idx[:, None])[:, 0]
                    den = xin.flatten(1).gather(1, idx[:, None])[:, 0]
                    masks.setdefault(i, []).append(num / den)
                hooks.append(blk.drop_path.register_forward_hook(grab))
        def grab_do(m, inp, out):
            masks["dropout"] = (out.detach() != 0).float() / (1 - do)
        hooks.append(model.head.dropout.register_forward_hook(grab_do))
    with torch.set_grad_enabled(backward):
        logits, extra = model([x], {})
    for h in hooks:
        h.remove()
    loss = torch.nn.functional.cross_entropy(logits, y)
    if backward:
        model.zero_grad()
        loss.backward()

    # ---- restatement on the same tensors -------------------------------------------
    p = {k: v.clone().requires_grad_(backward) for k, v in sd.items()}
    drop_scales = None
    dropout_keep = None
    if drop:
        drop_scales = [None if i not in masks else (masks[i][0], masks[i][1])
                       for i in range(spec.depth)]
        dropout_keep = masks["dropout"]
        for i, m in masks.items():
            if i != "dropout":
                arrays["dp_attn_%d" % i] = m[0].numpy()
                arrays["dp_mlp_%d" % i] = m[1].numpy()
        arrays["dropout_keep"] = dropout_keep.numpy().astype(np.float32) \
            if dropout_keep.numel() <= 65 * 768 * 4 else None
        if arrays["dropout_keep"] is None:
            del arrays["dropout_keep"]
    taps = {}
    with torch.set_grad_enabled(backward):
        lg, ex = R.forward(p, spec, x, training=True, drop_scales=drop_scales,
                           dropout_keep=dropout_keep, taps=taps)
    ls = R.video_loss(lg, y)
    if backward:
        ls.backward()

    agree["logits"] = maxabs(lg, logits)
    agree["loss"] = maxabs(ls, loss)
    for k in extra:
        agree[k] = maxabs(ex[k], extra[k])
    for i in range(spec.depth):
        agree["block%d_rel" % i] = relerr(taps["block%d" % i], taps_ref["block%d" % i])
    if backward:
        worst, worst_name = 0.0, ""
        gmax = max(float(v.grad.abs().max()) for v in model.parameters())
        for k, v in model.named_parameters():
            e = relerr(p[k].grad, v.grad, floor=1e-5 * gmax)
            if e > worst:
                worst, worst_name = e, k
        agree["grad_rel_worst"] = worst
        print("   worst grad:", worst_name, worst, float(dict(model.named_parameters())[worst_name].grad.abs().max()))

    store(arrays, digests, "logits", logits)
    store(arrays, digests, "loss", loss)
    for k in extra:
        store(arrays, digests, k, extra[k])
    for i in range(spec.depth):
        store(arrays, digests, "block%d" % i, taps_ref["block%d" % i])
    if backward:
        for k, v in model.named_parameters():
            store(arrays, digests, "grad:" + k, v.grad, sample=True)

    # ---- eval mode -------------------------------------------------------------------
    if eval_too:
        model.eval()
        with torch.no_grad():
            pe, ee = model([x], {})
            p0 = {k: v.detach() for k, v in p.items()}
            pr, er = R.forward(p0, spec, x, training=False)
        agree["eval_probs"] = maxabs(pr, pe)
        store(arrays, digests, "eval_probs", pe)
        for k in ee:
            agree["eval_" + k] = maxabs(er[k], ee[k])
            store(arrays, digests, "eval_" + k, ee[k])

    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrays)
    manifest["cases"][name] = {
        "num_frames": num_frames, "crop": crop, "batch": batch, "frames_path": frames_path,
        "drop": drop, "backward": backward, "tokens": int(taps_ref["block0"].shape[1])
        if not spec.blocks[0].pools_q else None,
        "restatement_vs_reference_maxabs": agree, "digests": digests,
    }
    print(name, "worst disagreement:", max(agree.values()), {k: "%.2e" % v for k, v in agree.items() if v > 1e-4}, flush=True)
    return cfg, model, spec, sd, extra


def run_op_cases(out_dir, manifest):
    """Per-op known-answer tests taken from the reference's own functions."""
    ref_shim.install()
    from slowfast.models import attention as A
    arrays, agree = {}, {}
    # (a) attention_pool with depthwise conv, strides 1/2/4/8, and the max-pool skip
    for s in (1, 2, 4, 8):
        T, H, W, h, O = 2, 8, 8, 2, 3
        x = P.tensor("kat:pool:x:%d" % s, (2, h, 1 + T * H * W + O, 96), 1.0)
        w = P.tensor("kat:pool:w:%d" % s, (96, 1, 3, 3, 3), 0.3)
        nw = P.tensor("kat:pool:nw", (96,), 0.2, 1.0)
        nb = P.tensor("kat:pool:nb", (96,), 0.1)
        conv = torch.nn.Conv3d(96, 96, 3, stride=(1, s, s), padding=1, groups=96, bias=False)
        conv.weight.data.copy_(w)
        norm = torch.nn.LayerNorm(96, eps=1e-6)
        norm.weight.data.copy_(nw)
        norm.bias.data.copy_(nb)
        xr = x.clone().requires_grad_(True)
        out_ref, thw = A.attention_pool(xr, conv, [T, H, W], has_cls_embed=True, norm=norm)
        out_ref.backward(P.tensor("kat:pool:g:%d" % s, tuple(out_ref.shape), 1.0))
        xo = x.clone().requires_grad_(True)
        wo = w.clone().requires_grad_(True)
        out, thw2 = R.pool_tokens(xo, (T, H, W), (1, s, s), wo, nw, nb, O)
        out.backward(P.tensor("kat:pool:g:%d" % s, tuple(out.shape), 1.0))
        assert tuple(thw) == tuple(thw2)
        agree["pool_s%d" % s] = maxabs(out, out_ref)
        agree["pool_s%d_dx" % s] = maxabs(xo.grad, xr.grad)
        agree["pool_s%d_dw" % s] = maxabs(wo.grad, conv.weight.grad)
        arrays["pool_s%d_out" % s] = out_ref.detach().numpy()
        arrays["pool_s%d_dx" % s] = xr.grad.numpy()
        arrays["pool_s%d_dw" % s] = conv.weight.grad.numpy()
        arrays["pool_s%d_gain" % s] = R.object_gain(w, (1, s, s)).numpy()
    T, H, W, O = 2, 8, 8, 3
    x = P.tensor("kat:skip:x", (2, 1 + T * H * W + O, 96), 1.0)
    mp = torch.nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
    out_ref, _ = A.attention_pool(x, mp, [T, H, W], has_cls_embed=True)
    agree["skip"] = maxabs(R.maxpool_skip(x, (T, H, W), (1, 2, 2), O), out_ref)
    arrays["skip_out"] = out_ref.numpy()
    # (b) rel-pos bias incl. the interpolation branch and T'=1
    for tag, q_thw, k_thw, rows_sp, rows_t in (
            ("same", (2, 4, 4), (2, 4, 4), 7, 3),
            ("kvpool", (2, 8, 8), (2, 2, 2), 15, 3),
            ("interp", (2, 5, 5), (2, 3, 3), 7, 3),      # needs 9 rows, owns 7
            ("t1", (1, 4, 4), (1, 2, 2), 7, 5)):         # T'=1: table 5 rows -> 1
        Lq = q_thw[0] * q_thw[1] * q_thw[2]
        Lk = k_thw[0] * k_thw[1] * k_thw[2]
        q = P.tensor("kat:rel:q:" + tag, (2, 2, 1 + Lq + 3, 96), 1.0)
        k = P.tensor("kat:rel:k:" + tag, (2, 2, 1 + Lk + 3, 96), 1.0)
        rh = P.tensor("kat:rel:h:" + tag, (rows_sp, 96), 0.3)
        rw = P.tensor("kat:rel:w:" + tag, (rows_sp, 96), 0.3)
        rt = P.tensor("kat:rel:t:" + tag, (rows_t, 96), 0.3)
        attn = torch.zeros(2, 2, 1 + Lq + 3, 1 + Lk + 3)
        attn = A.cal_rel_pos_spatial(attn, q, k, True, q_thw, k_thw, rh, rw)
        attn = A.cal_rel_pos_temporal(attn, q, True, q_thw, k_thw, rt)
        bias = R.rel_pos_bias(q, q_thw, k_thw, rh, rw, rt)
        agree["rel_" + tag] = maxabs(bias, attn[:, :, 1:1 + Lq, 1:1 + Lk])
        assert float(attn[:, :, 0].abs().max()) == 0 and float(attn[:, :, :, 0].abs().max()) == 0
        arrays["rel_%s_bias" % tag] = attn[:, :, 1:1 + Lq, 1:1 + Lk].numpy()
    np.savez_compressed(os.path.join(out_dir, "ops.npz"), **arrays)
    manifest["ops"] = {"restatement_vs_reference_maxabs": agree}
    print("ops worst disagreement:", max(agree.values()), flush=True)


def run_loss_optim_cases(out_dir, manifest):
    """HAOG losses, lr policy, param groups and one clip+AdamW step on the tiny model."""
    torch.manual_seed(0)
    cfg, model, shapes, sd = build_reference(4, 64)
    spec = R.make_spec(num_frames=4, crop=64, drop_path_rate=0.0, dropout_rate=0.0)
    from slowfast.models import losses as L
    from slowfast.models import optimizer as O
    from slowfast.utils import lr_policy, misc
    arrays, agree, info = {}, {}, {}
    # image-rank step: T=1 images, HAOG losses
    B = 3
    x = P.frames(B, 1, 64, tag="img")
    meta = P.haog_meta(B)
    model.train()
    logits, extra = model([x], {})
    lf = L.VideoImageLoss(cfg)
    lf._is_vid = False
    lf.train()
    parts = lf(logits, extra, None, meta)
    lam = misc.get_lambdas_dict(cfg)
    total = sum(lam[k] * v for k, v in parts.items())
    model.zero_grad()
    total.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lg, ex = R.forward(p, spec, x, training=True)
    tot, prt = R.image_loss(ex, meta, R.loss_weights(cfg.SVIT.LAMBDA_NODES, cfg.SVIT.LAMBDA_EDGES))
    tot.backward()
    agree["image_total"] = maxabs(tot, total)
    for k in parts:
        agree["image_" + k] = maxabs(prt[k], parts[k])
        info["image_" + k] = float(parts[k].detach())
    info["image_total"] = float(total)
    info["lambdas"] = {k: float(v) for k, v in lam.items()}
    worst = 0.0
    gmax = max(float(v.grad.abs().max()) for v in model.parameters())
    for k, v in model.named_parameters():
        worst = max(worst, relerr(p[k].grad, v.grad, floor=1e-5 * gmax))
    agree["image_grad_rel_worst"] = worst
    info["image_grad_digest"] = {k: P.digest(v.grad) for k, v in model.named_parameters()
                                 if k.startswith("head.") or k in ("cls_token", "object_queries",
                                                                   "pos_embed_temporal")}
    # lr policy samples
    eps = [0.0, 0.013, 1.0, 12.5, 25.0, 49.99]
    info["lr_epochs"] = eps
    info["lr_values"] = [float(lr_policy.get_lr_at_epoch(cfg, e)["lr"]) for e in eps]
    agree["lr"] = max(abs(R.cosine_lr(e, cfg.SOLVER.BASE_LR, cfg.SOLVER.COSINE_END_LR,
                                      cfg.SOLVER.MAX_EPOCH, cfg.SOLVER.WARMUP_EPOCHS,
                                      cfg.SOLVER.WARMUP_START_LR) - v)
                      for e, v in zip(eps, info["lr_values"]))
    # param groups + one clip + AdamW step (train_net.py:133-151)
    opt = O.construct_optimizer(model, cfg)
    wd_by_id = {}
    for g in opt.param_groups:
        for q in g["params"]:
            wd_by_id[id(q)] = g["weight_decay"]
    info["n_wd_zero"] = sum(1 for k, v in model.named_parameters() if wd_by_id[id(v)] == 0.0)
    info["n_wd"] = sum(1 for k, v in model.named_parameters() if wd_by_id[id(v)] > 0.0)
    for k, v in model.named_parameters():
        assert R.weight_decay_of(k, tuple(v.shape), cfg.SOLVER.WEIGHT_DECAY) == wd_by_id[id(v)], k
    lr = 1.5e-4
    for g in opt.param_groups:
        g["lr"] = lr
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.CLIP_GRAD_L2NORM)
    opt.step()
    pw = {k: v.detach().clone() for k, v in p.items()}
    gr = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pw.items()}
    n2 = R.clip_and_adamw_step(pw, gr, st, lr, 1, cfg.SOLVER.CLIP_GRAD_L2NORM,
                               lambda n, s: R.weight_decay_of(n, s, cfg.SOLVER.WEIGHT_DECAY))
    agree["grad_norm"] = abs(n2 - float(norm)) / float(norm)
    info["grad_norm"] = float(norm)
    worst = 0.0
    for k, v in model.named_parameters():
        worst = max(worst, maxabs(pw[k], v))
    agree["adamw_param_maxabs"] = worst
    info["adamw_param_digest"] = {k: P.digest(v) for k, v in model.named_parameters()
                                  if v.numel() <= 200000}
    info["adamw_lr"] = lr
    np.savez_compressed(os.path.join(out_dir, "loss_optim.npz"), **arrays)
    manifest["loss_optim"] = {"restatement_vs_reference_maxabs": agree, "info": info}
    print("loss/optim worst disagreement:", max(agree.values()), agree, flush=True)


def run_meter_case(out_dir, manifest):
    """The reference's own TestMeter / topks_correct / uniform_crop on a seeded stream of clip
    predictions: 12 videos x (2 ensemble views x 3 crops), 29 classes,
    batches of 8 clips in dataset order and in a shuffled order, "sum" and "max"."""
    ref_shim.install()
    import importlib.util
    import slowfast.utils.meters as meters
    import slowfast.utils.metrics as metrics
    # slowfast.datasets is a stub package (its __init__ pulls decoders the image lacks): point it
    # at the real directory so that the unmodified transform.py (and the two augmentation files it
    # imports) load from the reference tree; torchvision.transforms is plumbing uniform_crop never
    # touches
    tv = sys.modules["torchvision"]
    tv.transforms = ref_shim._mod("torchvision.transforms")
    tv.transforms.functional = ref_shim._mod("torchvision.transforms.functional")
    sys.modules["slowfast.datasets"].__path__ = [ref_shim.REFERENCE_ROOT + "/slowfast/datasets"]
    import slowfast.datasets.transform as transform
    import meter_ref
    V, views, crops, C, bs = 12, 2, 3, 29, 8
    n = V * views * crops
    u = P.hash_uniform("meter:preds", n * C).astype(np.float32).reshape(n, C)
    probs = torch.softmax(torch.from_numpy(u) * 3.0, dim=1)
    labels_v = torch.from_numpy(((P.hash_uniform("meter:labels", V) + 1) * 0.5 * C).astype(np.int64)).clamp_(0, C - 1)
    # make the label the strongest class for about half of the videos so top-1/5 are not trivial
    for v in range(0, V, 2):
        probs[v * views * crops:(v + 1) * views * crops, labels_v[v]] += 0.08
    order = {"ordered": np.arange(n),
             "shuffled": np.argsort(P.hash_uniform("meter:perm", n), kind="stable")}
    arrays = {"probs": probs.numpy(), "labels_v": labels_v.numpy()}
    agree = {}
    for oname, perm in order.items():
        arrays["perm_" + oname] = perm.astype(np.int64)
        for method in ("sum", "max"):
            m = meters.TestMeter(V, views * crops, C, (n + bs - 1) // bs, ensemble_method=method)
            r = meter_ref.TestMeterRef(V, views * crops, C, ensemble_method=method)
            for a in range(0, n, bs):
                ids = torch.from_numpy(perm[a:a + bs].astype(np.int64))
                lab = labels_v[ids // (views * crops)]
                m.update_stats(probs[ids], lab, ids)
                r.update_stats(probs[ids].numpy(), lab.numpy(), ids.numpy())
            m.finalize_metrics(ks=(1, 5))
            key = "%s_%s" % (oname, method)
            arrays[key + "_video_preds"] = m.video_preds.numpy()
            arrays[key + "_clip_count"] = m.clip_count.numpy()
            arrays[key + "_video_labels"] = m.video_labels.numpy()
            correct = [int(c) for c in metrics.topks_correct(m.video_preds, m.video_labels, (1, 5))]
            arrays[key + "_topk_correct"] = np.array(correct, np.int64)
            rstats, rcorrect = r.finalize_metrics((1, 5))
            assert np.array_equal(r.video_preds, m.video_preds.numpy()), key       # bit-exact
            assert np.array_equal(r.clip_count, m.clip_count.numpy()) and rcorrect == correct
            assert rstats["top1_acc"] == m.stats["top1_acc"] and rstats["top5_acc"] == m.stats["top5_acc"]
            agree[key] = {"top1_acc": m.stats["top1_acc"], "top5_acc": m.stats["top5_acc"],
                          "correct": correct}
    # uniform_crop of the reference: offsets recovered from a coordinate image
    crops_tab = []
    for (h, w, size) in [(224, 298, 224), (312, 415, 312), (300, 224, 224), (224, 224, 224), (225, 301, 224)]:
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        img = torch.stack([yy, xx]).float().unsqueeze(0)            # [1, 2, H, W]
        for sidx in range(3):
            c, _ = transform.uniform_crop(img, size, sidx)
            y0, x0 = int(c[0, 0, 0, 0]), int(c[0, 1, 0, 0])
            assert (y0, x0) == meter_ref.uniform_crop_offsets(h, w, size, sidx)
            crops_tab.append([h, w, size, sidx, y0, x0])
    arrays["crop_offsets"] = np.array(crops_tab, np.int64)
    np.savez_compressed(os.path.join(out_dir, "meter.npz"), **arrays)
    manifest["meter"] = {"videos": V, "ensemble_views": views, "spatial_crops": crops, "classes": C,
                         "batch": bs, "results": agree}
    print("meter", agree)


def run_layout_case(out_dir, manifest):
    """Checkpoint / optimizer-state layout of the reference: the order of
    `named_parameters()`, the param groups `construct_optimizer` builds, and the structure of the
    file `cu.save_checkpoint` writes -- names, shapes, dtypes and a few digests (data, no code)."""
    import shutil
    import tempfile
    ref_shim.install()
    import slowfast.models.optimizer as optim
    import slowfast.utils.checkpoint as cu
    cfg, model, shapes, sd = build_reference(16, 224)
    names = [n for n, _ in model.named_parameters()]
    opt = optim.construct_optimizer(model, cfg)
    for i, (n, p) in enumerate(model.named_parameters()):        # one step so the state exists
        p.grad = 1e-3 * P.tensor("layout:grad:" + n, tuple(p.shape))
    opt.step()
    tmp = tempfile.mkdtemp(prefix="svit_layout_")
    try:
        path = cu.save_checkpoint(tmp, model, opt, 4, cfg, scaler=torch.cuda.amp.GradScaler(enabled=False))
        ck = torch.load(path, map_location="cpu", weights_only=False)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    osd = ck["optimizer_state"]
    name_of = {id(p): n for n, p in model.named_parameters()}
    order = [name_of[id(p)] for g in opt.param_groups for p in g["params"]]   # state index -> name
    groups = [{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in g.items()}
              for g in osd["param_groups"]]
    st0 = osd["state"][0]
    layout = {
        "file_name": os.path.basename(path),
        "top_level_keys": sorted(ck.keys()), "epoch": ck["epoch"],
        "cfg_type": type(ck["cfg"]).__name__,
        "named_parameters": names,
        "model_state_keys": list(ck["model_state"].keys()),
        "shapes": {n: list(shapes[n]) for n in names},
        "param_groups": groups,
        "state_entry_keys": sorted(st0.keys()),
        "state_step": {"type": type(st0["step"]).__name__, "value": float(st0["step"]),
                       "dtype": str(getattr(st0["step"], "dtype", ""))},
        "scaler_state": {k: (v if not torch.is_tensor(v) else float(v)) for k, v in ck["scaler_state"].items()},
        "optimizer_order": order,
        "digests": {order[j]: {"index": j, "param": P.digest(ck["model_state"][order[j]]),
                               "exp_avg": P.digest(osd["state"][j]["exp_avg"]),
                               "exp_avg_sq": P.digest(osd["state"][j]["exp_avg_sq"])}
                    for j in range(0, len(order), 57)},
    }
    json.dump(layout, open(os.path.join(out_dir, "layout.json"), "w"), indent=1, sort_keys=True)
    manifest["layout"] = {"file": "layout.json", "params": len(names),
                          "groups": [len(g["params"]) for g in groups]}
    print("layout", manifest["layout"], layout["state_step"], layout["top_level_keys"])


def _grads_of(total, named):
    gs = torch.autograd.grad(total, [v for _, v in named], retain_graph=True, allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(named, gs)}


def run_consistency_case(out_dir, manifest):
    """The reference's own frame-clip consistency term (losses.py:127-136) on the tiny model:
    clip forward with grad, the no-grad single-frame pass of tools/train_net.py:105-110, then
    `VideoImageLoss._consistency_loss` called directly with the l1 / l2 keys present in its
    lambda table (as released `get_lambdas_dict` never produces them).
    Records the values, the total CE + LAMBDA_CON * consistency, and its gradients."""
    torch.manual_seed(0)
    cfg, model, shapes, sd = build_reference(4, 64, extra=("TRAIN.FORWARD_VIDEO_FRAMES", True))
    spec = R.make_spec(num_frames=4, crop=64, drop_path_rate=0.0, dropout_rate=0.0)
    from slowfast.models import losses as L
    from slowfast.utils import misc
    B = 2
    x, y = P.frames(B, 4, 64), P.labels(B)
    model.train()
    logits, extra = model([x], {})
    with torch.no_grad():
        _fp, fextra = model([x.transpose(1, 2).flatten(0, 1).unsqueeze(2)], {})
    lf = L.VideoImageLoss(cfg)
    lf._is_vid = True
    lf.train()
    lam = float(cfg.SVIT.LAMBDA_CON)
    named = list(model.named_parameters())
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lg, ex = R.forward(p, spec, x, training=True)
    with torch.no_grad():
        _, fex = R.forward({k: v.detach() for k, v in p.items()}, spec,
                           x.transpose(1, 2).flatten(0, 1).unsqueeze(2), training=True)
    arrays, digests, agree, info = {}, {}, {}, {"lambda_con": lam, "batch": B}
    store(arrays, digests, "frames_obj_desc", fextra["obj_desc"], sample=True)
    agree["frames_obj_desc"] = maxabs(fex["obj_desc"], fextra["obj_desc"])
    for mode in ("l1", "l2"):
        key = "video_image_desc_%s_loss" % mode
        lf._lambda = dict(misc.get_lambdas_dict(cfg))
        lf._lambda[key] = lam
        parts = lf._consistency_loss(extra, fextra)
        assert list(parts) == [key]
        ce = lf.ce_loss(logits, y)
        total = ce + lam * parts[key]
        grads = _grads_of(total, named)
        con = R.consistency_loss(ex, fex, mode)
        tot = R.video_loss(lg, y) + lam * con
        rgr = _grads_of(tot, list(p.items()))
        agree[mode + "_value"] = maxabs(con, parts[key])
        agree[mode + "_total"] = maxabs(tot, total)
        gmax = max(float(g.abs().max()) for g in grads.values())
        agree[mode + "_grad_rel_worst"] = max(relerr(rgr[k], grads[k], floor=1e-5 * gmax) for k in grads)
        info[mode] = {"value": float(parts[key]), "ce": float(ce), "total": float(total)}
        for k, g in grads.items():
            store(arrays, digests, "%s:grad:%s" % (mode, k), g, sample=True)
    np.savez_compressed(os.path.join(out_dir, "consistency.npz"), **arrays)
    manifest["consistency"] = {"num_frames": 4, "crop": 64, "restatement_vs_reference_maxabs": agree,
                               "info": info, "digests": digests}
    print("consistency worst disagreement:", max(agree.values()), info, flush=True)


HOT_CFG_SECTIONS = ("DATA", "MODEL", "MVIT", "SVIT", "SOLVER", "TRAIN", "TEST", "IMAGE_TRAIN")


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def run_cfg_case(out_dir, manifest):
    """The reference's `get_cfg()` + configs/ssv2.yaml, hot-path sections only, as plain JSON
    (defaults.py:12-1173; ssv2.yaml).  Also what two yacs-style coercion traps of that yaml
    become after the merge: `PATCH_KERNEL: (3, 7, 7)` and `BASE_LR: 2e-4` are YAML strings."""
    ref_shim.install()
    from slowfast.config.defaults import get_cfg
    import yaml
    cfg = get_cfg()
    defaults = {k: _plain(cfg[k]) for k in HOT_CFG_SECTIONS}
    top_defaults = {k: _plain(v) for k, v in cfg.items() if not isinstance(v, dict)}
    cfg.merge_from_file(ref_shim.REFERENCE_ROOT + "/configs/ssv2.yaml")
    merged = {k: _plain(cfg[k]) for k in HOT_CFG_SECTIONS}
    top = {k: _plain(v) for k, v in cfg.items() if not isinstance(v, dict)}
    raw = yaml.safe_load(open(ref_shim.REFERENCE_ROOT + "/configs/ssv2.yaml"))
    traps = {"MVIT.PATCH_KERNEL": {"yaml_type": type(raw["MVIT"]["PATCH_KERNEL"]).__name__,
                                   "yaml_value": raw["MVIT"]["PATCH_KERNEL"],
                                   "merged": _plain(cfg.MVIT.PATCH_KERNEL),
                                   "merged_type": type(cfg.MVIT.PATCH_KERNEL).__name__},
             "SOLVER.BASE_LR": {"yaml_type": type(raw["SOLVER"]["BASE_LR"]).__name__,
                                "yaml_value": raw["SOLVER"]["BASE_LR"],
                                "merged": cfg.SOLVER.BASE_LR,
                                "merged_type": type(cfg.SOLVER.BASE_LR).__name__}}
    out = {"sections": list(HOT_CFG_SECTIONS), "defaults": defaults, "top_level_defaults": top_defaults,
           "merged": merged, "top_level": top, "traps": traps,
           "yaml_sections": {k: sorted(v) for k, v in raw.items() if isinstance(v, dict)}}
    json.dump(out, open(os.path.join(out_dir, "cfg.json"), "w"), indent=1, sort_keys=True)
    manifest["cfg"] = {"file": "cfg.json", "sections": list(HOT_CFG_SECTIONS), "traps": traps}
    print("cfg", traps, flush=True)


def run_module_cases(out_dir, manifest):
    """Known-answer tests of the reference's own modules, taken from the tiny
    model's submodules with closed-form inputs and upstream gradients: PatchEmbed
    (stem_helper.py:290-320), MultiScaleAttention (attention.py:331-466) of blocks 1 and 15,
    MultiScaleBlock (attention.py:557-571) of blocks 0 (plain), 1 and 3 (dim change + q pooling),
    15, and SViTHead train / eval (video_model_builder.py:507-551).  Outputs, input gradients and
    every parameter gradient are stored (digest + strided sample) and compared with the
    restatement (oracle/svit_ref.py)."""
    torch.manual_seed(0)
    cfg, model, shapes, sd = build_reference(4, 64)
    spec = R.make_spec(num_frames=4, crop=64, drop_path_rate=0.0, dropout_rate=0.0)
    model.train()
    arrays, digests, agree, meta = {}, {}, {}, {}
    B, O = 2, 4
    n_obj = 4 * O

    def pgrads(mod, prefix):
        return {prefix + k: v.grad for k, v in mod.named_parameters()}

    def rest(names):
        return {k: sd[k].clone().requires_grad_(True) for k in names}

    # ---- PatchEmbed -------------------------------------------------------------------------
    x = P.frames(B, 4, 64, tag="kat")
    model.zero_grad()
    tok, shp = model.patch_embed(x)
    g = P.tensor("kat:patch:g", tuple(tok.shape), 1.0)
    tok.backward(g)
    pr = rest(["patch_embed.proj.weight", "patch_embed.proj.bias"])
    y = torch.nn.functional.conv3d(x, pr["patch_embed.proj.weight"], pr["patch_embed.proj.bias"],
                                   stride=spec.patch_stride, padding=spec.patch_pad)
    tk = y.flatten(2).transpose(1, 2)
    tk.backward(g)
    agree["patch_out"] = maxabs(tk, tok)
    store(arrays, digests, "patch:out", tok, sample=True)
    for k, v in pgrads(model.patch_embed, "patch_embed.").items():
        agree["patch_d" + k.split(".")[-1]] = relerr(pr[k].grad, v)
        store(arrays, digests, "patch:grad:" + k, v, sample=True)
    meta["patch"] = {"conv_shape": [int(v) for v in shp], "batch": B}

    # shapes entering each block of the tiny model
    thw_in, dims = {}, {}
    thw = (2, 16, 16)
    for b in spec.blocks:
        thw_in[b.index], dims[b.index] = thw, b.dim_in
        thw = (thw[0], R.pooled_size(thw[1], b.stride_q[1]), R.pooled_size(thw[2], b.stride_q[2]))

    # ---- MultiScaleAttention / MultiScaleBlock -------------------------------------------------
    for kind, idxs in (("attn", (1, 15)), ("block", (0, 1, 3, 15))):
        for i in idxs:
            b = spec.blocks[i]
            t, hh, ww = thw_in[i]
            N = 1 + t * hh * ww + n_obj
            tag = "%s%d" % (kind, i)
            xin = P.tensor("kat:%s:x" % tag, (B, N, dims[i]), 1.0)
            xr = xin.clone().requires_grad_(True)
            model.zero_grad()
            mod = model.blocks[i].attn if kind == "attn" else model.blocks[i]
            out, thw_o = mod(xr, list(thw_in[i]))
            g = P.tensor("kat:%s:g" % tag, tuple(out.shape), 1.0)
            out.backward(g)
            pre = "blocks.%d." % i + ("attn." if kind == "attn" else "")
            names = [k for k in sd if k.startswith(pre)]
            pr = rest(names)
            xo = xin.clone().requires_grad_(True)
            if kind == "attn":
                o2, thw2 = R.attention(pr, pre, b, xo, thw_in[i], n_obj)
            else:
                o2, thw2 = R.block(pr, b, xo, thw_in[i], n_obj)
            o2.backward(g)
            assert tuple(thw2) == tuple(thw_o)
            agree[tag + "_out"] = relerr(o2, out)
            agree[tag + "_dx"] = relerr(xo.grad, xr.grad)
            gmax = max(float(v.grad.abs().max()) for v in mod.parameters())
            worst = 0.0
            for k, v in pgrads(mod, pre).items():
                worst = max(worst, relerr(pr[k].grad, v, floor=1e-5 * gmax))
                store(arrays, digests, "%s:grad:%s" % (tag, k), v, sample=True)
            agree[tag + "_dparam_worst"] = worst
            store(arrays, digests, tag + ":out", out, sample=True)
            store(arrays, digests, tag + ":dx", xr.grad, sample=True)
            meta[tag] = {"thw_in": list(thw_in[i]), "thw_out": [int(v) for v in thw_o], "N": N,
                         "dim_in": dims[i], "out_shape": list(out.shape), "n_obj": n_obj}

    # ---- SViTHead ----------------------------------------------------------------------------
    feat = P.tensor("kat:head:x", (B, 1 + n_obj, 768), 1.0)
    hn = [k for k in sd if k.startswith("head.")]
    for training in (True, False):
        tag = "head_train" if training else "head_eval"
        model.head.train(training)
        fr = feat.clone().requires_grad_(True)
        model.zero_grad()
        lg, ex = model.head(fr)
        pr = rest(hn)
        fo = feat.clone().requires_grad_(True)
        lg2, ex2 = R.head(pr, spec, fo, 4, training)
        outs = {"logits": (lg, lg2), "pred_bboxes": (ex["pred_bboxes"], ex2["pred_bboxes"]),
                "pred_contact_state": (ex["pred_contact_state"], ex2["pred_contact_state"]),
                "obj_desc": (ex["obj_desc"], ex2["obj_desc"])}
        tot = tot2 = 0.0
        for k, (a, a2) in outs.items():
            agree["%s_%s" % (tag, k)] = maxabs(a2, a)
            store(arrays, digests, "%s:%s" % (tag, k), a, sample=True)
            gk = P.tensor("kat:head:g:" + k, tuple(a.shape), 1.0)
            tot = tot + (a * gk).sum()
            tot2 = tot2 + (a2 * gk).sum()
        if training:
            tot.backward()
            tot2.backward()
            agree[tag + "_dx"] = relerr(fo.grad, fr.grad)
            store(arrays, digests, tag + ":dx", fr.grad, sample=True)
            for k, v in pgrads(model.head, "head.").items():
                agree[tag + "_d" + k] = relerr(pr[k].grad, v)
                store(arrays, digests, "%s:grad:%s" % (tag, k), v, sample=True)
    model.head.train(True)
    np.savez_compressed(os.path.join(out_dir, "modules.npz"), **arrays)
    manifest["modules"] = {"num_frames": 4, "crop": 64, "batch": B, "meta": meta,
                           "restatement_vs_reference_maxabs": agree, "digests": digests}
    print("modules worst disagreement:", max(agree.values()),
          {k: "%.2e" % v for k, v in agree.items() if v > 1e-4}, flush=True)


def _cos(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def run_yardstick_cases(manifest):
    """Round 6: the noise floor of the gradient tolerances.  The reference's backward once in fp32
    and once under ref_shim.autocast_emulation() (bf16 operands / results of every matrix op); per-tensor cosine of the
    two gradients = what a correct bf16-GEMM implementation scores.  Same inputs as the parity tests that use them:
    tiny (the smoke), tiny_frames (T' = 1 path), tiny_image (tests/test_model_gpu.py::test_image_rank_step_parity_vs_oracle),
    c2 (the headline clip)."""
    ref_shim.install()
    from slowfast.models import losses as L
    from slowfast.utils import misc
    out = {}
    for name, nf, crop, batch, kind in (("tiny", 4, 64, 2, "video"), ("tiny_frames", 4, 64, 3, "frames"),
                                        ("tiny_image", 4, 64, 3, "image"), ("c2", 16, 224, 1, "video")):
        torch.manual_seed(0)
        cfg, model, shapes, sd = build_reference(nf, crop)
        model.train()
        x = P.frames(batch, nf if kind == "video" else 1, crop)
        y = P.labels(batch)
        meta = P.haog_meta(batch)
        lam = misc.get_lambdas_dict(cfg)
        lf = L.VideoImageLoss(cfg)
        lf._is_vid = False
        lf.train()

        def step():
            model.zero_grad()
            logits, extra = model([x], {})
            if kind == "image":
                parts = lf(logits, extra, None, meta)
                loss = sum(lam[k] * v for k, v in parts.items())
            else:
                loss = torch.nn.functional.cross_entropy(logits, y)
            loss.backward()
            return logits.detach().clone(), {k: v.grad.detach().clone() for k, v in model.named_parameters()}
        lg32, g32 = step()
        mode = ref_shim.autocast_emulation()
        with mode:
            lg16, g16 = step()
        gmax = max(float(v.abs().max()) for v in g32.values())
        cos = {k: round(_cos(g16[k], g32[k]), 6) for k in g32 if float(g32[k].abs().max()) >= 1e-4 * gmax}
        num = sum(float((g16[k].double() * g32[k].double()).sum()) for k in g32)
        da = sum(float((g16[k].double() ** 2).sum()) for k in g32) ** 0.5
        db = sum(float((g32[k].double() ** 2).sum()) for k in g32) ** 0.5
        worst = min(cos, key=cos.get)
        out[name] = {"num_frames": nf, "crop": crop, "batch": batch, "kind": kind,
                     "rounded_ops": type(mode).calls, "logits_maxabs": float((lg16 - lg32).abs().max()),
                     "logits_cos": _cos(lg16, lg32), "grad_cos_global": num / (da * db),
                     "grad_cos_worst": [worst, cos[worst]], "autocast_emulation_cos": cos}
        type(mode).calls = 0
        rel = sorted((v, k) for k, v in cos.items() if "rel_pos" in k)[:4]
        print("yardstick", name, "worst", worst, cos[worst], "global %.5f" % out[name]["grad_cos_global"],
              "logits maxabs %.4f" % out[name]["logits_maxabs"], "rel-pos worst:", rel, flush=True)
    manifest["yardstick"] = {
        "note": "per-tensor cosine (bf16-emulated reference backward vs the fp32 reference backward): the noise floor "
                "of a correct bf16-GEMM implementation; oracle/ref_shim.py::autocast_emulation", "cases": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    mpath = os.path.join(args.out, "manifest.json")
    manifest = {"cases": {}}
    if args.only and os.path.exists(mpath):
        manifest = json.load(open(mpath))
    manifest.update({
        "generator": "oracle/gen_golden.py", "torch": torch.__version__,
        "note": "arrays/digests produced by running the unmodified reference on CPU fp32 "
                "with the closed-form tensors of oracle/procedural.py"})
    want = set(args.only.split(",")) if args.only else None

    def on(n):
        return want is None or n in want
    if on("ops"):
        run_op_cases(args.out, manifest)
    if on("tiny"):
        run_model_case("tiny", 4, 64, 2, args.out, manifest, eval_too=True)
    if on("tiny_drop"):
        run_model_case("tiny_drop", 4, 64, 4, args.out, manifest, drop=True)
    if on("tiny_odd"):
        run_model_case("tiny_odd", 4, 88, 2, args.out, manifest)
    if on("tiny_frames"):
        run_model_case("tiny_frames", 4, 64, 3, args.out, manifest, frames_path=True)
    if on("loss_optim"):
        run_loss_optim_cases(args.out, manifest)
    if on("meter"):
        run_meter_case(args.out, manifest)
    if on("layout"):
        run_layout_case(args.out, manifest)
    if on("c1"):
        run_model_case("c1", 8, 224, 1, args.out, manifest, eval_too=True)
    if on("c2_fwd"):
        run_model_case("c2_fwd", 16, 224, 1, args.out, manifest, backward=False)
    if on("c2"):       # the config the headline metric is quoted on: forward AND backward
        run_model_case("c2", 16, 224, 1, args.out, manifest)
    if on("c2_drop"):  # the bench regime: B = 2, DropPath 0.4 + head dropout 0.5, the reference's own masks
        run_model_case("c2_drop", 16, 224, 2, args.out, manifest, drop=True)
    if on("consistency"):
        run_consistency_case(args.out, manifest)
    if on("cfg"):
        run_cfg_case(args.out, manifest)
    if on("modules"):
        run_module_cases(args.out, manifest)
    # round 4: the reference itself on the remaining BASELINE.json shapes (forward; the restatement is compared with
    # it on each) -- C4 long clip, C5 312^2 crop in eval mode (probabilities), and the bench workload's batch of 8
    if on("c4_fwd"):
        run_model_case("c4_fwd", 32, 224, 1, args.out, manifest, backward=False)
    if on("c5_eval"):
        run_model_case("c5_eval", 16, 312, 1, args.out, manifest, backward=False, eval_too=True)
    if on("c2_b8_fwd"):
        run_model_case("c2_b8_fwd", 16, 224, 8, args.out, manifest, backward=False)
    if on("c2_frames"):
        run_model_case("c2_frames", 16, 224, 2, args.out, manifest, backward=False,
                       frames_path=True)
    if on("yardstick"):
        run_yardstick_cases(manifest)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    print("wrote", mpath)


if __name__ == "__main__":
    main()
