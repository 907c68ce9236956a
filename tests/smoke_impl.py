"""One tiny SViT forward+backward through the HIP path on cuda:0, checked against the CPU
oracle (used by __graft_entry__.smoke() and by tests/test_model_gpu.py)."""
import torch

from oracle import procedural as P
from oracle import svit_ref as R


def build_hip_model(num_frames, crop, drop=False, train=True):
    from svit_amd import config
    from svit_amd.model import build_model
    cfg = config.ssv2_cfg(num_frames=num_frames, crop=crop)
    if not drop:
        cfg.MVIT.DROPPATH_RATE = 0.0
        cfg.MODEL.DROPOUT_RATE = 0.0
    model = build_model(cfg)
    spec = R.make_spec(num_frames=num_frames, crop=crop,
                       drop_path_rate=cfg.MVIT.DROPPATH_RATE, dropout_rate=cfg.MODEL.DROPOUT_RATE)
    sd = P.state_dict(R.param_shapes(spec))
    model.load_state_dict(sd, strict=True)
    model.train(train)
    return cfg, model, spec, sd


def cosine(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def compare_step(num_frames=4, crop=64, batch=2, frames_path=False, verbose=False, image=False):
    """-> dict of parity numbers (HIP bf16 path vs fp32 oracle) for one video CE step, or
    (image=True) one image-rank step: still images, HAOG box / contact losses."""
    cfg, model, spec, sd = build_hip_model(num_frames, crop)
    x = P.frames(batch, 1 if (frames_path or image) else num_frames, crop)
    y = P.labels(batch)
    logits, extra = model([x.cuda()], {})
    parts = {}
    if image:
        from svit_amd import losses
        meta = P.haog_meta(batch)
        fn = losses.VideoImageLoss(cfg, is_video_rank=False)
        parts = fn(logits, extra, None, {k: v.cuda() for k, v in meta.items()})
        loss = fn.total(parts)
    else:
        loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    model.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lg, ex = R.forward(p, spec, x, training=True)
    if image:
        ls, ref_parts = R.image_loss(ex, meta, R.loss_weights(cfg.SVIT.LAMBDA_NODES,
                                                              cfg.SVIT.LAMBDA_EDGES))
    else:
        ls = R.video_loss(lg, y)
    ls.backward()
    out = {"logits_maxabs": float((logits.detach().cpu() - lg.detach()).abs().max()),
           "logits_cos": cosine(logits.detach(), lg.detach()),
           "loss_abs": abs(float(loss.detach()) - float(ls.detach())),
           "obj_desc_cos": cosine(extra["obj_desc"].detach(), ex["obj_desc"].detach()),
           "obj_desc_maxabs": float((extra["obj_desc"].detach().cpu() - ex["obj_desc"].detach()).abs().max())}
    if image:
        out["loss_rel"] = out["loss_abs"] / abs(float(ls.detach()))
        out["parts_abs"] = {k: abs(float(parts[k].detach()) - float(v.detach()))
                            for k, v in ref_parts.items()}
    named = dict(model.named_parameters())
    gmax = max(float(v.grad.abs().max()) for v in p.values() if v.grad is not None)
    worst, worst_name = 1.0, ""
    ratio_lo, ratio_hi, ratio_name = 1.0, 1.0, ""          # tensors carrying >= 1 % of the gradient norm
    small_lo, small_hi, small_name = 1.0, 1.0, ""          # the rest (few-sample rel-pos tables, ...)
    ref_total = sum(float((v.grad.double() ** 2).sum()) for v in p.values() if v.grad is not None) ** 0.5
    num = den_a = den_b = 0.0
    for k, v in p.items():
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = named[k].grad.detach().cpu()
        num += float((got.double() * ref.double()).sum())
        den_a += float((got.double() ** 2).sum())
        den_b += float((ref.double() ** 2).sum())
        if float(ref.abs().max()) < 1e-4 * gmax:   # mathematically ~zero (e.g. norm_k.bias)
            assert float(got.abs().max()) < 2e-2 * gmax, (k, float(got.abs().max()), gmax)
            continue
        c = cosine(got, ref)
        if c < worst:
            worst, worst_name = c, k
        # a scale error (a DropPath factor applied twice, mean-vs-sum) leaves the cosine at 1:
        # the norms must agree too
        r = float(got.double().norm() / ref.double().norm())
        if float(ref.double().norm()) >= 1e-2 * ref_total:
            if r < ratio_lo or r > ratio_hi:
                ratio_name = k
            ratio_lo, ratio_hi = min(ratio_lo, r), max(ratio_hi, r)
        else:
            if r < small_lo or r > small_hi:
                small_name = k
            small_lo, small_hi = min(small_lo, r), max(small_hi, r)
        if verbose:
            print("%-40s cos %.5f  |ref| %.3e |got| %.3e" % (k, c, float(ref.norm()), float(got.norm())))
    out["grad_cos_worst"] = worst
    out["grad_cos_worst_name"] = worst_name
    out["grad_cos_global"] = num / ((den_a ** 0.5) * (den_b ** 0.5) + 1e-30)
    out["grad_norm_ratio_min"], out["grad_norm_ratio_max"] = ratio_lo, ratio_hi
    out["grad_norm_ratio_name"] = ratio_name
    out["grad_norm_ratio_small"] = (small_lo, small_hi, small_name)
    out["grad_norm_ratio_global"] = (den_a / den_b) ** 0.5
    return out


# stated tolerance of the bf16 HIP path against the fp32 oracle (BASELINE.json north_star;
# SURVEY.md 8(c)): logits max-abs <= 0.05 and cosine >= 0.999; per-tensor grad cosine >= 0.99.
# gradient norm ratio |got| / |ref|: [0.97, 1.03] per tensor that carries >= 1 % of the global
# gradient norm, [0.94, 1.06] for the smaller ones (measured: the outliers are rel-pos tables
# with <= 0.6 % of the norm in the tiny T' = 1 case, scattered on both sides of 1 -- bf16 noise
# on few samples, not a scale error), [0.99, 1.01] for the whole gradient.
TOL = {"logits_maxabs": 0.05, "logits_cos": 0.999, "grad_cos": 0.99, "obj_desc_cos": 0.999,
       "grad_norm_ratio": (0.97, 1.03), "grad_norm_ratio_small": (0.94, 1.06),
       "grad_norm_ratio_global": (0.99, 1.01)}


def check(res):
    assert res["logits_maxabs"] <= TOL["logits_maxabs"], res
    assert res["logits_cos"] >= TOL["logits_cos"], res
    assert res["obj_desc_cos"] >= TOL["obj_desc_cos"], res
    assert res["grad_cos_worst"] >= TOL["grad_cos"], res
    assert res["grad_cos_global"] >= 0.995, res
    lo, hi = TOL["grad_norm_ratio"]
    assert lo <= res["grad_norm_ratio_min"] and res["grad_norm_ratio_max"] <= hi, res
    lo, hi = TOL["grad_norm_ratio_small"]
    assert lo <= res["grad_norm_ratio_small"][0] and res["grad_norm_ratio_small"][1] <= hi, res
    lo, hi = TOL["grad_norm_ratio_global"]
    assert lo <= res["grad_norm_ratio_global"] <= hi, res


def run():
    res = compare_step(4, 64, 2)
    print("smoke:", res)
    check(res)
