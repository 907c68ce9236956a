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
    # scale test per tensor: s = <got, ref> / <ref, ref> is the factor by which the gradient is
    # reproduced along the reference direction (a DropPath factor applied twice or mean-vs-sum moves
    # s and leaves the cosine at 1).  The part of `got` orthogonal to ref is noise of norm
    # nu = |got| sqrt(1 - cos^2); its projection on the ref direction has standard deviation
    # nu / sqrt(n) for a tensor of n elements, so  |s - 1| <= 3 % + 4 nu / (sqrt(n) |ref|)  is an
    # absolute criterion that follows each tensor's own measured bf16 noise and sample count
    # (a [15, 96] rel-pos table with cosine 0.993 gets 3 % + 1.3 %, a 2-D weight 3 % + ~0).
    scale_excess, scale_name, scale_worst = -1.0, "", (1.0, 0.0)
    per_tensor = {}
    ref_total = sum(float((v.grad.double() ** 2).sum()) for v in p.values() if v.grad is not None) ** 0.5
    num = den_a = den_b = 0.0
    for k, v in p.items():
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = named[k].grad.detach().cpu()
        dot = float((got.double() * ref.double()).sum())
        num += dot
        den_a += float((got.double() ** 2).sum())
        den_b += float((ref.double() ** 2).sum())
        if float(ref.abs().max()) < 1e-4 * gmax:   # mathematically ~zero (e.g. norm_k.bias)
            assert float(got.abs().max()) < 2e-2 * gmax, (k, float(got.abs().max()), gmax)
            continue
        c = cosine(got, ref)
        per_tensor[k] = c
        if c < worst:
            worst, worst_name = c, k
        rn = float(ref.double().norm())
        sc = dot / (rn * rn)
        nu = float(got.double().norm()) * max(0.0, 1.0 - c * c) ** 0.5
        tol = SCALE_TOL + 4.0 * nu / (max(1, ref.numel()) ** 0.5 * rn)
        if abs(sc - 1.0) - tol > scale_excess:
            scale_excess, scale_name, scale_worst = abs(sc - 1.0) - tol, k, (sc, tol)
        if verbose:
            print("%-40s cos %.5f  scale %.4f (tol %.4f)  |ref| %.3e |got| %.3e"
                  % (k, c, sc, tol, float(ref.norm()), float(got.norm())))
    out["grad_cos_worst"] = worst
    out["grad_cos_per_tensor"] = per_tensor
    out["grad_cos_worst_name"] = worst_name
    out["grad_cos_global"] = num / ((den_a ** 0.5) * (den_b ** 0.5) + 1e-30)
    out["grad_scale_excess"] = scale_excess            # <= 0: every tensor's scale is inside its band
    out["grad_scale_worst"] = (scale_name,) + scale_worst
    out["grad_norm_ratio_global"] = (den_a / den_b) ** 0.5
    return out


# stated tolerance of the bf16 HIP path against the fp32 oracle (BASELINE.json north_star;
# SURVEY.md 8(c)): logits max-abs <= 0.05 and cosine >= 0.999; per-tensor grad cosine >= 0.99.
# gradient scale per tensor (s = <got, ref> / <ref, ref>): |s - 1| <= 3 % + 4 sigma, sigma = the
# standard deviation of s that the tensor's own orthogonal (noise) component implies (compare_step);
# norm ratio of the whole gradient within [0.99, 1.01].
SCALE_TOL = 0.03
TOL = {"logits_maxabs": 0.05, "logits_cos": 0.999, "grad_cos": 0.99, "obj_desc_cos": 0.999,
       "grad_scale": SCALE_TOL, "grad_norm_ratio_global": (0.99, 1.01)}


# Per-tensor gradient criterion where the golden manifest holds a YARDSTICK for the case (round 6):
# manifest["yardstick"]["cases"][name]["autocast_emulation_cos"][tensor] = cosine, against the fp32 reference, of the REFERENCE's own
# backward with every matrix op's operands and results rounded to bf16 (oracle/ref_shim.py::autocast_emulation) -- what a correct
# bf16-GEMM implementation scores.  The HIP path must stay within YARD_RATIO x that noise power per tensor (+ an absolute floor for
# tensors both compute almost exactly).  Measured on MI355X (profiles/r06_yardstick.txt): median ratio 0.8-0.9, maximum 2.0 over the
# 405 tensors of the four cases; the rel-pos tables -- always the worst tensors of a step -- read 0.9938 / 0.9950 / 0.9917 / 0.9893
# against the yardstick's 0.9930 / 0.9951 / 0.9932 / 0.9789 (c2, tiny, tiny_frames, tiny_image).  Cases without a yardstick keep
# the flat per-tensor bar TOL["grad_cos"].
YARD_RATIO, YARD_FLOOR = 3.0, 2e-4
YARD_CASES = {(4, 64, 2, False, False): "tiny", (4, 64, 3, True, False): "tiny_frames", (4, 64, 3, False, True): "tiny_image",
              (16, 224, 1, False, False): "c2"}


def yardstick_for(num_frames, crop, batch, frames_path=False, image=False):
    """the manifest's yardstick case for these step arguments, or None"""
    import json
    import os
    name = YARD_CASES.get((num_frames, crop, batch, bool(frames_path), bool(image)))
    if name is None:
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest.json")
    with open(path) as f:
        return json.load(f)["yardstick"]["cases"][name]


def check(res, yardstick=None):
    assert res["logits_maxabs"] <= TOL["logits_maxabs"], res
    assert res["logits_cos"] >= TOL["logits_cos"], res
    assert res["obj_desc_cos"] >= TOL["obj_desc_cos"], res
    if yardstick is None:
        assert res["grad_cos_worst"] >= TOL["grad_cos"], res
    else:
        yc = yardstick["autocast_emulation_cos"]
        bad = [(k, c, yc[k]) for k, c in res["grad_cos_per_tensor"].items()
               if k in yc and (1.0 - c) > YARD_RATIO * (1.0 - yc[k]) + YARD_FLOOR]
        assert not bad, ("gradient tensors noisier than %.1f x the bf16 yardstick" % YARD_RATIO, bad[:8])
        assert res["grad_cos_worst"] >= 0.97, res        # (a backstop no tensor of any case comes near)
    assert res["grad_cos_global"] >= 0.995, res
    assert res["grad_scale_excess"] <= 0.0, res
    lo, hi = TOL["grad_norm_ratio_global"]
    assert lo <= res["grad_norm_ratio_global"] <= hi, res


def run():
    res = compare_step(4, 64, 2)
    print("smoke:", {k: v for k, v in res.items() if k != "grad_cos_per_tensor"})
    check(res, yardstick_for(4, 64, 2))
