"""Whole-model parity of the HIP path (through build_model -> C ABI) against the CPU oracle and
the golden vectors taken from the reference.  Needs a real MI355X."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import procedural as P
from oracle import svit_ref as R
from tests import smoke_impl as S


@pytest.mark.parametrize("frames,crop,batch,frames_path", [
    (4, 64, 2, False),     # tiny
    (4, 88, 2, False),     # odd sizes: rel-pos tables interpolated in blocks >= 3
    (4, 64, 3, True),      # T=1 frames path of a 4-frame model (rel_pos_t interpolated to 1 row)
    (8, 224, 1, False),    # BASELINE config 1 (C1)
    (16, 224, 1, False),   # BASELINE config 2/3: the shape the headline metric is quoted on
    (16, 224, 2, False),   # ... and with two clips (cross-clip reductions of every weight gradient)
    (32, 224, 1, False),   # BASELINE config 4: long clip (T' = 16, J = 30 / 44 bias columns)
    (16, 312, 1, False),   # BASELINE config 5: 312^2 crop (78x78 patches, interpolated tables)
])
def test_step_parity_vs_oracle(frames, crop, batch, frames_path):
    res = S.compare_step(frames, crop, batch, frames_path)
    print({k: v for k, v in res.items() if k != "grad_cos_per_tensor"})
    # per-tensor gradient bar: the manifest's bf16 yardstick where the case has one (tiny, the T = 1 path, the headline clip),
    # the flat 0.99 elsewhere (tests/smoke_impl.py::check)
    S.check(res, S.yardstick_for(frames, crop, batch, frames_path))


def test_image_rank_step_parity_vs_oracle():
    """a still-image batch [B,3,1,S,S] with the HAOG box / contact losses
    (fused svit_haog_loss) -- losses and every parameter gradient against the fp32 oracle, whose
    image loss is itself pinned by the reference's numbers (tests/test_oracle_golden.py)."""
    res = S.compare_step(4, 64, 3, image=True)
    print({k: v for k, v in res.items() if k != "grad_cos_per_tensor"})
    S.check(res, S.yardstick_for(4, 64, 3, image=True))
    assert res["loss_rel"] < 2e-2, res
    assert all(v < 2e-2 for v in res["parts_abs"].values()), res


@pytest.mark.parametrize("name", ["tiny", "c1", "tiny_odd", "c2_fwd", "tiny_frames", "c2_frames",
                                  "c4_fwd", "c5_eval", "c2_b8_fwd"])
def test_against_reference_golden(name, manifest, golden_dir):
    """Same closed-form weights/inputs as oracle/gen_golden.py fed to the reference: its own
    logits / box heads, incl. the headline 16x224^2 clip (c2_fwd), the single-frame path and
    (round 4) the reference's own outputs at the remaining BASELINE.json shapes: C4 32x224^2
    (c4_fwd), C5 16x312^2 in eval mode = the probabilities the 3-crop test folds (c5_eval), and the
    bench workload's batch of 8 clips of 16x224^2 (c2_b8_fwd)."""
    case = manifest["cases"][name]
    cfg, model, spec, sd = S.build_hip_model(case["num_frames"], case["crop"])
    x = P.frames(case["batch"], 1 if case.get("frames_path") else case["num_frames"], case["crop"])
    arrays = np.load(os.path.join(golden_dir, name + ".npz"))
    logits, extra = model([x.cuda()], {})
    ref = torch.from_numpy(arrays["logits"])
    assert float((logits.detach().cpu() - ref).abs().max()) <= S.TOL["logits_maxabs"]
    assert S.cosine(logits.detach(), ref) >= S.TOL["logits_cos"]
    d = case["digests"]["obj_desc"]
    got = P.digest(extra["obj_desc"].detach().cpu())
    assert abs(got["l2"] - d["l2"]) / d["l2"] < 2e-2
    np.testing.assert_allclose(got["head"], d["head"], atol=0.08)
    for key, atol in (("pred_bboxes", 3e-2), ("pred_contact_state", 5e-2)):
        got_t = extra[key].detach().cpu()
        if key in arrays:
            np.testing.assert_allclose(got_t.numpy(), arrays[key], atol=atol)
        else:      # more than 1024 elements (the batch of 8): the fixture holds the reference's digest
            dk, gk = case["digests"][key], P.digest(got_t)
            assert abs(gk["l2"] - dk["l2"]) / dk["l2"] < 2e-2, key
            np.testing.assert_allclose(gk["head"], dk["head"], atol=atol)
            np.testing.assert_allclose(gk["strided"], dk["strided"], atol=atol)
    # eval mode: probabilities
    if "eval_probs" in arrays:
        model.eval()
        with torch.no_grad():
            probs, ex = model([x.cuda()], {})
        np.testing.assert_allclose(probs.cpu().numpy(), arrays["eval_probs"], atol=4e-3)
        np.testing.assert_allclose(ex["pred_bboxes"].cpu().numpy(), arrays["eval_pred_bboxes"], atol=3e-2)


def _grad_vs_golden(named_grads, digests, arrays, prefix):
    """HIP gradients against the REFERENCE's own (digest l2 + strided sample or full tensor):
    per-tensor norm ratio in [0.97, 1.03] and cosine >= 0.99 on what the fixture holds."""
    gmax = max(digests[prefix + k]["l2"] for k in named_grads)
    worst_cos, worst_full, worst_ratio = (1.0, ""), (1.0, ""), (0.0, "")
    for k, g in named_grads.items():
        d = digests[prefix + k]
        g = g.detach().float().cpu()
        if d["l2"] < 1e-4 * gmax:      # mathematically ~zero (e.g. norm_k.bias)
            assert float(g.norm()) < 2e-2 * gmax, (k, float(g.norm()), gmax)
            continue
        ratio = float(g.double().norm()) / d["l2"]
        if abs(ratio - 1) > abs(worst_ratio[0] - 1) or worst_ratio[1] == "":
            worst_ratio = (ratio, k)
        if prefix + k in arrays:       # stored whole (small tensors; since round 4 every rel-pos table)
            c = S.cosine(g, torch.from_numpy(arrays[prefix + k]).reshape(g.shape))
            if c < worst_full[0]:
                worst_full = (c, k)
        else:
            c = S.cosine(P.sample_of(g), torch.from_numpy(arrays["sample:" + prefix + k]))
            if c < worst_cos[0]:
                worst_cos = (c, k)
    print("worst cosine: sampled", worst_cos, "full tensors", worst_full, "worst norm ratio", worst_ratio)
    # 0.985 on the 256-element strided samples (the fixture cannot hold 34 M gradients; a cosine
    # estimated from 256 elements scatters by ~ (1 - c) / sqrt(128) around the full tensor's); the
    # full-tensor criterion (cosine >= 0.99, scale within 3 % + 4 sigma) is tests/smoke_impl.py's
    assert worst_cos[0] >= 0.985, worst_cos
    assert worst_full[0] >= 0.99, worst_full       # every element compared: the stated per-tensor tolerance
    assert 0.97 <= worst_ratio[0] <= 1.03, worst_ratio


def test_headline_config_gradients_vs_reference_golden(manifest, golden_dir):
    """16x224^2 (the config the metric is quoted on): logits, loss and ALL 405 parameter gradients
    of the HIP step against the reference's own forward+backward (tests/golden/c2.npz, fp32 CPU)."""
    case = manifest["cases"]["c2"]
    cfg, model, spec, sd = S.build_hip_model(16, 224)
    x, y = P.frames(case["batch"], 16, 224), P.labels(case["batch"])
    arrays = np.load(os.path.join(golden_dir, "c2.npz"))
    logits, extra = model([x.cuda()], {})
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    model.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    ref = torch.from_numpy(arrays["logits"])
    assert float((logits.detach().cpu() - ref).abs().max()) <= S.TOL["logits_maxabs"]
    assert abs(float(loss.detach()) - float(arrays["loss"])) < 2e-2
    _grad_vs_golden({k: v.grad for k, v in model.named_parameters()}, case["digests"], arrays, "grad:")


@pytest.mark.parametrize("mode", ["l1", "l2"])
def test_consistency_loss_vs_reference_golden(mode, manifest, golden_dir):
    """clip forward + the no-grad single-frame pass (train_net.py:105-110) +
    the frame-clip consistency term, all through the HIP path, against the numbers the reference's
    own `VideoImageLoss._consistency_loss` (losses.py:127-136) produced: frames obj_desc, the
    loss value, CE + LAMBDA_CON * consistency, and its 405 gradients."""
    from svit_amd import losses
    c = manifest["consistency"]
    arrays = np.load(os.path.join(golden_dir, "consistency.npz"))
    cfg, model, spec, sd = S.build_hip_model(4, 64)
    cfg.TRAIN.FORWARD_VIDEO_FRAMES = True
    cfg.SVIT.CONSISTENCY = mode
    B = c["info"]["batch"]
    x, y = P.frames(B, 4, 64).cuda(), P.labels(B).cuda()
    logits, extra = model([x], {})
    with torch.no_grad():
        fp, fe = model([x.transpose(1, 2).flatten(0, 1).unsqueeze(2)], {})
    d = c["digests"]["frames_obj_desc"]
    got = P.digest(fe["obj_desc"].detach().cpu())
    assert abs(got["l2"] - d["l2"]) / d["l2"] < 2e-2
    assert S.cosine(P.sample_of(fe["obj_desc"].detach().cpu()),
                    torch.from_numpy(arrays["sample:frames_obj_desc"])) >= 0.999
    extra = dict(extra)
    extra["frames_output"] = {"preds": fp, "extra_preds": fe}
    fn = losses.VideoImageLoss(cfg, is_video_rank=True)
    parts = fn(logits, extra, y, {})
    key = "video_image_desc_%s_loss" % mode
    assert set(parts) == {"loss_ce", key}
    info = c["info"][mode]
    assert abs(float(parts[key].detach()) - info["value"]) / info["value"] < 2e-2
    total = fn.total(parts)
    assert abs(float(total.detach()) - info["total"]) < 3e-2
    model.zero_grad(set_to_none=True)
    total.backward()
    torch.cuda.synchronize()
    _grad_vs_golden({k: v.grad for k, v in model.named_parameters()}, c["digests"], arrays,
                    mode + ":grad:")


def test_droppath_and_dropout_masks(manifest, golden_dir):
    """DropPath per-sample factors and head dropout mask taken from the reference's own run."""
    case = manifest["cases"]["tiny_drop"]
    cfg, model, spec, sd = S.build_hip_model(4, 64, drop=True)
    arrays = np.load(os.path.join(golden_dir, "tiny_drop.npz"))
    x = P.frames(case["batch"], 4, 64)
    ds = []
    for i in range(16):
        if "dp_attn_%d" % i in arrays:
            ds.append((torch.from_numpy(arrays["dp_attn_%d" % i]).cuda().contiguous(),
                       torch.from_numpy(arrays["dp_mlp_%d" % i]).cuda().contiguous()))
        else:
            ds.append(None)
    keep = torch.from_numpy(arrays["dropout_keep"]).cuda()
    logits, extra = model([x.cuda()], {}, drop_scales=ds, dropout_keep=keep)
    ref = torch.from_numpy(arrays["logits"])
    assert float((logits.detach().cpu() - ref).abs().max()) <= 2 * S.TOL["logits_maxabs"]
    assert S.cosine(logits.detach(), ref) >= S.TOL["logits_cos"]


def test_bench_regime_droppath_vs_reference_golden(manifest, golden_dir):
    """The regime bench.py times -- 16x224^2, B = 2, DropPath 0.4 and head dropout 0.5 ON -- against
    the reference's own forward + backward with the reference's own masks replayed
    (tests/golden/c2_drop.npz: per-block per-sample DropPath factors, the dropout keep mask, logits,
    loss, norms + strided samples of all 405 gradients)."""
    case = manifest["cases"]["c2_drop"]
    cfg, model, spec, sd = S.build_hip_model(16, 224, drop=True)
    arrays = np.load(os.path.join(golden_dir, "c2_drop.npz"))
    x, y = P.frames(case["batch"], 16, 224), P.labels(case["batch"])
    ds = []
    for i in range(16):
        if "dp_attn_%d" % i in arrays:
            ds.append((torch.from_numpy(arrays["dp_attn_%d" % i]).cuda().contiguous(),
                       torch.from_numpy(arrays["dp_mlp_%d" % i]).cuda().contiguous()))
        else:
            ds.append(None)
    keep = torch.from_numpy(arrays["dropout_keep"]).cuda()
    logits, extra = model([x.cuda()], {}, drop_scales=ds, dropout_keep=keep)
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    model.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    ref = torch.from_numpy(arrays["logits"])
    assert float((logits.detach().cpu() - ref).abs().max()) <= 2 * S.TOL["logits_maxabs"]
    assert S.cosine(logits.detach(), ref) >= S.TOL["logits_cos"]
    assert abs(float(loss.detach()) - float(arrays["loss"])) < 3e-2
    _grad_vs_golden({k: v.grad for k, v in model.named_parameters()}, case["digests"], arrays, "grad:")


def test_state_dict_layout_and_interface():
    cfg, model, spec, sd = S.build_hip_model(16, 224)
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ours == R.param_shapes(spec) and len(ours) == 405
    assert sum(p.numel() for p in model.parameters()) == 34373560
    assert hasattr(model, "no_weight_decay") and model.no_weight_decay() == []
    assert [list(r) for r in cfg.MVIT.POOL_KV_STRIDE][:2] == [[0, 1, 8, 8], [1, 1, 4, 4]]
    with pytest.raises(Exception):
        model([torch.zeros(1, 3, 8, 224, 224).cuda()], {})   # wrong clip length for this cfg


def test_zero_decay_pos_cls_layout_and_state_shape_check():
    """MVIT.ZERO_DECAY_POS_CLS (the default of defaults.py; ssv2.yaml turns it off): the three
    top-level tokens move to the zero-decay group of the flat layout AND of the optimizer-state
    order, and a state whose order does not fit is refused instead of broadcast."""
    from svit_amd import config, optim
    from svit_amd.model import build_model
    cfg = config.ssv2_cfg(4, 64)
    cfg.MVIT.ZERO_DECAY_POS_CLS = True
    cfg.MVIT.DROPPATH_RATE = 0.0
    model = build_model(cfg)
    flat = model.flat
    for n in ("cls_token", "object_queries", "pos_embed_temporal"):
        assert flat.slots[n][0] >= flat.n_decay, n
    assert flat.slots["blocks.0.attn.rel_pos_h"][0] < flat.n_decay
    opt = optim.construct_optimizer(model, cfg)
    dec, rest = opt._order()
    assert rest[:3] == ["cls_token", "pos_embed_temporal", "object_queries"]
    x, y = P.frames(2, 4, 64), P.labels(2)
    logits, _ = model([x.cuda()], {})
    torch.nn.functional.cross_entropy(logits, y.cuda()).backward()
    opt.step()
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    # the ssv2-recipe order (tokens decayed) must not load into this layout silently
    cfg2 = config.ssv2_cfg(4, 64)
    other = optim.construct_optimizer(build_model(cfg2), cfg2)
    other.step_count = 1
    with pytest.raises(ValueError):
        opt.load_state_dict(other.state_dict())


def test_fused_optimizer_step_matches_oracle():
    """clip_grad_norm_(1.0) + AdamW on the flat buffers vs the oracle's restatement."""
    from svit_amd import optim
    cfg, model, spec, sd = S.build_hip_model(4, 64)
    x, y = P.frames(2, 4, 64), P.labels(2)
    opt = optim.construct_optimizer(model, cfg)
    optim.set_lr(opt, 1.5e-4)
    logits, _ = model([x.cuda()], {})
    loss = torch.nn.functional.cross_entropy(logits, y.cuda())
    opt.zero_grad()
    loss.backward()
    grads = {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters()}
    before = {k: v.detach().cpu().clone() for k, v in model.named_parameters()}
    opt.step()
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in before.items()}
    norm = R.clip_and_adamw_step(before, grads, st, 1.5e-4, 1)
    assert abs(opt.grad_norm() - norm) / norm < 1e-4
    for k, v in model.named_parameters():
        if float(grads[k].abs().max()) < 1e-7:
            continue
        assert float((v.detach().cpu() - before[k]).abs().max()) < 3e-6, k


def test_full_size_properties_of_the_bench_workload():
    """Size-independent checks at the BASELINE workload (8 clips of 16x224^2, where the fp32 oracle
    is too slow to run): eval probabilities are distributions, box heads are in range, the
    forward is bit-reproducible, clips do not interact (batch permutation), and a few replayed
    training steps on a fixed batch reduce the loss."""
    from svit_amd import optim
    from svit_amd.graph import GraphedTrainStep
    cfg, model, spec, sd = S.build_hip_model(16, 224, train=False)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 3, 16, 224, 224, generator=g).cuda()
    y = torch.randint(0, 174, (8,), generator=g).cuda()
    with torch.no_grad():
        p1, e1 = model([x], {})
        p2, e2 = model([x], {})
        perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4]).cuda()
        p3, e3 = model([x[perm]], {})
    assert torch.equal(p1, p2) and torch.equal(e1["obj_desc"], e2["obj_desc"])          # deterministic
    assert float((p1.sum(1) - 1).abs().max()) < 1e-5 and float(p1.min()) >= 0            # softmax rows
    assert float((e1["pred_contact_state"].sum(-1) - 1).abs().max()) < 1e-5
    assert float(e1["pred_bboxes"].min()) >= 0 and float(e1["pred_bboxes"].max()) <= 1
    assert tuple(e1["obj_desc"].shape) == (8, 16, 4, 768)
    assert float((p3 - p1[perm]).abs().max()) < 1e-6                                       # no cross-clip coupling
    assert float((e3["obj_desc"] - e1["obj_desc"][perm]).abs().max()) < 1e-3
    model.train()
    opt = optim.construct_optimizer(model, cfg)
    optim.set_lr(opt, 1e-4)
    step = GraphedTrainStep(model, lambda p, e, l: torch.nn.functional.cross_entropy(p, l), [x], y)
    losses = []
    for _ in range(6):
        loss, _ = step([x], y)
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and losses[-1] < losses[0] - 0.05, losses


def test_grouped_second_stage_reductions_equal_ungrouped():
    """Engine.RED_GROUP (second-stage reductions of 4 blocks in one launch, rotating scratch slots)
    against RED_GROUP = 1 (one launch per block), with the gradient-ready hook firing INSIDE a group (ranks 2 and 6 of
    a depth-16 model = after blocks 14 and 10, both mid-group) so that `ready()` flushes a part-filled queue: flat.grad
    must be bit-equal (deterministic mode: no fp32 atomics meet anywhere)."""
    from svit_amd.engine import Engine
    grads = []
    for rg in (1, 4, 0):
        cfg, model, spec, sd = S.build_hip_model(4, 64)
        model.engine = Engine(model.plan, model.flat, red_group=rg)
        assert model.engine.RED_GROUP == max(1, rg)
        model.engine.deterministic = True
        model.engine.refresh_weights()
        fired = []
        model._grad_ready_hook = fired.append
        model._grad_ready_ranks = {2, 6, model.flat.n_ranks - 1}
        x = P.frames(2, 4, 64).cuda()
        logits, _ = model([x], {})
        model.flat.grad.zero_()
        torch.nn.functional.cross_entropy(logits, P.labels(2).cuda()).backward()
        torch.cuda.synchronize()
        assert fired == list(range(model.flat.n_ranks))
        grads.append(model.flat.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert float(grads[0].abs().max()) > 0


def test_batch_that_outgrows_the_fixed_pooling_scratch():
    """Round 6: the product library has no streaming fallback for the pooling conv backward, so a step whose fused-kernel plan needs
    more partial rows than the engine's fixed scratch region (9 M floats: 8 clips of 16x224^2; 16 clips outgrow it) must get a
    buffer of the plan's size (svit_pool_conv_bwd_workspace, Engine._wgrad_ws) instead of an error.  Here the fixed region is
    shrunk to 1000 floats on the tiny model so that EVERY block takes the grown buffers: the gradient must be bit-equal to the
    default engine's (deterministic mode); `bench.py --batch 16 / 32` is the full-size run (profiles/r06_throughput_vs_batch.txt)."""
    grads = []
    for shrink in (False, True):
        cfg, model, spec, sd = S.build_hip_model(4, 64)
        eng = model.engine
        eng.deterministic = True
        if shrink:
            lo, _ = eng._red_regions["wgrad"]
            eng._red_regions["wgrad"] = (lo, lo + 1000)
        model.flat.grad.zero_()
        logits, _ = model([P.frames(3, 4, 64).cuda()], {})
        torch.nn.functional.cross_entropy(logits, P.labels(3).cuda()).backward()
        torch.cuda.synchronize()
        assert bool(eng._wgrad_big) == shrink and all(n > 1000 for n in eng._wgrad_need.values())
        grads.append(model.flat.grad.clone())
    assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0


def test_pool_conv_bwd_refuses_a_workspace_that_is_too_small():
    """svit_pool_conv_bwd_workspace is exact: a workspace of that many floats runs the fused kernel, one float less is refused
    (SVIT_ERR_SHAPE: the product library has no other conv backward -- a silent fallback used to hide this)."""
    from svit_amd import hip, ops
    B, h, thw, O, strides = 2, 4, (8, 14, 14), 5, (1, 2, 2)
    need = ops.pool_conv_bwd_workspace(B, h, thw, O, strides)
    assert need > 0 and need % (3 * 27 * 96) == 0
    N = 1 + thw[0] * thw[1] * thw[2] + O
    qkv = torch.randn(B, N, 3, h, 96, device="cuda").bfloat16()
    ws_ = [torch.randn(96, 27, device="cuda") * 0.2 for _ in range(3)]
    dpres = [torch.randn(B, h, 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + O, 96, device="cuda").bfloat16() for s in strides]
    dws = [torch.zeros(96, 27, device="cuda") for _ in range(3)]
    ops.pool_conv_bwd_qkv(dpres, ws_, torch.empty_like(qkv), qkv, dws, B, h, thw, O, strides, ws=torch.empty(need, device="cuda"))
    assert hip.load().svit_debug_pool_bwd_path() == 1
    with pytest.raises(hip.SvitHipError):
        ops.pool_conv_bwd_qkv(dpres, ws_, torch.empty_like(qkv), qkv, dws, B, h, thw, O, strides, ws=torch.empty(need - 1, device="cuda"))
