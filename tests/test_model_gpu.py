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
    (32, 224, 1, False),   # BASELINE config 4: long clip (T' = 16, J = 30 / 44 bias columns)
    (16, 312, 1, False),   # BASELINE config 5: 312^2 crop (78x78 patches, interpolated tables)
])
def test_step_parity_vs_oracle(frames, crop, batch, frames_path):
    res = S.compare_step(frames, crop, batch, frames_path)
    print(res)
    S.check(res)


def test_image_rank_step_parity_vs_oracle():
    """a still-image batch [B,3,1,S,S] with the HAOG box / contact losses
    (fused svit_haog_loss) -- losses and every parameter gradient against the fp32 oracle, whose
    image loss is itself pinned by the reference's numbers (tests/test_oracle_golden.py)."""
    res = S.compare_step(4, 64, 3, image=True)
    print(res)
    S.check(res)
    assert res["loss_rel"] < 2e-2, res
    assert all(v < 2e-2 for v in res["parts_abs"].values()), res


@pytest.mark.parametrize("name", ["tiny", "c1", "tiny_odd", "c2_fwd", "tiny_frames", "c2_frames"])
def test_against_reference_golden(name, manifest, golden_dir):
    """Same closed-form weights/inputs as oracle/gen_golden.py fed to the reference: its own
    logits / box heads, incl. the headline 16x224^2 clip (c2_fwd) and the single-frame path."""
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
    np.testing.assert_allclose(extra["pred_bboxes"].detach().cpu().numpy(), arrays["pred_bboxes"], atol=3e-2)
    np.testing.assert_allclose(extra["pred_contact_state"].detach().cpu().numpy(),
                               arrays["pred_contact_state"], atol=5e-2)
    # eval mode: probabilities
    if "eval_probs" in arrays:
        model.eval()
        with torch.no_grad():
            probs, ex = model([x.cuda()], {})
        np.testing.assert_allclose(probs.cpu().numpy(), arrays["eval_probs"], atol=4e-3)
        np.testing.assert_allclose(ex["pred_bboxes"].cpu().numpy(), arrays["eval_pred_bboxes"], atol=3e-2)


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


def test_state_dict_layout_and_interface():
    cfg, model, spec, sd = S.build_hip_model(16, 224)
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ours == R.param_shapes(spec) and len(ours) == 405
    assert sum(p.numel() for p in model.parameters()) == 34373560
    assert hasattr(model, "no_weight_decay") and model.no_weight_decay() == []
    assert [list(r) for r in cfg.MVIT.POOL_KV_STRIDE][:2] == [[0, 1, 8, 8], [1, 1, 4, 4]]
    with pytest.raises(Exception):
        model([torch.zeros(1, 3, 8, 224, 224).cuda()], {})   # wrong clip length for this cfg


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
