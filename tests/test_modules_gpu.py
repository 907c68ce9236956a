"""Module-level known-answer tests of the HIP path against the REFERENCE's own modules
(tests/golden/modules.npz, recorded by oracle/gen_golden.py::run_module_cases from the tiny
model's PatchEmbed / MultiScaleBlock / SViTHead with closed-form inputs and upstream gradients;
SURVEY.md 8(c) G1).  Everything goes through the C ABI (svit_amd.ops / engine).  Needs an MI355X.

bf16 tolerance: output cosine >= 0.999 and norm ratio within 2 %; gradients cosine >= 0.985 on
the stored 256-element samples (>= 0.99 on full tensors) and norm ratio within 3 %.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import procedural as P
from tests import smoke_impl as S


def _check(name, got, digests, arrays, cos_min, ratio_tol):
    got = got.detach().float().cpu()
    d = digests[name]
    assert got.numel() == d["n"], (name, got.shape, d["n"])
    ratio = float(got.double().norm()) / d["l2"]
    if name in arrays:
        c = S.cosine(got, torch.from_numpy(arrays[name]))
    else:
        c = S.cosine(P.sample_of(got), torch.from_numpy(arrays["sample:" + name]))
    assert c >= cos_min, (name, c)
    assert abs(ratio - 1) <= ratio_tol, (name, ratio)
    return c, ratio


@pytest.fixture(scope="module")
def tiny():
    cfg, model, spec, sd = S.build_hip_model(4, 64)
    return cfg, model, spec, sd


def test_patch_embed_vs_reference(tiny, manifest, golden_dir):
    """stem_helper.py:290-320: conv3d (3,7,7)/(2,4,4)/(1,3,3) as im2col + NT GEMM; weight / bias
    gradients by the TN GEMM."""
    from svit_amd import hip, ops
    cfg, model, spec, sd = tiny
    m = manifest["modules"]
    a = np.load(os.path.join(golden_dir, "modules.npz"))
    eng, flat = model.engine, model.flat
    eng.refresh_weights()
    x = P.frames(2, 4, 64, tag="kat").cuda()
    cols, (To, Ho, Wo) = ops.im2col_patch(x)
    assert [2, 96, To, Ho, Wo] == m["meta"]["patch"]["conv_shape"]
    tok = ops.gemm_nt(cols, eng.patch_w16, flat.p("patch_embed.proj.bias"), hip.EPI_F32)
    _check("patch:out", tok, m["digests"], a, 0.999, 0.02)
    g = P.tensor("kat:patch:g", (2, To * Ho * Wo, 96), 1.0).cuda()
    dw = torch.zeros(96, 441, device="cuda")
    db = torch.zeros(96, device="cuda")
    ops.gemm_tn(g.view(-1, 96).bfloat16().contiguous(), cols, dw, dbias=db)
    torch.cuda.synchronize()
    _check("patch:grad:patch_embed.proj.weight", dw, m["digests"], a, 0.99, 0.03)
    _check("patch:grad:patch_embed.proj.bias", db, m["digests"], a, 0.99, 0.03)


@pytest.mark.parametrize("index", [0, 1, 3, 15])
def test_multiscale_block_vs_reference(index, tiny, manifest, golden_dir):
    """attention.py:557-571 (+ :331-466): one MultiScaleBlock forward and backward through the
    engine's launch schedule -- plain block (0), dim change + q pooling (1, 3), last block (15)."""
    cfg, model, spec, sd = tiny
    m = manifest["modules"]
    a = np.load(os.path.join(golden_dir, "modules.npz"))
    tag = "block%d" % index
    meta = m["meta"][tag]
    eng, flat = model.engine, model.flat
    blk = eng.plan.blocks[index]
    eng.refresh_weights()
    flat.grad.zero_()
    x = P.tensor("kat:%s:x" % tag, (2, meta["N"], meta["dim_in"]), 1.0).cuda()
    with torch.no_grad():
        out, thw, sv = eng._block_fwd(blk, x, tuple(meta["thw_in"]), meta["n_obj"], None, True)
        assert list(thw) == meta["thw_out"] and list(out.shape) == meta["out_shape"]
        _check(tag + ":out", out, m["digests"], a, 0.999, 0.02)
        g = P.tensor("kat:%s:g" % tag, tuple(out.shape), 1.0).cuda()
        dx, _ = eng._block_bwd(blk, sv, g, g.bfloat16(), meta["n_obj"], None)
        eng._flush_tn()
        eng._join()
    torch.cuda.synchronize()
    _check(tag + ":dx", dx, m["digests"], a, 0.99, 0.03)
    pre = "blocks.%d." % index
    names = [k for k in sd if k.startswith(pre)]
    gmax = max(m["digests"]["%s:grad:%s" % (tag, k)]["l2"] for k in names)
    worst = (1.0, "")
    for k in names:
        key = "%s:grad:%s" % (tag, k)
        if m["digests"][key]["l2"] < 1e-4 * gmax:
            assert float(flat.g(k).norm()) < 2e-2 * gmax, k
            continue
        c, r = _check(key, flat.g(k), m["digests"], a, 0.985, 0.03)
        if c < worst[0]:
            worst = (c, k)
    print(tag, "worst parameter-gradient cosine", worst)


@pytest.mark.parametrize("training", [True, False])
def test_head_vs_reference(training, tiny, manifest, golden_dir):
    """video_model_builder.py:507-551: logits / probabilities, box and contact heads, obj_desc."""
    cfg, model, spec, sd = tiny
    m = manifest["modules"]
    a = np.load(os.path.join(golden_dir, "modules.npz"))
    tag = "head_train" if training else "head_eval"
    head = model.head
    head.train(training)
    try:
        feat = P.tensor("kat:head:x", (2, 17, 768), 1.0).cuda().requires_grad_(True)
        lg, ex = head(feat, T=4)
        outs = {"logits": lg, "pred_bboxes": ex["pred_bboxes"],
                "pred_contact_state": ex["pred_contact_state"], "obj_desc": ex["obj_desc"]}
        tot = 0.0
        for k, v in outs.items():
            ref = a["%s:%s" % (tag, k)] if "%s:%s" % (tag, k) in a else None
            if ref is not None:
                np.testing.assert_allclose(v.detach().cpu().numpy(), ref, atol=2e-4, rtol=1e-4)
            else:
                _check("%s:%s" % (tag, k), v, m["digests"], a, 0.99999, 1e-3)
            tot = tot + (v * P.tensor("kat:head:g:" + k, tuple(v.shape), 1.0).cuda()).sum()
        if training:
            for p in head.parameters():
                p.grad = None
            tot.backward()
            _check(tag + ":dx", feat.grad, m["digests"], a, 0.9999, 5e-3)
            for k, p in head.named_parameters():
                _check("%s:grad:head.%s" % (tag, k), p.grad, m["digests"], a, 0.9999, 5e-3)
                p.grad = None
    finally:
        head.train(True)
