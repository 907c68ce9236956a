"""The CPU restatement (oracle/svit_ref.py) replayed against the golden vectors that
oracle/gen_golden.py took from the unmodified reference.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import procedural as P
from oracle import svit_ref as R

torch.set_num_threads(min(8, os.cpu_count() or 1))


def close_digest(t, d, rtol=2e-4, atol=1e-6):
    got = P.digest(t)
    assert got["n"] == d["n"]
    scale = max(d["absmean"], atol)
    assert abs(got["absmean"] - d["absmean"]) <= rtol * scale + atol
    assert abs(got["l2"] - d["l2"]) <= rtol * max(d["l2"], atol) + atol
    assert abs(got["proj"] - d["proj"]) <= rtol * max(d["l2"], atol) + atol
    np.testing.assert_allclose(got["head"], d["head"], rtol=0, atol=rtol * 25 * scale + atol)
    np.testing.assert_allclose(got["strided"], d["strided"], rtol=0, atol=rtol * 25 * scale + atol)


def close_sample(t, ref, rel=3e-3):
    """strided sample of a large tensor (oracle/procedural.py::sample_of) against the golden one"""
    got = P.sample_of(t).numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=rel * float(np.abs(ref).max()) + 1e-7)


def run_case(case, name, golden_dir, backward):
    spec = R.make_spec(num_frames=case["num_frames"], crop=case["crop"],
                       drop_path_rate=0.4 if case["drop"] else 0.0,
                       dropout_rate=0.5 if case["drop"] else 0.0)
    sd = P.state_dict(R.param_shapes(spec))
    p = {k: v.clone().requires_grad_(backward) for k, v in sd.items()}
    B = case["batch"]
    x = P.frames(B, 1 if case["frames_path"] else case["num_frames"], case["crop"])
    y = P.labels(B)
    arrays = np.load(os.path.join(golden_dir, name + ".npz"))
    drop_scales = dropout_keep = None
    if case["drop"]:
        drop_scales = []
        for i in range(spec.depth):
            if "dp_attn_%d" % i in arrays:
                drop_scales.append((torch.from_numpy(arrays["dp_attn_%d" % i]),
                                    torch.from_numpy(arrays["dp_mlp_%d" % i])))
            else:
                drop_scales.append(None)
        dropout_keep = torch.from_numpy(arrays["dropout_keep"])
    taps = {}
    logits, extra = R.forward(p, spec, x, training=True, drop_scales=drop_scales,
                              dropout_keep=dropout_keep, taps=taps)
    loss = R.video_loss(logits, y)
    if backward:
        loss.backward()
    return spec, p, logits, extra, loss, taps, arrays


@pytest.mark.parametrize("name", ["tiny", "tiny_drop", "tiny_odd", "tiny_frames", "c1",
                                  "c2_fwd", "c2_frames", "c2", "c2_drop",
                                  "c4_fwd", "c5_eval", "c2_b8_fwd"])   # round 4: the reference at C4 / C5 / B = 8
def test_model_case(name, manifest, golden_dir):
    case = manifest["cases"][name]
    for k, v in case["restatement_vs_reference_maxabs"].items():
        assert v < 5e-3, (k, v)  # recorded when the fixtures were made (pins the restatement)
    spec, p, logits, extra, loss, taps, arrays = run_case(case, name, golden_dir, case["backward"])
    dg = case["digests"]
    np.testing.assert_allclose(logits.detach().numpy(), arrays["logits"], atol=2e-4, rtol=1e-4)
    assert abs(float(loss.detach()) - float(arrays["loss"])) < 1e-4
    for k in ("obj_desc", "pred_bboxes", "pred_contact_state"):
        close_digest(extra[k], dg[k])
    for i in range(spec.depth):
        close_digest(taps["block%d" % i], dg["block%d" % i])
    if case["backward"]:
        gmax = max(dg["grad:" + k]["l2"] for k in p)
        for k, v in p.items():
            d = dg["grad:" + k]
            g = v.grad if v.grad is not None else torch.zeros_like(v)
            if d["l2"] < 1e-5 * gmax:  # mathematically-zero grads (e.g. attn.norm_k.bias)
                assert float(g.norm()) < 1e-4 * gmax
                continue
            close_digest(g, d, rtol=1e-3)
            if "grad:" + k in arrays:
                # rel-pos tables (stored whole since round 4) are scatter-added over up to 25 k query rows in
                # fp32: the two summation orders differ by up to 3.9e-3 of the tensor's scale (manifest:
                # grad_rel_worst, pinned < 5e-3 above); everything else small agrees to 2e-3
                tol = 5e-3 if "rel_pos_" in k else 2e-3
                np.testing.assert_allclose(g.numpy(), arrays["grad:" + k], rtol=0,
                                           atol=tol * float(np.abs(arrays["grad:" + k]).max()) + 1e-7)
            if "sample:grad:" + k in arrays:      # strided sample of a large gradient
                close_sample(g, arrays["sample:grad:" + k])


@pytest.mark.parametrize("name", ["tiny", "c1", "c5_eval"])
def test_eval_mode(name, manifest, golden_dir):
    case = manifest["cases"][name]
    spec = R.make_spec(num_frames=case["num_frames"], crop=case["crop"], drop_path_rate=0.0,
                       dropout_rate=0.0)
    sd = P.state_dict(R.param_shapes(spec))
    x = P.frames(case["batch"], case["num_frames"], case["crop"])
    arrays = np.load(os.path.join(golden_dir, name + ".npz"))
    with torch.no_grad():
        probs, extra = R.forward(sd, spec, x, training=False)
    np.testing.assert_allclose(probs.numpy(), arrays["eval_probs"], atol=1e-5)
    np.testing.assert_allclose(probs.sum(1).numpy(), 1.0, atol=1e-5)
    np.testing.assert_allclose(extra["pred_bboxes"].numpy(), arrays["eval_pred_bboxes"], atol=1e-5)
    np.testing.assert_allclose(extra["pred_contact_state"].numpy(),
                               arrays["eval_pred_contact_state"], atol=1e-5)


def test_op_kats(golden_dir):
    a = np.load(os.path.join(golden_dir, "ops.npz"))
    for s in (1, 2, 4, 8):
        T, H, W, h, O = 2, 8, 8, 2, 3
        x = P.tensor("kat:pool:x:%d" % s, (2, h, 1 + T * H * W + O, 96), 1.0).requires_grad_(True)
        w = P.tensor("kat:pool:w:%d" % s, (96, 1, 3, 3, 3), 0.3).requires_grad_(True)
        nw = P.tensor("kat:pool:nw", (96,), 0.2, 1.0)
        nb = P.tensor("kat:pool:nb", (96,), 0.1)
        out, _ = R.pool_tokens(x, (T, H, W), (1, s, s), w, nw, nb, O)
        out.backward(P.tensor("kat:pool:g:%d" % s, tuple(out.shape), 1.0))
        np.testing.assert_allclose(out.detach().numpy(), a["pool_s%d_out" % s], atol=2e-5)
        np.testing.assert_allclose(x.grad.numpy(), a["pool_s%d_dx" % s], atol=5e-5)
        np.testing.assert_allclose(w.grad.numpy(), a["pool_s%d_dw" % s], atol=5e-4)
        np.testing.assert_allclose(R.object_gain(w.detach(), (1, s, s)).numpy(),
                                   a["pool_s%d_gain" % s], atol=1e-6)
    x = P.tensor("kat:skip:x", (2, 1 + 2 * 8 * 8 + 3, 96), 1.0)
    np.testing.assert_array_equal(R.maxpool_skip(x, (2, 8, 8), (1, 2, 2), 3).numpy(), a["skip_out"])
    for tag, q_thw, k_thw, rows_sp, rows_t in (
            ("same", (2, 4, 4), (2, 4, 4), 7, 3), ("kvpool", (2, 8, 8), (2, 2, 2), 15, 3),
            ("interp", (2, 5, 5), (2, 3, 3), 7, 3), ("t1", (1, 4, 4), (1, 2, 2), 7, 5)):
        Lq = q_thw[0] * q_thw[1] * q_thw[2]
        q = P.tensor("kat:rel:q:" + tag, (2, 2, 1 + Lq + 3, 96), 1.0)
        rh = P.tensor("kat:rel:h:" + tag, (rows_sp, 96), 0.3)
        rw = P.tensor("kat:rel:w:" + tag, (rows_sp, 96), 0.3)
        rt = P.tensor("kat:rel:t:" + tag, (rows_t, 96), 0.3)
        bias = R.rel_pos_bias(q, q_thw, k_thw, rh, rw, rt)
        np.testing.assert_allclose(bias.numpy(), a["rel_%s_bias" % tag], atol=2e-5)


def test_losses_lr_optimizer(manifest):
    lo = manifest["loss_optim"]
    info = lo["info"]
    for k, v in lo["restatement_vs_reference_maxabs"].items():
        assert v < 5e-3, (k, v)
    spec = R.make_spec(num_frames=4, crop=64, drop_path_rate=0.0, dropout_rate=0.0)
    sd = P.state_dict(R.param_shapes(spec))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = P.frames(3, 1, 64, tag="img")
    meta = P.haog_meta(3)
    _, extra = R.forward(p, spec, x, training=True)
    total, parts = R.image_loss(extra, meta, R.loss_weights(3.7, 0.3))
    total.backward()
    assert abs(float(total.detach()) - info["image_total"]) < 1e-4
    for k, v in parts.items():
        assert abs(float(v.detach()) - info["image_" + k]) < 1e-5
    lam = dict(info["lambdas"])
    # as released the consistency weight sits under a key no loss ever produces
    assert lam.pop("video_image_boxes_l1_loss") == 1.5
    assert lam == {k: pytest.approx(v) for k, v in R.loss_weights(3.7, 0.3).items()}
    for k, d in info["image_grad_digest"].items():
        g = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        close_digest(g, d, rtol=1e-3)  # image rank: head.projection / cls grads are exact zeros
    for e, v in zip(info["lr_epochs"], info["lr_values"]):
        assert R.cosine_lr(e) == pytest.approx(v, rel=1e-12)
    shapes = R.param_shapes(spec)
    assert sum(R.weight_decay_of(k, s) == 0.0 for k, s in shapes.items()) == info["n_wd_zero"] == 234
    assert sum(R.weight_decay_of(k, s) > 0.0 for k, s in shapes.items()) == info["n_wd"] == 171
    pw = {k: v.detach().clone() for k, v in p.items()}
    gr = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pw.items()}
    norm = R.clip_and_adamw_step(pw, gr, st, info["adamw_lr"], 1)
    assert norm == pytest.approx(info["grad_norm"], rel=1e-4)
    for k, d in info["adamw_param_digest"].items():
        if float(gr[k].abs().max()) < 1e-6:  # Adam amplifies rounding noise on ~zero grads
            continue
        close_digest(pw[k], d, rtol=2e-4, atol=2e-6)


def test_shape_tables():
    """token counts and rel-pos rows for the five BASELINE configs."""
    s = R.make_spec(16, 224)
    assert [(b.dim_in, b.dim_out, b.heads) for b in s.blocks][:4] == [
        (96, 96, 1), (96, 192, 2), (192, 192, 2), (192, 384, 4)]
    assert [b.stride_kv for b in s.blocks][:4] == [(1, 8, 8), (1, 4, 4), (1, 4, 4), (1, 2, 2)]
    assert [b.rel_sp_rows for b in s.blocks] == [111, 55, 55] + [27] * 12 + [13]
    assert all(b.rel_t_rows == 15 for b in s.blocks)
    shapes = R.param_shapes(s)
    assert len(shapes) == 405 and sum(int(np.prod(v)) for v in shapes.values()) == 34373560
    s8 = R.make_spec(8, 224)
    assert sum(int(np.prod(v)) for v in R.param_shapes(s8).values()) == 34360504
    s5 = R.make_spec(16, 312)
    assert [b.rel_sp_rows for b in s5.blocks][3] == 37 and s5.blocks[15].rel_sp_rows == 17


def test_meter_restatement_vs_reference_golden(manifest, golden_dir):
    """oracle/meter_ref.py (TestMeter, topks_correct, uniform_crop offsets, view table) against
    the numbers the reference's own classes produced (oracle/gen_golden.py::run_meter_case), and
    the host-side copies in svit_amd/evaluate.py."""
    from oracle import meter_ref
    from svit_amd import config, evaluate
    g = np.load(os.path.join(golden_dir, "meter.npz"))
    m = manifest["meter"]
    num_clips = m["ensemble_views"] * m["spatial_crops"]
    for order in ("ordered", "shuffled"):
        perm = g["perm_" + order]
        for method in ("sum", "max"):
            r = meter_ref.TestMeterRef(m["videos"], num_clips, m["classes"], method)
            for a in range(0, len(perm), m["batch"]):
                ids = perm[a:a + m["batch"]]
                r.update_stats(g["probs"][ids], g["labels_v"][ids // num_clips], ids)
            key = "%s_%s" % (order, method)
            assert np.array_equal(r.video_preds, g[key + "_video_preds"])          # bit-exact
            assert np.array_equal(r.clip_count, g[key + "_clip_count"])
            assert np.array_equal(r.video_labels, g[key + "_video_labels"])
            stats, correct = r.finalize_metrics((1, 5))
            assert correct == list(g[key + "_topk_correct"])
            assert stats["top1_acc"] == m["results"][key]["top1_acc"]
            assert stats["top5_acc"] == m["results"][key]["top5_acc"]
    for h, w, size, sidx, y0, x0 in g["crop_offsets"]:
        assert meter_ref.uniform_crop_offsets(int(h), int(w), int(size), int(sidx)) == (y0, x0)
        assert evaluate.uniform_crop_offsets(int(h), int(w), int(size), int(sidx)) == (y0, x0)
    cfg = config.ssv2_cfg(16, 224)
    assert evaluate.unique_views(cfg) == (3, 10)
    views = meter_ref.test_views(2, 10, 3)
    assert len(views) == 60 and views[31] == (1, 1) and {s for _, s in views} == {0, 1, 2}
    with pytest.raises(AssertionError):
        bad = meter_ref.TestMeterRef(2, 3, 4)
        bad.update_stats(np.ones((2, 4), np.float32), [1, 2], [0, 1])


def test_consistency_loss_vs_reference_golden(manifest, golden_dir):
    """oracle.consistency_loss (+ the T=1 frames pass that feeds it) against the numbers the
    reference's own `VideoImageLoss._consistency_loss` produced (losses.py:127-136), values and
    the gradients of CE + LAMBDA_CON * consistency."""
    c = manifest["consistency"]
    for k, v in c["restatement_vs_reference_maxabs"].items():
        assert v < 5e-3, (k, v)
    a = np.load(os.path.join(golden_dir, "consistency.npz"))
    spec = R.make_spec(num_frames=4, crop=64, drop_path_rate=0.0, dropout_rate=0.0)
    sd = P.state_dict(R.param_shapes(spec))
    B, lam = c["info"]["batch"], c["info"]["lambda_con"]
    x, y = P.frames(B, 4, 64), P.labels(B)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits, extra = R.forward(p, spec, x, training=True)
    with torch.no_grad():
        _, fextra = R.forward(sd, spec, x.transpose(1, 2).flatten(0, 1).unsqueeze(2), training=True)
    assert tuple(fextra["obj_desc"].shape) == (B * 4, 1, 4, 768)
    close_digest(fextra["obj_desc"], c["digests"]["frames_obj_desc"])
    for mode in ("l1", "l2"):
        con = R.consistency_loss(extra, fextra, mode)
        assert abs(float(con.detach()) - c["info"][mode]["value"]) < 2e-5
        total = R.video_loss(logits, y) + lam * con
        assert abs(float(total.detach()) - c["info"][mode]["total"]) < 1e-4
        names = list(p)
        grads = torch.autograd.grad(total, [p[k] for k in names], retain_graph=True, allow_unused=True)
        dg = c["digests"]
        gmax = max(dg["%s:grad:%s" % (mode, k)]["l2"] for k in names)
        for k, g in zip(names, grads):
            d = dg["%s:grad:%s" % (mode, k)]
            g = g if g is not None else torch.zeros_like(p[k])
            if d["l2"] < 1e-5 * gmax:
                assert float(g.norm()) < 1e-4 * gmax
                continue
            close_digest(g, d, rtol=1e-3)


def _module_inputs(meta, tag, dim):
    return P.tensor("kat:%s:x" % tag, (2, meta["N"], dim), 1.0)


def test_module_kats_vs_reference_golden(manifest, golden_dir):
    """PatchEmbed, MultiScaleAttention, MultiScaleBlock and SViTHead of the reference (outputs,
    input gradients, parameter gradients recorded by oracle/gen_golden.py::run_module_cases)
    against the restatement's functions on the same closed-form tensors."""
    m = manifest["modules"]
    for k, v in m["restatement_vs_reference_maxabs"].items():
        assert v < 5e-3, (k, v)
    a = np.load(os.path.join(golden_dir, "modules.npz"))
    dg = m["digests"]
    spec = R.make_spec(num_frames=4, crop=64, drop_path_rate=0.0, dropout_rate=0.0)
    sd = P.state_dict(R.param_shapes(spec))

    def check(key, t, rtol=1e-3):
        close_digest(t, dg[key], rtol=rtol)
        if "sample:" + key in a:
            close_sample(t, a["sample:" + key])
        elif key in a:
            np.testing.assert_allclose(t.detach().numpy(), a[key], rtol=0,
                                       atol=3e-3 * float(np.abs(a[key]).max()) + 1e-7)

    # PatchEmbed
    x = P.frames(2, 4, 64, tag="kat")
    w = sd["patch_embed.proj.weight"].clone().requires_grad_(True)
    b = sd["patch_embed.proj.bias"].clone().requires_grad_(True)
    y = torch.nn.functional.conv3d(x, w, b, stride=spec.patch_stride, padding=spec.patch_pad)
    assert list(y.shape) == m["meta"]["patch"]["conv_shape"]
    tok = y.flatten(2).transpose(1, 2)
    tok.backward(P.tensor("kat:patch:g", tuple(tok.shape), 1.0))
    check("patch:out", tok)
    check("patch:grad:patch_embed.proj.weight", w.grad)
    check("patch:grad:patch_embed.proj.bias", b.grad)
    # attention / block
    for tag, meta in m["meta"].items():
        if not (tag.startswith("attn") or tag.startswith("block")):
            continue
        kind = "attn" if tag.startswith("attn") else "block"
        i = int(tag[len(kind):])
        blk = spec.blocks[i]
        pre = "blocks.%d." % i + ("attn." if kind == "attn" else "")
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
        xin = _module_inputs(meta, tag, meta["dim_in"]).requires_grad_(True)
        fn = R.attention if kind == "attn" else R.block
        args = (pr, pre, blk, xin, tuple(meta["thw_in"]), meta["n_obj"]) if kind == "attn" else \
            (pr, blk, xin, tuple(meta["thw_in"]), meta["n_obj"])
        out, thw = fn(*args)
        assert list(thw) == meta["thw_out"] and list(out.shape) == meta["out_shape"]
        out.backward(P.tensor("kat:%s:g" % tag, tuple(out.shape), 1.0))
        check(tag + ":out", out)
        check(tag + ":dx", xin.grad)
        gmax = max(dg["%s:grad:%s" % (tag, k)]["l2"] for k in pr)
        for k, v in pr.items():
            if dg["%s:grad:%s" % (tag, k)]["l2"] < 1e-5 * gmax:
                continue
            check("%s:grad:%s" % (tag, k), v.grad, rtol=2e-3)
    # head
    feat = P.tensor("kat:head:x", (2, 17, 768), 1.0)
    for training in (True, False):
        tag = "head_train" if training else "head_eval"
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("head.")}
        f = feat.clone().requires_grad_(True)
        lg, ex = R.head(pr, spec, f, 4, training)
        outs = {"logits": lg, "pred_bboxes": ex["pred_bboxes"],
                "pred_contact_state": ex["pred_contact_state"], "obj_desc": ex["obj_desc"]}
        tot = 0.0
        for k, v in outs.items():
            check("%s:%s" % (tag, k), v)
            tot = tot + (v * P.tensor("kat:head:g:" + k, tuple(v.shape), 1.0)).sum()
        if training:
            tot.backward()
            check(tag + ":dx", f.grad)
            for k, v in pr.items():
                check("%s:grad:%s" % (tag, k), v.grad)


def test_cfg_schema_pinned_to_reference(manifest, golden_dir, tmp_path):
    """tests/golden/cfg.json = the reference's get_cfg() + configs/ssv2.yaml (hot-path sections,
    defaults.py:12-1173).  `ssv2_cfg()` must equal the merged tree key for key, `get_cfg()` the
    defaults, and `merge_from_file` must coerce the two yacs traps of that yaml the same way."""
    import json
    from svit_amd import config
    ref = json.load(open(os.path.join(golden_dir, "cfg.json")))

    def plain(v):
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v

    def compare(ours, theirs, path):
        missing = [k for k in theirs if k not in ours]
        assert not missing, "%s: keys missing in svit_amd.config: %s" % (path, missing)
        for k, v in theirs.items():
            if isinstance(v, dict):
                compare(ours[k], v, path + "." + k)
            else:
                assert plain(ours[k]) == v, ("%s.%s" % (path, k), plain(ours[k]), v)
                assert type(plain(ours[k])) is type(v), ("%s.%s" % (path, k), type(ours[k]), type(v))

    base = config.get_cfg()
    for sec in ref["sections"]:
        compare(base[sec], ref["defaults"][sec], sec)
    cfg = config.ssv2_cfg(num_gpus=ref["top_level"]["NUM_GPUS"])
    for sec in ref["sections"]:
        compare(cfg[sec], ref["merged"][sec], sec)
        extra = set(cfg[sec]) - set(ref["merged"][sec])
        assert extra <= {"CONSISTENCY"}, (sec, extra)     # the build's one added key
    for k in ("NUM_GPUS", "RNG_SEED", "DDP_FIND_UNUSED_PARAMETERS", "OUTPUT_DIR", "DIST_BACKEND",
              "NUM_SHARDS", "SHARD_ID", "LOG_PERIOD"):
        assert plain(cfg[k]) == ref["top_level"][k], k
    # the two yacs traps, through OUR merge_from_file, written exactly as the yaml writes them
    t = ref["traps"]
    assert t["MVIT.PATCH_KERNEL"]["yaml_type"] == "str" and t["SOLVER.BASE_LR"]["yaml_type"] == "str"
    y = tmp_path / "traps.yaml"
    y.write_text("MVIT:\n  PATCH_KERNEL: %s\nSOLVER:\n  BASE_LR: %s\n"
                 % (t["MVIT.PATCH_KERNEL"]["yaml_value"], t["SOLVER.BASE_LR"]["yaml_value"]))
    c2 = config.get_cfg()
    c2.merge_from_file(str(y))
    assert plain(c2.MVIT.PATCH_KERNEL) == t["MVIT.PATCH_KERNEL"]["merged"]
    assert type(c2.MVIT.PATCH_KERNEL).__name__ == t["MVIT.PATCH_KERNEL"]["merged_type"]
    assert c2.SOLVER.BASE_LR == t["SOLVER.BASE_LR"]["merged"] and isinstance(c2.SOLVER.BASE_LR, float)


@pytest.mark.skipif(not os.path.exists("/root/reference/configs/ssv2.yaml"),
                    reason="reference checkout only exists in the build container")
def test_reference_yaml_merges_to_the_recorded_tree(golden_dir):
    """svit_amd.config.get_cfg().merge_from_file(<the reference's own configs/ssv2.yaml>) gives the
    tree the reference's get_cfg() + merge gave (tests/golden/cfg.json); sections this build does
    not know (AUG, MIXUP, DATA_LOADER, ...) merge without error."""
    import json
    from svit_amd import config
    ref = json.load(open(os.path.join(golden_dir, "cfg.json")))
    cfg = config.get_cfg()
    cfg.merge_from_file("/root/reference/configs/ssv2.yaml")
    want = config.ssv2_cfg(num_gpus=ref["top_level"]["NUM_GPUS"])
    for sec in ref["sections"]:
        for k, v in ref["merged"][sec].items():
            got = cfg[sec][k]
            got = list(got) if isinstance(got, tuple) else got
            assert got == v, (sec, k, got, v)
            assert want[sec][k] == v
    assert cfg.NUM_GPUS == 8 and "AUG" in cfg and "MIXUP" in cfg


def test_bf16_yardstick_is_recorded(manifest):
    """Round 6: the golden manifest carries, for the four step cases the GPU parity tests use it for, the per-tensor cosine of the
    REFERENCE's own backward under bf16 matrix operands against its fp32 backward (oracle/gen_golden.py::run_yardstick_cases) --
    the noise floor tests/smoke_impl.py::check measures the HIP path against.  Here: present, complete, and in the band a bf16
    step lives in (every tensor >= 0.97, the worst tensors are rel-pos tables, the global cosine >= 0.99)."""
    from tests import smoke_impl as S
    y = manifest["yardstick"]["cases"]
    assert set(y) == {"tiny", "tiny_frames", "tiny_image", "c2"}
    for name, c in y.items():
        cos = c["autocast_emulation_cos"]
        assert len(cos) >= 370, (name, len(cos))             # 405 tensors minus the mathematically-zero gradients (norm_k.bias, unused heads)
        assert min(cos.values()) >= 0.97 and max(cos.values()) <= 1.0 + 1e-9, (name, min(cos.values()))
        assert "rel_pos" in c["grad_cos_worst"][0], c["grad_cos_worst"]
        assert c["grad_cos_global"] >= 0.99 and c["rounded_ops"] > 100
    # the test-side lookup finds exactly these cases
    assert S.yardstick_for(4, 64, 2) is y["tiny"] or S.yardstick_for(4, 64, 2) == y["tiny"]
    assert S.yardstick_for(4, 64, 3, frames_path=True) == y["tiny_frames"]
    assert S.yardstick_for(4, 64, 3, image=True) == y["tiny_image"]
    assert S.yardstick_for(16, 224, 1) == y["c2"] and S.yardstick_for(8, 224, 1) is None


def test_autocast_emulation_rounds_matrix_ops_only():
    """oracle/ref_shim.py::autocast_emulation (the yardstick's mode): linear / matmul / einsum / conv3d see bf16-rounded operands,
    return bf16-rounded results and bf16-rounded gradients; everything else stays fp32."""
    import torch
    import torch.nn.functional as F
    from oracle import ref_shim
    torch.manual_seed(0)
    x = torch.randn(5, 16, requires_grad=True)
    w = torch.randn(8, 16, requires_grad=True)
    r = lambda t: t.to(torch.bfloat16).to(torch.float32)
    mode = ref_shim.autocast_emulation()
    with mode:
        y = F.linear(x, w)
        z = (x @ w.t()) + 1.0 / 3.0            # the add is not a matrix op: not re-rounded
        e = torch.einsum("ik,jk->ij", x, w)
        s = torch.softmax(x, -1)
    assert type(mode).calls == 3
    ref = r(F.linear(r(x), r(w)))
    assert torch.equal(y, ref) and torch.equal(e, ref) and torch.equal(z, ref + 1.0 / 3.0)
    assert torch.equal(s, torch.softmax(x, -1))
    g = torch.randn_like(y)
    y.backward(g)
    gx = r(r(g) @ r(w).detach())
    assert torch.equal(x.grad, gx)
    assert torch.equal(w.grad, r(r(g).t() @ r(x).detach()))
