"""Host-side loss module (svit_amd/losses.py): the frame-clip consistency switch of SURVEY.md 8(f)
rank 1.  As released the reference weights a key no loss produces (losses.py:127-136,
misc.py:421-422); SVIT.CONSISTENCY = "l1" / "l2" makes the term live.  CPU only."""
import pytest
import torch

from svit_amd import config, losses


def _cfg(mode):
    cfg = config.ssv2_cfg(num_frames=4, crop=64)
    cfg.SVIT.CONSISTENCY = mode
    return cfg


def _extra(B=2, T=4):
    g = torch.Generator().manual_seed(3)
    vid = torch.randn(B, T, 4, 768, generator=g, requires_grad=True)
    frm = torch.randn(B * T, 1, 4, 768, generator=g)
    return {"obj_desc": vid, "frames_output": {"preds": None, "extra_preds": {"obj_desc": frm}}}, vid, frm


def test_as_released_consistency_is_dead():
    cfg = _cfg("")
    lam = losses.get_lambdas_dict(cfg)
    assert lam["video_image_boxes_l1_loss"] == cfg.SVIT.LAMBDA_CON
    assert not any(k.startswith("video_image_desc") for k in lam)
    fn = losses.VideoImageLoss(cfg)
    extra, vid, _ = _extra()
    logits, y = torch.randn(2, 174), torch.tensor([3, 7])
    d = fn(logits, extra, y, {})
    assert set(d) == {"loss_ce"}                       # the frames pass changes nothing


@pytest.mark.parametrize("mode", ["l1", "l2"])
def test_consistency_switch(mode):
    cfg = _cfg(mode)
    fn = losses.VideoImageLoss(cfg)
    extra, vid, frm = _extra()
    logits, y = torch.randn(2, 174), torch.tensor([3, 7])
    d = fn(logits, extra, y, {})
    key = "video_image_desc_%s_loss" % mode
    tar = frm.reshape(vid.shape)
    ref = (vid - tar).abs().mean() if mode == "l1" else ((vid - tar) ** 2).mean()
    assert torch.allclose(d[key], ref)
    total = fn.total(d)
    assert torch.allclose(total, d["loss_ce"] + cfg.SVIT.LAMBDA_CON * ref)
    total.backward()
    assert vid.grad is not None and float(vid.grad.abs().sum()) > 0      # target side is detached


def test_bad_mode_raises():
    with pytest.raises(NotImplementedError):
        losses.get_lambdas_dict(_cfg("huber"))


def test_rank_roles_of_the_published_recipe():
    """slowfast/datasets/loader.py:186-201 with configs/ssv2.yaml: GPUs 0-6 train 9 clips each,
    GPU 7 trains 63 still images."""
    from svit_amd import dp
    cfg = config.ssv2_cfg(num_frames=16, crop=224, num_gpus=8)
    roles = [dp.rank_role(cfg, r) for r in range(8)]
    assert [r.is_image for r in roles] == [False] * 7 + [True]
    assert [r.data_rank for r in roles] == [0, 1, 2, 3, 4, 5, 6, 0]
    assert [r.replicas for r in roles] == [7] * 7 + [1]
    assert [r.batch_size for r in roles] == [9] * 7 + [63]
    cfg.IMAGE_TRAIN.GPU_IDS = [2, 5]
    cfg.IMAGE_TRAIN.BATCH_SIZE, cfg.TRAIN.BATCH_SIZE = 64, 48
    roles = [dp.rank_role(cfg, r) for r in range(8)]
    assert [r.is_image for r in roles] == [False, False, True, False, False, True, False, False]
    assert roles[5].data_rank == 1 and roles[5].batch_size == 32 and roles[6].data_rank == 4
    assert roles[6].batch_size == 8 and roles[6].replicas == 6


def test_image_rank_loss_host_path_matches_oracle():
    """the boolean-index formulation kept for host tensors == oracle restatement (pinned by the
    reference's numbers in tests/test_oracle_golden.py)."""
    from oracle import procedural as P
    from oracle import svit_ref as R
    cfg = config.ssv2_cfg(num_frames=4, crop=64)
    meta = P.haog_meta(5)
    g = torch.Generator().manual_seed(5)
    extra = {"pred_bboxes": torch.cat((torch.randn(5, 1, 4, 1, generator=g),
                                       torch.rand(5, 1, 4, 4, generator=g) * 0.5 + 0.2), -1),
             "pred_contact_state": torch.randn(5, 1, 2, 5, generator=g)}
    fn = losses.VideoImageLoss(cfg, is_video_rank=False)
    fn.train()
    d = fn(None, extra, None, meta)
    total, parts = R.image_loss(extra, meta, R.loss_weights(cfg.SVIT.LAMBDA_NODES, cfg.SVIT.LAMBDA_EDGES))
    assert set(d) == set(parts)
    for k in parts:
        assert torch.allclose(d[k], parts[k], atol=1e-6), k
    assert torch.allclose(fn.total(d), total, atol=1e-5)


def test_nan_checks():
    """misc.check_nan_losses (reference semantics) and the periodic device-side watch."""
    from svit_amd import misc
    misc.check_nan_losses(torch.tensor(1.5))
    with pytest.raises(RuntimeError, match="NaN"):
        misc.check_nan_losses(torch.tensor(float("nan")), extra_msg="x")
    w = misc.NanWatch("cpu", period=4)
    for v in (0.5, 0.25, float("nan")):
        w.update(torch.tensor(v))                 # no sync, no raise before the period ends
    with pytest.raises(RuntimeError, match="first at step 2"):
        w.update(torch.tensor(0.1))
    ok = misc.NanWatch("cpu", period=2)
    for v in (1.0, 2.0, 3.0, 4.0):
        ok.update(torch.tensor(v))
    ok.check()


def test_cross_entropy_host_path_is_torch():
    """losses.cross_entropy: host tensors (and anything that is not fp32 [B, C] on the GPU) take F.cross_entropy; the fused launch
    (svit_ce_loss) is the device path only -- tests/test_kernels_gpu.py::test_ce_loss_fused."""
    import torch
    import torch.nn.functional as F
    from svit_amd import losses
    x = torch.randn(5, 11, requires_grad=True)
    y = torch.tensor([0, 3, 10, -100, 7])
    a = losses.cross_entropy(x, y)
    assert torch.equal(a, F.cross_entropy(x, y))
    a.backward()
    assert x.grad is not None and float(x.grad[3].abs().max()) == 0.0
