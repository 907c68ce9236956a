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
