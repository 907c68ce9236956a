"""Losses of the SViT recipe (slowfast/models/losses.py:50-93,119-168; slowfast/utils/box_ops.py:
10-77; slowfast/utils/misc.py:412-423).  A few hundred scalars per step.  On the device the two losses a
step really computes are one launch each way -- the image ranks' HAOG losses (round 1) and, since round 6, the video ranks'
cross entropy (`cross_entropy` below: the replayed step spent four stock launches + autograd's fills on [B, 174] numbers);
host tensors (configuration / unit tests of the plumbing) take plain fp32 torch ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def get_lambdas_dict(cfg):
    """misc.get_lambdas_dict, including the as-released dead consistency key."""
    ret = {"loss_ce": 1, "boxes_l1_loss": 5 * cfg.SVIT.LAMBDA_NODES,
           "boxes_bce_loss": 1 * cfg.SVIT.LAMBDA_NODES, "boxes_giou_loss": 2 * cfg.SVIT.LAMBDA_NODES,
           "loss_contact_state": cfg.SVIT.LAMBDA_EDGES}
    if cfg.TRAIN.FORWARD_VIDEO_FRAMES:
        ret["video_image_boxes_l1_loss"] = cfg.SVIT.LAMBDA_CON
        # extension: make the consistency term live
        mode = getattr(cfg.SVIT, "CONSISTENCY", "")
        if mode in ("l1", "l2"):
            ret["video_image_desc_%s_loss" % mode] = cfg.SVIT.LAMBDA_CON
        elif mode:
            raise NotImplementedError("SVIT.CONSISTENCY must be '', 'l1' or 'l2'")
    return ret


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def generalized_box_iou_diag(a, b):
    """diag(generalized_box_iou(a, b)) for matched xyxy boxes (box_ops.py:56-77)."""
    assert (a[:, 2:] >= a[:, :2]).all() and (b[:, 2:] >= b[:, :2]).all()
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, 2:], b[:, 2:]) - torch.max(a[:, :2], b[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    whc = (torch.max(a[:, 2:], b[:, 2:]) - torch.min(a[:, :2], b[:, :2])).clamp(min=0)
    area_c = whc[:, 0] * whc[:, 1]
    return inter / union - (area_c - union) / area_c


def boxes_loss_(pred, tar):
    """losses.py:50-93 for 4-d targets: (L1, BCE on objectness, 1-GIoU)."""
    tar_mask = 1 - torch.all(tar == 0, dim=-1).float()
    loss_mask = F.binary_cross_entropy_with_logits(pred[..., 0], tar_mask, reduction="none").mean()
    if tar_mask.sum() > 0:
        m = tar_mask.bool()
        src, dst = pred[..., 1:][m], tar[m]
        loss_l1 = F.l1_loss(src, dst, reduction="mean")
        loss_giou = (1 - generalized_box_iou_diag(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(dst))).mean()
    else:
        loss_l1 = torch.tensor(0, device=pred.device, requires_grad=True, dtype=torch.float32)
        loss_giou = torch.tensor(0, device=pred.device, requires_grad=True, dtype=torch.float32)
    return loss_l1, loss_mask, loss_giou


class _HaogLossHip(torch.autograd.Function):
    """The four HAOG losses in one launch on the device (svit_haog_loss, include/svit_hip.h):
    arithmetic masks instead of the reference's `pred[mask]` / `if mask.sum() > 0`, i.e. no host
    sync and fixed shapes, so an image rank's step can be replayed as a HIP graph."""

    @staticmethod
    def forward(ctx, pred, tar, contact, contact_tar):
        from . import ops
        pred, contact = pred.contiguous(), contact.contiguous()
        losses, unit = ops.haog_loss_fwd(pred, tar.contiguous().float(), contact,
                                         contact_tar.contiguous().long())
        ctx.unit, ctx.shapes = unit, (pred.shape, contact.shape)
        parts, stats = losses[:4].clone(), losses[4:].clone()
        ctx.mark_non_differentiable(stats)
        return parts, stats          # parts = (l1, bce, giou, contact CE)

    @staticmethod
    def backward(ctx, d_parts, _):
        from . import ops
        dpred, dcontact = ops.haog_loss_bwd(d_parts.contiguous().float(), ctx.unit, *ctx.shapes)
        return dpred, None, dcontact, None


class _CrossEntropyHip(torch.autograd.Function):
    """nn.CrossEntropyLoss(reduction="mean") forward + unit gradient in one launch (svit_ce_loss, include/svit_hip.h)."""

    @staticmethod
    def forward(ctx, logits, labels):
        from . import ops
        loss, dlogits = ops.ce_loss(logits, labels)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * d_loss, None


def cross_entropy(logits, labels):
    """F.cross_entropy(logits, labels) (mean, ignore_index -100) -- the fused launch for fp32 logits on the GPU."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and labels.dtype == torch.int64:
        return _CrossEntropyHip.apply(logits, labels)
    return F.cross_entropy(logits, labels)


class VideoImageLoss(nn.Module):
    """losses.py:119-168.  `is_video_rank` replaces the reference's local-rank test
    (`du.get_local_rank() not in cfg.IMAGE_TRAIN.GPU_IDS`)."""

    def __init__(self, cfg, reduction="mean", is_video_rank=True):
        super().__init__()
        self.cfg, self.reduction = cfg, reduction
        self.ce_loss = nn.CrossEntropyLoss(reduction=reduction)
        self._lambda = get_lambdas_dict(cfg)
        self._is_vid = is_video_rank

    def is_vid(self):
        return self._is_vid or (not self.training)

    def _consistency_loss(self, extra_preds, frames_extra_preds):
        ret = {}
        pred = extra_preds["obj_desc"]
        tar = frames_extra_preds["obj_desc"].reshape(pred.shape).detach()
        if "video_image_desc_l1_loss" in self._lambda:
            ret["video_image_desc_l1_loss"] = F.l1_loss(pred, tar, reduction=self.reduction)
        if "video_image_desc_l2_loss" in self._lambda:
            ret["video_image_desc_l2_loss"] = F.mse_loss(pred, tar, reduction=self.reduction)
        return ret

    def _haog_loss(self, extra_preds, metadata):
        if extra_preds["pred_bboxes"].is_cuda and self.reduction == "mean":   # product path: one fused launch
            parts, stats = _HaogLossHip.apply(
                extra_preds["pred_bboxes"], metadata["haog_bboxes"],
                extra_preds["pred_contact_state"], metadata["contact_state"])
            self.last_stats = stats     # device-side: #boxes, #contacts, #out-of-range targets
            return {"boxes_l1_loss": parts[0], "boxes_bce_loss": parts[1],
                    "boxes_giou_loss": parts[2], "loss_contact_state": parts[3]}
        # host tensors (configuration / unit tests of the loss plumbing): plain torch ops
        l1, bce, giou = boxes_loss_(extra_preds["pred_bboxes"], metadata["haog_bboxes"])
        ret = {"boxes_l1_loss": l1, "boxes_bce_loss": bce, "boxes_giou_loss": giou}
        pred = extra_preds["pred_contact_state"].flatten(0, 2)
        tar = metadata["contact_state"].flatten()
        mask = tar >= 0
        ret["loss_contact_state"] = (self.ce_loss(pred[mask], tar[mask]) if mask.sum() > 0 else
                                     torch.tensor(0, device=pred.device, requires_grad=True,
                                                  dtype=torch.float32))
        return ret

    def forward(self, x, extra_preds, y, metadata):
        ret = {}
        if self.is_vid():
            ret["loss_ce"] = cross_entropy(x, y) if self.reduction == "mean" else self.ce_loss(x, y)
            if self.cfg.TRAIN.FORWARD_VIDEO_FRAMES and "frames_output" in extra_preds:
                ret.update(self._consistency_loss(extra_preds,
                                                  extra_preds["frames_output"]["extra_preds"]))
        else:
            ret.update(self._haog_loss(extra_preds, metadata))
        return ret

    def total(self, loss_dict):
        """tools/train_net.py:124."""
        return sum(self._lambda[k] * v for k, v in loss_dict.items() if k in self._lambda)


def get_loss_func(cfg, state="train"):
    if cfg.MODEL.LOSS_FUNC != "video_image_loss":
        raise NotImplementedError("Loss {} is not supported".format(cfg.MODEL.LOSS_FUNC))
    return lambda **kw: VideoImageLoss(cfg, **kw)
