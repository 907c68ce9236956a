"""Optimiser tail of the training step on the flat parameter buffers.

Replaces `scaler.unscale_ -> clip_grad_norm_(1.0) -> AdamW.step` (tools/train_net.py:136-151,
slowfast/models/optimizer.py:15-112) by three HBM-bound launches: one sum-of-squares reduction
over the flat grad buffer and one fused clip+AdamW kernel per weight-decay group.  No host
synchronisation: the clip coefficient is computed on the device from the reduced norm.
"""
import math

import torch

from . import ops


def get_lr_at_epoch(cfg, cur_epoch):
    """slowfast/utils/lr_policy.py:9-66 (cosine policy with optional linear warm-up)."""
    s = cfg.SOLVER

    def cosine(ep):
        offset = s.WARMUP_EPOCHS if s.COSINE_AFTER_WARMUP else 0.0
        assert s.COSINE_END_LR < s.BASE_LR
        return s.COSINE_END_LR + (s.BASE_LR - s.COSINE_END_LR) * (
            math.cos(math.pi * (ep - offset) / (s.MAX_EPOCH - offset)) + 1.0) * 0.5
    if s.LR_POLICY != "cosine":
        raise NotImplementedError("svit_amd implements SOLVER.LR_POLICY == 'cosine'")
    lr = cosine(cur_epoch)
    if cur_epoch < s.WARMUP_EPOCHS:
        lr_end = cosine(s.WARMUP_EPOCHS)
        alpha = (lr_end - s.WARMUP_START_LR) / s.WARMUP_EPOCHS
        lr = cur_epoch * alpha + s.WARMUP_START_LR
    return {"lr": lr}


class FusedClipAdamW:
    """torch.optim-like surface (`param_groups`, `zero_grad`, `step`, `state_dict`) over the
    model's FlatParams.  Semantics = clip_grad_norm_(max_norm) + torch.optim.AdamW(eps=1e-8)."""

    def __init__(self, model, lr, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 clip_grad_l2norm=None, grad_scale=1.0):
        core = model.module if hasattr(model, "module") else model
        self.model, self.flat = core, core.flat
        if self.flat is None:
            raise RuntimeError("FusedClipAdamW needs a finalized (on-GPU) SViT")
        self.betas, self.eps = betas, eps
        self.clip = clip_grad_l2norm
        self.grad_scale = grad_scale
        n = self.flat.total
        dev = self.flat.data.device
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.sumsq = torch.zeros(1, device=dev)
        self.step_count = 0
        nd = self.flat.n_decay
        self.param_groups = [
            {"lr": lr, "weight_decay": weight_decay, "range": (0, nd)},
            {"lr": lr, "weight_decay": 0.0, "range": (nd, n)},
        ]

    def zero_grad(self, set_to_none=False):
        self.flat.grad.zero_()

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        f = self.flat
        sumsq = None
        if self.clip is not None and self.clip > 0:
            self.sumsq.zero_()
            ops.sumsq(f.grad, self.sumsq)
            sumsq = self.sumsq
        for g in self.param_groups:
            a, b = g["range"]
            if b <= a:
                continue
            ops.adamw_step(f.data[a:b], f.grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], sumsq,
                           float(self.clip or 0.0), g["lr"], self.betas[0], self.betas[1], self.eps,
                           g["weight_decay"], self.step_count, self.grad_scale)

    def grad_norm(self):
        """host value of the last clipped step's global grad norm (forces a sync; logging only)."""
        return float(self.sumsq.sqrt()) * self.grad_scale

    # ---- checkpoint format of torch.optim.AdamW as the reference builds it -------------------
    def _order(self):
        """state index -> parameter name: the reference's two groups (decayed, then 1-D / bias;
        slowfast/models/optimizer.py:39-72), each in named_parameters() order."""
        named = [(n, tuple(p.shape)) for n, p in self.model.named_parameters()]
        wd = self.model.weight_decayed       # same predicate that laid out the flat buffers
        dec = [n for n, s in named if wd(n, s)]
        return dec, [n for n, s in named if not wd(n, s)]

    def state_dict(self):
        """The dict torch.optim.AdamW.state_dict() yields for the reference's optimizer (what a
        released .pyth holds under "optimizer_state"): per-parameter step / exp_avg / exp_avg_sq
        (views of the flat moment buffers) and two param groups of indices."""
        dec, rest = self._order()
        state = {}
        if self.step_count > 0:
            for j, n in enumerate(dec + rest):
                state[j] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.flat.view(self.exp_avg, n),
                            "exp_avg_sq": self.flat.view(self.exp_avg_sq, n)}
        groups, base = [], 0
        for g, names in zip(self.param_groups, (dec, rest)):
            if not names:
                continue
            groups.append({"lr": g["lr"], "betas": tuple(self.betas), "eps": self.eps,
                           "weight_decay": g["weight_decay"], "amsgrad": False, "maximize": False,
                           "foreach": None, "capturable": False, "differentiable": False,
                           "fused": None, "decoupled_weight_decay": True,
                           "params": list(range(base, base + len(names)))})
            base += len(names)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        dec, rest = self._order()
        order = dec + rest
        n_listed = sum(len(g["params"]) for g in sd["param_groups"])
        if n_listed != len(order):
            raise ValueError("optimizer state lists %d parameters, the model has %d" % (n_listed, len(order)))
        state = sd["state"]
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        step = 0
        for j, n in enumerate(order):
            ent = state.get(j, state.get(str(j)))
            if ent is None:
                continue
            want = self.flat.slots[n][2]
            for key in ("exp_avg", "exp_avg_sq"):     # copy_ would silently broadcast [96] -> [1,1,96]
                if tuple(ent[key].shape) != tuple(want):
                    raise ValueError("optimizer state %d (%s): %s has shape %s, the parameter %s -- the "
                                     "checkpoint's parameter order / decay groups differ from this model's"
                                     % (j, n, key, tuple(ent[key].shape), tuple(want)))
            self.flat.view(self.exp_avg, n).copy_(ent["exp_avg"])
            self.flat.view(self.exp_avg_sq, n).copy_(ent["exp_avg_sq"])
            step = max(step, int(ent["step"]))
        self.step_count = step
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g["lr"] = src["lr"]


def construct_optimizer(model, cfg):
    """slowfast/models/optimizer.py:15-112 for the configuration the SViT recipe uses."""
    if cfg.SOLVER.OPTIMIZING_METHOD != "adamw" or not cfg.SOLVER.ZERO_WD_1D_PARAM:
        raise NotImplementedError("svit_amd fuses the configs/ssv2.yaml solver (adamw, ZERO_WD_1D_PARAM)")
    return FusedClipAdamW(model, lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY,
                          clip_grad_l2norm=cfg.SOLVER.CLIP_GRAD_L2NORM)


def set_lr(optimizer, new_lr):
    for g in optimizer.param_groups:
        g["lr"] = new_lr["lr"] if isinstance(new_lr, dict) else new_lr
