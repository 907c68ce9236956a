"""Host-side mirror of the reference's model interface for the SViT hot path.

    MODEL_REGISTRY.get("SViT")(cfg)  /  build_model(cfg, gpu_id=None)
    model(inputs: List[Tensor], metadata=None, bboxes=None) -> (preds, extra_preds)

Same names, argument meaning, outputs, state_dict layout and error behaviour as
slowfast/models/build.py:20-75 and slowfast/models/video_model_builder.py:24-551 (SURVEY.md
8(b), Appendix D), but the backbone is ONE autograd node whose forward/backward are the HIP
launch schedules of svit_amd/engine.py.  There is no PyTorch fallback for the backbone: without
a GPU or without libsvit_hip.so the forward raises.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import arch, hip, ops
from .engine import Engine, FlatParams


class Registry:
    """fvcore-style registry (slowfast/models/build.py:9)."""

    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        def deco(o):
            self._map[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name):
        if name not in self._map:
            raise KeyError("No object named '%s' found in '%s' registry!" % (name, self._name))
        return self._map[name]


MODEL_REGISTRY = Registry("MODEL")


def _weight_decayed(name, shape, skip=()):
    """slowfast/models/optimizer.py:35-52 with ZERO_WD_1D_PARAM: 1-D tensors, biases and the
    names `model.no_weight_decay()` lists are not decayed; everything else is.  The reference
    tests `name in skip` on the FULL dotted name, so of the names SViT.no_weight_decay() returns
    (video_model_builder.py:267-289) only the top-level cls_token / object_queries /
    pos_embed_temporal ever match -- "rel_pos_h" never equals "blocks.0.attn.rel_pos_h"."""
    return not (name in skip or len(shape) == 1 or name.endswith(".bias"))


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class _Backbone(torch.autograd.Function):
    """video -> LayerNorm-ed tokens.  Parameter gradients are accumulated by the engine straight
    into the flat grad buffer that every `param.grad` is a view of."""

    @staticmethod
    def forward(ctx, model, video, drop_scales, need_grad, anchor):
        with torch.no_grad():
            y, st = model.engine.forward(video, drop_scales, save=need_grad)
        ctx.model, ctx.st = model, (st if need_grad else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        model = ctx.model
        if ctx.st is None:
            raise RuntimeError("backward through an SViT forward that was run without grad")
        model._attach_grads()
        with torch.no_grad():
            model.engine.backward(ctx.st, dy, on_ready=model._grad_ready_hook,
                                  ready_ranks=model._grad_ready_ranks)
        ctx.st = None
        return None, None, None, None, None


class _HeadFn(torch.autograd.Function):
    """The training-mode head as ONE launch each way (svit_head_fwd / svit_head_bwd, csrc/head.hip) instead of
    ~45 stock ATen launches: the backward writes the eight parameter gradients straight into the flat gradient
    buffer's views (like the backbone's node does) and returns d(tokens) whole."""

    @staticmethod
    def forward(ctx, model, tokens, T, O, keep):
        head = model.head
        tokens = tokens.contiguous()
        with torch.no_grad():
            outs = ops.head_fwd(tokens, T, O, keep, [(w.data, b.data) for w, b in head.param_pairs()])
        ctx.model, ctx.T, ctx.O = model, T, O
        ctx.save_for_backward(tokens, keep, outs[1])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, dlogits, dboxes, dcontact, dxobj):
        tokens, keep, boxes = ctx.saved_tensors
        model = ctx.model
        model._attach_grads()      # (zeroes the flat buffer first if zero_grad() dropped the views)
        pairs = model.head.param_pairs()
        with torch.no_grad():
            dtok = ops.head_bwd(tokens, ctx.T, ctx.O, keep, [(w.data, b.data) for w, b in pairs], boxes,
                                (dlogits, dboxes, dcontact, dxobj), [(w.grad, b.grad) for w, b in pairs])
        return None, dtok, None, None, None


class SViTHead(nn.Module):
    """slowfast/models/video_model_builder.py:408-551 -- tiny ([B,65,768]) fp32 torch ops."""

    def __init__(self, cfg, dim_in, num_classes, dropout_rate=0.0, act_func="softmax"):
        super().__init__()
        self.T = cfg.DATA.NUM_FRAMES
        self.dropout_rate = dropout_rate
        if act_func not in ("softmax", "sigmoid"):
            raise NotImplementedError("{} is not supported as an activationfunction.".format(act_func))
        self.act_func = act_func
        self.projection = nn.Linear(dim_in, num_classes, bias=True)
        self.boxes_mlp = nn.Sequential(nn.Linear(dim_in, 4, bias=True), nn.Sigmoid())
        self.boxes_bce_mlp = nn.Linear(dim_in, 1, bias=True)
        self.contact_mlp = nn.Linear(dim_in, 5, bias=True)

    def param_pairs(self):
        """(weight, bias) of the class projection, the box MLP, the objectness Linear and the contact MLP."""
        return [(self.projection.weight, self.projection.bias), (self.boxes_mlp[0].weight, self.boxes_mlp[0].bias),
                (self.boxes_bce_mlp.weight, self.boxes_bce_mlp.bias), (self.contact_mlp.weight, self.contact_mlp.bias)]

    def forward(self, x, T=None, dropout_keep=None):
        T = self.T if T is None else T
        if self.dropout_rate > 0.0 and self.training:
            x = x * dropout_keep if dropout_keep is not None else F.dropout(x, self.dropout_rate, True)
        B = x.size(0)
        # the reference touches every head parameter so that DDP sees a grad on every rank
        # (video_model_builder.py:514).  Here every parameter's .grad is a view of the flat grad
        # buffer (SViT._attach_grads), so heads the loss does not reach already hold exact zeros.
        cls, xobj = x[:, 0], x[:, 1:]
        extra = {"obj_desc": xobj.reshape(B, T, -1, xobj.size(-1))}
        logits = self.projection(cls)
        if not self.training:
            logits = logits.softmax(dim=1) if self.act_func == "softmax" else logits.sigmoid()
        xobj = xobj.reshape(B, T, -1, xobj.size(-1))
        boxes = self.boxes_mlp(xobj)
        boxes_bce = self.boxes_bce_mlp(xobj)
        contact = self.contact_mlp(xobj[:, :, :2])
        if not self.training:
            boxes_bce = boxes_bce.sigmoid()
            contact = contact.softmax(dim=-1)
        extra["pred_bboxes"] = torch.cat((boxes_bce, boxes), dim=-1)
        extra["pred_contact_state"] = contact
        return logits, extra


@MODEL_REGISTRY.register()
class SViT(nn.Module):
    """MI355X-native SViT (reference: slowfast/models/video_model_builder.py:24-398)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.plan = arch.build_plan(cfg)
        self.O = cfg.SVIT.O
        shapes = arch.param_shapes(self.plan)
        self._shapes = shapes
        # parameters first live as ordinary CPU tensors with the reference's init
        # (video_model_builder.py:244-265, attention.py:317-327); finalize() moves them into the
        # flat device buffers.
        self._names = []
        for name, shape in shapes.items():
            if name.startswith("head."):
                continue
            t = torch.zeros(shape)
            leaf = name.split(".")[-1]
            if name in ("cls_token", "pos_embed_temporal", "object_queries") or leaf.startswith("rel_pos_"):
                _trunc_normal_(t)
            elif "pool_" in name or name == "patch_embed.proj.weight":
                fan_in = int(math.prod(shape[1:]))
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            elif name == "patch_embed.proj.bias":
                bound = 1.0 / math.sqrt(3 * 3 * 7 * 7)
                nn.init.uniform_(t, -bound, bound)
            elif leaf == "weight" and len(shape) == 2:
                _trunc_normal_(t)
            elif leaf == "weight":          # LayerNorm gains
                t.fill_(1.0)
            self._register(name, t)
        self.head = SViTHead(cfg, self.plan.final_dim, self.plan.num_classes,
                             dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT)
        for m in self.head.modules():
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight)
                nn.init.constant_(m.bias, 0)
        self.engine = None
        self.fused_head = True      # (False = the ATen head in training too; tests / tools flip the attribute)
        self._head_ones = None
        self._rng_state = None      # int64 [3] on the device: {seed, draw number, ticket} of svit_step_draws
        self._head_keep = None      # the head's dropout factors drawn together with the stochastic-depth factors
        self.fused_draws = True     # (False = torch's own rand / floor / dropout launches; tests / tools flip the attribute)
        self.flat = None
        self._grad_ready_hook = None   # set by the data-parallel wrapper (svit_amd/dp.py)
        self._grad_ready_ranks = None  # ranks at which that hook launches collectives
        self._anchor = None
        self._keep = None
        self._grad_views = None

    # -- parameters are registered under the reference's dotted names ---------------------------
    def _register(self, name, tensor):
        parts = name.split(".")
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(tensor))
        self._names.append(name)

    def _param(self, name):
        mod = self
        for p in name.split("."):
            mod = getattr(mod, p)
        return mod

    def no_weight_decay(self):
        """video_model_builder.py:267-289."""
        names = []
        if self.cfg.MVIT.ZERO_DECAY_POS_CLS:
            names.extend(["rel_pos_h", "rel_pos_w", "rel_pos_hw", "rel_pos_t", "cls_token",
                          "object_queries", "pos_embed_temporal"])
        return names

    def decay_skip(self):
        """The skip list `construct_optimizer` really applies (optimizer.py:35-37): it asks
        `hasattr(model, "no_weight_decay")` of the object build_model returned, which is the DDP
        wrapper when NUM_GPUS > 1 -- there the attribute does not exist and nothing is skipped."""
        return frozenset(self.no_weight_decay()) if self.cfg.NUM_GPUS <= 1 else frozenset()

    def weight_decayed(self, name, shape):
        return _weight_decayed(name, shape, self.decay_skip())

    # -- device placement ------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        p0 = self.cls_token
        if p0.is_cuda and (self.flat is None or self.flat.data.device != p0.device):
            self.finalize()
        return out

    def finalize(self):
        """Move every parameter (head included) into the flat fp32 buffer on the current device
        and make `param.data` / `param.grad` views of the flat data / grad buffers."""
        named = dict(self.named_parameters())
        dev = self.cls_token.device
        if dev.type != "cuda":
            raise hip.SvitHipError("SViT (svit_amd) runs on an MI355X only: move the model to a "
                                   "GPU (build_model with cfg.NUM_GPUS >= 1); no CPU fallback")
        hip.load()
        shapes = {n: tuple(p.shape) for n, p in named.items()}
        flat = FlatParams(shapes, self.weight_decayed, dev, arch.readiness_rank)
        for n, p in named.items():
            flat.p(n).copy_(p.data)
            p.data = flat.p(n)
            p.grad = None
        self.flat = flat
        self.engine = Engine(self.plan, flat)
        self._grad_views = None
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        return self

    def _attach_grads(self):
        """Make every `.grad` a view of the flat grad buffer before the engine accumulates into
        it.  Called at the start of the backbone's backward, i.e. after the (tiny) head has
        already back-propagated through ordinary autograd: grads it produced into fresh tensors
        (after `optimizer.zero_grad()` dropped ours, tools/train_net.py:133) are adopted."""
        flat = self.flat
        if self._grad_views is None:
            self._grad_views = [(p, flat.g(n)) for n, p in self.named_parameters()]
        views = self._grad_views
        fresh = views[0][0].grad is not views[0][1]
        if fresh:
            flat.grad.zero_()
        for p, v in views:
            g = p.grad
            if g is v:
                continue
            if g is not None:
                if fresh:
                    v.copy_(g)
                else:
                    v.add_(g)
            p.grad = v

    # -- forward ---------------------------------------------------------------------------------
    def sample_drop_scales(self, batch, device, Tx=None):
        """Per-sample stochastic-depth factors floor(keep + U[0,1)) / keep (common.py:46-59).  Round 6: ONE launch
        (svit_step_draws) that also draws the training head's dropout factors when `Tx` says how many tokens the head
        will see (head_train picks them up) -- five stock launches before.  Philox keyed by torch.initial_seed() at the
        first draw; the draw number lives on the device and advances with every launch, also under HIP-graph replay."""
        blocks = self.plan.blocks
        self._head_keep = None
        if not self.training or all(b.drop_path <= 0.0 for b in blocks):
            return [None] * len(blocks)
        if self._keep is None or self._keep.device != torch.device(device):
            self._keep = torch.tensor([1.0 - b.drop_path for b in blocks], device=device).view(-1, 1, 1)
        if not self.fused_draws:
            # one launch chain for all blocks: rows (attention branch, MLP branch) per block
            m = torch.floor(self._keep + torch.rand((len(blocks), 2, batch), device=device)) / self._keep
        else:
            if self._rng_state is None or self._rng_state.device != torch.device(device):
                self._rng_state = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device)
            shp, p = None, float(self.head.dropout_rate)
            if Tx is not None and self.fused_head and p > 0.0:
                shp = (batch, 1 + Tx * self.O, self.plan.final_dim)
            from . import ops
            m, drop = ops.step_draws(self._rng_state, self._keep.view(-1), 2 * batch,
                                     n_drop=shp[0] * shp[1] * shp[2] if shp else 0, p_drop=p)
            m = m.view(len(blocks), 2, batch)
            if shp:
                self._head_keep = drop.view(shp)
        return [(m[i, 0], m[i, 1]) if b.drop_path > 0.0 else None for i, b in enumerate(blocks)]

    def head_train(self, tokens, Tx, dropout_keep=None):
        """Training-mode head on the final norm's output tokens f32 [B,N,C] as one launch each way
        (csrc/head.hip through _HeadFn); eval keeps the ATen ops (softmax / sigmoid outputs).  Also what
        graph.GraphedTrainStep calls."""
        n_obj = Tx * self.O
        keep = dropout_keep
        shp = (tokens.shape[0], 1 + n_obj, tokens.shape[2])
        if keep is None and self._head_keep is not None and tuple(self._head_keep.shape) == shp:
            keep, self._head_keep = self._head_keep, None          # drawn with this pass's stochastic-depth factors
        elif keep is None and self.head.dropout_rate > 0.0:
            if self._head_ones is None or tuple(self._head_ones.shape) != shp:
                self._head_ones = torch.ones(shp, device=tokens.device)
            keep = F.dropout(self._head_ones, self.head.dropout_rate, True)     # mask / (1 - p), one launch
        elif keep is not None:
            keep = keep.to(torch.float32).expand(tokens.shape[0], 1 + n_obj, tokens.shape[2]).contiguous()
        logits, boxes, contact, xobj = _HeadFn.apply(self, tokens, Tx, self.O, keep)
        return logits, {"obj_desc": xobj, "pred_bboxes": boxes, "pred_contact_state": contact}

    def forward(self, x, metadata=None, bboxes=None, drop_scales=None, dropout_keep=None):
        if self.engine is None:
            raise hip.SvitHipError("SViT (svit_amd) has no CPU path: build it with "
                                   "build_model(cfg) / call .cuda() on an MI355X first")
        x = x[0]
        if x.dim() == 4:  # image
            x = x.unsqueeze(2)
        Tx = x.shape[2]
        self.engine.refresh_weights()
        if drop_scales is None:
            drop_scales = self.sample_drop_scales(x.shape[0], x.device, Tx=Tx)
        need_grad = torch.is_grad_enabled()
        tokens = _Backbone.apply(self, x, drop_scales, need_grad, self._anchor)
        n_obj = Tx * self.O
        if self.training and self.fused_head:
            return self.head_train(tokens, Tx, dropout_keep)
        feat = torch.cat((tokens[:, :1], tokens[:, -n_obj:]), dim=1)
        return self.head(feat, T=Tx, dropout_keep=dropout_keep)


def build_model(cfg, gpu_id=None):
    """slowfast/models/build.py:20-75: same contract (GPU-count asserts, registry lookup,
    `.cuda(device)`, data-parallel wrap when NUM_GPUS > 1)."""
    if torch.cuda.is_available():
        assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    else:
        assert cfg.NUM_GPUS == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if cfg.NUM_GPUS:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)
    if cfg.NUM_GPUS > 1:
        from .dp import DataParallel
        model = DataParallel(model, device_ids=[cur_device], output_device=cur_device,
                             find_unused_parameters=cfg.DDP_FIND_UNUSED_PARAMETERS)
    return model
