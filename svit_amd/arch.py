"""Static architecture plan of SViT derived from the cfg tree.

Mirrors the constructor arithmetic of the reference (slowfast/models/video_model_builder.py:
133-232, slowfast/models/attention.py:308-327, slowfast/models/utils.py:16-29) so that the
state_dict layout and every per-block shape are identical.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

HEAD_DIM = 96


def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:16-29."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


def pooled(n, stride):
    """output length of a 3-tap / pad-1 window at this stride (conv pool and max-pool skip)."""
    return (n - 1) // stride + 1


@dataclass
class BlockPlan:
    index: int
    dim_in: int
    dim_out: int
    heads: int
    stride_q: Tuple[int, int, int]
    stride_kv: Tuple[int, int, int]
    rel_sp_rows: int
    rel_t_rows: int
    drop_path: float

    @property
    def has_proj(self):
        return self.dim_in != self.dim_out

    @property
    def pools_q(self):
        return self.stride_q[1] > 1


@dataclass
class Plan:
    num_frames: int
    crop: int
    in_chans: int
    embed_dim: int
    num_classes: int
    objects: int
    patch_kernel: Tuple[int, int, int]
    patch_stride: Tuple[int, int, int]
    patch_pad: Tuple[int, int, int]
    mlp_ratio: float
    dropout_rate: float
    blocks: List[BlockPlan] = field(default_factory=list)
    final_dim: int = 768


def build_plan(cfg) -> Plan:
    """Read exactly the cfg keys SViT.__init__ reads (SURVEY.md 8(b) 'Constructor contract');
    like the reference it WRITES cfg.MVIT.POOL_KV_STRIDE (video_model_builder.py:156-165)."""
    mv = cfg.MVIT
    assert cfg.DATA.TRAIN_CROP_SIZE == cfg.DATA.TEST_CROP_SIZE
    if mv.NORM != "layernorm":
        raise NotImplementedError("Only supports layernorm.")
    if mv.MODE != "conv":
        raise NotImplementedError("svit_amd implements MVIT.MODE == 'conv' (configs/ssv2.yaml)")
    if mv.PATCH_2D or mv.POOL_FIRST or mv.SEPARATE_QKV or mv.USE_ABS_POS or mv.NORM_STEM:
        raise NotImplementedError("svit_amd implements the configs/ssv2.yaml MViTv2 variant only")
    if not (mv.CLS_EMBED_ON and mv.REL_POS_SPATIAL and mv.REL_POS_TEMPORAL and mv.RESIDUAL_POOLING
            and mv.DIM_MUL_IN_ATT and mv.QKV_BIAS):
        raise NotImplementedError("svit_amd implements the configs/ssv2.yaml MViTv2 variant only")
    if tuple(mv.PATCH_KERNEL) != (3, 7, 7) or tuple(mv.PATCH_STRIDE) != (2, 4, 4) or \
            tuple(mv.PATCH_PADDING) != (1, 3, 3) or list(mv.POOL_KVQ_KERNEL) != [3, 3, 3]:
        raise NotImplementedError("patch kernel (3,7,7)/(2,4,4)/(1,3,3) and 3x3x3 pools only")
    if mv.EMBED_DIM != HEAD_DIM * mv.NUM_HEADS:
        raise NotImplementedError("head_dim must be 96")
    if cfg.DETECTION.ENABLE:
        raise NotImplementedError("DETECTION.ENABLE is dead code in the reference (head_helper)")
    depth = mv.DEPTH
    plan = Plan(num_frames=cfg.DATA.NUM_FRAMES, crop=cfg.DATA.TRAIN_CROP_SIZE,
                in_chans=cfg.DATA.INPUT_CHANNEL_NUM[0], embed_dim=mv.EMBED_DIM,
                num_classes=cfg.MODEL.NUM_CLASSES, objects=cfg.SVIT.O,
                patch_kernel=tuple(mv.PATCH_KERNEL), patch_stride=tuple(mv.PATCH_STRIDE),
                patch_pad=tuple(mv.PATCH_PADDING), mlp_ratio=mv.MLP_RATIO,
                dropout_rate=cfg.MODEL.DROPOUT_RATE)
    if plan.in_chans != 3 or plan.mlp_ratio != 4.0:
        raise NotImplementedError("3 input channels and MLP_RATIO 4.0 only")
    dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
    for i, m in mv.DIM_MUL:
        dim_mul[i] = m
    for i, m in mv.HEAD_MUL:
        head_mul[i] = m
    stride_q = [[] for _ in range(depth)]
    for row in mv.POOL_Q_STRIDE:
        stride_q[row[0]] = list(row[1:])
    if mv.POOL_KV_STRIDE_ADAPTIVE is not None:
        skv = list(mv.POOL_KV_STRIDE_ADAPTIVE)
        out = []
        for i in range(depth):
            if len(stride_q[i]) > 0:
                skv = [max(skv[d] // stride_q[i][d], 1) for d in range(3)]
            out.append([i] + skv)
        cfg.MVIT.POOL_KV_STRIDE = out
    stride_kv = [[] for _ in range(depth)]
    if cfg.MVIT.POOL_KV_STRIDE is None:
        raise NotImplementedError("MVIT.POOL_KV_STRIDE or POOL_KV_STRIDE_ADAPTIVE must be set")
    for row in cfg.MVIT.POOL_KV_STRIDE:
        stride_kv[row[0]] = list(row[1:])
    input_size = [plan.num_frames // plan.patch_stride[0], plan.crop // plan.patch_stride[1],
                  plan.crop // plan.patch_stride[2]]
    if depth > 1:
        dpr = [mv.DROPPATH_RATE * i / (depth - 1) for i in range(depth)]
    else:
        dpr = [0.0]
    heads, dim = mv.NUM_HEADS, mv.EMBED_DIM
    for i in range(depth):
        if len(stride_q[i]) == 0 or len(stride_kv[i]) == 0:
            raise NotImplementedError("every block needs POOL_Q_STRIDE / POOL_KV_STRIDE entries")
        sq, skv_i = tuple(stride_q[i]), tuple(stride_kv[i])
        if sq[0] != 1 or skv_i[0] != 1 or sq[1] != sq[2] or skv_i[1] != skv_i[2] or sq[1] not in (1, 2):
            raise NotImplementedError("temporal stride 1, square spatial strides, q stride 1 or 2")
        heads = round_width(heads, head_mul[i])
        dim_out = round_width(dim, dim_mul[i], divisor=round_width(heads, head_mul[i]))
        if dim_out != heads * HEAD_DIM:
            raise NotImplementedError("head_dim must stay 96 in every block")
        size = input_size[1]
        plan.blocks.append(BlockPlan(
            index=i, dim_in=dim, dim_out=dim_out, heads=heads, stride_q=sq, stride_kv=skv_i,
            rel_sp_rows=2 * max(size // sq[1], size // skv_i[1]) - 1,
            rel_t_rows=2 * input_size[0] - 1, drop_path=float(dpr[i])))
        input_size = [s // st for s, st in zip(input_size, sq)]
        dim = dim_out
    plan.final_dim = dim
    return plan


def param_shapes(plan: Plan):
    """Ordered {name: shape} in the reference's registration order."""
    s = {}
    s["cls_token"] = (1, 1, plan.embed_dim)
    s["pos_embed_temporal"] = (1, plan.num_frames, plan.embed_dim)
    s["object_queries"] = (1, plan.objects, plan.embed_dim)
    s["patch_embed.proj.weight"] = (plan.embed_dim, plan.in_chans) + tuple(plan.patch_kernel)
    s["patch_embed.proj.bias"] = (plan.embed_dim,)
    for b in plan.blocks:
        p = "blocks.%d." % b.index
        s[p + "norm1.weight"] = (b.dim_in,)
        s[p + "norm1.bias"] = (b.dim_in,)
        s[p + "attn.rel_pos_h"] = (b.rel_sp_rows, HEAD_DIM)
        s[p + "attn.rel_pos_w"] = (b.rel_sp_rows, HEAD_DIM)
        s[p + "attn.rel_pos_t"] = (b.rel_t_rows, HEAD_DIM)
        s[p + "attn.qkv.weight"] = (3 * b.dim_out, b.dim_in)
        s[p + "attn.qkv.bias"] = (3 * b.dim_out,)
        s[p + "attn.proj.weight"] = (b.dim_out, b.dim_out)
        s[p + "attn.proj.bias"] = (b.dim_out,)
        for r in "qkv":      # attention.py:263-304 registers pool_q, norm_q, pool_k, norm_k, ...
            s[p + "attn.pool_%s.weight" % r] = (HEAD_DIM, 1, 3, 3, 3)
            s[p + "attn.norm_%s.weight" % r] = (HEAD_DIM,)
            s[p + "attn.norm_%s.bias" % r] = (HEAD_DIM,)
        s[p + "norm2.weight"] = (b.dim_out,)
        s[p + "norm2.bias"] = (b.dim_out,)
        hid = int(b.dim_out * plan.mlp_ratio)
        s[p + "mlp.fc1.weight"] = (hid, b.dim_out)
        s[p + "mlp.fc1.bias"] = (hid,)
        s[p + "mlp.fc2.weight"] = (b.dim_out, hid)
        s[p + "mlp.fc2.bias"] = (b.dim_out,)
        if b.has_proj:
            s[p + "proj.weight"] = (b.dim_out, b.dim_in)
            s[p + "proj.bias"] = (b.dim_out,)
    d = plan.final_dim
    s["norm.weight"] = (d,)
    s["norm.bias"] = (d,)
    s["head.projection.weight"] = (plan.num_classes, d)
    s["head.projection.bias"] = (plan.num_classes,)
    s["head.boxes_mlp.0.weight"] = (4, d)
    s["head.boxes_mlp.0.bias"] = (4,)
    s["head.boxes_bce_mlp.weight"] = (1, d)
    s["head.boxes_bce_mlp.bias"] = (1,)
    s["head.contact_mlp.weight"] = (5, d)
    s["head.contact_mlp.bias"] = (5,)
    return s


def readiness_rank(name, depth):
    """Order in which parameter gradients become final during backward (0 = first): head and
    final norm, then blocks depth-1 .. 0, then the stem.  The flat grad buffer is laid out in
    this order so that data-parallel all-reduce buckets are contiguous slices that can be
    launched while earlier blocks are still back-propagating."""
    if name.startswith("head.") or name.startswith("norm."):
        return 0
    if name.startswith("blocks."):
        return 1 + (depth - 1 - int(name.split(".")[1]))
    return depth + 1
