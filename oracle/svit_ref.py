"""CPU restatement (fp32, plain PyTorch CPU ops) of the reference's SViT forward/backward.

TEST INFRASTRUCTURE ONLY.  This file is the *oracle* of the build: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and only as the
checker / the timed CPU baseline -- never from the product path (svit_amd/), which fails
loudly when its HIP library is missing.

Parity status: PINNED.  oracle/gen_golden.py imports the unmodified reference on CPU
(this container only), loads the closed-form weights of oracle/procedural.py into it and
(a) asserts this restatement matches it tensor-for-tensor (max-abs recorded in
tests/golden/manifest.json), (b) writes the golden vectors the tests replay.

Each function cites the reference file:line it restates (paths relative to
/root/reference).  The formulation is deliberately different from the reference's module
tree: one functional pass over a flat {name: tensor} state_dict with the reference's
parameter names, token-major everywhere, closed-form object gain,
explicit rel-pos index tables -- i.e. the same formulation the HIP engine uses, so a
disagreement localises to a kernel.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

HEAD_DIM = 96
LN_EPS = 1e-6  # slowfast/models/video_model_builder.py:69


# --------------------------------------------------------------------------------------
# Architecture spec (slowfast/models/video_model_builder.py:133-232, models/utils.py:16-29)
# --------------------------------------------------------------------------------------
def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:16-29."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


@dataclass
class BlockSpec:
    index: int
    dim_in: int
    dim_out: int
    heads: int
    stride_q: Tuple[int, int, int]
    stride_kv: Tuple[int, int, int]
    in_thw: Tuple[int, int, int]      # ctor-time input size (sizes rel-pos tables)
    rel_sp_rows: int
    rel_t_rows: int
    drop_path: float

    @property
    def has_proj(self):
        return self.dim_in != self.dim_out

    @property
    def pools_q(self):
        return any(s > 1 for s in self.stride_q)


@dataclass
class SViTSpec:
    num_frames: int = 16
    crop: int = 224
    in_chans: int = 3
    embed_dim: int = 96
    depth: int = 16
    num_classes: int = 174
    objects: int = 4
    patch_kernel: Tuple[int, int, int] = (3, 7, 7)
    patch_stride: Tuple[int, int, int] = (2, 4, 4)
    patch_pad: Tuple[int, int, int] = (1, 3, 3)
    mlp_ratio: float = 4.0
    drop_path_rate: float = 0.4
    dropout_rate: float = 0.5
    dim_mul: Tuple[Tuple[int, float], ...] = ((1, 2.0), (3, 2.0), (14, 2.0))
    head_mul: Tuple[Tuple[int, float], ...] = ((1, 2.0), (3, 2.0), (14, 2.0))
    q_stride: Tuple[Tuple[int, int, int, int], ...] = ()
    kv_stride_adaptive: Tuple[int, int, int] = (1, 8, 8)
    blocks: List[BlockSpec] = field(default_factory=list)
    final_dim: int = 768


def make_spec(num_frames=16, crop=224, depth=16, drop_path_rate=0.4, dropout_rate=0.5,
              num_classes=174, dim_mul=((1, 2.0), (3, 2.0), (14, 2.0)),
              head_mul=((1, 2.0), (3, 2.0), (14, 2.0)), q_pool_blocks=(1, 3, 14),
              kv_stride_adaptive=(1, 8, 8)) -> SViTSpec:
    """Block table of SViT.__init__ (video_model_builder.py:133-232) for configs/ssv2.yaml-style
    settings: every block has a (3,3,3) q/kv pool conv; q stride (1,2,2) on q_pool_blocks."""
    spec = SViTSpec(num_frames=num_frames, crop=crop, depth=depth, num_classes=num_classes,
                    drop_path_rate=drop_path_rate, dropout_rate=dropout_rate,
                    dim_mul=tuple(dim_mul), head_mul=tuple(head_mul),
                    kv_stride_adaptive=tuple(kv_stride_adaptive))
    dm = [1.0] * (depth + 1)
    hm = [1.0] * (depth + 1)
    for i, m in dim_mul:
        dm[i] = m
    for i, m in head_mul:
        hm[i] = m
    stride_q = [(1, 2, 2) if i in q_pool_blocks else (1, 1, 1) for i in range(depth)]
    spec.q_stride = tuple((i,) + stride_q[i] for i in range(depth))
    # adaptive kv stride, video_model_builder.py:156-165
    skv = list(kv_stride_adaptive)
    stride_kv = []
    for i in range(depth):
        skv = [max(skv[d] // stride_q[i][d], 1) for d in range(3)]
        stride_kv.append(tuple(skv))
    input_size = [num_frames // spec.patch_stride[0], crop // spec.patch_stride[1],
                  crop // spec.patch_stride[2]]
    dpr = torch.linspace(0, drop_path_rate, depth).tolist() if depth > 1 else [0.0]
    heads, dim = 1, spec.embed_dim
    for i in range(depth):
        heads = round_width(heads, hm[i])
        dim_out = round_width(dim, dm[i], divisor=round_width(heads, hm[i]))
        size = input_size[1]
        q_size = size // stride_q[i][1]
        kv_size = size // stride_kv[i][1]
        spec.blocks.append(BlockSpec(
            index=i, dim_in=dim, dim_out=dim_out, heads=heads, stride_q=stride_q[i],
            stride_kv=stride_kv[i], in_thw=tuple(input_size),
            rel_sp_rows=2 * max(q_size, kv_size) - 1, rel_t_rows=2 * input_size[0] - 1,
            drop_path=float(dpr[i])))
        input_size = [s // st for s, st in zip(input_size, stride_q[i])]
        dim = dim_out
    spec.final_dim = dim
    return spec


def param_shapes(spec: SViTSpec) -> Dict[str, Tuple[int, ...]]:
    """state_dict layout, SURVEY.md Appendix D (verified against the reference in gen_golden)."""
    s: Dict[str, Tuple[int, ...]] = {}
    s["cls_token"] = (1, 1, spec.embed_dim)
    s["pos_embed_temporal"] = (1, spec.num_frames, spec.embed_dim)
    s["object_queries"] = (1, spec.objects, spec.embed_dim)
    s["patch_embed.proj.weight"] = (spec.embed_dim, spec.in_chans) + tuple(spec.patch_kernel)
    s["patch_embed.proj.bias"] = (spec.embed_dim,)
    for b in spec.blocks:
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
        for r in "qkv":
            s[p + "attn.pool_%s.weight" % r] = (HEAD_DIM, 1, 3, 3, 3)
            s[p + "attn.norm_%s.weight" % r] = (HEAD_DIM,)
            s[p + "attn.norm_%s.bias" % r] = (HEAD_DIM,)
        s[p + "norm2.weight"] = (b.dim_out,)
        s[p + "norm2.bias"] = (b.dim_out,)
        hid = int(b.dim_out * spec.mlp_ratio)
        s[p + "mlp.fc1.weight"] = (hid, b.dim_out)
        s[p + "mlp.fc1.bias"] = (hid,)
        s[p + "mlp.fc2.weight"] = (b.dim_out, hid)
        s[p + "mlp.fc2.bias"] = (b.dim_out,)
        if b.has_proj:
            s[p + "proj.weight"] = (b.dim_out, b.dim_in)
            s[p + "proj.bias"] = (b.dim_out,)
    d = spec.final_dim
    s["norm.weight"] = (d,)
    s["norm.bias"] = (d,)
    s["head.projection.weight"] = (spec.num_classes, d)
    s["head.projection.bias"] = (spec.num_classes,)
    s["head.boxes_mlp.0.weight"] = (4, d)
    s["head.boxes_mlp.0.bias"] = (4,)
    s["head.boxes_bce_mlp.weight"] = (1, d)
    s["head.boxes_bce_mlp.bias"] = (1,)
    s["head.contact_mlp.weight"] = (5, d)
    s["head.contact_mlp.bias"] = (5,)
    return s


# --------------------------------------------------------------------------------------
# Pooling of q / k / v and of the skip path (slowfast/models/attention.py:13-65)
# --------------------------------------------------------------------------------------
def pooled_size(n, stride):
    """3-tap, pad-1 window with the given stride (conv pool and max-pool skip alike)."""
    return (n - 1) // stride + 1


def object_gain(weight, stride):
    """Closed form of the object branch of attention_pool (attention.py:45-53): an object
    token replicated to a 3x3x3 cube, zero-padded, convolved with the depthwise kernel at
    this stride and averaged is  obj[c] * g[c].   weight: [96,1,3,3,3] -> g: [96]."""
    def counts(s):
        n_out = pooled_size(3, s)
        n = [0, 0, 0]
        for o in range(n_out):
            for tap in range(3):
                if 0 <= o * s - 1 + tap < 3:
                    n[tap] += 1
        return torch.tensor(n, dtype=weight.dtype, device=weight.device), n_out
    nt, pt = counts(stride[0])
    nh, ph = counts(stride[1])
    nw, pw = counts(stride[2])
    w = weight[:, 0]
    g = torch.einsum("cijk,i,j,k->c", w, nt, nh, nw)
    return g / float(pt * ph * pw)


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def pool_tokens(x, thw, stride, conv_w, norm_w, norm_b, n_obj):
    """attention_pool with a depthwise Conv3d (attention.py:13-65).
    x: [B,h,1+THW+O,96] -> [B,h,1+T'H'W'+O,96], then LayerNorm(96) over ALL tokens."""
    B, h, N, C = x.shape
    T, H, W = thw
    L = T * H * W
    assert N == 1 + L + n_obj
    cls, patch, obj = x[:, :, :1], x[:, :, 1:1 + L], x[:, :, 1 + L:]
    vol = patch.reshape(B * h, T, H, W, C).permute(0, 4, 1, 2, 3)
    vol = F.conv3d(vol, conv_w, None, stride=stride, padding=1, groups=C)
    thw_out = tuple(vol.shape[2:])
    patch = vol.reshape(B, h, C, -1).transpose(2, 3)
    obj = obj * object_gain(conv_w, stride)
    out = torch.cat([cls, patch, obj], dim=2)
    return layer_norm(out, norm_w, norm_b), thw_out


def maxpool_skip(x, thw, stride, n_obj):
    """Skip path of MultiScaleBlock (attention.py:549-555,562-564): MaxPool3d with kernel
    s+1 (=3) / stride s / pad s//2 on patch tokens; cls and object tokens untouched."""
    if all(s == 1 for s in stride):
        return x
    B, N, C = x.shape
    T, H, W = thw
    L = T * H * W
    cls, patch, obj = x[:, :1], x[:, 1:1 + L], x[:, 1 + L:]
    vol = patch.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    k = [s + 1 if s > 1 else s for s in stride]
    vol = F.max_pool3d(vol, k, stride, [kk // 2 for kk in k])
    patch = vol.reshape(B, C, -1).transpose(1, 2)
    return torch.cat([cls, patch, obj], dim=1)


# --------------------------------------------------------------------------------------
# Decomposed relative-position bias (slowfast/models/attention.py:68-183)
# --------------------------------------------------------------------------------------
def resize_table(table, rows):
    """get_rel_pos (attention.py:68-81): F.interpolate(mode='linear', align_corners=False)
    along the row axis when the table length differs from the needed 2*max(q,k)-1."""
    L = table.shape[0]
    if L == rows:
        return table
    scale = L / rows
    pos = (torch.arange(rows, dtype=torch.float32, device=table.device) + 0.5) * scale - 0.5
    pos = pos.clamp(min=0.0)
    i0 = pos.floor().long().clamp(max=L - 1)
    i1 = (i0 + 1).clamp(max=L - 1)
    lam = (pos - i0.float()).unsqueeze(1).to(table.dtype)
    return table[i0] * (1 - lam) + table[i1] * lam


def rel_index(q_n, k_n):
    """dist table of cal_rel_pos_spatial/temporal (attention.py:100-119,156-163):
    dist[q,k] = q*max(k/q,1) - k*max(q/k,1) + (k_n-1)*max(q/k,1), truncated to long."""
    q_ratio = max(k_n / q_n, 1.0)
    k_ratio = max(q_n / k_n, 1.0)
    d = torch.arange(q_n)[:, None] * q_ratio - torch.arange(k_n)[None, :] * k_ratio
    d = d + (k_n - 1) * k_ratio
    return d.long()


def rel_pos_bias(q, q_thw, k_thw, rel_h, rel_w, rel_t):
    """Bias added to attn[:, :, 1:1+Lq, 1:1+Lk] (attention.py:84-183).  q is the UN-scaled,
    pooled+normed query [B,h,Nq,96].  Returns [B,h,Lq,Lk]."""
    B, h, _, C = q.shape
    qt, qh, qw = q_thw
    kt, kh, kw = k_thw
    Rh = resize_table(rel_h, 2 * max(qh, kh) - 1)[rel_index(qh, kh)]  # [qh,kh,C]
    Rw = resize_table(rel_w, 2 * max(qw, kw) - 1)[rel_index(qw, kw)]  # [qw,kw,C]
    Rt = resize_table(rel_t, 2 * max(qt, kt) - 1)[rel_index(qt, kt)]  # [qt,kt,C]
    qp = q[:, :, 1:1 + qt * qh * qw].reshape(B, h, qt, qh, qw, C)
    bh = torch.einsum("bntyxc,ykc->bntyxk", qp, Rh)
    bw = torch.einsum("bntyxc,xkc->bntyxk", qp, Rw)
    bt = torch.einsum("bntyxc,tkc->bntyxk", qp, Rt)
    bias = (bt[..., :, None, None] + bh[..., None, :, None] + bw[..., None, None, :])
    return bias.reshape(B, h, qt * qh * qw, kt * kh * kw)


# --------------------------------------------------------------------------------------
# Blocks (slowfast/models/attention.py:331-466, 557-571; common.py:26-34, 46-59)
# --------------------------------------------------------------------------------------
def attention(p, pre, b: BlockSpec, xn, thw, n_obj, taps=None):
    """MultiScaleAttention.forward (attention.py:331-466)."""
    B, N, _ = xn.shape
    h = b.heads
    qkv = F.linear(xn, p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, h, HEAD_DIM).permute(2, 0, 3, 1, 4)
    q, q_thw = pool_tokens(qkv[0], thw, b.stride_q, p[pre + "pool_q.weight"],
                           p[pre + "norm_q.weight"], p[pre + "norm_q.bias"], n_obj)
    k, k_thw = pool_tokens(qkv[1], thw, b.stride_kv, p[pre + "pool_k.weight"],
                           p[pre + "norm_k.weight"], p[pre + "norm_k.bias"], n_obj)
    v, _ = pool_tokens(qkv[2], thw, b.stride_kv, p[pre + "pool_v.weight"],
                       p[pre + "norm_v.weight"], p[pre + "norm_v.bias"], n_obj)
    Lq = q_thw[0] * q_thw[1] * q_thw[2]
    Lk = k_thw[0] * k_thw[1] * k_thw[2]
    scores = (q * HEAD_DIM ** -0.5) @ k.transpose(-2, -1)
    bias = rel_pos_bias(q, q_thw, k_thw, p[pre + "rel_pos_h"], p[pre + "rel_pos_w"],
                        p[pre + "rel_pos_t"])
    scores = torch.cat([
        scores[:, :, :1],
        torch.cat([scores[:, :, 1:1 + Lq, :1],
                   scores[:, :, 1:1 + Lq, 1:1 + Lk] + bias,
                   scores[:, :, 1:1 + Lq, 1 + Lk:]], dim=3),
        scores[:, :, 1 + Lq:]], dim=2)
    prob = scores.softmax(dim=-1)
    out = prob @ v
    out = torch.cat([out[:, :, :1], out[:, :, 1:] + q[:, :, 1:]], dim=2)  # residual pooling
    out = out.transpose(1, 2).reshape(B, -1, b.dim_out)
    if taps is not None:
        taps[pre + "q"], taps[pre + "k"], taps[pre + "v"] = q, k, v
        taps[pre + "ctx"] = out
    out = F.linear(out, p[pre + "proj.weight"], p[pre + "proj.bias"])
    return out, q_thw


def drop_path(x, keep_scale):
    """common.py:46-59 with the per-sample factor mask/keep_prob supplied by the caller
    (None = identity: eval mode or rate 0)."""
    if keep_scale is None:
        return x
    return x * keep_scale.reshape(-1, *([1] * (x.dim() - 1)))


def block(p, b: BlockSpec, x, thw, n_obj, dp_attn=None, dp_mlp=None, taps=None):
    """MultiScaleBlock.forward (attention.py:557-571), DIM_MUL_IN_ATT=True."""
    pre = "blocks.%d." % b.index
    xn = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    xa, thw_new = attention(p, pre + "attn.", b, xn, thw, n_obj, taps)
    skip = F.linear(xn, p[pre + "proj.weight"], p[pre + "proj.bias"]) if b.has_proj else x
    skip = maxpool_skip(skip, thw, b.stride_q, n_obj)
    x = skip + drop_path(xa, dp_attn)
    xn2 = layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
    hid = F.gelu(F.linear(xn2, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
    x = x + drop_path(F.linear(hid, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"]), dp_mlp)
    return x, thw_new


# --------------------------------------------------------------------------------------
# Whole model (slowfast/models/video_model_builder.py:315-398, 507-551)
# --------------------------------------------------------------------------------------
def embed_tokens(p, spec: SViTSpec, frames):
    """PatchEmbed + cls + object tokens (stem_helper.py:317-320, video_model_builder.py:315-363)."""
    if frames.dim() == 4:
        frames = frames.unsqueeze(2)
    B, _, Tx = frames.shape[:3]
    y = F.conv3d(frames, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"],
                 stride=spec.patch_stride, padding=spec.patch_pad)
    H, W = y.shape[-2:]
    # NB: T comes from the config, not the tensor (video_model_builder.py:322)
    T = spec.num_frames // spec.patch_stride[0] if Tx > 1 else Tx
    tok = y.flatten(2).transpose(1, 2)
    cls = p["cls_token"].expand(B, -1, -1)
    obj = p["object_queries"].unsqueeze(1).expand(B, Tx, -1, -1)
    if Tx > 1:
        obj = obj + p["pos_embed_temporal"].unsqueeze(2)
    else:
        obj = obj + p["pos_embed_temporal"].sum() * 0
    x = torch.cat([cls, tok, obj.flatten(1, 2)], dim=1)
    return x, (T, H, W), Tx * spec.objects, Tx


def head(p, spec: SViTSpec, feat, Tx, training, dropout_keep=None):
    """SViTHead.forward (video_model_builder.py:507-551).  feat: [B,1+O,768] (cls, objects).
    dropout_keep: optional {0, 1/(1-p)} tensor shaped like feat (training only)."""
    if training and dropout_keep is not None:
        feat = feat * dropout_keep
    B = feat.shape[0]
    cls, obj = feat[:, 0], feat[:, 1:]
    extra = {"obj_desc": obj.reshape(B, Tx, -1, obj.shape[-1])}
    logits = F.linear(cls, p["head.projection.weight"], p["head.projection.bias"])
    obj = obj.reshape(B, Tx, -1, obj.shape[-1])
    coords = torch.sigmoid(F.linear(obj, p["head.boxes_mlp.0.weight"], p["head.boxes_mlp.0.bias"]))
    score = F.linear(obj, p["head.boxes_bce_mlp.weight"], p["head.boxes_bce_mlp.bias"])
    contact = F.linear(obj[:, :, :2], p["head.contact_mlp.weight"], p["head.contact_mlp.bias"])
    if not training:
        logits = logits.softmax(dim=1)
        score = score.sigmoid()
        contact = contact.softmax(dim=-1)
    extra["pred_bboxes"] = torch.cat([score, coords], dim=-1)
    extra["pred_contact_state"] = contact
    return logits, extra


def forward(p, spec: SViTSpec, frames, training=True, drop_scales=None, dropout_keep=None,
            taps=None):
    """SViT.forward.  drop_scales: optional list of (attn_scale, mlp_scale) per block, each a
    [B] tensor mask/keep_prob (None entries = no drop)."""
    x, thw, n_obj, Tx = embed_tokens(p, spec, frames)
    if taps is not None:
        taps["tokens"] = x
    for b in spec.blocks:
        dpa = dpm = None
        if drop_scales is not None and drop_scales[b.index] is not None:
            dpa, dpm = drop_scales[b.index]
        x, thw = block(p, b, x, thw, n_obj, dpa, dpm, taps)
        if taps is not None:
            taps["block%d" % b.index] = x
    x = layer_norm(x, p["norm.weight"], p["norm.bias"])
    feat = torch.cat([x[:, :1], x[:, -n_obj:]], dim=1)
    if taps is not None:
        taps["feat"] = feat
    return head(p, spec, feat, Tx, training, dropout_keep)


def sample_drop_scales(spec: SViTSpec, batch, generator=None):
    """Per-sample stochastic-depth factors, common.py:46-59: floor(keep + U[0,1)) / keep."""
    out = []
    for b in spec.blocks:
        if b.drop_path <= 0.0:
            out.append(None)
            continue
        keep = 1.0 - b.drop_path
        pair = []
        for _ in range(2):
            m = torch.floor(keep + torch.rand(batch, generator=generator))
            pair.append(m / keep)
        out.append(tuple(pair))
    return out


# --------------------------------------------------------------------------------------
# Losses (slowfast/models/losses.py:50-93,119-168; slowfast/utils/box_ops.py:10-77)
# --------------------------------------------------------------------------------------
def cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def giou_pairs(a, b):
    """Diagonal of generalized_box_iou (box_ops.py:56-77) for matched xyxy pairs."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, :2], b[:, :2])
    rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    iou = inter / union
    lt_c = torch.min(a[:, :2], b[:, :2])
    rb_c = torch.max(a[:, 2:], b[:, 2:])
    wh_c = (rb_c - lt_c).clamp(min=0)
    area_c = wh_c[:, 0] * wh_c[:, 1]
    return iou - (area_c - union) / area_c


def haog_losses(extra, meta):
    """VideoImageLoss._haog_loss + boxes_loss_ (losses.py:50-93,138-155)."""
    pred, tar = extra["pred_bboxes"], meta["haog_bboxes"]
    valid = 1.0 - torch.all(tar == 0, dim=-1).float()
    out = {"boxes_bce_loss": F.binary_cross_entropy_with_logits(pred[..., 0], valid)}
    if valid.sum() > 0:
        m = valid.bool()
        src, dst = pred[..., 1:][m], tar[m]
        out["boxes_l1_loss"] = (src - dst).abs().mean()
        out["boxes_giou_loss"] = (1 - giou_pairs(cxcywh_to_xyxy(src), cxcywh_to_xyxy(dst))).mean()
    else:
        out["boxes_l1_loss"] = pred.new_zeros(())
        out["boxes_giou_loss"] = pred.new_zeros(())
    cp = extra["pred_contact_state"].flatten(0, 2)
    ct = meta["contact_state"].flatten()
    keep = ct >= 0
    out["loss_contact_state"] = (F.cross_entropy(cp[keep], ct[keep]) if keep.sum() > 0
                                 else pred.new_zeros(()))
    return out


def loss_weights(lambda_nodes=3.7, lambda_edges=0.3):
    """misc.get_lambdas_dict (slowfast/utils/misc.py:412-423) for configs/ssv2.yaml."""
    return {"loss_ce": 1.0, "boxes_l1_loss": 5 * lambda_nodes, "boxes_bce_loss": lambda_nodes,
            "boxes_giou_loss": 2 * lambda_nodes, "loss_contact_state": lambda_edges}


def video_loss(logits, labels):
    """Video-rank loss as released: CE only (losses.py:156-168)."""
    return F.cross_entropy(logits, labels)


def consistency_loss(extra, frames_extra, mode="l1"):
    """VideoImageLoss._consistency_loss (losses.py:127-136): the clip's object descriptors
    against those of the no-grad single-frame pass (train_net.py:105-110), reshaped
    [B*T,1,O,d] -> [B,T,O,d] and detached; mean-reduced L1 ("l1") or squared error ("l2")."""
    pred = extra["obj_desc"]
    tar = frames_extra["obj_desc"].reshape(pred.shape).detach()
    if mode == "l1":
        return F.l1_loss(pred, tar, reduction="mean")
    if mode == "l2":
        return F.mse_loss(pred, tar, reduction="mean")
    raise ValueError(mode)


def image_loss(extra, meta, weights=None):
    w = weights or loss_weights()
    parts = haog_losses(extra, meta)
    return sum(w[k] * v for k, v in parts.items()), parts


# --------------------------------------------------------------------------------------
# Optimiser pieces (slowfast/models/optimizer.py:39-108, utils/lr_policy.py:9-66,
# tools/train_net.py:133-151)
# --------------------------------------------------------------------------------------
def weight_decay_of(name, shape, wd=1e-4):
    """optimizer.py:39-60 with ZERO_WD_1D_PARAM=True and an empty skip list."""
    return 0.0 if (len(shape) == 1 or name.endswith(".bias")) else wd


def cosine_lr(epoch_float, base_lr=2e-4, end_lr=2e-6, max_epoch=50, warmup_epochs=0.0,
              warmup_start=2e-6):
    """lr_policy.get_lr_at_epoch (lr_policy.py:9-66), cosine, COSINE_AFTER_WARMUP."""
    offset = warmup_epochs
    lr = end_lr + (base_lr - end_lr) * (
        math.cos(math.pi * (epoch_float - offset) / (max_epoch - offset)) + 1.0) * 0.5
    if epoch_float < warmup_epochs:
        lr_end = cosine_lr(warmup_epochs, base_lr, end_lr, max_epoch, warmup_epochs, warmup_start)
        alpha = (lr_end - warmup_start) / warmup_epochs
        lr = epoch_float * alpha + warmup_start
    return lr


def clip_and_adamw_step(params, grads, state, lr, step, max_norm=1.0, wd_of=weight_decay_of,
                        betas=(0.9, 0.999), eps=1e-8):
    """clip_grad_norm_(max_norm) then torch.optim.AdamW semantics (train_net.py:144-151,
    optimizer.py:102-108).  params/grads/state: dicts by name; state[name]=(m,v). In place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for name, w in params.items():
        g = grads[name] * coef
        m, v = state[name]
        w.mul_(1 - lr * wd_of(name, tuple(w.shape)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m, denom, value=-lr / bc1)
    return float(total)
