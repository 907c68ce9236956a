"""Closed-form (RNG-free) weights and inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The golden vectors under
tests/golden/ were produced by loading exactly these tensors into the *reference*
(oracle/gen_golden.py); the tests rebuild the same tensors from the same formula and
feed them to the CPU restatement (oracle/svit_ref.py) and to the HIP path.

Every element is a pure function of (tensor name, flat index): a 64-bit integer mix
(murmur3 finaliser) mapped to [-1, 1).  No torch/numpy RNG state is involved, so the
values are identical on any machine and any library version.
"""
import zlib

import numpy as np
import torch

_M1 = np.uint64(0xFF51AFD7ED558CCD)
_M2 = np.uint64(0xC4CEB9FE1A85EC53)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fmix64(x):
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(33))
        x = x * _M1
        x = x ^ (x >> np.uint64(33))
        x = x * _M2
        x = x ^ (x >> np.uint64(33))
    return x


def hash_uniform(name, n, offset=0):
    """float64 array of n values in [-1, 1), element i a pure function of (name, offset+i)."""
    seed = np.uint64(zlib.crc32(name.encode("utf-8")))
    i = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = _fmix64(i * _GOLD + seed * _M1 + np.uint64(1))
    top = (x >> np.uint64(40)).astype(np.float64)  # 24 bits
    return top / float(1 << 23) - 1.0


def tensor(name, shape, amp=1.0, center=0.0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    v = hash_uniform(name, n) * amp + center
    return torch.from_numpy(v.astype(np.float32)).reshape(shape).to(dtype)


def _amp_center(name, shape):
    """Scale per parameter kind: near the reference's init scale but with NON-zero biases
    and non-unit LayerNorm gains so every term of every formula is exercised."""
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    if parent.startswith("norm") or name.startswith("norm."):
        return (0.2, 1.0) if leaf == "weight" else (0.1, 0.0)
    if "pool_" in parent:
        return 0.3, 0.0  # depthwise 3x3x3 conv: default init bound is 1/sqrt(27)
    if name == "patch_embed.proj.weight":
        return 0.08, 0.0
    if leaf == "bias":
        return 0.05, 0.0
    if leaf in ("rel_pos_h", "rel_pos_w", "rel_pos_t"):
        return 0.25, 0.0  # larger than init so the bias visibly moves the softmax
    if name in ("cls_token", "object_queries", "pos_embed_temporal"):
        return 0.5, 0.0
    if len(shape) == 2:  # nn.Linear weight [out, in]
        return 1.7 / np.sqrt(shape[1]), 0.0
    return 0.05, 0.0


def state_dict(shapes):
    """shapes: {name: shape} -> {name: fp32 tensor} with the procedural values."""
    out = {}
    for name, shape in shapes.items():
        amp, center = _amp_center(name, tuple(shape))
        out[name] = tensor("param:" + name, tuple(shape), amp, center)
    return out


def frames(batch, num_frames, crop, tag="clip"):
    """Synthetic normalised frames [B,3,T,S,S], roughly unit variance."""
    return tensor("input:%s:%d:%d:%d" % (tag, batch, num_frames, crop),
                  (batch, 3, num_frames, crop, crop), amp=1.7)


def labels(batch, num_classes=174, tag="label"):
    v = hash_uniform("input:%s:%d" % (tag, batch), batch)
    return torch.from_numpy(((v + 1.0) * 0.5 * num_classes).astype(np.int64)).clamp_(0, num_classes - 1)


def haog_meta(batch, frames_t=1, objects=4, tag="haog"):
    """Synthetic HAOG targets: cxcywh boxes in (0,1), ~25% rows empty (all-zero);
    contact_state in {-1,0,..,4}."""
    u = hash_uniform("input:%s:box:%d" % (tag, batch), batch * frames_t * objects * 4)
    u = torch.from_numpy(u.astype(np.float32)).reshape(batch, frames_t, objects, 4)
    box = torch.empty_like(u)
    box[..., :2] = 0.5 + 0.25 * u[..., :2]
    box[..., 2:] = 0.25 + 0.15 * u[..., 2:]
    e = hash_uniform("input:%s:empty:%d" % (tag, batch), batch * frames_t * objects)
    empty = torch.from_numpy(e).reshape(batch, frames_t, objects) > 0.5
    box[empty] = 0.0
    c = hash_uniform("input:%s:contact:%d" % (tag, batch), batch * 2)
    contact = torch.from_numpy(np.floor((c + 1.0) * 3.0).astype(np.int64) - 1).reshape(batch, 2)
    return {"haog_bboxes": box, "contact_state": contact}


def sample_of(t, k=256):
    """Strided sample (<= k elements) of a large tensor: what the golden files keep of tensors too
    big to store whole (gradients of the 16x224^2 model, module KAT outputs)."""
    t = t.detach().reshape(-1)
    return t[::max(t.numel() // k, 1)][:k]


def digest(t, k=8):
    """Small fingerprint of a tensor: stats + first/strided elements + a hashed projection."""
    t = t.detach().to(torch.float64).reshape(-1)
    n = t.numel()
    proj = torch.from_numpy(hash_uniform("digest", n))
    stride = max(n // k, 1)
    return {
        "n": n,
        "mean": float(t.mean()),
        "absmean": float(t.abs().mean()),
        "l2": float(t.norm()),
        "proj": float((t * proj).sum()),
        "head": [float(v) for v in t[:k]],
        "strided": [float(v) for v in t[::stride][:k]],
    }
