#!/usr/bin/env python3
"""Which path svit_pool_conv_bwd_qkv takes (1 = the fused plane-walk kernel, 0 = the streaming launches) at every block of every
configuration the parity suite runs -- the evidence behind moving the streaming conv-backward kernels to the diagnostic build
(round 6).   python tools/diag/pool_bwd_paths.py    (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import smoke_impl as S
from oracle import procedural as P
from svit_amd import hip, ops

lib = hip.load()
paths = []
orig = ops.pool_conv_bwd_qkv


def spy(*a, **k):
    r = orig(*a, **k)
    paths.append(lib.svit_debug_pool_bwd_path())
    return r


ops.pool_conv_bwd_qkv = spy
import svit_amd.engine as E
E.ops.pool_conv_bwd_qkv = spy
bad = 0
for name, (nf, crop, batch, frames) in {"tiny 4x64^2": (4, 64, 2, False), "odd 4x88^2": (4, 88, 2, False), "T'=1 4x64^2": (4, 64, 3, True),
                                        "C1 8x224^2": (8, 224, 1, False), "C2 16x224^2 B=8": (16, 224, 8, False),
                                        "C4 32x224^2 B=4": (32, 224, 4, False), "C5 16x312^2 B=4": (16, 312, 4, False),
                                        "16x224^2 B=1": (16, 224, 1, False), "image rank 63 stills 224^2": (16, 224, 63, True)}.items():
    cfg, model, spec, sd = S.build_hip_model(nf, crop)
    x = P.frames(batch, 1 if frames else nf, crop)
    paths.clear()
    logits, extra = model([x.cuda()], {})
    logits.float().sum().backward()
    torch.cuda.synchronize()
    print("%-28s blocks 15..0: %s" % (name, "".join(str(p) for p in paths)), flush=True)
    bad += sum(1 for p in paths if p != 1)
    del model
    torch.cuda.empty_cache()
print("launches that fell back to the streaming kernels:", bad)
