#!/usr/bin/env python3
"""Why every rel-pos table gradient sits at cosine ~0.995 against the fp32 oracle while every other tensor is at ~0.9999
(profiles/r05_relpos_cos.txt).  CPU experiment on synthetic tensors with the statistics of block 0 at
random init (near-uniform attention, residual pooling): the attention backward's  dS = P o (dP - delta)  with
delta = rowsum(dO o O), O = ctx - q taken from the STORED bf16 ctx = bf16(O + q)  (what a flash-style backward does)
against delta from the exact O.  A row's delta error is COMMON-MODE over the keys: it cancels in dq = dS K (keys have
zero mean) but not in d(bias)[q, j] = sum of dS over the key group j, which is what the table gradients are made of."""
import torch

torch.manual_seed(0)
B, Nq, gh, gw, gt, C = 2, 3136, 7, 7, 8, 96          # keys on a gh x gw x gt grid, queries random positions
Nk = gh * gw * gt
bf = lambda x: x.to(torch.bfloat16).to(torch.float64)
q = torch.nn.functional.layer_norm(torch.randn(B, Nq, C, dtype=torch.float64), (C,))
k = torch.nn.functional.layer_norm(torch.randn(B, Nk, C, dtype=torch.float64), (C,))
v = torch.nn.functional.layer_norm(torch.randn(B, Nk, C, dtype=torch.float64), (C,))
Rh = (torch.randn(gh, C, dtype=torch.float64) * 0.02).requires_grad_(True)     # one table row per key row (dist index = key row here)
ky = torch.arange(Nk) // (gw * gt) % gh
onehot = torch.nn.functional.one_hot(ky, gh).to(torch.float64)                 # [Nk, gh]
dO = torch.randn(B, Nq, C, dtype=torch.float64)


def table_grad(delta_from):
    """d(loss)/dRh with dS = P (dP - delta); delta_from: 'exact' | 'stored_bf16_ctx' | 'bf16_everything'"""
    relq = q @ Rh.detach().t()                                                # [B, Nq, gh]
    S = (q @ k.transpose(1, 2)) * C ** -0.5 + relq @ onehot.t()
    P = torch.softmax(S, -1)
    O = P @ v
    dP = dO @ v.transpose(1, 2)
    if delta_from == "exact":
        delta = (dO * O).sum(-1, keepdim=True)
    else:
        ctx = bf(O + q)                                                       # residual pooling, stored bf16
        delta = (bf(dO) * (ctx - bf(q))).sum(-1, keepdim=True)
    if delta_from == "bf16_everything":
        P, dP = bf(P), (bf(dO) @ bf(v).transpose(1, 2))
    dS = P * (dP - delta)
    if delta_from == "bf16_everything":
        dS = bf(dS)
    drelq = dS @ onehot                                                       # [B, Nq, gh]: group sums over the keys of a row
    dq = dS @ k * C ** -0.5
    return torch.einsum("bqj,bqc->jc", drelq, q), dq


ref, dq_ref = table_grad("exact")
cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
for mode in ("stored_bf16_ctx", "bf16_everything"):
    g, dq = table_grad(mode)
    print("%-18s table-gradient cosine %.5f   dq cosine %.6f" % (mode, cos(g, ref), cos(dq, dq_ref)))
O = torch.softmax((q @ k.transpose(1, 2)) * C ** -0.5, -1) @ v
print("|O| rms %.3f vs |q| rms %.3f: ctx = O + q stored in bf16 keeps O to %.1f %% per element" %
      (float(O.pow(2).mean().sqrt()), float(q.pow(2).mean().sqrt()), 100 * float((bf(O + q) - bf(q) - O).pow(2).mean().sqrt() / O.pow(2).mean().sqrt())))
