"""Diagnostic for the round-1 finding "pooling-conv wgrad partial sums are not reproducible when
co-scheduled with GEMM workgroups of another queue" (DESIGN.md section 4).

Victim = svit_pool_conv_wgrad_qkv on a side stream (its per-workgroup partial rows are read back,
so the second-stage reduce is out of the picture); partner = six launches of another kernel on
the main stream.  For every (victim placement, partner) pair: how many of N runs differ from the
first, WHICH rows differ (workgroup, which of q/k/v, whether the row belongs to a workgroup that
exists at all) and by how much.

    python tools/stress_concurrency.py [runs]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from svit_amd import hip, ops

torch.manual_seed(0)
DEV = "cuda"
dev0 = torch.device("cuda", 0)
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, h, thw, O, sq, skv = 8, 4, (8, 14, 14), 64, 1, 2
N = 1 + thw[0] * thw[1] * thw[2] + O
qkv = (torch.randn(B, N, 3, h, 96, device=DEV) * 0.5).bfloat16()
strides = (sq, skv, skv)
dpres = []
for s in strides:
    Nout = 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + O
    dpres.append((torch.randn(B, h, Nout, 96, device=DEV) * 0.5).bfloat16())
M = 13064
a16 = (torch.randn(M, 384, device=DEV) * 0.5).bfloat16()
w16 = (torch.randn(1536, 384, device=DEV) * 0.1).bfloat16()
Nq, Nk = 1633, 457
qa = (torch.randn(B, h, Nq, 128, device=DEV) * 0.3).bfloat16()
ka = (torch.randn(B, h, Nk, 128, device=DEV) * 0.3).bfloat16()
vv = (torch.randn(B, h, Nk, 96, device=DEV) * 0.3).bfloat16()
big = torch.randn(64 * 1024 * 1024, device=DEV)
nt_out = torch.empty((M, 1536), device=DEV, dtype=torch.bfloat16)
tn_out = torch.zeros(384, 384, device=DEV)
ROW = 3 * 27 * 96          # floats per workgroup row: [which][c][tap]
WS_ROWS = 1024



def _streaming():
    """the streaming conv-backward wrappers (diagnostic build only since round 6: tools/diag/pool_streaming.py)"""
    from tools.diag import pool_streaming
    return pool_streaming


def partner(kind):
    if kind == "nt":
        ops.gemm_nt(a16, w16, None, hip.EPI_BF16, out=nt_out)
    elif kind == "attn":
        ops.attn_fwd(qa, ka, vv, 96 ** -0.5)
    elif kind == "tn":
        ops.gemm_tn(a16[:, :384], a16, tn_out)
    elif kind == "stream":          # HBM-streaming elementwise kernel: no LDS, no MFMA
        big.mul_(1.0000001)
    elif kind == "none":
        pass


def victim(ws):
    ws.fill_(-7.0)                 # rows no workgroup writes stay recognisable
    dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
    _streaming().pool_conv_wgrad_qkv(dpres, qkv, dws, B, h, thw, O, strides, ws=ws)
    return ws[:WS_ROWS * ROW].clone(), torch.stack(dws)


qkv0 = qkv.clone()
dp0 = [d.clone() for d in dpres]
side = torch.cuda.Stream()
ws_side = torch.empty(8 * 1024 * 1024, device=DEV)
torch.cuda.synchronize()

for placement in ("victim on side stream", "victim on main stream"):
    for pk in ("none", "stream", "nt", "tn", "attn"):
        ref_rows = ref_dw = None
        bad_rows = bad_dw = 0
        detail = []
        for it in range(RUNS):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            if placement.endswith("side stream"):
                with torch.cuda.stream(side):
                    rows, dw = victim(ws_side)
                for _ in range(6):
                    partner(pk)
            else:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        partner(pk)
                rows, dw = victim(ws_side)
            main.wait_stream(side)
            torch.cuda.synchronize()
            if ref_rows is None:
                ref_rows, ref_dw = rows.clone(), dw.clone()
                written = (ref_rows.view(WS_ROWS, 3, -1) != -7.0).any(dim=2)     # [row, which]
                continue
            if not torch.equal(dw, ref_dw):
                bad_dw += 1
            if not torch.equal(rows, ref_rows):
                bad_rows += 1
                if len(detail) < 3:
                    d = (rows - ref_rows).view(WS_ROWS, 3, -1).abs()
                    hit = (d.amax(dim=2) > 0).nonzero()
                    unwritten = sum(1 for r, w in hit.tolist() if not bool(written[r, w]))
                    detail.append("run %d: %d (row, which) pairs differ, %d of them in rows no workgroup "
                                  "writes; first %s; max |diff| %.3g (|ref| max %.3g)"
                                  % (it, len(hit), unwritten, hit[:6].tolist(), float(d.max()),
                                     float(ref_rows[ref_rows != -7.0].abs().max())))
        print("%-22s partner %-6s: partial rows differ in %d/%d runs, reduced dw in %d/%d; rows written %d; "
              "inputs intact %s" % (placement, pk, bad_rows, RUNS - 1, bad_dw, RUNS - 1,
                                    int(written.any(dim=1).sum()),
                                    torch.equal(qkv, qkv0) and all(torch.equal(a, b) for a, b in zip(dpres, dp0))))
        for line in detail:
            print("     ", line)
