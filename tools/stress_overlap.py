"""pool-conv weight grads come from a deterministic two-stage reduction: any run-to-run change of
them means the side-stream overlap raced.  python tools/stress_overlap.py [frames crop batch iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import config
from svit_amd.model import build_model
fr, crop, B, iters = [int(a) for a in (sys.argv[1:5] + ["16", "224", "8", "12"][len(sys.argv) - 1:])]
cfg = config.ssv2_cfg(num_frames=fr, crop=crop, num_gpus=1)
cfg.MVIT.DROPPATH_RATE = 0.0; cfg.MODEL.DROPOUT_RATE = 0.0
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0); model.train()
x = torch.randn(B, 3, fr, crop, crop, device="cuda"); y = torch.randint(0, 174, (B,), device="cuda")
model.engine.attn_q_splits = 1          # no atomics upstream of the pool / norm grads
names = [n for n, _ in model.named_parameters() if ".pool_" in n or "norm" in n]
def grads(overlap):
    model.engine.overlap_wgrad = overlap
    model.flat.grad.zero_()
    logits, _ = model([x], {})
    torch.nn.functional.cross_entropy(logits, y).backward()
    torch.cuda.synchronize()
    return {n: model.flat.g(n).clone() for n in names}
ref = grads(False)
for ov in (False, True):
    bad = 0
    for it in range(iters):
        g = grads(ov)
        diff = [(float((g[n] - ref[n]).abs().max()), n) for n in names if not torch.equal(g[n], ref[n])]
        if diff:
            bad += 1
            if bad <= 3: print("  overlap", ov, "iter", it, sorted(diff, reverse=True)[:3])
    print("overlap=%s: %d/%d steps with changed pool/norm grads" % (ov, bad, iters))
