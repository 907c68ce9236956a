import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import procedural as P
from tests import smoke_impl as S
from svit_amd.graph import GraphedTrainStep

def ce(p, e, l): return torch.nn.functional.cross_entropy(p, l)
cfg, model, spec, sd = S.build_hip_model(4, 64)
xa, ya = P.frames(2, 4, 64).cuda(), P.labels(2).cuda()
model.flat.grad.zero_()
logits, _ = model([xa], {})
torch.nn.functional.cross_entropy(logits, ya).backward()
torch.cuda.synchronize()
ref = model.flat.grad.clone()
step = GraphedTrainStep(model, ce, [xa], ya)
for it in range(2):
    step([xa], ya)
    torch.cuda.synchronize()
    g = model.flat.grad
    bad = []
    for n, p in model.named_parameters():
        a, b = model.flat.g(n), None
        off = a.data_ptr() - g.data_ptr()
        r = ref.view(-1)[off // 4: off // 4 + a.numel()].view_as(a)
        d = float((a - r).abs().max())
        if not d <= 1e-3 * max(1e-6, float(r.abs().max())):
            bad.append((n, d, float(r.abs().max())))
    print("replay", it, "bad params:", len(bad))
    for b in bad[:40]:
        print("   ", b)
