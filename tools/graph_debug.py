import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import procedural as P
from tests import smoke_impl as S
from svit_amd.graph import GraphedTrainStep

def ce(p, e, l): return torch.nn.functional.cross_entropy(p, l)
cfg, model, spec, sd = S.build_hip_model(4, 64)
xa, ya = P.frames(2, 4, 64).cuda(), P.labels(2).cuda()
xb = (xa.flip(0) * 0.5 + 0.1).contiguous(); yb = (ya + 3) % 174
def eager(x, y):
    model.flat.grad.zero_()
    logits, _ = model([x], {})
    torch.nn.functional.cross_entropy(logits, y).backward()
    torch.cuda.synchronize()
    return model.flat.grad.clone()
ra, rb = eager(xa, ya), eager(xb, yb)
print("eager noise", float((eager(xa, ya) - ra).abs().max()), float((eager(xb, yb) - rb).abs().max()))
step = GraphedTrainStep(model, ce, [xa], ya)
print("segments:", [k for k, _ in step.segments])
for it, (x, y, ref) in enumerate([(xa, ya, ra), (xb, yb, rb)] * 15):
    from svit_amd import engine as E
    E.SNAP.clear()
    step([x], y)
    torch.cuda.synchronize()
    for bi, snaps, finals in E.SNAP:
        for w, (s, f) in enumerate(zip(snaps, finals)):
            if not torch.equal(s, f):
                print("replay", it, "SNAPDIFF block", bi, "which", w, float((s.float() - f.float()).abs().max()))
    g = model.flat.grad
    worst = []
    for n, p in model.named_parameters():
        a = model.flat.g(n)
        off = (a.data_ptr() - g.data_ptr()) // 4
        r = ref.view(-1)[off: off + a.numel()].view_as(a)
        worst.append((float((a - r).abs().max()), float(r.abs().max()), n))
    worst.sort(reverse=True)
    if worst[0][0] > 1e-6: print("replay", it, "worst:", [w for w in worst[:6] if w[0] > 1e-6])

import gc
stor = {}
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            s = o.untyped_storage()
            stor[s.data_ptr()] = max(stor.get(s.data_ptr(), 0), s.nbytes())
    except Exception:
        pass
iv = sorted((p, p + n) for p, n in stor.items() if n > 0)
ov = 0
for (a0, a1), (b0, b1) in zip(iv, iv[1:]):
    if b0 < a1:
        ov += 1
        if ov <= 5: print("OVERLAP live storages", hex(a0), a1 - a0, hex(b0), b1 - b0)
print("live cuda storages", len(iv), "overlaps", ov)
