#!/usr/bin/env python3
"""Per-tensor gradient cosines of the rel-pos tables (HIP bf16 path vs fp32 oracle) at the smoke configuration and with
more clips -- where does blocks.0.attn.rel_pos_h's 0.5 % go?  GPU box."""
import io
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import smoke_impl as S

for frames, crop, batch in ((4, 64, 2), (4, 64, 8), (8, 224, 1)):
    buf = io.StringIO()
    with redirect_stdout(buf):
        res = S.compare_step(frames, crop, batch, verbose=True)
    rows = [l for l in buf.getvalue().splitlines() if "rel_pos" in l]
    rows.sort(key=lambda l: float(l.split("cos")[1].split()[0]))
    print("== %dx%d^2, %d clip(s): worst %s %.5f, global %.5f" % (frames, crop, batch, res["grad_cos_worst_name"],
                                                                  res["grad_cos_worst"], res["grad_cos_global"]))
    for l in rows[:8]:
        print("   " + l)
