#!/usr/bin/env python3
"""HIP bf16 step against the bf16 yardstick of tests/golden/manifest.json (the reference's own backward with every matrix op's operands and
results rounded to bf16, oracle/gen_golden.py::run_yardstick_cases): per gradient tensor, cosine against the fp32 oracle for both, and the
noise-power ratio (1 - cos_hip) / (1 - cos_yardstick).     python tools/diag/yardstick_compare.py [case ...] [--pool-frame N]   (GPU box)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import smoke_impl as S
from svit_amd import hip

args = sys.argv[1:]
pf = None
if "--pool-frame" in args:
    i = args.index("--pool-frame")
    pf = int(args[i + 1])
    del args[i:i + 2]
man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["yardstick"]["cases"]
lib = hip.load()
for name in (args or list(man)):
    c = man[name]
    if pf is not None:
        assert lib.svit_debug_set_pool(3, pf) == 0
    res = S.compare_step(c["num_frames"], c["crop"], c["batch"], frames_path=c["kind"] == "frames", image=c["kind"] == "image")
    lib.svit_debug_reset()
    yc, hc = c["autocast_emulation_cos"], res["grad_cos_per_tensor"]
    rows = sorted(((1 - hc[k]) / max(1 - yc[k], 1e-7), k) for k in hc if k in yc)
    print("== %s%s: HIP worst %.5f (%s) global %.5f | yardstick worst %.5f (%s) global %.5f | logits maxabs HIP %.4f yardstick %.4f"
          % (name, "" if pf is None else " [pool frame knob %d]" % pf, res["grad_cos_worst"], res["grad_cos_worst_name"], res["grad_cos_global"],
             c["grad_cos_worst"][1], c["grad_cos_worst"][0], c["grad_cos_global"], res["logits_maxabs"], c["logits_maxabs"]))
    print("   noise-power ratio (1 - cos_hip) / (1 - cos_yardstick): median %.2f, max %.2f; the ten largest:" % (rows[len(rows) // 2][0], rows[-1][0]))
    for r, k in rows[-10:][::-1]:
        print("     %6.2f  %-44s hip %.5f  yardstick %.5f" % (r, k, hc[k], yc[k]))
    rel = [(hc[k], yc[k], k) for k in hc if "rel_pos" in k and k in yc]
    w = min(rel)
    print("   worst rel-pos table: %s hip %.5f yardstick %.5f" % (w[2], w[0], w[1]), flush=True)
