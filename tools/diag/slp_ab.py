#!/usr/bin/env python3
"""A/B of the step time with and without -fno-slp-vectorize: builds a second
library WITH the SLP vectoriser into gpurun_out/, counts the in-place cross-half packed-fp32
instructions in both, and runs bench.py alternately against both (SVIT_HIP_LIB).
    python tools/diag/slp_ab.py    (GPU box; ~3 min)"""
import concurrent.futures
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svit_amd import build as B

out_dir = os.path.join(ROOT, "gpurun_out", "slp_build")
os.makedirs(out_dir, exist_ok=True)
flags = [f for f in B.FLAGS if f != "-fno-slp-vectorize"]


def cc(src):
    obj = os.path.join(out_dir, src.replace(".hip", ".o"))
    subprocess.check_call([B.HIPCC] + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj])
    return obj


with concurrent.futures.ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, B.SOURCES))
slp_lib = os.path.join(ROOT, "gpurun_out", "libsvit_slp.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", slp_lib] + objs)
for name, lib in (("no-slp (shipped)", B.LIB), ("slp", slp_lib)):
    text, n = B.disassemble(lib)
    import re
    print("%-18s %d code objects, %d packed-fp32 instructions, %d in-place cross-half"
          % (name, n, len(re.findall(r"v_pk_(?:mul|add|fma)_f32", text)), len(B.hazardous_packed_f32(text))), flush=True)
res = {"no-slp": [], "slp": []}
for rnd in range(3):
    for name, lib in (("no-slp", B.LIB), ("slp", slp_lib)):
        env = dict(os.environ, SVIT_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                            "--no-cpu-baseline", "--no-kernel-trace"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "bench failed:", r.stderr[-500:])
            continue
        d = json.loads(line[-1])
        res[name].append(d["ms_per_step"])
        print("round %d %-7s %.3f ms/step  loss %.4f" % (rnd, name, d["ms_per_step"], d["loss"]), flush=True)
print(json.dumps(res))
