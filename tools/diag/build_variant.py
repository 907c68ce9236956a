#!/usr/bin/env python3
"""Build a diagnostic variant of libsvit_hip.so: ONE source (or a comma-separated list) recompiled with extra -D flags,
linked with the product objects of the others.   python tools/diag/build_variant.py <tag> <source.hip[,other.hip]> -DFOO=1 [-DBAR ...]
-> tools/diag/libsvit_diag_<tag>.so   (use with SVIT_HIP_LIB=<path>; git-ignored, travels with gpurun)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svit_amd import build as b
tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
out_dir = os.path.join(ROOT, "tools", "diag", "build")
os.makedirs(out_dir, exist_ok=True)
mine = {}
for one in src.split(","):
    obj = os.path.join(out_dir, "%s_%s.o" % (one.replace(".hip", ""), tag))
    r = subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, one), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    mine[one] = obj
objs = [mine.get(s, os.path.join(b.OUT_DIR, s.replace(".hip", ".o"))) for s in b.SOURCES]
lib = os.path.join(ROOT, "tools", "diag", "libsvit_diag_%s.so" % tag)
r = subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
print(lib)
