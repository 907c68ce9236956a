#!/usr/bin/env python3
"""Build a diagnostic variant of the WHOLE library with changed compiler flags:
   python tools/diag/build_all_variant.py <tag> [--drop FLAG ...] [--add FLAG ...]  -> tools/diag/libsvit_diag_<tag>.so"""
import concurrent.futures, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svit_amd import build as b
tag = sys.argv[1]
drop, add, mode = [], [], None
for a in sys.argv[2:]:
    if a in ("--drop", "--add"):
        mode = a
    elif mode == "--drop":
        drop.append(a)
    else:
        add.append(a)
flags = [f for f in b.FLAGS if f not in drop] + add
out_dir = os.path.join(ROOT, "tools", "diag", "build")
os.makedirs(out_dir, exist_ok=True)


def comp(src):
    obj = os.path.join(out_dir, "%s_%s.o" % (src.replace(".hip", ""), tag))
    r = subprocess.run([b.HIPCC] + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr)
    return obj


with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(comp, b.SOURCES))
lib = os.path.join(ROOT, "tools", "diag", "libsvit_diag_%s.so" % tag)
r = subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr)
print(lib, "flags:", " ".join(flags))
