"""Build libsvit_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m svit_amd.build            # incremental
    python -m svit_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting svit_amd/lib/libsvit_hip.so travels to the GPU box with the source tree.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libsvit_hip.so")
SOURCES = ["gemm_nt.hip", "gemm_tn.hip", "norm.hip", "misc.hip", "pool.hip", "attn_fwd.hip", "attn_fwd2.hip", "attn_bwd.hip",
           "loss.hip", "meter.hip", "input.hip"]
HEADERS = ["common.h", "attn_common.h", "gemm_epilogue.h", os.path.join("..", "..", "include", "svit_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math",
         "-fno-finite-math-only", "-Wno-unused-result"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _newer(obj, deps):
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
    return obj


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OUT_DIR):
            if f.endswith((".o", ".so")):
                os.remove(os.path.join(OUT_DIR, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
