"""Build libsvit_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m svit_amd.build            # incremental
    python -m svit_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting svit_amd/lib/libsvit_hip.so travels to the GPU box with the source tree.
"""
import concurrent.futures
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libsvit_hip.so")
SOURCES = ["gemm_nt.hip", "gemm_tn.hip", "norm.hip", "misc.hip", "pool.hip", "attn_fwd.hip", "attn_bwd.hip",
           "loss.hip", "meter.hip", "input.hip"]
HEADERS = ["common.h", "attn_common.h", "gemm_epilogue.h", os.path.join("..", "..", "include", "svit_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent fp32 multiplies / adds into v_pk_*_f32.  One of
# the forms it emits -- destination pair == a source pair, with op_sel / op_sel_hi crossing the
# halves of that pair, e.g. `v_pk_mul_f32 v[10:11], v[4:5], v[10:11] op_sel:[0,1]` -- returned wrong
# low halves on MI355X whenever waves of another kernel (the TN weight-gradient GEMM) were
# co-resident on the SIMD: the pooling-conv weight gradient lost exactly the three taps computed
# from such a result (profiles/r02_wgrad_overlap_rootcause.md; tools/diag/).  Scalar fp32 VALU code is
# also what the CDNA guide recommends beside MFMAs.  check_isa() below refuses a build that still
# contains the form.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math",
         "-fno-finite-math-only", "-fno-slp-vectorize", "-Wno-unused-result"]
TMP_DIR = os.path.join(OUT_DIR, "tmp")


_PK = re.compile(r"\s*(v_pk_(?:mul|add|fma)_f32)\s+v\[(\d+):\d+\],\s*(.*)")


def _sel(rest, name, n, default):
    m = re.search(name + r":\[([0-9,]+)\]", rest)
    v = [int(x) for x in m.group(1).split(",")] if m else []
    return v + [default] * (n - len(v))


def hazardous_packed_f32(asm_text):
    """[(line number, instruction)] of packed-fp32 VALU instructions whose destination pair is also
    a source pair read ACROSS its halves (low result from the high register or vice versa)."""
    bad = []
    for ln, line in enumerate(asm_text.splitlines(), 1):
        m = _PK.match(line)
        if not m:
            continue
        d0, rest = int(m.group(2)), m.group(3)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest.split(" op_sel")[0].split(" neg_")[0])]
        sel = _sel(rest, "op_sel", len(ops), 0) if re.search(r"op_sel:\[", rest) else [0] * len(ops)
        sel_hi = _sel(rest, "op_sel_hi", len(ops), 1)
        for i, o in enumerate(ops):
            mm = re.match(r"v\[(\d+):\d+\]", o)
            if mm and int(mm.group(1)) == d0 and (sel[i] == 1 or sel_hi[i] == 0):
                bad.append((ln, line.strip()))
                break
    return bad


def check_isa():
    """scan the device assembly kept by the last build (lib/tmp/*.s)"""
    found = []
    for src in SOURCES:
        path = os.path.join(TMP_DIR, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if os.path.exists(path):
            with open(path) as f:
                found += [(src, ln, ins) for ln, ins in hazardous_packed_f32(f.read())]
    if found:
        raise RuntimeError("in-place cross-half packed-fp32 instructions in the device code "
                           "(wrong results beside other kernels on gfx950):\n" +
                           "\n".join("%s:%d %s" % x for x in found[:10]))
    return True


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _newer(obj, deps + [os.path.abspath(__file__)]):
        # -save-temps keeps the device assembly next to a scratch object for check_isa()
        tmp_obj = os.path.join(TMP_DIR, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", tmp_obj]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=TMP_DIR)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        os.replace(tmp_obj, obj)
        for f in os.listdir(TMP_DIR):          # keep only the device assembly
            stem = src.replace(".hip", "")
            if (f.startswith(stem + "-") or f.startswith(stem + ".")) and not f.endswith("gfx950.s"):
                os.remove(os.path.join(TMP_DIR, f))
    return obj


def build(force=False, verbose=False):
    os.makedirs(TMP_DIR, exist_ok=True)
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
    check_isa()
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
