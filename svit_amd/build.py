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
           "loss.hip", "meter.hip", "input.hip", "head.hip"]
HEADERS = ["common.h", "attn_common.h", "gemm_epilogue.h", os.path.join("..", "..", "include", "svit_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent fp32 multiplies / adds into v_pk_*_f32.  The
# forms it emits that read a VGPR pair across its halves (op_sel / op_sel_hi), e.g.
# `v_pk_mul_f32 v[10:11], v[4:5], v[10:11] op_sel:[0,1]`, return wrong results on MI355X at a rate
# of ~6e-6 whenever waves of the TN weight-gradient GEMM -- or of hipBLASLt's GEMM -- are co-resident
# (standalone reproducer tools/diag/slp_repro.hip, profiles/r03_packed_fp32_hazard.md; round 2 found
# it as three lost taps of the pooling-conv weight gradient, profiles/r02_wgrad_overlap_rootcause.md).
# The SLP build is also 0.3 % SLOWER per step (14.57 vs 14.51 ms, tools/diag/slp_ab.py), as the CDNA
# guide predicts for packed fp32 beside MFMAs.  check_isa() below refuses a library that contains
# such an instruction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffast-math",
         "-fno-finite-math-only", "-fno-slp-vectorize", "-Wno-unused-result"]
TMP_DIR = os.path.join(OUT_DIR, "tmp")


_PK = re.compile(r"\s*(v_pk_(?:mul|add|fma)_f32)\s+v\[(\d+):\d+\],\s*(.*)")


def _sel(rest, name, n, default):
    m = re.search(name + r":\[([0-9,]+)\]", rest)
    v = [int(x) for x in m.group(1).split(",")] if m else []
    return v + [default] * (n - len(v))


def hazardous_packed_f32(asm_text):
    """[(line number, instruction)] of packed-fp32 VALU instructions that read a VGPR pair ACROSS its
    halves (low result from the high register or the high result from the low one: op_sel 1 /
    op_sel_hi 0 on a VGPR-pair source).  Round 3 (tools/diag/slp_repro.hip,
    profiles/r03_packed_fp32_hazard.md): every such form -- destination == source or not -- returned
    wrong results at a rate of ~6e-6 while waves of the TN GEMM or of hipBLASLt's GEMM were
    co-resident; no cross-half selection, or a cross-half selection on an SGPR pair (scalar
    broadcast), never did.  Round 2 had only caught the in-place form."""
    bad = []
    for ln, line in enumerate(asm_text.splitlines(), 1):
        m = _PK.match(line)
        if not m:
            continue
        rest = m.group(3).split("//")[0]
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest.split(" op_sel")[0].split(" neg_")[0])]
        sel = _sel(rest, "op_sel", len(ops), 0) if re.search(r"op_sel:\[", rest) else [0] * len(ops)
        sel_hi = _sel(rest, "op_sel_hi", len(ops), 1)
        for i, o in enumerate(ops):
            if re.match(r"v\[(\d+):\d+\]", o) and (sel[i] == 1 or sel_hi[i] == 0):
                bad.append((ln, line.split("//")[0].strip()))
                break
    return bad


LLVM_OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def disassemble(lib=None):
    """Device code of the SHIPPED library as text: llvm-objdump unbundles the gfx950 code objects of
    the .so (into a scratch copy's directory) and disassembles each.  -> (text, number of objects)"""
    import shutil
    import tempfile
    lib = lib or LIB
    tmp = tempfile.mkdtemp(prefix="svit_isa_")
    try:
        cp = os.path.join(tmp, "lib.so")
        shutil.copy(lib, cp)
        r = subprocess.run([LLVM_OBJDUMP, "--offloading", cp], capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError("llvm-objdump --offloading failed: %s" % r.stderr[-300:])
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f)
        text = []
        for f in objs:
            d = subprocess.run([LLVM_OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)],
                               capture_output=True, text=True)
            if d.returncode != 0:
                raise RuntimeError("llvm-objdump -d failed on %s: %s" % (f, d.stderr[-300:]))
            text.append(d.stdout)
        return "\n".join(text), len(objs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check_isa(lib=None):
    """Scan the device code of the library that SHIPS (disassembled from the .so itself, so the check
    means the same on the GPU box, where the build's scratch assembly does not travel) for in-place
    cross-half packed-fp32 instructions.  Raises if it finds one -- or if it could not look: a gate
    that scanned nothing must not read as a pass.  -> number of code objects scanned."""
    lib = lib or LIB
    if not os.path.exists(LLVM_OBJDUMP):
        raise RuntimeError("check_isa: %s not found -- the library's device code was NOT checked" % LLVM_OBJDUMP)
    text, nobj = disassemble(lib)
    if nobj < len(SOURCES) or "v_mfma_f32_32x32x16_bf16" not in text:
        raise RuntimeError("check_isa: only %d gfx950 code objects found in %s (expected %d) -- not checked"
                           % (nobj, lib, len(SOURCES)))
    found = hazardous_packed_f32(text)
    if found:
        raise RuntimeError("in-place cross-half packed-fp32 instructions in the device code "
                           "(wrong results beside other kernels on gfx950):\n" +
                           "\n".join("%d %s" % x for x in found[:10]))
    return nobj


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
