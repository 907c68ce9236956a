"""The C-ABI library loads and exports every symbol include/svit_hip.h declares (no GPU needed,
no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "svit_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    # declarations of DIAGNOSTIC builds (#ifdef SVIT_DIAG_... blocks: entry points the product library does not export)
    text = re.sub(r"#ifdef SVIT_DIAG_\w+.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svit_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding():
    from svit_amd import hip
    assert header_functions() == list(hip.EXPORTS)


def test_library_loads_and_exports_everything():
    import __graft_entry__
    __graft_entry__.build()
    from svit_amd import hip
    lib = hip.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.svit_version() >= 1
    assert lib.svit_arch() == b"gfx950"


def test_library_exports_nothing_the_header_does_not_declare():
    """Every `svit_*` function symbol of the shipped .so is declared in include/svit_hip.h (the tuning knobs of
    tools/ sit in its diagnostics block) -- no undeclared entry points."""
    import shutil
    import subprocess
    import __graft_entry__
    lib = __graft_entry__.build()
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("svit_")})
    assert exported == header_functions()


def test_argument_validation_without_gpu():
    """Host-side shape/argument checks reject bad calls before any launch."""
    import ctypes as C
    from svit_amd import hip
    lib = hip.load()
    g = hip.GemmArgs()
    assert lib.svit_gemm_nt(C.byref(g), None) == -4          # null pointers
    g.A, g.W, g.out = 16, 16, 16
    g.M, g.N, g.K, g.lda, g.ldw, g.ldo = 8, 100, 96, 96, 96, 100
    assert lib.svit_gemm_nt(C.byref(g), None) == -2          # N % 96 != 0
    a = hip.AttnFwdArgs()
    a.qa = a.ka = a.v = a.ctx = a.lse2 = 16
    a.B, a.heads, a.Nq, a.Nk, a.DA = 1, 1, 4, 4, 96
    assert lib.svit_attn_fwd(C.byref(a), None) == -2         # DA must be 128 or 160


def test_product_path_has_no_oracle_import():
    """svit_amd/ must never import the CPU oracle (test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "svit_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_no_cross_half_packed_fp32_in_the_device_code():
    """profiles/r03_packed_fp32_hazard.md: `v_pk_*_f32` reading a VGPR pair across its halves
    (op_sel / op_sel_hi) gave wrong results beside the TN GEMM and beside hipBLASLt on MI355X,
    destination == source or not.  check_isa() disassembles the SHIPPED .so and refuses it; the
    scanner itself is checked on the offending forms and on the two measured-clean ones."""
    from svit_amd import build
    bad = build.hazardous_packed_f32(
        "\tv_pk_mul_f32 v[10:11], v[4:5], v[10:11] op_sel:[0,1]\n"
        "\tv_pk_mul_f32 v[10:11], v[10:11], v[16:17] op_sel_hi:[1,0]\n"
        "\tv_pk_fma_f32 v[76:77], v[74:75], s[12:13], v[76:77] op_sel_hi:[1,0,0]\n"
        "\tv_pk_add_f32 v[2:3], v[4:5], v[6:7]\n"
        "\tv_pk_mul_f32 v[8:9], v[6:7], v[4:5] op_sel:[0,1]          // 000000001C34: D3B14008\n"
        "\tv_pk_fma_f32 v[20:21], v[50:51], s[10:11], v[20:21] op_sel_hi:[1,0,1]\n")
    assert [ln for ln, _ in bad] == [1, 2, 3, 5]
    assert "-fno-slp-vectorize" in build.FLAGS
    build.build()
    assert build.check_isa() >= len(build.SOURCES)      # number of code objects actually scanned
