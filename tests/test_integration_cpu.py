"""INTEGRATION.md section A, executed: the three registry lines applied to the IMPORTED reference
(slowfast/models/build.py:9,20-75, video_model_builder.py:24,156-165).  Build container only -- the
reference never travels, so the test skips where /root/reference is absent (GPU box).  Runs in a child
process: the import shim installs stand-in `fvcore` / `torchvision` modules that must not leak into
the other tests of this process."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def _section_a_lines():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## A."):text.index("## B.")]
    return re.search(r"```python\n(.*?)```", sec, re.S).group(1)


def test_section_a_is_what_the_check_executes():
    """The lines the check script runs are the lines the document shows."""
    doc = [l.split("#")[0].strip() for l in _section_a_lines().splitlines()]
    script = open(os.path.join(ROOT, "oracle", "check_integration_a.py")).read()
    for line in doc[1:]:           # (the first line is the reference's own existing import)
        if line:
            assert any(line == s.split("#")[0].strip() for s in script.splitlines()), line


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")
def test_registry_swap_on_the_imported_reference():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "check_integration_a.py")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ref_class"] == "slowfast.models.video_model_builder.SViT"
    assert out["hip_class"] == "svit_amd.model.SViT"
    assert out["cfg_class"].startswith("fvcore.common.config") or out["cfg_class"].endswith("CfgNode")
    assert out["n_ref"] == out["n_hip"] == 405 and out["numel"] == 34373560
    assert out["same_names_in_order"]
    assert out["shape_mismatches"] == [] and out["dtype_mismatches"] == []
    assert out["pool_kv_stride_hip"] == out["pool_kv_stride_ref"] and len(out["pool_kv_stride_hip"]) == 16
    assert out["no_weight_decay_hip"] == out["no_weight_decay_ref"]
    assert out["strict_load"] == [[], []] and out["round_trip_bit_equal"]
    assert out["optimizer_groups_hip"] == out["optimizer_groups_ref"]
    assert out["cpu_forward"] == "SvitHipError"
