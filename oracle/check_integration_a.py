#!/usr/bin/env python3
"""Executes INTEGRATION.md section A against the IMPORTED reference (build container only).

TEST INFRASTRUCTURE ONLY (run by tests/test_integration_cpu.py in a child process; needs
/root/reference, which never travels).  Steps:
  1. import the unmodified reference's `slowfast.models` (plumbing stand-ins: oracle/ref_shim.py) and build
     its own SViT from its own `get_cfg()` + configs/ssv2.yaml through its own `build_model`
     (slowfast/models/build.py:20-75) -> state_dict layout, cfg.MVIT.POOL_KV_STRIDE as the constructor
     leaves it (video_model_builder.py:156-165), no_weight_decay();
  2. apply the three lines of INTEGRATION.md section A to the reference's MODEL_REGISTRY;
  3. call the reference's `build_model` again on a fresh reference cfg: it must now return
     `svit_amd.model.SViT`, with the same 405 names / order / shapes / dtypes, the same POOL_KV_STRIDE
     written into the cfg, the same no_weight_decay(), and it must load the reference model's
     state_dict with strict=True (and hand it back bit for bit).
Prints one JSON line."""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
import slowfast.models as ref_models  # noqa: E402  (the reference's package: registers its SViT)
from slowfast.models import MODEL_REGISTRY, build_model  # noqa: E402

torch.manual_seed(0)
cfg_ref = ref_shim.reference_cfg()
ref_model = build_model(cfg_ref)
ref_cls = type(ref_model)
ref_sd = ref_model.state_dict()

# ---- INTEGRATION.md section A, verbatim ------------------------------------------------------------
import svit_amd.model as _hip                    # noqa: E402
MODEL_REGISTRY._obj_map["SViT"] = _hip.SViT      # fvcore Registry: replace the ATen-op SViT
# ----------------------------------------------------------------------------------------------------

cfg_hip = ref_shim.reference_cfg()
hip_model = build_model(cfg_hip)                 # the REFERENCE's build_model, the reference's cfg class
hip_sd = hip_model.state_dict()

out = {
    "ref_class": ref_cls.__module__ + "." + ref_cls.__name__,
    "hip_class": type(hip_model).__module__ + "." + type(hip_model).__name__,
    "n_ref": len(ref_sd), "n_hip": len(hip_sd),
    "same_names_in_order": list(ref_sd.keys()) == list(hip_sd.keys()),
    "shape_mismatches": [k for k in ref_sd if k in hip_sd and tuple(ref_sd[k].shape) != tuple(hip_sd[k].shape)],
    "dtype_mismatches": [k for k in ref_sd if k in hip_sd and ref_sd[k].dtype != hip_sd[k].dtype],
    "numel": int(sum(v.numel() for v in hip_sd.values())),
    "pool_kv_stride_ref": [list(x) for x in cfg_ref.MVIT.POOL_KV_STRIDE],
    "pool_kv_stride_hip": [list(x) for x in cfg_hip.MVIT.POOL_KV_STRIDE],
    "no_weight_decay_ref": sorted(ref_model.no_weight_decay()),
    "no_weight_decay_hip": sorted(hip_model.no_weight_decay()),
    "cfg_class": type(cfg_hip).__module__ + "." + type(cfg_hip).__name__,
}
missing, unexpected = hip_model.load_state_dict(ref_sd, strict=True)
out["strict_load"] = (list(missing), list(unexpected))
back = hip_model.state_dict()
out["round_trip_bit_equal"] = all(torch.equal(back[k], ref_sd[k]) for k in ref_sd)
# the reference's optimizer construction reads only named_parameters() / no_weight_decay(): same groups
from slowfast.models import optimizer as ref_optim  # noqa: E402
g_ref = ref_optim.construct_optimizer(ref_model, cfg_ref).param_groups
g_hip = ref_optim.construct_optimizer(hip_model, cfg_hip).param_groups
out["optimizer_groups_ref"] = [(len(g["params"]), g["weight_decay"]) for g in g_ref]
out["optimizer_groups_hip"] = [(len(g["params"]), g["weight_decay"]) for g in g_hip]
# without a GPU the HIP model must refuse to run, not fall back
try:
    hip_model([torch.zeros(1, 3, 16, 224, 224)])
    out["cpu_forward"] = "ran (WRONG: there must be no CPU path)"
except Exception as e:  # noqa: BLE001
    out["cpu_forward"] = type(e).__name__
print(json.dumps(out))
