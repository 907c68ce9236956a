"""Import shim for running the UNMODIFIED reference on CPU in the build container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py (this container, where
/root/reference exists) to produce the golden vectors under tests/golden/.
Nothing here is imported by the product path, and nothing from /root/reference
is copied into this repository: the shim only provides stand-ins for *plumbing*
packages the image lacks (iopath, fvcore, simplejson, torchvision) so that the
reference's own model/loss/optimizer files import and run on CPU.

Recipe: SURVEY.md Appendix B.
"""
import ast
import copy
import sys
import types

sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference tree

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _PathMgr:
    def open(self, path, mode="r", **kw):
        return open(path, mode)

    def exists(self, path):
        import os
        return os.path.exists(path)

    def mkdirs(self, path):
        import os
        os.makedirs(path, exist_ok=True)

    def ls(self, path):
        import os
        return os.listdir(path)

    def get_local_path(self, path, **kw):
        return path


class _PathManagerFactory:
    @staticmethod
    def get(key=None, **kw):
        return _PathMgr()


class _Registry:
    """Stand-in for fvcore.common.registry.Registry with fvcore's own attribute and method names
    (`_obj_map`, `_do_register`, duplicate names refused, KeyError on a missing name), so that what
    INTEGRATION.md section A does to the reference's MODEL_REGISTRY can be executed here."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, \
            "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map


class _CfgNode(dict):
    """Minimal yacs/fvcore CfgNode: attribute access, clone, merge with coercion."""

    def __init__(self, init=None):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = _CfgNode(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _coerce(new, old):
        if isinstance(new, str):
            try:
                new = ast.literal_eval(new)
            except (ValueError, SyntaxError):
                pass
        if old is None or new is None or type(old) is type(new):
            return new
        if isinstance(old, float) and isinstance(new, (int, str)):
            return float(new)
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        return new

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self:
                    self[k] = _CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v, self.get(k))

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for key, val in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = self._coerce(val, node.get(parts[-1]))

    def dump(self):
        import yaml
        def plain(n):
            return {k: plain(v) if isinstance(v, dict) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))


class _Timer:
    def __init__(self):
        import time
        self._t = time.time()

    def reset(self):
        import time
        self._t = time.time()

    def seconds(self):
        import time
        return time.time() - self._t

    def pause(self):
        pass

    def resume(self):
        pass


def install():
    """Install the stand-in modules and put the reference on sys.path."""
    if "slowfast" in sys.modules:
        return
    _mod("iopath")
    _mod("iopath.common")
    _mod("iopath.common.file_io", PathManagerFactory=_PathManagerFactory, g_pathmgr=_PathMgr())
    _mod("fvcore")
    _mod("fvcore.common")
    _mod("fvcore.nn")
    _mod("fvcore.common.registry", Registry=_Registry)
    _mod("fvcore.common.config", CfgNode=_CfgNode)
    _mod("fvcore.common.timer", Timer=_Timer)
    _mod("fvcore.nn.activation_count", activation_count=None)
    _mod("fvcore.nn.flop_count", flop_count=None)
    _mod("fvcore.nn.precise_bn", get_bn_modules=None, update_bn_stats=None)
    import json
    _mod("simplejson", loads=json.loads,
         dumps=lambda o, **kw: json.dumps(o, sort_keys=kw.get("sort_keys", False)))
    _mod("torchvision")
    _mod("torchvision.ops")
    _mod("torchvision.ops.boxes",
         box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    ds = _mod("slowfast.datasets")
    ds.__path__ = []
    _mod("slowfast.datasets.utils", pack_pathway_output=lambda cfg, frames: [frames])
    _mod("slowfast.datasets.ava_helper")     # AVA plumbing imported (not used) by utils/meters.py
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_cfg(num_frames=16, crop=224, overrides=()):
    """get_cfg() + configs/ssv2.yaml + CPU overrides, from the reference itself."""
    install()
    from slowfast.config.defaults import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(REFERENCE_ROOT + "/configs/ssv2.yaml")
    cfg.NUM_GPUS = 0
    cfg.DATA.NUM_FRAMES = num_frames
    cfg.DATA.TRAIN_CROP_SIZE = crop
    cfg.DATA.TEST_CROP_SIZE = crop
    if overrides:
        cfg.merge_from_list(list(overrides))
    return cfg


# ---- bf16 yardstick (round 6): the reference under an emulation of CUDA autocast, on CPU ---------------------
# tests/golden/manifest.json["yardstick"] records what a CORRECT bf16-GEMM implementation of the step scores against
# the fp32 reference, per gradient tensor -- the noise floor the HIP path's per-tensor tolerances are set against.
# Plumbing only: the reference's modules run unmodified; a TorchFunctionMode rounds the operands and the result of
# every op CUDA autocast runs in bf16 (linear, matmul / bmm / einsum, conv3d) to bf16, forward and backward (the
# incoming gradient is rounded as a bf16 output's gradient is, the operand gradients as bf16 tensors' are);
# accumulation, LayerNorm, softmax, GELU and the residual stream stay fp32.  Elementwise ops on bf16 intermediates
# (q * scale, attn + rel-pos bias) are NOT re-rounded here, which autocast would do: the yardstick is a lower bound
# on autocast's own noise.
def autocast_emulation():
    import torch
    import torch.nn.functional as F
    from torch.overrides import TorchFunctionMode
    from torch.utils._pytree import tree_map

    class _Round(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.to(torch.bfloat16).to(torch.float32)

        @staticmethod
        def backward(ctx, g):
            return g.to(torch.bfloat16).to(torch.float32)

    def rnd(t):
        if isinstance(t, torch.Tensor) and t.dtype == torch.float32:
            return _Round.apply(t)
        return t

    ops = {F.linear, torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.bmm, torch.einsum,
           torch.conv3d, F.conv3d}

    class Mode(TorchFunctionMode):
        calls = 0

        def __torch_function__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            if func in ops:
                Mode.calls += 1
                return rnd(func(*tree_map(rnd, args), **tree_map(rnd, kwargs)))
            return func(*args, **kwargs)

    return Mode()
