#!/usr/bin/env python3
"""bench.py with attributes of the Engine flipped right after its construction -- the in-step A/B runner for schedule switches that
live in svit_amd/engine.py (the library knobs go through tools/bench_knobs.py):
    python tools/diag/engine_attr_ab.py --attr fused_qln=0 -- --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace     (GPU box)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
rest = argv[argv.index("--") + 1:] if "--" in argv else []
mine = argv[:argv.index("--")] if "--" in argv else argv
attrs = [mine[i + 1] for i, a in enumerate(mine) if a == "--attr"]
import svit_amd.engine as E
orig = E.Engine.__init__


def init(self, *a, **k):
    orig(self, *a, **k)
    for spec in attrs:
        name, _, val = spec.partition("=")
        assert hasattr(self, name), name
        setattr(self, name, type(getattr(self, name))(int(val)))


E.Engine.__init__ = init
import io, contextlib
import bench
sys.argv = ["bench.py"] + rest
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
for line in buf.getvalue().splitlines():
    if line.startswith("{"):
        out = json.loads(line)
        out["engine_attrs"] = attrs
        print(json.dumps(out))
