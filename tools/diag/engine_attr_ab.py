#!/usr/bin/env python3
"""bench.py with attributes of the Engine flipped right after its construction -- the in-step A/B runner for schedule switches that
live in svit_amd/engine.py (the library knobs go through tools/bench_knobs.py):
    python tools/diag/engine_attr_ab.py [--attr overlap_wgrad=1] [--init red_group=8] -- --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace     (GPU box)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
rest = argv[argv.index("--") + 1:] if "--" in argv else []
mine = argv[:argv.index("--")] if "--" in argv else argv
attrs = [mine[i + 1] for i, a in enumerate(mine) if a == "--attr"]
inits = dict((mine[i + 1].split("=")[0], int(mine[i + 1].split("=")[1])) for i, a in enumerate(mine) if a == "--init")   # constructor keywords (e.g. red_group=8)
import svit_amd.engine as E
orig = E.Engine.__init__


def init(self, *a, **k):
    k.update(inits)
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
        out["engine_attrs"] = attrs + ["%s=%d" % kv for kv in inits.items()]
        print(json.dumps(out))
