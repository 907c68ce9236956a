#!/usr/bin/env python3
"""bench.py with tuning knobs set first -- the in-step A/B runner (the library reads no environment variable; the
declared svit_debug_* entry points of include/svit_hip.h are the only switches, and the product bench never calls them).

    python tools/bench_knobs.py --set pool:0=1 --set attn:0=1 -- --steps 20 --warmup 5 --no-cpu-baseline

families: nt:<key>=<v> (svit_debug_set), tn_tile=<mode>, tn=<step_us_x100>,<tbs_x100>, pool:<key>=<v>
(svit_debug_set_pool), attn:<key>=<v> (svit_attn_debug_set).  The line bench.py prints gets a "knobs" entry.  GPU box."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def apply(lib, spec):
    fam, _, val = spec.partition("=")
    if fam == "tn_tile":
        rc = lib.svit_debug_set_tn_tile(int(val))
    elif fam == "tn":
        a, b = val.split(",")
        rc = lib.svit_debug_set_tn(int(a), int(b))
    else:
        name, _, key = fam.partition(":")
        fn = {"nt": lib.svit_debug_set, "pool": lib.svit_debug_set_pool, "attn": lib.svit_attn_debug_set}[name]
        rc = fn(int(key), int(val))
    if rc != 0:
        raise SystemExit("knob %s refused (rc %d)" % (spec, rc))


def main():
    argv = sys.argv[1:]
    rest = argv[argv.index("--") + 1:] if "--" in argv else []
    mine = argv[:argv.index("--")] if "--" in argv else argv
    sets = [mine[i + 1] for i, a in enumerate(mine) if a == "--set"]
    from svit_amd import hip
    lib = hip.load()
    for s in sets:
        apply(lib, s)
    import io
    import contextlib
    import bench
    sys.argv = ["bench.py"] + rest
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    for line in buf.getvalue().splitlines():
        if line.startswith("{"):
            out = json.loads(line)
            out["knobs"] = sets
            print(json.dumps(out))
        else:
            print(line)


if __name__ == "__main__":
    main()
