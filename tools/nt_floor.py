#!/usr/bin/env python3
"""Floor table of the step's NT GEMMs: one row per distinct (M, N, K, epilogue) -- calls, microseconds
inside the REPLAYED step (rocprofv3 kernel trace, matched to the schedule by position), algorithmic bytes / 6.3 TB/s (the
achievable HBM rate), flops / (2.5 PFLOP/s x 0.79: the matrix pipe at the ~1.9 GHz the chip holds under this load), the
ratio to the larger floor, and (us - floor) x calls, sorted by that loss.

    python tools/nt_floor.py --metas gpurun_out/nt_metas.json -- <bench.py args>     # one eager traced step: the ordered (M, N, K, epilogue, bytes) list
    python tools/nt_floor.py --table gpurun_out/nt_metas.json <rocprofv3 output dir>   # join with the kernel trace of a replayed run
(tools/nt_floor.sh runs both on the GPU box.)"""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EPI = {0: "bf16", 1: "gelu", 2: "resid", 3: "f32", 4: "dgelu", 5: "relq"}


def dump_metas(path, rest):
    """bench.py's own eager traced step (its kernel-trace leg), with the ordered svit_gemm_nt metas written to `path`."""
    import bench
    orig = bench.kernel_report

    def wrapper(trace, batch):
        json.dump([list(m) for n, e0, e1, m in trace if n == "svit_gemm_nt"], open(path, "w"))
        return orig(trace, batch)      # (sets kernel_report.phases on the module-level name, i.e. on this wrapper)
    bench.kernel_report = wrapper
    sys.argv = ["bench.py"] + rest
    bench.main()


def last_step(d):
    f = max(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    ends = [i for i, j in zip(adam, adam[1:] + [None]) if j is None or j != i + 1]
    return rows[ends[-2] + 1:ends[-1] + 1]


def table(meta_path, trace_dir):
    metas = json.load(open(meta_path))
    seg = last_step(trace_dir)
    nt = [(re.search(r"gemm_nt_(v2|ring)_kernel<([^>]*)>", r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
          for r in seg if "gemm_nt" in r["Kernel_Name"]]
    assert len(nt) == len(metas), "schedule mismatch: %d launches in the trace, %d calls in the eager step" % (len(nt), len(metas))
    agg = defaultdict(lambda: [0, 0.0, set()])
    for (m, us), meta in zip(nt, metas):
        _, M, N, K, epi, nbytes = meta[:6]
        k = (M, N, K, epi, nbytes)
        agg[k][0] += 1
        agg[k][1] += us
        agg[k][2].add("%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else "?")
    rows = []
    for (M, N, K, epi, nbytes), (calls, us, kern) in agg.items():
        t = us / calls
        f_hbm = nbytes / 6.3e12 * 1e6
        f_mfma = 2.0 * M * N * K / (2.5e15 * 0.79) * 1e6
        floor = max(f_hbm, f_mfma)
        rows.append((((t - floor) * calls), M, N, K, EPI.get(epi, str(epi)), calls, t, f_hbm, f_mfma, t / floor, sorted(kern)))
    rows.sort(reverse=True)
    tot = sum(r[6] * r[5] for r in rows)
    print("NT GEMMs of one replayed step: %d launches, %.1f us; sum of floors %.1f us; lost %.1f us" %
          (len(nt), tot, sum(max(r[7], r[8]) * r[5] for r in rows), sum(r[0] for r in rows)))
    print("%7s %5s %5s %-6s %5s %8s %8s %8s %6s %9s  %s" % ("M", "N", "K", "epi", "calls", "us/call", "hbm us", "mfma us", "ratio", "lost us", "kernel"))
    for lost, M, N, K, epi, calls, t, fh, fm, ratio, kern in rows:
        print("%7d %5d %5d %-6s %5d %8.1f %8.1f %8.1f %6.2f %9.1f  %s" % (M, N, K, epi, calls, t, fh, fm, ratio, lost, " ".join(kern)))


if __name__ == "__main__":
    if sys.argv[1] == "--metas":
        rest = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
        dump_metas(sys.argv[2], rest)
    else:
        table(sys.argv[2], sys.argv[3])
