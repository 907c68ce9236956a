#!/usr/bin/env python3
"""Busy time vs wall span from a rocprofv3 kernel trace (…_kernel_trace.csv).

    python tools/trace_gaps.py gpurun_out/prof/<host>/<pid>_kernel_trace.csv [n_last_dispatches]

Prints the device-busy fraction over the last N dispatches, the idle-gap total, and the kernels
that most often precede long gaps (launch-bound stretches)."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[-n_last:]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    gaps = defaultdict(lambda: [0, 0])
    gap_total = 0
    end = rows[0][0]
    for s, e, name in rows:
        if s > end:
            g = s - end
            gap_total += g
            gaps[prev][0] += g
            gaps[prev][1] += 1
        busy += max(0, e - max(s, end))
        end = max(end, e)
        prev = name.split("(")[0][:70]
    print("dispatches %d span %.3f ms busy %.3f ms (%.1f%%) idle %.3f ms" %
          (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, gap_total / 1e6))
    print("idle time by preceding kernel:")
    for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print("  %8.1f us  %5d gaps  avg %6.2f us  %s" % (g / 1e3, c, g / 1e3 / c, k))
    per_step(path)


def per_step(path):
    """One line per training step (delimited by the optimizer's last kernel): wall span against the sum of
    kernel durations, the idle time before the step's first dispatch and inside it.  The window above mixes
    in the warm-up -> capture -> timed-region boundaries (each a host synchronisation); this does not."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"]))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    ends = [i for i, j in zip(adam, adam[1:] + [None]) if j is None or j != i + 1]    # last adamw of each step
    print("per step (last optimizer kernel to last optimizer kernel):")
    for a, b in zip(ends[:-1], ends[1:]):
        seg = rows[a + 1:b + 1]
        span = seg[-1][1] - seg[0][0]
        ksum = sum(e - s for s, e, _, _ in seg)
        gaps = [y[0] - x[1] for x, y in zip(seg[:-1], seg[1:])]
        print("  %4d dispatches  span %.3f ms  kernels %.3f ms  idle before %.1f us  idle inside %.1f us (largest %.1f)"
              % (len(seg), span / 1e6, ksum / 1e6, (seg[0][0] - rows[a][1]) / 1e3,
                 sum(g for g in gaps if g > 0) / 1e3, max(gaps) / 1e3))


def window(path, needle, before, after):
    """Timeline (start us, duration us, gap before us, name) around the last dispatch of the
    first run of kernels matching `needle` in the final step."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if needle in r[2]]
    # first match of the last cluster (matches further than 5 ms apart start a new cluster)
    start = idx[-1]
    for a, b in zip(reversed(idx[:-1]), reversed(idx[1:])):
        if rows[b][0] - rows[a][0] > 5_000_000:
            break
        start = a
    lo, hi = max(0, start - before), min(len(rows), start + after)
    t0 = rows[lo][0]
    for i in range(lo, hi):
        s, e, n = rows[i]
        gap = s - rows[i - 1][1] if i else 0
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:60]
        print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short))


def around(path, needle, limit):
    """which kernels run right before / after each dispatch matching `needle` (histogram)"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sh = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:44]
    hist = defaultdict(int)
    for i, r in enumerate(rows):
        if needle in r[2] and 0 < i < len(rows) - 1:
            hist[(sh(rows[i - 1][2]), sh(rows[i + 1][2]), (r[1] - r[0]) // 1000)] += 1
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:limit]:
        print("%5d x  prev=%-46s next=%-46s dur~%d us" % (v, k[0], k[1], k[2]))


def blocks(path):
    """backward of the last step split at the fc2-dgrad GEMMs (epilogue 4 = first kernel of a
    transformer block's backward): per block total and its five longest kernels"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sh = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:34]
    # epilogue id of an NT GEMM dispatch: the 6th template argument of gemm_nt_v2_kernel<RB, NB, WM, WN, STAGES, EPI, BK>,
    # the last of gemm_nt_ring_kernel<RB, NB, WM, WN, NL, STAGES, EPI> (csrc/gemm_nt.hip); 4 = SVIT_EPI_DGELU = fc2 dgrad
    import re

    def epi(name):
        m = re.search(r"gemm_nt_(v2|ring)_kernel<([^>]*)>", name)
        if not m:
            return None
        a = [x.strip() for x in m.group(2).split(",")]
        return int(a[5] if m.group(1) == "v2" else a[-1])
    marks = [i for i, r in enumerate(rows) if epi(r[2]) == 4]
    if len(marks) < 16:
        sys.exit("trace_gaps --blocks: only %d fc2-dgrad (SVIT_EPI_DGELU) dispatches found in %s -- kernel names changed?"
                 % (len(marks), path))
    marks = marks[-16:]
    last_adam = max(i for i, r in enumerate(rows) if "adamw_kernel" in r[2])
    bounds = marks + [last_adam]
    for b, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in rows[lo:hi]:
            agg[sh(n)][0] += e - s
            agg[sh(n)][1] += 1
        tot = sum(v[0] for v in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:6]
        print("bwd block %2d: %7.1f us | " % (15 - b, tot / 1e3) +
              "  ".join("%s %.0f(%d)" % (k, v[0] / 1e3, v[1]) for k, v in top))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--blocks":
        blocks(sys.argv[1])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "--around":
        around(sys.argv[1], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "--window":
        window(sys.argv[1], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    main()
