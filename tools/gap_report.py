#!/usr/bin/env python3
"""GPU idle time of the bench step from a rocprofv3 kernel trace: gaps between consecutive kernels of the LAST timed step.

    python tools/gap_report.py <..._kernel_trace.csv> [n_top]

The last step is found as the span between the last two launches of the loss's norm kernel (one per step)."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find(">(")
    return (name[:cut + 1] if cut >= 0 else name.split("(")[0])[:70]


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y"))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]) + " [" + grid + "]"))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "pcg_update_kernel" in r[2] or "double, float, double, 0, 5," in r[2]]
    # steps: pcg_update launches come in bursts of (fwd its + adj its); split bursts by gaps > 20 ms
    starts = [marks[0]]
    for a, b in zip(marks, marks[1:]):
        if rows[b][0] - rows[a][1] > 8e6:
            starts.append(b)
    print("solve bursts found:", len(starts))
    # take the window from the 3rd-last burst start to the last burst start = one fwd + one adj solve + glue
    if len(starts) < 5:
        lo, hi = 0, len(rows) - 1
    else:
        lo, hi = starts[-5], starts[-3]
    win = rows[lo:hi]
    span = (win[-1][1] - win[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in win) / 1e6
    print(f"window: {len(win)} kernels, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms")
    per = defaultdict(lambda: [0, 0.0, 0.0])
    for s_, e_, n_ in win:
        per[n_][0] += 1
        per[n_][1] += (e_ - s_) / 1e6
        per[n_][2] = max(per[n_][2], (e_ - s_) / 1e6)
    print("kernel time inside the window:")
    acc = 0.0
    for n_, (c, t, mx) in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
        acc += t
        print(f"  {t:7.3f} ms ({100 * t / busy:5.1f} %, cum {100 * acc / busy:5.1f})  n={c:4d}  max {mx:6.3f}  {n_}")
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(win, win[1:]):
        gaps.append(((s1 - e0) / 1e3, n0, n1))
    hist = defaultdict(lambda: [0, 0.0])
    for g, n0, n1 in gaps:
        key = "<5us" if g < 5 else "<10us" if g < 10 else "<20us" if g < 20 else "<50us" if g < 50 else "<200us" if g < 200 else ">=200us"
        hist[key][0] += 1
        hist[key][1] += max(g, 0.0)
    for k in ("<5us", "<10us", "<20us", "<50us", "<200us", ">=200us"):
        print(f"  gaps {k:8s}: {hist[k][0]:5d}  total {hist[k][1] / 1e3:7.3f} ms")
    top = sorted(gaps, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]
    for g, n0, n1 in top:
        print(f"  {g:9.1f} us   after {n0}   before {n1}")
    by_pair = defaultdict(lambda: [0, 0.0])
    for g, n0, n1 in gaps:
        by_pair[(n0[:40], n1[:40])][0] += 1
        by_pair[(n0[:40], n1[:40])][1] += max(g, 0.0)
    print("largest idle by kernel pair:")
    for (n0, n1), (c, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:20]:
        print(f"  {t / 1e3:7.3f} ms in {c:4d} gaps   {n0} -> {n1}")


if __name__ == "__main__":
    main()
