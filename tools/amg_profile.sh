#!/bin/bash
# kernel times of one steady general-path solve (tools/amg_bench.py under rocprofv3):  bash tools/amg_profile.sh outdir [N B gamma]
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-amgprof}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o amg -- python3 $R/tools/amg_bench.py ${@:-512 64 1} > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out ${@:-512 64 1} <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(f))))
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)
sel = rows[gaps[0][1] + 1:]                      # the second (steady) solve: after the longest idle gap
agg = {}
for s, e, n, g in sel:
    k = (re.sub(r"\(anonymous namespace\)::|void ", "", n)[:48], g)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
print("steady solve: %d launches, span %.2f ms, busy %.2f ms" % (len(sel), (sel[-1][1] - sel[0][0]) / 1e6, sum(a[1] for a in agg.values())))
# algorithmic bytes per node and sample of the FINE-level launch of each kernel (fp32-stored cycle, batch-shared matrix:
# gathers are L2 hits and not counted); the fine level's launches are the longest ones of a (kernel, grid) group
N, B = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 64)
nb = (N + 1) ** 2 * B
ALG = {"ellw_kernel<0, float": 12, "ellw_kernel<1, float": 12, "ellw_kernel<2, double": 16, "amg_update_kernel": 52,
       "cg_update_p_kernel<float>": 20, "ellw_jacobi0_kernel<float": 8, "sa_prolong_add_kernel<float>": 9, "agg_restrict_kernel<float>": 5}
longest = {}
for s, e, n, g in sel:
    k = (re.sub(r"\(anonymous namespace\)::|void ", "", n)[:48], g)
    longest.setdefault(k, []).append((e - s) / 1e3)
for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:18]:
    d = sorted(longest[k])
    top = sum(d[-max(1, len(d) // 10):]) / max(1, len(d) // 10)      # mean of the longest tenth: the fine level
    alg = next((v for kk, v in ALG.items() if k[0].startswith(kk)), None) if k[1] == 524288 else None
    roof = "  fine level %6.1f us = %4.2f TB/s algorithmic (%d B per node-sample)" % (top, alg * nb / top / 1e6, alg) if alg else ""
    print("%8.3f ms %5d  avg %7.1f us  grid %8d %s%s" % (a[1], a[0], a[1] / a[0] * 1e3, k[1], k[0], roof))
PY
