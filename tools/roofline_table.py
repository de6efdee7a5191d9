#!/usr/bin/env python3
"""Per-kernel roofline table from a rocprofv3 kernel trace of `bench.py` (default workload).

    python tools/roofline_table.py gpurun_out/prof_dir/*/..._kernel_trace.csv > profiles/rNN_roofline_table.md

For every strip/vector kernel instance launched on the FINE level (identified by its grid size) the
algorithmic bytes per launch (8 n Bp = one fp64 vector pass) are divided by the
average duration.  Coarse-level launches are summed separately."""
import collections
import csv
import sys

N, BP = 1025 * 1025, 256
PASS = 8.0 * N * BP
# (substring of the kernel name, template-argument pattern) -> (label, algorithmic passes per launch)
RULES = [
    ("dia_strip_kernel<double, float, double, 0, 4", "fused CG step (p = z + beta p stored, p.Ap; A p kept in registers, x formed at the end), z/p fp32", 1.5),
    ("dia_strip_kernel<double, float, double, 0, 5", "residual update, A p recomputed from p: r -= alpha A p, r.r, fp32 copy of r", 3.0),
    ("dia_strip_kernel<double, double, double, 0, 4", "fused CG step, z/p fp64", 4.0),
    ("dia_strip_kernel<double, float, double, 0, 2", "fused CG step with the x update (isolated launches of bench.py only)", 4.5),
    ("cgstep2_kernel", "fused CG step, two samples per lane (p = z + beta p stored, p.Ap with a packed-fp32 stencil; A p never stored)", 1.5),
    ("pcg_update_kernel", "r -= alpha Ap, r.r (+ fp32 copy of r)", 3.5),
    ("fused_pre_kernel", "fused PRE pass: two sweeps from 0 + residual + restriction (fp32, two samples per lane)", 1.125),
    ("fused_post_kernel", "fused POST pass: prolongation + correction + two sweeps (fp32, two samples per lane)", 1.625),
    ("dia_strip2_kernel<2, 0, 3, true", "first two Jacobi sweeps from 0 (fp32, two samples per lane)", 1.0),
    ("dia_strip2_kernel<1, 3", "residual + restriction, residual never stored (fp32, two samples per lane)", 1.125),
    ("dia_strip2_kernel<2, 1", "prolong + correct + Jacobi sweep (fp32, two samples per lane)", 1.625),
    ("dia_strip_kernel<float, float, double, 2, 0, 3, true, true", "first two Jacobi sweeps from 0 (fp32)", 1.0),
    ("dia_strip_kernel<float, float, double, 1, 3", "residual + restriction, residual never stored (fp32)", 1.125),
    ("dia_strip_kernel<float, float, double, 1, 0", "residual (fp32)", 1.5),
    ("dia_strip_kernel<float, float, double, 2, 1", "prolong + correct + Jacobi sweep (fp32)", 1.625),
    ("dia_strip_kernel<float, float, double, 2, 0, 3, true, false", "Jacobi sweep (fp32)", 1.5),
    ("dia_strip_kernel<double, double, double, 2, 0, 3, true, true", "first two Jacobi sweeps from 0 (fp64)", 2.0),
    ("dia_strip_kernel<double, double, double, 1, 3", "residual + restriction (fp64)", 2.25),
    ("dia_strip_kernel<double, double, double, 1, 0", "residual (fp64)", 3.0),
    ("dia_strip_kernel<double, double, double, 2, 1", "prolong + correct + Jacobi sweep (fp64)", 3.25),
    ("dia_strip_kernel<double, double, double, 2, 0, 3, true, false", "Jacobi sweep (fp64)", 3.0),
    ("pcg_axpy_kernel", "x += alpha p (flush)", 2.5),
    ("pcg_finish_kernel", "x += sum_j alpha_j p_j + V(r) (end of a solve; 5 directions)", 5.0),
    ("mg_dense_solve_kernel", "dense coarse solve (33^2 level)", 0.0),
    ("mg_dense_mfma_kernel", "dense coarse solve (33^2 level) on MFMA", 0.0),
    ("pcg_init_kernel", "b.b (x, r are set after the full-multigrid start: b is read, nothing is written)", 1.0),
    ("to_node_major_kernel", "(B,n) -> (n,Bp)", 2.0),
    ("to_sample_major_kernel", "(n,Bp) -> (B,n)", 2.0),
    ("dia_shared_apply_kernel", "F = M f - lift", 2.0),
    ("dia_strip_kernel<double, double, double, 0, 0, 4", "F = M f - lift (strip kernel)", 2.0),
    ("dia_strip_kernel<double, double, double, 0, 0, 3", "dL/dkappa bilinear form lambda^T K_1 u", 2.0),
    ("pcg_cvt_kernel", "fp32 copy of rs * b", 1.5),
    ("pcg_setx_kernel", "x = x0 + e0 (two fp32 vectors -> fp64)", 2.0),
    ("lattice_grad_kappa_kernel", "per-element gradient, lattice strip pass", 4.0),
    ("grad_kappa_shared_kernel", "gradient of a shared kappa field (summed over the batch)", 2.0),
]


def main(path):
    per = collections.defaultdict(list)
    total = 0
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            total += d
            cut = name.find(">(")
            per[name[:cut + 1] if cut >= 0 else name.split("(")[0]].append(d)
    print("| kernel | launches (fine level) | avg ms | algorithmic GB/s | frac of 8 TB/s | share of GPU time |")
    print("|---|---|---|---|---|---|")
    seen = 0
    for key, label, passes in RULES:
        for name, durs in per.items():
            if key in name:
                top = max(durs)
                fine = [d for d in durs if d > 0.6 * top]        # fine-level launches dominate the duration
                avg = sum(fine) / len(fine)
                gbs = passes * PASS / (avg * 1e-9) / 1e9
                share = sum(durs) / total
                seen += sum(durs)
                if gbs > 8000 and passes > 0 and "b.b" not in label:
                    # faster than the HBM peak at fine-level bytes: this instantiation only ever ran on coarser levels
                    print(f"| `{label}` -- coarser levels only in this run ({name[:60]}) | {len(durs)} | {avg / 1e6:.3f} (largest level) | | | {100 * share:.1f} % |")
                else:
                    print(f"| `{label}` | {len(fine)} | {avg / 1e6:.3f} | {gbs:.0f} | {gbs / 8000:.2f} | {100 * share:.1f} % (all levels) |")
    print(f"| everything else (coarse-level simple kernels, scalars, torch ops) | | | | | {100 * (total - seen) / total:.1f} % |")


if __name__ == "__main__":
    main(sys.argv[1])
