#!/usr/bin/env python3
"""Reduce the two rocprofv3 PMC passes over tools/pmc_traffic.py to HBM bytes per launch.

    python tools/pmc_reduce.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> [out.json]

Counters are in KB.  FETCH_SIZE is doubled (gfx950 reports half of the streamed read bytes,
MI355X_MICROARCH.md HBM section); the factor is re-checked on the two calibration kernels of known
traffic that the script launches (a device-to-device copy and a Jacobi sweep from a zero guess).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

KEEP = ("__amd_rocclr_copyBuffer", "dia_strip_kernel", "dia_strip2_kernel", "fused_pre_kernel", "fused_post_kernel", "cgstep2_kernel", "mg_prolong2_kernel", "mg_restrict2_kernel", "mg_dense_mfma_kernel", "dia_jacobi_kernel", "pcg_update_kernel", "pcg_setx_kernel",
        "pcg_cvt_kernel", "pcg_axpy_kernel", "pcg_finish_kernel", "to_node_major_kernel", "to_sample_major_kernel",
        "cg_spmv_kernel", "ell_jacobi_kernel", "ell_galerkin_kernel", "ell_residual_out_kernel", "agg_restrict_kernel",
        "sa_prolong_add_kernel", "amg_update_kernel", "assemble_rows_kernel", "lattice_grad_kappa_kernel",
        "grad_kappa_shared_kernel", "lattice_assemble_kernel")


def short(name):
    """Kernel symbol without the `void`, the anonymous namespace and the argument list (template arguments may contain
    parentheses themselves: `float __vector(2)`)."""
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    cut = name.find(">(")
    if cut >= 0:
        return name[:cut + 1].strip()
    return name.split("(")[0].strip()


def collect(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in KEEP):
                    acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    # a name can cover launches of very different sizes (small helper copies): keep the full-size ones
    big = {k: [x for x in v if x > 0.5 * max(v)] for k, v in acc.items()}
    return {k: sum(v) / len(v) for k, v in big.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    log = os.path.join(os.path.dirname(sys.argv[1].rstrip("/")), os.path.basename(sys.argv[1].rstrip("/")) + ".log")
    pass_bytes = None
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("pass_bytes"):
                pass_bytes = int(line.split()[1])
    if pass_bytes is None:
        pass_bytes = 1025 * 1025 * 8 * 256       # one fp64 vector of the bench workload
    out = {"pass_bytes": pass_bytes, "kernels": {}}
    try:        # algorithmic passes per launch of the kernels of the bench step (tools/roofline_table.py)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from roofline_table import RULES
    except Exception:
        RULES = []
    for k in sorted(set(fetch) & set(write)):
        rd, wr = 2.0 * fetch[k] * 1024.0, write[k] * 1024.0
        out["kernels"][k] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k], "read_bytes_corrected": rd,
                             "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                             "passes": (rd + wr) / pass_bytes, "read_passes": rd / pass_bytes,
                             "write_passes": wr / pass_bytes}
        for key, label, passes in RULES:
            if key in k:
                out["kernels"][k].update(label=label, algorithmic_passes=passes,
                                         traffic_over_algorithmic=((rd + wr) / pass_bytes / passes) if passes > 0 else None)
                break
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
