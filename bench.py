#!/usr/bin/env python3
"""Headline benchmark: differentiable 2D P1 Poisson solves/sec (fwd + adjoint).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / SURVEY 8(d) C4): FEMesh.rectangle(1024, 1024), 256 samples
per GPU, one scalar kappa_b ~ U(0.5, 2.0) per sample (seed 4096 + rank), f == 1,
L = mean_b sum_i u_b[i]^2.  One "step" = assemble K(kappa_b), F; forward solve; dL/du;
adjoint solve; dL/dkappa contraction -- for every sample of the batch.  Inputs are resident
in HBM before the timed region.  N > 1: one process per GPU (torch.distributed, RCCL), the
batch is sharded (weak scaling: 256 per GPU), the only collective is the loss all-reduce.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the
dominant kernel and `cpu_baseline` (the CPU oracle timed on the host cores, rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mesh", type=int, default=1024, help="N of the N x N mesh (bench contract: 1024)")
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU (bench contract: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (clean kernel traces)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--tol", type=float, default=None, help="override the solver's relative residual tolerance")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this device index (needs --backend gloo)")
    ap.add_argument("--kappa", choices=["sample", "element"], default="sample",
                    help="sample: one scalar kappa per sample (the contract workload); element: a log-normal "
                         "per-element field per sample, exp(0.3 randn) (SURVEY 8(d) C3/C4 variant)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.all_on_device is not None:
        local_rank = args.all_on_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from diffhe import FEMesh, DifferentiableFESolver, _hip
    from diffhe.plan import get_plan, padded_batch

    N, B = args.mesh, args.batch
    mesh = FEMesh.rectangle(N, N)
    n = mesh.n_nodes
    gen = torch.Generator().manual_seed(4096 + rank)
    if args.kappa == "sample":
        kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=torch.float64)).to(dev).requires_grad_(True)
    else:
        gdev = torch.Generator(device=dev).manual_seed(2025 + rank)
        kappa = torch.exp(0.3 * torch.randn(B, mesh.n_elements, generator=gdev, dtype=torch.float64, device=dev))
        kappa.requires_grad_(True)
    f = torch.ones(B, n, dtype=torch.float64, device=dev)
    solver = DifferentiableFESolver(mesh, kappa, device=dev, **({"tol": args.tol} if args.tol else {}))
    plan = get_plan(mesh, dev)

    iters = []

    def step():
        kappa.grad = None
        u = solver(f)
        loss = (u ** 2).sum(dim=1).mean()
        loss.backward()
        info = solver.last_info
        iters.append((info.iterations, info.adj_iterations, info.max_relres, info.adj_max_relres,
                      info.not_converged))
        return loss.detach(), u

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    # dominant kernel (fused CG step) timed with HIP events INSIDE the solver loop, over the timed region only
    import ctypes
    prof_ms, prof_n = ctypes.c_double(0.0), ctypes.c_longlong(0)
    if rank == 0:
        _hip.lib().diffhe_lattice_pcg_profile(1, None, None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, u = step()
        if world > 1:
            dist.all_reduce(loss)       # the only collective: scalar loss (per-sample kappa)
    sync_all()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        _hip.lib().diffhe_lattice_pcg_profile(0, ctypes.byref(prof_ms), ctypes.byref(prof_n))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = world * B * args.steps / elapsed
    timed_iters = list(iters[-args.steps:]) if args.steps else list(iters)

    # secondary measurement, same run: V-cycle vectors stored in fp64 instead of fp32
    variant = None
    if (world == 1 and solver.last_info.path == "lattice-mgpcg" and solver.mg.get("fp32") and args.kappa == "sample"
            and not args.no_variants):
        solver.mg["fp32"] = 0
        step()
        torch.cuda.synchronize(dev)
        tv = time.perf_counter()
        step()
        torch.cuda.synchronize(dev)
        tv = time.perf_counter() - tv
        variant = {"vcycle_storage_fp64": {"value_per_gpu": round(B / tv, 3), "ms_per_step": round(1e3 * tv, 3),
                                           "iters_fwd": solver.last_info.iterations}}
        solver.mg["fp32"] = 1
        # ... and the same workload with an independent per-element kappa FIELD per sample (SURVEY 8(d) C3/C4
        # variant): one matrix per sample and level, nothing shared or factored across the batch
        gdev = torch.Generator(device=dev).manual_seed(2025)
        kap_e = torch.exp(0.3 * torch.randn(B, mesh.n_elements, generator=gdev, dtype=torch.float64, device=dev))
        kap_e.requires_grad_(True)
        solver_e = DifferentiableFESolver(mesh, kap_e, device=dev, **({"tol": args.tol} if args.tol else {}))

        def step_e():
            kap_e.grad = None
            ue = solver_e(f)
            (ue ** 2).sum(dim=1).mean().backward()

        step_e()
        torch.cuda.synchronize(dev)
        tv = time.perf_counter()
        step_e()
        torch.cuda.synchronize(dev)
        tv = time.perf_counter() - tv
        variant["kappa_element_field"] = {
            "value_per_gpu": round(B / tv, 3), "ms_per_step": round(1e3 * tv, 3),
            "iters_fwd": solver_e.last_info.iterations, "iters_adj": solver_e.last_info.adj_iterations,
            "tol": solver_e.tol, "not_converged": solver_e.last_info.not_converged}
        del solver_e, kap_e
        torch.cuda.empty_cache()

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: the batched operator apply of the CG loop ----
        L = _hip.lib()
        Bp = padded_batch(B)
        roof = None
        traffic = None
        copy_gbs = None
        # the fused CG step (and its roofline entry) exists for wave-sized batches on strip-sized meshes only
        roofline_ok = (solver.last_info.path == "lattice-mgpcg" and args.kappa == "sample" and Bp >= 64
                       and N >= 191) or solver.last_info.path == "ell-pcg"
        st = torch.cuda.current_stream(dev).cuda_stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def time_launch(launch):
            for _ in range(3):
                launch()
            e0.record()
            for _ in range(args.kernel_reps):
                launch()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / args.kernel_reps

        other = None
        if roofline_ok and solver.last_info.path == "lattice-mgpcg":
            # The kernel with the largest share of the step (profiles/r01_mgpcg_final_kernel_stats.csv) is the
            # fused CG step dia_strip_kernel<M_APPLY, F_PUPD> (p = z + beta p, x += alpha p_old, Ap = A p, p.Ap);
            # it is timed alone on the operator this workload assembles (shared unit matrix + kappa_b scale).
            from diffhe.solver import _Engine, K_SAMPLE
            eng = _Engine(plan, solver.tol, solver.max_iter, solver.check_every, solver.assembly)
            vals, Bv, scale, _, _ = eng.lattice_assemble(kappa.detach(), K_SAMPLE, B, Bp)
            arr = eng.lattice_levels(vals)
            f32 = bool(solver.mg.get("fp32"))
            z = torch.rand((n, Bp), dtype=torch.float32 if f32 else torch.float64, device=dev)
            x = torch.rand((n, Bp), dtype=torch.float64, device=dev)
            p_in = torch.rand((n, Bp), dtype=z.dtype, device=dev)       # direction stored in z's precision
            p_out = torch.empty_like(p_in)
            Ap = torch.empty_like(x)
            ab = torch.rand(2, Bp, dtype=torch.float64, device=dev)
            part = torch.empty(L.diffhe_lattice_blocks(n, Bp) * Bp, dtype=torch.float64, device=dev)
            dur = time_launch(lambda: _hip.check(L.diffhe_lattice_cg_step(
                arr, Bv, _hip.ptr(scale), _hip.ptr(z), int(f32), _hip.ptr(p_in), _hip.ptr(p_out), _hip.ptr(x),
                _hip.ptr(ab[0]), _hip.ptr(ab[1]), 0, _hip.ptr(Ap), _hip.ptr(part), Bp, st), "diffhe_lattice_cg_step"))
            zb = 4.0 if f32 else 8.0
            alg_bytes = (3 * zb + 24.0) * n * Bp  # read z, p, x; write p, Ap, x (matrix batch-shared: amortised)
            kname = "dia_strip_kernel<M_APPLY,F_PUPD> (fused CG step: p-update + x-update + operator apply + dot)"
            pmc_key = "F_PUPD"
            # second kernel family: one weighted-Jacobi sweep of the V-cycle on the fine level, fp64 storage
            rhs = torch.rand((n, Bp), dtype=torch.float64, device=dev)
            dur_j = time_launch(lambda: _hip.check(L.diffhe_lattice_smooth(
                arr, Bv, _hip.ptr(scale), _hip.ptr(rhs), _hip.ptr(x), _hip.ptr(Ap), 0.8, Bp, st),
                "diffhe_lattice_smooth"))
            copy_dur = time_launch(lambda: Ap.copy_(x))
            copy_gbs = 16.0 * n * Bp / copy_dur / 1e9
            other = [{"kernel": "dia_strip_kernel<M_JACOBI> (fine-level Jacobi sweep, fp64 storage)",
                      "achieved": round(24.0 * n * Bp / dur_j / 1e9, 1), "frac": round(24.0 * n * Bp / dur_j / 8e12, 4),
                      "bytes_per_launch": 24.0 * n * Bp, "avg_launch_ms": round(dur_j * 1e3, 4)}]
            try:   # HBM bytes per launch from the PMC passes committed under profiles/ (same kernels, same sizes)
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                    pmc = json.load(fh)
                if pmc["pass_bytes"] == 8 * n * Bp:
                    for k_, v_ in pmc["kernels"].items():
                        if "dia_strip_kernel<double, float, double, 0, 2" in k_ and f32:
                            traffic = v_["hbm_bytes_per_launch"]
                        if "dia_strip_kernel<double, double, double, 2, 0" in k_:
                            other[0]["traffic"] = v_["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            del z, x, p_in, p_out, Ap, rhs, vals
        elif roofline_ok and solver.last_info.path == "ell-pcg":
            W = plan.W
            vals = torch.rand((W, n, Bp), dtype=torch.float64, device=dev)
            x = torch.rand((n, Bp), dtype=torch.float64, device=dev)
            y = torch.empty_like(x)
            part = torch.empty(L.diffhe_grad_kappa_blocks(n, Bp) * Bp, dtype=torch.float64, device=dev)
            dur = time_launch(lambda: _hip.check(L.diffhe_ell_apply(
                _hip.ptr(vals), _hip.ptr(plan.cols), _hip.ptr(x), _hip.ptr(y), _hip.ptr(part), n, W, Bp, Bp, st),
                "diffhe_ell_apply"))
            alg_bytes = (8.0 * W + 16.0) * n * Bp       # read W values + p, write Ap
            kname = "cg_spmv_kernel"
            del vals, x, y, part
        if roofline_ok:
            # `achieved` = the kernel's average duration INSIDE the solver, HIP events over the timed region (cold
            # caches between the V-cycle and the residual update; this is the figure the rocprofv3 kernel stats of
            # the same command show); the back-to-back launches above are reported next to it as `isolated_*`.
            iso = alg_bytes / dur / 1e9
            in_loop = prof_n.value > 0 and solver.last_info.path == "lattice-mgpcg"
            dur_used = (prof_ms.value * 1e-3 / prof_n.value) if in_loop else dur
            achieved = alg_bytes / dur_used / 1e9
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": alg_bytes, "avg_launch_ms": round(dur_used * 1e3, 4),
                    "launches_timed": int(prof_n.value) if in_loop else None,
                    "isolated_launch_ms": round(dur * 1e3, 4), "isolated_achieved": round(iso, 1),
                    "isolated_frac": round(iso / HBM_PEAK_GBS, 4),
                    "stream_copy_gbs": round(copy_gbs, 1) if copy_gbs else None, "other_kernels": other}

        # ---- CPU baseline: the oracle (port of the reference algorithm, sparse LU) -----------
        cpu = None
        parity = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import p1_oracle as orc
            bn = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64)
            bv = np.fromiter(mesh.dirichlet_nodes.values(), dtype=np.float64)
            nodes, elements = mesh.nodes.numpy(), mesh.elements.numpy()
            kb = float(kappa[0].detach()) if args.kappa == "sample" else kappa[0].detach().cpu().numpy()
            tc = time.perf_counter()
            uo, dk, _ = orc.solve_with_adjoint(nodes, elements, bn, bv, kb, np.ones(n), lambda u_: 2 * u_ / B,
                                               sparse=True)
            tc = time.perf_counter() - tc
            cpu = {"value": round(1.0 / tc, 4), "unit": "solves/s", "cores": 1, "kind": "port",
                   "sample": f"1 sample of the same workload ({N}x{N}, fwd+adjoint, scipy SuperLU), {tc:.1f} s"}
            ug = u[0].detach().cpu().numpy()
            parity = {"u_rel_err": float(np.max(np.abs(ug - uo)) / np.max(np.abs(uo))),
                      "dkappa_rel_err": (float(abs(float(kappa.grad[0]) - dk.sum()) / abs(dk.sum()))
                                         if args.kappa == "sample" else
                                         float(np.max(np.abs(kappa.grad[0].cpu().numpy() - dk)) / np.max(np.abs(dk))))}

        it = np.array(timed_iters, dtype=np.float64)
        out = {
            "metric": "FEM solves/sec (fwd+adjoint), 2D P1 Poisson 1024^2 mesh, batch=256 per GPU",
            "value": round(value, 4), "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C4: rectangle({N},{N}), {B} samples/GPU, "
                                   + ("kappa_b~U(0.5,2) scalar per sample, " if args.kappa == "sample" else
                                      "per-element log-normal kappa field per sample, ")
                                   + "f=1, L=mean_b sum u^2, fwd+adjoint", "mesh": f"{N}x{N}", "batch_per_gpu": B,
                       "global_batch": B * world, "solver": solver.last_info.path, "tol": solver.tol,
                       "stop": ("per sample |r| <= max(tol |b|, 0.5 u |A| |x0|): tol, floored at half the residual level "
                                "fp64 can attain (true residual of this workload stalls at ~3e-11 |b|)"
                                if solver.mg.get("floor", 1) else "per sample |r| <= tol |b|"),
                       "multigrid": {k: v for k, v in solver.mg.items() if v is not None},
                       "operator": ("K_b = kappa_b * K_1: one shared matrix + a per-sample scale; every sample is "
                                    "solved by its own PCG (no u(1)/kappa shortcut); per-element-field figure in "
                                    "variants.kappa_element_field") if args.kappa == "sample" else
                                   "one assembled matrix per sample and level",
                       "precision": "fp64 arithmetic, CG vectors, residuals and dots; V-cycle (preconditioner) "
                                    "vectors stored fp32" if solver.mg.get("fp32") else "fp64 throughout",
                       "parallelism": f"batch-sharded x{world}"},
            "solver_iters": {"fwd": int(it[:, 0].max()), "adj": int(it[:, 1].max()),
                             "max_relres_fwd": float(it[:, 2].max()), "max_relres_adj": float(it[:, 3].max()),
                             "not_converged": int(it[:, 4].max())},
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_oracle": parity, "variants": variant,
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
