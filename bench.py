#!/usr/bin/env python3
"""Headline benchmark: differentiable 2D P1 Poisson solves/sec (fwd + adjoint).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / SURVEY 8(d) C4): FEMesh.rectangle(1024, 1024), 256 samples
per GPU, one scalar kappa_b ~ U(0.5, 2.0) per sample (seed 4096 + rank), f == 1,
L = mean_b sum_i u_b[i]^2.  One "step" = assemble K(kappa_b), F; forward solve; dL/du;
adjoint solve; dL/dkappa contraction -- for every sample of the batch.  Inputs are resident
in HBM before the timed region.

N > 1: one process per GPU over torch.distributed (RCCL).  Under `torch.distributed.run` the ranks come
from the environment; a plain `python bench.py --gpus N` starts the N rank processes ITSELF (fresh children,
before this process touches the GPU) and relays rank 0's JSON line.  The batch is sharded (weak scaling:
256 per GPU) through `diffhe.distributed.ShardedBatchSolve`; the headline's only collective is the fused loss
all-reduce (per-sample kappa needs no gradient reduction).  `variants.shared_kappa_allreduce` -- reported at
EVERY N -- is the same workload with ONE per-element kappa field shared by the batch: its 16.8 MB gradient
really crosses the reduce-scatter + all-gather of BASELINE config 4's "RCCL grad all-reduce".

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the
dominant kernel (+ the step-level figure), `cpu_baseline` (CPU restatements of the reference path timed on
the host cores, rank 0, N = 1), parity against the oracle, and `variants`: the other BASELINE configs
(2: 1D 10^4 x 1024, 3: 512^2 x 256, 5: kappa recovery by Adam) and the secondary forms of the headline workload.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_PROFILE = "r04_pmc_traffic.json"   # profiles/: rocprofv3 --pmc passes over this command, reduced by tools/pmc_reduce.py


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mesh", type=int, default=1024, help="N of the N x N mesh (bench contract: 1024)")
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU (bench contract: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (clean kernel traces)")
    ap.add_argument("--only-variant", default=None, help="run just this entry of `variants` (development / profiling)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--variant-reps", type=int, default=3, help="repetitions per variant (the median is reported)")
    ap.add_argument("--tol", type=float, default=None, help="override the solver's relative residual tolerance")
    ap.add_argument("--max-iter", type=int, default=None, help="cap the CG iterations (development: kernel experiments)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--all-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this device index (needs --backend gloo)")
    ap.add_argument("--kappa", choices=["sample", "element", "shared-field"], default="sample",
                    help="sample: one scalar kappa per sample (the contract workload); element: a log-normal "
                         "per-element field per sample, exp(0.3 randn) (SURVEY 8(d) C3/C4 variant); shared-field: ONE "
                         "such field for the whole batch -- its (m,) gradient is all-reduced over the ranks")
    ap.add_argument("--layout", choices=["node", "sample"], default="node",
                    help="node: f and u are kept (n, B), the solver's own layout (DifferentiableFESolver.forward(..., "
                         "layout='node')); sample: the reference API's (B, n), transposed in and out every step")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercise launch, rendezvous, barriers and the max-over-ranks timing only "
                         "(value is null); used by the CPU tests of the multi-rank launch path")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and wait.
    Nothing in this process has initialised the GPU (no torch.cuda call, torch is not even imported yet), and the
    children are new interpreters, not re-execs.  Rank 0 inherits stdout, so its JSON line is this command's."""
    port = _free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes():
    """NUMA node of every GPU in KFD enumeration order (= HIP device order unless *_VISIBLE_DEVICES remaps it), read from
    sysfs only -- no HIP call, so it may run before the process touches the GPU.  [] when the topology is not readable."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    out = []
    try:
        for name in sorted(os.listdir(base), key=int):
            props = {}
            for line in open(os.path.join(base, name, "properties")):
                k, _, v = line.partition(" ")
                props[k] = v.strip()
            if int(props.get("simd_count", "0")) == 0:
                continue                                   # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
            try:
                out.append(int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read()))
            except (OSError, ValueError):
                out.append(-1)
    except (OSError, ValueError):
        return []
    return out


def rank_cpu_slice(allowed, world, local_rank, gpu_numa=(), node_cpus=None):
    """Host cores of rank `local_rank`: the cores of its GPU's NUMA node (shared evenly with the other ranks whose GPU
    sits on the same node), else an even contiguous slice of the allowed cores.  Pure function (tested on CPU)."""
    allowed = sorted(allowed)
    node_cpus = node_cpus or {}
    if gpu_numa and local_rank < len(gpu_numa) and gpu_numa[local_rank] >= 0 and gpu_numa[local_rank] in node_cpus:
        node = gpu_numa[local_rank]
        peers = [r for r in range(min(world, len(gpu_numa))) if gpu_numa[r] == node]
        pool = [c for c in allowed if c in node_cpus[node]]
        if len(pool) >= len(peers):
            k = peers.index(local_rank)
            per = len(pool) // len(peers)
            return pool[k * per:(k + 1) * per]
    per = len(allowed) // world
    if per < 1:
        return allowed
    return allowed[local_rank * per:(local_rank + 1) * per]


def pin_rank(world, local_rank):
    """One process per GPU: keep this rank's host threads (Python, the HIP runtime's helper threads, autograd's
    thread) on the cores next to its GPU.  Every solve reads one int per CG iteration from the host; with N ranks
    wandering over all cores that wait grows.  Called before anything touches the GPU.  Never fatal."""
    try:
        allowed = os.sched_getaffinity(0)
        numa = gpu_numa_nodes()
        node_cpus = {}
        for node in set(numa):
            if node >= 0:
                try:
                    node_cpus[node] = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
                except OSError:
                    pass
        cpus = rank_cpu_slice(allowed, world, local_rank, numa, node_cpus)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return {"cores": len(cpus), "first": min(cpus) if cpus else None,
                "numa_node": numa[local_rank] if local_rank < len(numa) else None}
    except Exception as exc:   # noqa: BLE001 -- placement is an optimisation: whatever goes wrong, the rank runs unpinned
        return {"error": type(exc).__name__}


def _spread(xs):
    xs = sorted(float(x) for x in xs)
    return {"min": round(xs[0], 3), "median": round(statistics.median(xs), 3), "max": round(xs[-1], 3)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    affinity = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        affinity = pin_rank(world, local_rank)       # before torch is imported, before any GPU call
        # the mesh plan's host arrays (gather lists, reference-order integrals: ~5 s of numpy at 1024^2) are built by
        # rank 0 and mapped by the other ranks of the node (diffhe.plan.host_arrays)
        os.environ.setdefault("DIFFHE_PLAN_CACHE", os.path.join(
            "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"diffhe_plan_{os.environ.get('MASTER_PORT', '0')}"))

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.dry_run:
        return dry_run(args, rank, world, dist, torch)

    if args.all_on_device is not None:
        local_rank = args.all_on_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import ctypes
    from diffhe import FEMesh, DifferentiableFESolver, _hip
    from diffhe.distributed import ShardedBatchSolve
    from diffhe.plan import get_plan

    N, B = args.mesh, args.batch
    mesh = FEMesh.rectangle(N, N)
    n, m = mesh.n_nodes, mesh.n_elements
    node = args.layout == "node"
    tol_kw = {"tol": args.tol} if args.tol else {}
    if args.max_iter:
        tol_kw["max_iter"] = args.max_iter
    t_plan = time.perf_counter()
    if world > 1:           # rank 0 builds the plan (and fills the node-level cache) first, the others map it
        os.makedirs(os.environ["DIFFHE_PLAN_CACHE"], exist_ok=True)
        if rank == 0:
            get_plan(mesh, dev)
        dist.barrier()
    plan = get_plan(mesh, dev)
    t_plan = time.perf_counter() - t_plan
    L = _hip.lib()

    def make_kappa(kind, seed):
        if kind == "sample":
            g_ = torch.Generator().manual_seed(seed)
            return (0.5 + 1.5 * torch.rand(B, generator=g_, dtype=torch.float64)).to(dev).requires_grad_(True)
        g_ = torch.Generator(device=dev).manual_seed(seed)
        shape = (B, m) if kind == "element" else (m,)
        k_ = torch.exp(0.3 * torch.randn(*shape, generator=g_, dtype=torch.float64, device=dev))
        if kind == "element" and node:      # layout='node': per-sample fields are kept element-major (m, B), like f and u
            k_ = k_.t().contiguous()
        return k_.requires_grad_(True)

    # seeds of SURVEY 8(d): 4096 (+ rank) for the scalars, 2025 (+ rank) for per-sample fields; a SHARED field is the
    # same on every rank (seed 2025)
    kappa = make_kappa(args.kappa, {"sample": 4096 + rank, "element": 2025 + rank, "shared-field": 2025}[args.kappa])
    f = torch.ones((n, B) if node else (B, n), dtype=torch.float64, device=dev)
    sdim = 0 if node else 1                     # the dimension a sample's nodes run along
    solver = DifferentiableFESolver(mesh, kappa, device=dev, **tol_kw)

    def loss_sum(u_):
        # sum_b sum_i u_b[i]^2 as per-sample squared norms: one reduction pass forward, one elementwise pass backward
        # (the same loss as (u ** 2).sum(), 24 instead of 40 B per node of torch-side traffic)
        return torch.linalg.vector_norm(u_, dim=sdim).square().sum()

    def make_step(solver_, kappa_, f_, layout_node, shared):
        drv = ShardedBatchSolve(lambda kap, f_local: solver_(f_local, layout="node" if layout_node else "sample"))
        d = 0 if layout_node else 1

        def step_(time_collective=False):
            kappa_.grad = None
            loss_, u_ = drv.step_local(f_, B * world, lambda uu: torch.linalg.vector_norm(uu, dim=d).square().sum(), kappa_,
                                       kappa_ if shared else None, time_collective=time_collective)
            return loss_, u_
        return drv, step_

    # per-rank differentiable solve of this rank's shard: the HIP path (`solver` holds this rank's kappa shard)
    driver, step_raw = make_step(solver, kappa, f, node, args.kappa == "shared-field")
    assert driver.world == world
    iters = []

    def step():
        loss_, u_ = step_raw()
        info = solver.last_info
        iters.append((info.iterations, info.adj_iterations, info.max_relres, info.adj_max_relres, info.not_converged,
                      dict(info.stop_rules or {}), dict(info.adj_stop_rules or {})))
        return loss_, u_

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync_all()
    # The interpreter's full (generation-2) garbage collection walks every object torch has ever imported: 70-85 ms, once
    # in the 20 timed steps (timed_steps_ms.max showed it as one 165-180 ms step among 94 ms ones).  Park what exists
    # now in the permanent generation, as a long-running training process would; collection stays ENABLED.
    import gc
    gc.collect()
    gc.freeze()
    # main kernels timed with HIP events INSIDE the solver loop, over the timed region only;
    # algorithmic bytes of every launch of the timed steps from the library's own accounting
    prof_ms, prof_n = ctypes.c_double(0.0), ctypes.c_longlong(0)
    acc_bytes, acc_launches = ctypes.c_double(0.0), ctypes.c_longlong(0)
    if rank == 0:
        L.diffhe_lattice_pcg_profile(1, None, None)
    L.diffhe_traffic_account(1, None, None)
    seg0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, u = step()
        marks.append(time.perf_counter())           # host clock only: every solve already waits for its own status
    torch.cuda.synchronize(dev)
    own_elapsed = time.perf_counter() - t0          # this rank's own time, before it waits for the others
    seg1 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    step_ms = [1e3 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]
    sync_all()
    elapsed = time.perf_counter() - t0
    L.diffhe_traffic_account(1, ctypes.byref(acc_bytes), ctypes.byref(acc_launches))
    kprof = read_kprof(L, ctypes) if rank == 0 else []
    if rank == 0:
        L.diffhe_lattice_pcg_profile(0, ctypes.byref(prof_ms), ctypes.byref(prof_n))
    rank_ms = [1e3 * own_elapsed / max(args.steps, 1)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        own = torch.tensor([own_elapsed], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allr, own)
        rank_ms = [1e3 * float(x[0]) / max(args.steps, 1) for x in allr]
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = world * B * args.steps / elapsed
    timed_iters = list(iters[-args.steps:]) if args.steps else list(iters)
    lattice = solver.last_info.path == "lattice-mgpcg"
    head_info = solver.last_info
    u_main, kgrad_main = u, (kappa.grad.detach().clone() if kappa.grad is not None else None)

    # ------------------------------------------------------------------------------------------------------------
    # secondary measurements, same run.  Every entry: `variant_reps` repetitions after one warm call, MEDIAN reported.
    # ------------------------------------------------------------------------------------------------------------
    def timed(fn, reps=None):
        fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(reps or args.variant_reps):
            if world > 1:
                dist.barrier()
            sg = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
            tv = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - tv)
            if os.environ.get("DIFFHE_BENCH_DEBUG"):
                st_ = torch.cuda.memory_stats(dev)
                print(f"[timed] {1e3 * ts[-1]:.2f} ms, device allocations {st_.get('segment.all.allocated', 0) - sg}, "
                      f"reserved {st_.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB, "
                      f"retries {st_.get('num_alloc_retries', 0)}", file=sys.stderr, flush=True)
        return statistics.median(ts), ts

    variants = {}
    want = (lambda name: (not args.no_variants and args.only_variant in (None, name)))

    # (A) at EVERY N: one per-element kappa field SHARED by the batch -- the (m,) gradient crosses the RCCL
    #     reduce-scatter + all-gather (BASELINE config 4: "with RCCL grad all-reduce")
    if lattice and args.kappa == "sample" and want("shared_kappa_allreduce"):
        kap_s = make_kappa("shared-field", 2025)
        solver_s = DifferentiableFESolver(mesh, kap_s, device=dev, **tol_kw)
        drv_s, step_s = make_step(solver_s, kap_s, f, node, True)
        coll_ms, solve_ms = [], []

        def one():
            torch.cuda.synchronize(dev)
            ta = time.perf_counter()
            step_s(time_collective=True)
            torch.cuda.synchronize(dev)
            tot = 1e3 * (time.perf_counter() - ta)
            c = drv_s.last_collective.get("ms", 0.0)
            coll_ms.append(c)
            solve_ms.append(tot - c)

        tv, ts = timed(one)
        coll_ms, solve_ms = coll_ms[1:], solve_ms[1:]            # drop the warm call
        st_ = dict(drv_s.last_collective)
        tmax = torch.tensor([tv, statistics.median(coll_ms)], dtype=torch.float64, device=dev)
        per_rank = [1e3 * tv]
        if world > 1:
            allr = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(allr, tmax[:1].clone())
            per_rank = [1e3 * float(x[0]) for x in allr]
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        cms = float(tmax[1])
        nbytes = st_.get("bytes", 8 * m + 8)
        variants["shared_kappa_allreduce"] = {
            "what": "same mesh and batch, ONE log-normal per-element kappa field (m,) shared by all samples and ranks: "
                    "every step all-reduces [loss, dL/dkappa (m,)] over the ranks (diffhe.distributed.allreduce_sum_fused)",
            "value": round(world * B / float(tmax[0]), 3), "unit": "solves/s (whole job)",
            "ms_per_step": round(1e3 * float(tmax[0]), 3), "solve_ms": round(statistics.median(solve_ms), 3),
            "collective_ms": round(cms, 4), "collective_bytes": int(nbytes), "collective_path": st_.get("path"),
            "staged_bytes": int(st_.get("staged_bytes", 0)), "collectives_per_step": st_.get("collectives"),
            # bus bandwidth of an all-reduce: 2 (N-1)/N x bytes / time (each byte leaves and enters every rank once)
            "bus_gbs": (round(2.0 * (world - 1) / world * nbytes / (cms * 1e-3) / 1e9, 2) if world > 1 and cms > 0 else None),
            "per_rank_step_ms": _spread(per_rank), "reps": args.variant_reps,
            "iters_fwd": solver_s.last_info.iterations, "iters_adj": solver_s.last_info.adj_iterations,
            "not_converged": solver_s.last_info.not_converged,
            "note": "world = 1: no exchange (collective_ms = 0); the timing synchronises the device around the "
                    "collective, so solve_ms + collective_ms = ms_per_step with nothing overlapped"}
        del solver_s, kap_s, drv_s
        torch.cuda.empty_cache()

    single = world == 1 and lattice and args.kappa == "sample"
    if single and want("api_layout_sample") and node:
        # the reference API's (B, n) tensors: to_node_major / to_sample_major passes in and out of every solve
        f_s = torch.ones(B, n, dtype=torch.float64, device=dev)
        _, step_api = make_step(solver, kappa, f_s, False, False)
        tv, ts = timed(step_api)
        variants["api_layout_sample"] = {"value_per_gpu": round(B / tv, 3), "ms_per_step": round(1e3 * tv, 3),
                                         "ms_all": [round(1e3 * x, 2) for x in ts],
                                         "what": "same workload through the reference API's (B, n) layout"}
        del f_s
    if single and solver.mg.get("fp32") and want("vcycle_storage_fp64"):
        solver.mg["fp32"] = 0
        tv, ts = timed(step_raw)
        variants["vcycle_storage_fp64"] = {"value_per_gpu": round(B / tv, 3), "ms_per_step": round(1e3 * tv, 3),
                                           "ms_all": [round(1e3 * x, 2) for x in ts],
                                           "iters_fwd": solver.last_info.iterations,
                                           "what": "every vector of the solve STORED fp64 (the apples-to-apples figure "
                                                   "against an fp64 reference; the headline stores p and the V-cycle fp32)"}
        solver.mg["fp32"] = 1
    if single and want("kappa_element_field"):
        # the same workload with an independent per-element kappa FIELD per sample (SURVEY 8(d) C3/C4 variant):
        # one matrix per sample and level, nothing shared or factored across the batch
        kap_e = make_kappa("element", 2025)
        solver_e = DifferentiableFESolver(mesh, kap_e, device=dev, **tol_kw)
        _, step_e = make_step(solver_e, kap_e, f, node, False)
        tv, ts = timed(step_e)
        variants["kappa_element_field"] = {
            "value_per_gpu": round(B / tv, 3), "ms_per_step": round(1e3 * tv, 3), "ms_all": [round(1e3 * x, 2) for x in ts],
            "iters_fwd": solver_e.last_info.iterations, "iters_adj": solver_e.last_info.adj_iterations,
            "tol": solver_e.tol, "tol_energy": solver_e.last_info.tol_energy,
            "stopped_by": {"fwd": solver_e.last_info.stop_rules, "adj": solver_e.last_info.adj_stop_rules},
            "not_converged": solver_e.last_info.not_converged}
        del solver_e, kap_e
        torch.cuda.empty_cache()
    if single:
        for name, fn in (("config2_1d_10000_b1024", variant_config2), ("config3_512_b256", variant_config3),
                         ("config5_512_b64_adam", variant_config5), ("general_mesh_jittered_512_b64", variant_general_mesh)):
            if want(name):
                variants[name] = fn(args, torch, L, _hip, ctypes, dev, timed)
                torch.cuda.empty_cache()

    if rank == 0:
        roof = roofline(args, torch, L, _hip, solver, plan, kappa, n, B, N, dev, prof_ms.value, prof_n.value, kprof)
        if roof is not None:
            bps = acc_bytes.value / max(args.steps, 1)
            roof["step"] = {"bytes_per_step": bps, "launches_per_step": acc_launches.value / max(args.steps, 1),
                            "achieved": round(bps / (ms_per_step * 1e-3) / 1e9, 1),
                            "frac": round(bps / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "definition": "algorithmic bytes of EVERY launch of the timed steps (library accounting, "
                                          "DESIGN.md section 4; the loss's own torch kernels excluded) / step time"}
        cpu = parity = None
        if not args.no_cpu_baseline and world == 1 and args.kappa != "shared-field":
            cpu, parity = cpu_baseline_and_parity(args, np, torch, mesh, kappa, kgrad_main, u_main, B, N, node)
        elif world == 1 and args.kappa == "sample" and N & (N - 1) == 0:
            parity = {"vs_exact_solution": exact_dst_parity(np, torch, mesh, kappa, kgrad_main, u_main, B, N, node)}
        it = timed_iters
        fp32 = bool(solver.mg.get("fp32")) and lattice

        def rule_sum(idx):
            tot = {}
            for row in it:
                for k_, v_ in row[idx].items():
                    tot[k_] = tot.get(k_, 0) + v_
            return tot

        fwd_rules, adj_rules = rule_sum(5), rule_sum(6)
        tol_e = head_info.tol_energy
        if tol_e:
            stop = (f"per sample, whichever fires first: (i) ENERGY-NORM estimate sqrt(r.z / u^T A u) <= {tol_e:g} asked of "
                    f"the final iterate (the CG iterate's own estimate may be 1/0.3 of it; it then receives one more "
                    f"multigrid correction), guards: r.z > 0, |r| within 1e4 x of the target, <= 10 iterations, closed "
                    f"lattice; (ii) RESIDUAL |r| <= max(tol |b|, 0.5 u |A| |x0|), tol = {solver.tol:g}")
        else:
            stop = (f"per sample RESIDUAL rule alone: |r| <= max(tol |b|, 0.5 u |A| |x0|), tol = {solver.tol:g}"
                    if solver.mg.get("floor", 1) else f"per sample |r| <= tol |b|, tol = {solver.tol:g}")
        v64 = variants.get("vcycle_storage_fp64", {}).get("value_per_gpu")
        vel = variants.get("kappa_element_field", {}).get("value_per_gpu")
        out = {
            "metric": f"FEM solves/sec (fwd+adjoint), 2D P1 Poisson {N}^2 mesh, batch={B} per GPU",
            "value": round(value, 4), "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "dtype_detail": ("f64 iterate / residual / reductions; fp32-stored preconditioner with packed-fp32 arithmetic, fp32 "
                             "stencil in the CG step length only (config.precision); all-fp64 storage: "
                             "variants.vcycle_storage_fp64") if fp32 else "f64 storage and arithmetic throughout",
            "data": "synthetic",
            "config": {"workload": f"C4: rectangle({N},{N}), {B} samples/GPU, "
                                   + {"sample": "kappa_b~U(0.5,2) scalar per sample, ",
                                      "element": "per-element log-normal kappa field per sample, ",
                                      "shared-field": "ONE per-element log-normal kappa field shared by the batch (gradient "
                                                      "all-reduced), "}[args.kappa]
                                   + "f=1, L=mean_b sum u^2, fwd+adjoint", "mesh": f"{N}x{N}", "batch_per_gpu": B,
                       "global_batch": B * world, "solver": head_info.path,
                       "layout": ("f, u kept (n, B): DifferentiableFESolver.forward(f, layout='node'), no transposing pass "
                                  "(variants.api_layout_sample = the same through the reference API's (B, n))") if node else
                                 "reference API layout (B, n): one transposing pass in, one out, one for the cotangent",
                       "tol": solver.tol, "tol_energy": tol_e, "stop": stop,
                       "stopped_by": {"fwd": fwd_rules, "adj": adj_rules,
                                      "note": "samples x timed steps ended by each rule; 'cap' = iteration cap (not converged)"},
                       "residual_at_exit": {"fwd_max_true_relres": max(r_[2] for r_ in it) if it else None,
                                            "adj_max_true_relres": max(r_[3] for r_ in it) if it else None,
                                            "note": "|b - A x| / |b| of the RETURNED iterate, recomputed after the solve"},
                       "multigrid": {k: v for k, v in solver.mg.items() if v is not None},
                       "operator": ("FACTORED: K_b = kappa_b * K_1, one shared unit matrix + a per-sample scale (zero "
                                    "matrix traffic; not bit-identical to the reference's sum_e kappa_b k0_e, whose "
                                    "rounded diagonal moves the exact solution by ~cond * eps: "
                                    + (f"measured distance to the refined oracle in this run u {parity['u_rel_err']:.1e}, "
                                       f"dL/dkappa {parity['dkappa_rel_err']:.1e}" if parity and "u_rel_err" in parity
                                       else "3e-11 in u / 8e-11 in dL/dkappa at 1024^2, see parity_vs_oracle of a run with "
                                            "the CPU baseline")
                                    + "); every sample is solved by its own PCG, no u(1)/kappa shortcut.  The general "
                                    "case -- a per-element field per sample, one matrix per sample -- is "
                                    "variants.kappa_element_field") if args.kappa == "sample" else
                                   ("one assembled matrix per sample and level" if args.kappa == "element" else
                                    "one assembled matrix shared by the batch (per-element field, nothing factored)"),
                       "precision": head_info.precision or ("fp64 throughout" if not fp32 else "see solver.last_info"),
                       "solver_flags": head_info.flags, "vcycle_coefficients": head_info.coeff_storage,
                       "parallelism": f"batch-sharded x{world} (diffhe.distributed.ShardedBatchSolve, fused loss all-reduce)",
                       "plan_build_s": round(t_plan, 2), "rank_affinity": affinity},
            "solver_iters": {"fwd": max(r_[0] for r_ in it) if it else None, "adj": max(r_[1] for r_ in it) if it else None,
                             "max_relres_fwd": max(r_[2] for r_ in it) if it else None,
                             "max_relres_adj": max(r_[3] for r_ in it) if it else None,
                             "not_converged": max(r_[4] for r_ in it) if it else None},
            "per_rank_step_ms": _spread(rank_ms),
            "timed_steps_ms": {**_spread(step_ms), "all": [round(x, 2) for x in step_ms],
                               "device_allocations_in_timed_region": int(seg1 - seg0),
                               "note": "rank 0's host clock after each timed step (value = steps / their SUM, nothing dropped)"},
            "headline_note": (f"{round(value, 1)} solves/s = factored scalar-kappa operator, p and V-cycle vectors stored "
                              f"fp32; ALL vectors stored fp64: {v64} solves/s/GPU; per-element kappa field per sample "
                              f"(general case, one matrix per sample): {vel} solves/s/GPU") if single and variants else None,
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_oracle": parity, "variants": variants or None,
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if rank == 0 and os.environ.get("DIFFHE_PLAN_CACHE", "").startswith(("/dev/shm/diffhe_plan_", "/tmp/diffhe_plan_")):
            import shutil
            shutil.rmtree(os.environ["DIFFHE_PLAN_CACHE"], ignore_errors=True)
        dist.destroy_process_group()


def read_kprof(L, ctypes):
    out = []
    for kid in range(6):      # in-solver HIP-event totals of the six main kernels (forward solves, fine level)
        ms_, n_ = ctypes.c_double(0.0), ctypes.c_longlong(0)
        L.diffhe_lattice_kernel_profile(kid, ctypes.byref(ms_), ctypes.byref(n_))
        out.append((ms_.value, n_.value))
    return out


KERNEL_NAMES = ["fused CG step: p = z + beta p, Ap = A p, p.Ap (x formed at the end of the solve)",
                "pcg_update_kernel (r -= alpha A p, r.r, fp32 copy of r)",
                "first two Jacobi sweeps from a zero guess",
                "residual + restriction (residual never stored)",
                "prolongation + correction + sweep",
                "Jacobi sweep"]


KERNEL_NAMES_FUSED = [KERNEL_NAMES[0], KERNEL_NAMES[1],
                      "fused PRE pass: two sweeps from 0 + residual + restriction (x2 and the coarse rhs written; 9 B)",
                      KERNEL_NAMES[3],
                      "fused POST pass: prolongation + correction + both post-sweeps (+ r.z partials; 13 B)",
                      KERNEL_NAMES[5]]


def kernel_table(kprof, n, Bp, f32, fused=False, rupd=False):
    """Per-kernel roofline rows from the in-solver HIP-event totals (fine level, forward solves)."""
    tv = 4.0 if f32 else 8.0
    bytes_ = [(3 * tv + 8.0), 24.0 + (4.0 if f32 else 0.0), 2.0 * tv, 2.25 * tv, 3.25 * tv, 3.0 * tv]
    names = list(KERNEL_NAMES)
    if fused and f32:
        bytes_[2], bytes_[4], names = 9.0, 13.0, list(KERNEL_NAMES_FUSED)
    if rupd and f32:   # A p never stored: CG step z, p_old in, p out; residual update p, r in, r and its fp32 copy out
        bytes_[0], bytes_[1] = 12.0, 24.0
        names[0], names[1] = KERNEL_NAME_CGSTEP_NOAP, KERNEL_NAME_RUPD
    rows = []
    for nm, bpn, (ms_, n_) in zip(names, bytes_, kprof):
        if n_ > 0:
            avg = ms_ * 1e-3 / n_
            byt = bpn * n * Bp
            rows.append({"kernel": nm, "achieved": round(byt / avg / 1e9, 1), "frac": round(byt / avg / 1e9 / HBM_PEAK_GBS, 4),
                         "bytes_per_launch": byt, "avg_launch_ms": round(avg * 1e3, 4), "launches_timed": int(n_),
                         "total_ms_timed": round(ms_, 3)})
    rows.sort(key=lambda e: -e["total_ms_timed"])
    return rows


def variant_config2(args, torch, L, _hip, ctypes, dev, timed):
    """BASELINE config 2: FEMesh.line(10 000), kappa = 1 shared, f_b = 1 + 0.5 randn (seed 1234), B = 1024 RHS,
    L = 0.5 sum_b |u_b|^2 / B; fwd + adjoint through the Python boundary, both chain modes; the two kernels of a
    step timed alone with HIP events for the roofline entry (40 n B per differentiable solve)."""
    from diffhe import FEMesh, DifferentiableFESolver
    from diffhe.plan import get_plan
    mesh = FEMesh.line(10_000)
    n, B = mesh.n_nodes, 1024
    gen = torch.Generator().manual_seed(1234)
    f = (1 + 0.5 * torch.randn(B, n, generator=gen, dtype=torch.float64)).to(dev).requires_grad_(True)
    plan = get_plan(mesh, dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {"what": "1D Poisson, 10 000 P1 elements, 1024 right-hand sides, kappa = 1, fwd + adjoint (dL/df and dL/dkappa)"}
    g = torch.rand(B, n, dtype=torch.float64, device=dev)
    u = torch.empty(B, n, dtype=torch.float64, device=dev)
    df = torch.empty_like(u)
    part = torch.empty(B, plan.n_seg, dtype=torch.float64, device=dev)
    kap1 = torch.ones(1, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def ktime(fn):
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(args.kernel_reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / args.kernel_reps

    for chain, flags in (("reference", _hip.CHAIN_REFERENCE_ORDER), ("exact", 0)):
        kappa = torch.tensor(1.0, dtype=torch.float64, device=dev, requires_grad=True)
        solver = DifferentiableFESolver(mesh, kappa, device=dev, chain=chain)

        def step():
            kappa.grad = None
            f.grad = None
            uu = solver(f)
            (0.5 * (uu ** 2).sum() / B).backward()

        tv, ts = timed(step, reps=max(args.variant_reps, 5))
        msl = plan.max_seg_len
        tf = ktime(lambda: _hip.check(L.diffhe_chain1d_solve(_hip.ptr(plan.x), _hip.ptr(kap1), 0, 0, _hip.ptr(f.detach()), n,
                                                             _hip.ptr(plan.seg), plan.n_seg, _hip.ptr(plan.g), _hip.ptr(u), n,
                                                             n, B, msl, flags, None, st), "chain fwd"))
        ta = ktime(lambda: _hip.check(L.diffhe_chain1d_adjoint(_hip.ptr(plan.x), _hip.ptr(kap1), 0, 0, _hip.ptr(g), n,
                                                               _hip.ptr(u), n, _hip.ptr(plan.seg), plan.n_seg, _hip.ptr(df),
                                                               n, None, 0, _hip.ptr(part), n, B, msl, flags, None, st),
                                      "chain adj"))
        gbs = 40.0 * n * B / (tf + ta) / 1e9
        out[f"chain_{chain}" + ("_default" if chain == "reference" else "")] = {
            "solves_per_s": round(B / tv, 1), "ms_per_step": round(1e3 * tv, 4), "iterations": "direct (scan)",
            "what": ("the system the reference ASSEMBLES in fp64 (rounded diagonal), two scans" if chain == "reference" else
                     "plain scan of the unrounded system") + "; step = solver(f) + torch loss + backward()",
            "kernels_only_solves_per_s": round(B / (tf + ta), 1),
            "roofline": {"bound": "hbm (instruction-bound in practice: ~140 fp64 lane-instructions per element)",
                         "kernel": "chain_reg_kernel fwd + adjoint", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch_pair": 40.0 * n * B,
                         "fwd_us": round(tf * 1e6, 1), "adj_us": round(ta * 1e6, 1)}}
    return out


def _lattice_variant_roofline(L, ctypes, n, Bp, f32):
    rows = kernel_table(read_kprof(L, ctypes), n, Bp, f32, fused=bool(L.diffhe_lattice_fused_passes()) and Bp % 64 == 0,
                        rupd=bool(L.diffhe_lattice_recompute_ap()) and f32)
    if not rows:
        return None
    top = dict(rows[0])
    top.update(bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", traffic=None,
               dominance="largest in-solver total among the six main kernels (HIP events, fine level, forward solves)")
    return top


def variant_config3(args, torch, L, _hip, ctypes, dev, timed):
    """BASELINE config 3: rectangle(512, 512), 256 kappa samples (U(0.5, 2), seed 2024), f = 1, fwd + adjoint."""
    from diffhe import FEMesh, DifferentiableFESolver
    from diffhe.plan import padded_batch
    N, B = 512, 256
    mesh = FEMesh.rectangle(N, N)
    gen = torch.Generator().manual_seed(2024)
    kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=torch.float64)).to(dev).requires_grad_(True)
    f = torch.ones(mesh.n_nodes, B, dtype=torch.float64, device=dev)
    solver = DifferentiableFESolver(mesh, kappa, device=dev)

    def step():
        kappa.grad = None
        u = solver(f, layout="node")
        (torch.linalg.vector_norm(u, dim=0).square().sum() / B).backward()

    step()
    L.diffhe_lattice_pcg_profile(1, None, None)
    tv, ts = timed(step, reps=max(args.variant_reps, 5))
    roof = _lattice_variant_roofline(L, ctypes, mesh.n_nodes, padded_batch(B), bool(solver.mg.get("fp32")))
    L.diffhe_lattice_pcg_profile(0, None, None)
    info = solver.last_info
    return {"what": "2D 512x512, 256 kappa samples (scalar per sample), f = 1, fwd + adjoint, layout='node'",
            "solves_per_s": round(B / tv, 1), "ms_per_step": round(1e3 * tv, 3), "ms_all": [round(1e3 * x, 2) for x in ts],
            "iterations": {"fwd": info.iterations, "adj": info.adj_iterations}, "not_converged": info.not_converged,
            "stopped_by": {"fwd": info.stop_rules, "adj": info.adj_stop_rules}, "max_true_relres": info.max_relres,
            "roofline": roof}


def variant_config5(args, torch, L, _hip, ctypes, dev, timed):
    """BASELINE config 5, one GPU's shard: kappa recovery on 512^2, 64 samples (seed 5), kappa_0 = 1, Adam lr 0.1 on
    kappa.abs() exactly like the reference's examples/poisson_1d_demo.py:102-110; 10 optimisation steps timed (the
    full 100-step run is tests/test_baseline_configs.py and tools/kappa_recovery.py)."""
    from diffhe import FEMesh, DifferentiableFESolver
    from diffhe.plan import padded_batch
    N, B, STEPS = 512, 64, 10
    mesh = FEMesh.rectangle(N, N)
    gen = torch.Generator().manual_seed(5)
    k_true = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=torch.float64)).to(dev)
    f = torch.ones(B, mesh.n_nodes, dtype=torch.float64, device=dev)
    with torch.no_grad():
        u_data = DifferentiableFESolver(mesh, k_true, device=dev)(f)
    scale = 1.0 / float((u_data ** 2).mean())
    state = {}

    def run(warm=False):
        k = torch.ones(B, dtype=torch.float64, device=dev, requires_grad=True)
        opt = torch.optim.Adam([k], lr=0.1)
        its = [0, 0]
        for _ in range(STEPS):
            opt.zero_grad()
            # a new solver per step, as the reference's loop (warm starts live in the per-mesh plan: they carry over)
            solver = DifferentiableFESolver(mesh, k.abs(), device=dev, warm_start="forward" if warm else False)
            u = solver(f)
            loss = ((u - u_data) ** 2).mean(dim=1).sum() * scale
            loss.backward()
            opt.step()
            its[0] += solver.last_info.iterations
            its[1] += solver.last_info.adj_iterations
        state.update(its=its, loss=float(loss.detach()), info=solver.last_info, fp32=bool(solver.mg.get("fp32")))

    run()
    L.diffhe_lattice_pcg_profile(1, None, None)
    tv, ts = timed(run)
    roof = _lattice_variant_roofline(L, ctypes, mesh.n_nodes, padded_batch(B), state["fp32"])
    L.diffhe_lattice_pcg_profile(0, None, None)
    cold = dict(state)
    tw, _ = timed(lambda: run(True))
    warm = {"what": "the same loop with warm_start='forward' (each forward solve starts from the previous step's solution)",
            "solves_per_s": round(STEPS * B / tw, 1), "ms_per_adam_step": round(1e3 * tw / STEPS, 3),
            "iterations": {"fwd_mean": state["its"][0] / STEPS, "adj_mean": state["its"][1] / STEPS},
            "loss_after_10_steps": state["loss"]}
    state.update(cold)
    return {"warm_start_forward": warm,
            "what": f"kappa recovery, 512x512, 64 samples, the first {STEPS} Adam steps (lr 0.1 on kappa.abs()) through the "
                    "adjoint solve, reference API layout (B, n), a new solver object per step",
            "adam_steps_per_s": round(STEPS / tv, 2), "solves_per_s": round(STEPS * B / tv, 1),
            "ms_per_adam_step": round(1e3 * tv / STEPS, 3), "ms_all": [round(1e3 * x / STEPS, 3) for x in ts],
            "iterations": {"fwd_mean": state["its"][0] / STEPS, "adj_mean": state["its"][1] / STEPS},
            "loss_after_10_steps": state["loss"], "roofline": roof}


def variant_general_mesh(args, torch, L, _hip, ctypes, dev, timed):
    """The general (ELL / smoothed-aggregation PCG) path, which every mesh that is not a lattice takes: rectangle(512, 512)
    with its interior nodes jittered by +-0.25 h (seed 0; the connectivity stays, the solver is told method="ell" and
    sees an unstructured mesh), 64 samples, one scalar kappa per sample ~ U(0.5, 2), f = 1; forward + loss + backward."""
    import numpy as np
    from diffhe import FEMesh, DifferentiableFESolver
    N, B = 512, 64
    base = FEMesh.rectangle(N, N)
    rng = np.random.default_rng(0)
    nodes = base.nodes.numpy().copy()
    h = 1.0 / N
    inner = (nodes[:, 0] > 1e-9) & (nodes[:, 0] < 1 - 1e-9) & (nodes[:, 1] > 1e-9) & (nodes[:, 1] < 1 - 1e-9)
    nodes[inner] += rng.uniform(-0.25 * h, 0.25 * h, (int(inner.sum()), 2))
    mesh = FEMesh(nodes=torch.from_numpy(nodes), elements=base.elements, dirichlet_nodes=dict(base.dirichlet_nodes))
    kappa = torch.from_numpy(rng.uniform(0.5, 2.0, B)).to(dev).requires_grad_(True)
    f = torch.ones(B, mesh.n_nodes, dtype=torch.float64, device=dev)
    solver = DifferentiableFESolver(mesh, kappa, device=dev, method="ell")

    def fwd():
        with torch.no_grad():
            solver(f)

    def step():
        kappa.grad = None
        u = solver(f)
        (torch.linalg.vector_norm(u, dim=1).square().sum() / B).backward()

    tf, _ = timed(fwd)
    it_f = solver.last_info.iterations
    tv, ts = timed(step)
    info = solver.last_info
    return {"what": "jittered 512 x 512 triangulation through method='ell' (smoothed-aggregation PCG, the path of every "
                    "unstructured mesh), 64 samples, scalar kappa per sample, reference API layout (B, n)",
            "value_per_gpu": round(B / tv, 1), "ms_per_step": round(1e3 * tv, 3), "ms_all": [round(1e3 * x, 2) for x in ts],
            "forward_only_ms": round(1e3 * tf, 3), "forward_ms_per_iteration": round(1e3 * tf / max(it_f, 1), 4),
            "iters_fwd": info.iterations, "iters_adj": info.adj_iterations, "path": info.path,
            "factored": bool(info.factored), "max_relres": info.max_relres, "not_converged": info.not_converged}


def dry_run(args, rank, world, dist, torch):
    """Launch-path rehearsal without a GPU: rendezvous over gloo, the barrier-bracketed timed region, the
    max-over-ranks reduction and rank 0's JSON line -- with no solve and therefore no value."""
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))           # ranks finish at different times: the MAX must win
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [1e3 * own / max(args.steps, 1)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        allr = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, torch.tensor([own], dtype=torch.float64))
        rank_ms = [1e3 * float(x[0]) / max(args.steps, 1) for x in allr]
    if rank == 0:
        print(json.dumps({"metric": f"FEM solves/sec (fwd+adjoint), 2D P1 Poisson {args.mesh}^2 mesh, "
                                    f"batch={args.batch} per GPU", "value": None, "unit": "solves/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3), "higher_is_better": True,
                          "scaling": "weak", "per_rank_step_ms": _spread(rank_ms), "dry_run": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(symbols, pass_bytes):
    """HBM bytes per launch of the named kernel SYMBOLS from the committed PMC passes (profiles/PMC_PROFILE).
    Exact symbol match: a kernel whose template arguments changed since the profile was taken must not silently lose
    (or worse, inherit) its traffic figure -- a profile for this problem size that lacks a symbol is an ERROR.
    No profile committed for this size: every entry is None."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    if not os.path.exists(path):
        return {s_: None for s_ in symbols}, f"profiles/{PMC_PROFILE} not present: traffic not reported"
    with open(path) as fh:
        pmc = json.load(fh)
    if pmc.get("pass_bytes") != pass_bytes:
        return {s_: None for s_ in symbols}, f"profiles/{PMC_PROFILE} was taken at another problem size: traffic not reported"
    out = {}
    for s_ in symbols:
        if s_ not in pmc["kernels"]:
            raise RuntimeError(f"bench.py: kernel symbol {s_!r} is not in profiles/{PMC_PROFILE}: the kernel changed since "
                               f"the PMC passes were taken -- re-collect them (profiles/README.md) or fix KERNEL_SYMBOLS")
        out[s_] = pmc["kernels"][s_]["hbm_bytes_per_launch"]
    return out, (f"profiles/{PMC_PROFILE} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, exact "
                 f"kernel symbol and problem size; NOT measured in this run)")


# exact instantiations (rocprofv3 kernel names, `(anonymous namespace)::` stripped) of the six main kernels of the
# headline configuration: fp32-stored V-cycle, factored (batch-shared) 3-diagonal operator
KERNEL_SYMBOLS_FP32 = ["dia_strip_kernel<double, float, double, 0, 4, 3, true, false, 4, 7>", "pcg_update_kernel",
                       "dia_strip2_kernel<2, 0, 3, true, 4, false, false>",
                       "dia_strip2_kernel<1, 3, 3, false, 5, false, false>",
                       "dia_strip2_kernel<2, 1, 3, false, 4, false, false>",
                       "dia_strip_kernel<float, float, double, 2, 0, 3, true, false, 4, 1>"]


KERNEL_NAME_CGSTEP_NOAP = "fused CG step: p = z + beta p stored, p.Ap with A p kept in registers (never stored; 12 B)"
KERNEL_SYMBOL_CGSTEP2 = "cgstep2_kernel<float __vector(2), 3, 4, 8>"
KERNEL_NAME_RUPD = "residual update with A p recomputed from the stored p: r -= alpha A p, r.r, fp32 copy of r (24 B)"
KERNEL_SYMBOL_RUPD = "dia_strip_kernel<double, float, double, 0, 5, 3, true, false, 4, 5>"


KERNEL_SYMBOLS_FUSED = [KERNEL_SYMBOLS_FP32[0], "pcg_update_kernel", "fused_pre_kernel<float __vector(4), 3, 2, true, 4, 2>",
                        KERNEL_SYMBOLS_FP32[3], "fused_post_kernel<float __vector(2), 3, 4, true, true, false, 4, 4>",
                        KERNEL_SYMBOLS_FP32[5]]


def roofline(args, torch, L, _hip, solver, plan, kappa, n, B, N, dev, prof_ms, prof_n, kprof=()):
    """Roofline entry of the dominant kernel: algorithmic bytes per launch / its average duration in the solver."""
    from diffhe.plan import padded_batch
    Bp = padded_batch(B)
    path = solver.last_info.path
    # the strip kernels (and their in-solver event timing) exist for wave-sized batches on strip-sized meshes only
    if not (path == "lattice-mgpcg" and args.kappa == "sample" and Bp >= 64 and N >= 191 and len(kprof) == 6):
        return None
    f32 = bool(solver.mg.get("fp32"))
    fused = bool(L.diffhe_lattice_fused_passes()) and Bp % 64 == 0 and bool(solver.mg.get("fused", 1))
    rupd = bool(L.diffhe_lattice_recompute_ap()) and f32
    table = kernel_table(kprof, n, Bp, f32, fused, rupd)
    if not table:
        return None
    symbols = list(KERNEL_SYMBOLS_FUSED if fused else KERNEL_SYMBOLS_FP32)
    if rupd:
        symbols[1] = KERNEL_SYMBOL_RUPD
        if fused:      # the fp32 CG step (packed, two samples per lane, for multiples of 128)
            symbols[0] = KERNEL_SYMBOL_CGSTEP2
    have = [sy for sy, (ms_, n_) in zip(symbols, kprof) if n_ > 0]
    traffic, src = pmc_traffic(have, 8 * n * Bp) if f32 else ({}, "fp64 storage: no PMC profile")
    names = list(KERNEL_NAMES_FUSED if fused else KERNEL_NAMES)
    if rupd:
        names[0], names[1] = KERNEL_NAME_CGSTEP_NOAP, KERNEL_NAME_RUPD
    by_name = dict(zip(names, symbols))
    for row in table:
        row["symbol"] = by_name[row["kernel"]] if f32 else None
        row["traffic"] = traffic.get(row["symbol"]) if f32 else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x = torch.rand((n, Bp), dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    e0.record()
    for _ in range(args.kernel_reps):
        y.copy_(x)
    e1.record()
    e1.synchronize()
    copy_gbs = 16.0 * n * Bp / (e0.elapsed_time(e1) * 1e-3 / args.kernel_reps) / 1e9
    del x, y
    top = table[0]
    # the dominant PHASE next to the dominant kernel: the V-cycle's passes together outweigh any single kernel
    vnames = set(names[2:6])
    vc = [row for row in table if row["kernel"] in vnames]
    phase = None
    if vc:
        v_ms = sum(r_["total_ms_timed"] for r_ in vc)
        v_bytes = sum(r_["bytes_per_launch"] * r_["launches_timed"] for r_ in vc)
        all_ms = sum(r_["total_ms_timed"] for r_ in table)
        v_traffic = [r_["traffic"] for r_ in vc]
        phase = {"phase": "V-cycle (preconditioner) passes on the fine level: " + " + ".join(r_["kernel"].split(":")[0] for r_ in vc),
                 "total_ms_timed": round(v_ms, 3), "share_of_the_timed_kernels": round(v_ms / all_ms, 3),
                 "achieved": round(v_bytes / (v_ms * 1e-3) / 1e9, 1),
                 "frac": round(v_bytes / (v_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "traffic_over_algorithmic": (round(sum(t_ * r_["launches_timed"] for t_, r_ in zip(v_traffic, vc)) / v_bytes, 3)
                                              if all(t_ is not None for t_ in v_traffic) else None),
                 "note": "aggregate algorithmic bytes / aggregate in-solver time of these launches; the coarser levels' "
                         "launches of the same kernels are not in these events (profiles/*_step_budget.txt has them)"}
    return {"bound": "hbm", "kernel": top["kernel"], "symbol": top["symbol"], "achieved": top["achieved"],
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["frac"], "traffic": top["traffic"], "traffic_source": src,
            "bytes_per_launch": top["bytes_per_launch"], "avg_launch_ms": top["avg_launch_ms"],
            "launches_timed": top["launches_timed"],
            "dominance": "largest in-solver total among the six main kernels of an iteration (HIP events on the solve's "
                         "stream, fine-level launches of the forward solves in the timed steps); the dominant PHASE -- the "
                         "V-cycle's passes together -- is `dominant_phase`",
            "dominant_phase": phase,
            "stream_copy_gbs": round(copy_gbs, 1),
            "stream_ceiling_note": "profiles/r02_stream_bench.txt: 4.7-5.6 TB/s for 1R1W..2R2W mixes, 6.3-6.5 read-only",
            "other_kernels": table[1:]}


def exact_dst_parity(np, torch, mesh, kappa, kgrad, u, B, N, node):
    """h = 1/N a power of two: every element integral is exact, the UNROUNDED assembled matrix is kappa_b x the 5-point
    Laplacian, and DST-I gives that system's exact solution (scipy, fp64 transforms: ~4e-12 of its own).  This is what
    the factored operator K_b = kappa_b K_1 solves; the reference's matrix differs from it by the roundings of its
    scatter-add (diag = fl(kappa/2 + kappa/2 + kappa + ...)), worth ~cond * eps = 3e-11 (u) / 8e-11 (dL/dkappa) here."""
    from scipy.fft import dstn, idstn
    from oracle import p1_oracle as orc
    n = mesh.n_nodes
    F = orc.load_vector(mesh.nodes.numpy(), mesh.elements.numpy(), np.ones(n)).reshape(N + 1, N + 1)[1:-1, 1:-1]
    kk = np.arange(1, N)
    lam = 4.0 - 2.0 * np.cos(np.pi * kk / N)[:, None] - 2.0 * np.cos(np.pi * kk / N)[None, :]
    u1 = np.zeros((N + 1, N + 1))
    u1[1:-1, 1:-1] = idstn(dstn(F, type=1) / lam, type=1)
    ug = u.detach().t() if node else u.detach()
    u1 = torch.from_numpy(u1.ravel()).to(ug.device)
    kd = kappa.detach()
    e_all = torch.stack([(ug[b] - u1 / kd[b]).abs().max() / (u1 / kd[b]).abs().max() for b in range(B)])
    gref = -2.0 * (u1 ** 2).sum() / kd ** 3 / B          # dL/dkappa_b = -2 L_b / kappa_b / B exactly
    g_all = (kgrad - gref).abs() / gref.abs()
    return {"u_rel_err_max": float(e_all.max()), "dkappa_rel_err_max": float(g_all.max()), "samples_checked": B,
            "against": "exact DST-I solution of kappa_b x the 5-point Laplacian (the unrounded assembled system), every "
                       "sample of the batch"}


def _oracle_sample(job):
    """Pool worker of the sparse CPU baseline: one fwd + adjoint of the oracle (scipy SuperLU, 1 thread), then the
    same with iterative refinement (extended-precision residuals): the yardstick with margin."""
    nodes, elements, bn, bv, kb, B, n = job
    import numpy as np
    from oracle import p1_oracle as orc
    t = time.perf_counter()
    uo, dk, _ = orc.solve_with_adjoint(nodes, elements, bn, bv, kb, np.ones(n), lambda u_: 2 * u_ / B, sparse=True)
    t = time.perf_counter() - t
    ur, dkr, _ = orc.solve_with_adjoint(nodes, elements, bn, bv, kb, np.ones(n), lambda u_: 2 * u_ / B, sparse=True, refine=2)
    red = (lambda d: d.sum() if np.ndim(kb) == 0 else d)
    return uo, red(dk), t, ur, red(dkr)


def cpu_baseline_and_parity(args, np, torch, mesh, kappa, kgrad, u, B, N, node):
    """CPU restatements of the reference path on this box's host cores, and -- from the same oracle
    solves -- parity of the GPU result on several samples of the batch including the first and the last."""
    import multiprocessing as mp
    from oracle import p1_oracle as orc
    from oracle import torch_dense as td
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    n = mesh.n_nodes
    bn = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64)
    bv = np.fromiter(mesh.dirichlet_nodes.values(), dtype=np.float64)
    nodes, elements = mesh.nodes.numpy(), mesh.elements.numpy()

    # (ii) sparse flavour at the bench size: the oracle (reference-order assembly, SuperLU fwd + adjoint), one
    # sample per worker process on `workers` cores at once (SuperLU itself is single-threaded)
    workers = max(1, min(cores, 16, B))
    idx = sorted({int(round(i * (B - 1) / max(workers - 1, 1))) for i in range(workers)})   # first ... last
    kap = kappa.detach()
    if args.kappa == "element" and node:
        kap, kgrad = kap.t(), kgrad.t()              # (B, m) views of the element-major tensors
    kap = kap.cpu().numpy() if args.kappa == "sample" else kap
    jobs = [(nodes, elements, bn, bv, (float(kap[b]) if args.kappa == "sample" else kap[b].cpu().numpy()), B, n) for b in idx]
    if len(jobs) > 1:
        with mp.get_context("spawn").Pool(len(jobs)) as pool:
            res = pool.map(_oracle_sample, jobs)
    else:
        res = [_oracle_sample(jobs[0])]
    single = float(np.mean([r[2] for r in res]))
    tc = float(np.max([r[2] for r in res]))          # the plain (reference-kind) solves ran side by side: slowest worker
    ug = u.detach().t() if node else u.detach()      # (B, n) view
    kg = kgrad

    def errs(which_u, which_g):
        ue, ge = 0.0, 0.0
        for b, r_ in zip(idx, res):
            uo, dk = r_[which_u], r_[which_g]
            ue = max(ue, float(np.max(np.abs(ug[b].cpu().numpy() - uo)) / np.max(np.abs(uo))))
            if args.kappa == "sample":
                ge = max(ge, float(abs(float(kg[b]) - dk) / abs(dk)))
            else:
                ge = max(ge, float(np.max(np.abs(kg[b].cpu().numpy() - dk)) / np.max(np.abs(dk))))
        return ue, ge

    ue, ge = errs(0, 1)
    uer, ger = errs(3, 4)
    parity = {"u_rel_err": uer, "dkappa_rel_err": ger, "samples_checked": idx, "tolerance": 1e-10,
              "against": "oracle/p1_oracle.py, reference-order assembly + SuperLU + iterative refinement with extended-"
                         "precision residuals (refine=2: the exact solution of the matrix the reference assembles; pinned "
                         "to the reference's own LU results on fixtures G10/G11/G13), max over the checked samples",
              "vs_raw_lu": {"u_rel_err": ue, "dkappa_rel_err": ge,
                            "against": "the same oracle WITHOUT refinement (what torch.linalg.solve / SuperLU returns): its "
                                       "own forward error cond * eps (~3e-11 u, ~9e-11 dL/dkappa at 1024^2) is included"}}
    if args.kappa == "sample" and N & (N - 1) == 0:
        parity["vs_exact_solution"] = exact_dst_parity(np, torch, mesh, kappa, kgrad, u, B, N, node)

    # (i) "reference-faithful dense" flavour: vectorised assembly -> dense torch.linalg.solve -> autograd backward
    # (oracle/torch_dense.py), all cores through torch's intra-op threads, at the sizes a dense matrix reaches
    table = []
    for label, m_ in (("C1: 1D 20 elements", orc.mesh_line(20)), ("1D 1000 elements", orc.mesh_line(1000)),
                      ("2D 32x32", orc.mesh_rectangle(32, 32)), ("2D 64x64", orc.mesh_rectangle(64, 64))):
        nn_ = len(m_[0])
        td.differentiable_solve(*m_, 1.3, np.ones(nn_))                      # warm-up (LAPACK first call)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t0 < 0.5 and reps < 200):
            td.differentiable_solve(*m_, 1.3, np.ones(nn_))
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        table.append({"config": label, "flavour": "dense torch.linalg.solve + autograd (vectorised assembly)",
                      "solves_per_s": round(1.0 / dt, 3), "ms_per_solve": round(1e3 * dt, 3), "threads": cores})
    table.append({"config": f"bench size: 2D {N}x{N}", "flavour": "sparse: reference-order assembly + SuperLU fwd + adjoint",
                  "solves_per_s": round(len(jobs) / tc, 4), "ms_per_solve": round(1e3 * tc / len(jobs), 1),
                  "threads": len(jobs), "single_sample_s": round(single, 2)})
    cpu = {"value": round(len(jobs) / tc, 4), "unit": "solves/s", "cores": len(jobs), "kind": "port",
           "host_cores_available": cores,
           "sample": f"{len(jobs)} samples of the same workload ({N}x{N}, fwd+adjoint, oracle: scipy SuperLU), one per "
                     f"process on {len(jobs)} cores at once, {tc:.1f} s for the slowest ({single:.1f} s per sample on average; "
                     f"the refined re-solves used for parity are not in this time)",
           "table": table,
           "note": "baseline only; the reference's own Python loops cannot run this size at all (dense K = 8.8 TB)"}
    return cpu, parity


if __name__ == "__main__":
    main()
