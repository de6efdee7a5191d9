#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the authoring container (the reference lives at /root/reference
and never travels).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs are small .npz/.json files holding inputs and the reference's outputs
(data, not source).  Fixture ids follow SURVEY.md section 8(c): G1..G9.
"""
import hashlib
import json
import math
import os
import sys

import numpy as np
import torch

REF = os.environ.get("DIFFHE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from diffhe import FEMesh, DifferentiableFESolver, PhysicsLoss  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))
T64 = torch.float64


def mesh_arrays(mesh):
    bc_nodes = np.array(list(mesh.dirichlet_nodes.keys()), dtype=np.int64)
    bc_vals = np.array(list(mesh.dirichlet_nodes.values()), dtype=np.float64)
    return dict(nodes=mesh.nodes.numpy().copy(), elements=mesh.elements.numpy().copy(),
                bc_nodes=bc_nodes, bc_vals=bc_vals)


def rand(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, generator=g, dtype=T64)


def forcing(kind, mesh, seed=0):
    n = mesh.n_nodes
    if mesh.dim == 1:
        x = mesh.nodes[:, 0]
        if kind == "one":
            return torch.ones(n, dtype=T64)
        if kind == "sin":
            return (math.pi ** 2) * torch.sin(math.pi * x)
    else:
        x, y = mesh.nodes[:, 0], mesh.nodes[:, 1]
        if kind == "one":
            return torch.ones(n, dtype=T64)
        if kind == "lin":
            return x + 2 * y + 1
    if kind == "rand":
        return 1.0 + 0.5 * rand(n, seed)
    if kind == "zero":
        return torch.zeros(n, dtype=T64)
    raise ValueError(kind)


def loss_of(kind, u, data=None):
    if kind == "sum":
        return u.sum()
    if kind == "sumsq":
        return (u ** 2).sum()
    if kind == "mse":
        return ((u - data) ** 2).mean()
    raise ValueError(kind)


def solve_with_grads(mesh, kappa, f, loss_kind, data=None):
    k = torch.tensor(float(kappa), dtype=T64, requires_grad=True)
    ff = f.clone().requires_grad_(True)
    u = DifferentiableFESolver(mesh, k)(ff)
    L = loss_of(loss_kind, u, data)
    L.backward()
    return u.detach().numpy().copy(), float(L), float(k.grad), ff.grad.numpy().copy()


cases = []


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    cases.append(name)
    print("wrote", name)


# --- G1: config 1 -------------------------------------------------------------
mesh = FEMesh.line(20)
f = forcing("one", mesh)
with torch.no_grad():
    u = DifferentiableFESolver(mesh)(f)
x = mesh.nodes[:, 0]
save("g1_config1", **mesh_arrays(mesh), kappa=np.float64(1.0), f=f.numpy(), u=u.numpy(),
     exact=(x * (1 - x) / 2).numpy())

# --- G2: 1D forward ------------------------------------------------------------
i = 0
for N in (5, 10, 100):
    for (bl, br) in ((0.0, 0.0), (1.0, 2.0), (1.0, None)):
        for kappa in (1.0, 1.5, 2.0):
            for fk in ("one", "sin", "rand"):
                mesh = FEMesh.line(N, bc_left=bl, bc_right=br)
                f = forcing(fk, mesh, seed=100 + i)
                with torch.no_grad():
                    u = DifferentiableFESolver(mesh, kappa)(f)
                save(f"g2_1d_fwd_{i:03d}", **mesh_arrays(mesh), kappa=np.float64(kappa), f=f.numpy(), u=u.numpy())
                i += 1
# non-uniform interval + shifted domain
mesh = FEMesh.line(16, x_left=-1.0, x_right=3.0, bc_left=0.5, bc_right=-0.25)
f = forcing("rand", mesh, seed=7)
with torch.no_grad():
    u = DifferentiableFESolver(mesh, 0.7)(f)
save("g2_1d_fwd_shift", **mesh_arrays(mesh), kappa=np.float64(0.7), f=f.numpy(), u=u.numpy())

# --- G3: 1D grads ---------------------------------------------------------------
i = 0
for N in (5, 30, 200):
    for lk in ("sum", "sumsq", "mse"):
        for (bl, br) in ((0.0, 0.0), (1.0, 2.0)):
            mesh = FEMesh.line(N, bc_left=bl, bc_right=br)
            f = forcing("rand", mesh, seed=300 + i)
            data = 0.1 * rand(mesh.n_nodes, 900 + i)
            u, L, dk, df = solve_with_grads(mesh, 1.5, f, lk, data)
            save(f"g3_1d_grad_{i:03d}", **mesh_arrays(mesh), kappa=np.float64(1.5), f=f.numpy(), u=u,
                 loss_kind=np.array(lk), data=data.numpy(), loss=np.float64(L), dkappa=np.float64(dk), df=df)
            i += 1

# --- G4: 2D forward + grads ------------------------------------------------------
i = 0
specs = [dict(nx=4, ny=4), dict(nx=8, ny=8), dict(nx=16, ny=16),
         dict(nx=3, ny=2, x_range=(0.0, 3.0), y_range=(0.0, 1.0), bc_value=0.5)]
for spec in specs:
    for kappa in (1.0, 1.7):
        for fk in ("one", "lin", "rand"):
            if spec["nx"] == 16 and not (kappa == 1.7 and fk == "rand"):
                continue  # backward through 16x16 costs seconds each: keep one
            mesh = FEMesh.rectangle(**spec)
            f = forcing(fk, mesh, seed=400 + i)
            u, L, dk, df = solve_with_grads(mesh, kappa, f, "sumsq")
            save(f"g4_2d_{i:03d}", **mesh_arrays(mesh), kappa=np.float64(kappa), f=f.numpy(), u=u,
                 loss_kind=np.array("sumsq"), loss=np.float64(L), dkappa=np.float64(dk), df=df)
            i += 1
# forward-only at a larger size (dense 1089^2)
mesh = FEMesh.rectangle(32, 32)
f = forcing("rand", mesh, seed=499)
with torch.no_grad():
    u = DifferentiableFESolver(mesh, 1.3)(f)
save("g4_2d_fwd_32", **mesh_arrays(mesh), kappa=np.float64(1.3), f=f.numpy(), u=u.numpy())

# --- G5: assembled K, F before BCs ------------------------------------------------
for name, mesh, kappa in (("g5_asm_1d_5", FEMesh.line(5, bc_left=1.0, bc_right=2.0), 1.5),
                          ("g5_asm_2d_6", FEMesh.rectangle(6, 6), 1.0),
                          ("g5_asm_2d_3x2", FEMesh.rectangle(3, 2, (0.0, 3.0), (0.0, 1.0), 0.5), 1.7)):
    f = forcing("rand", mesh, seed=55)
    solver = DifferentiableFESolver(mesh, kappa)
    cap = {}
    orig = solver._apply_bc_and_solve

    def spy(K, F, _orig=orig, _cap=cap):
        _cap["K"], _cap["F"] = K.detach().numpy().copy(), F.detach().numpy().copy()
        return _orig(K, F)

    solver._apply_bc_and_solve = spy
    with torch.no_grad():
        u = solver(f)
    save(name, **mesh_arrays(mesh), kappa=np.float64(kappa), f=f.numpy(), K=cap["K"], F=cap["F"], u=u.numpy())

# --- G6: mesh pins ----------------------------------------------------------------
pins = {}
for name, mesh in (("line_10", FEMesh.line(10)),
                   ("line_7_shift", FEMesh.line(7, -1.0, 2.5, 0.25, None)),
                   ("rect_4_4", FEMesh.rectangle(4, 4)),
                   ("rect_3_2", FEMesh.rectangle(3, 2, (0.0, 3.0), (0.0, 1.0), 0.5))):
    save("g6_mesh_" + name, **mesh_arrays(mesh), free=np.array(mesh.free_nodes(), dtype=np.int64),
         repr=np.array(repr(mesh)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


for (nx, ny) in ((64, 64), (100, 37), (512, 512)):
    mesh = FEMesh.rectangle(nx, ny)
    ma = mesh_arrays(mesh)
    pins[f"rect_{nx}_{ny}"] = dict(nodes=sha(ma["nodes"]), elements=sha(ma["elements"]),
                                   bc_nodes=sha(ma["bc_nodes"]), n_bc=int(len(ma["bc_nodes"])))
for N in (10_000,):
    mesh = FEMesh.line(N)
    ma = mesh_arrays(mesh)
    pins[f"line_{N}"] = dict(nodes=sha(ma["nodes"]), elements=sha(ma["elements"]),
                             bc_nodes=sha(ma["bc_nodes"]), n_bc=int(len(ma["bc_nodes"])))
with open(os.path.join(OUT, "g6_mesh_sha256.json"), "w") as fh:
    json.dump(pins, fh, indent=1, sort_keys=True)

# --- G7: kappa-recovery trajectory (examples/poisson_1d_demo.py:88-112) -------------
mesh = FEMesh.line(30)
f = torch.ones(mesh.n_nodes, dtype=T64)
with torch.no_grad():
    u_data = DifferentiableFESolver(mesh, torch.tensor(2.0, dtype=T64))(f)
k_est = torch.tensor(1.0, dtype=T64, requires_grad=True)
opt = torch.optim.Adam([k_est], lr=0.1)
traj = []
for step in range(200):
    opt.zero_grad()
    u = DifferentiableFESolver(mesh, k_est.abs())(f)
    loss = ((u - u_data) ** 2).mean()
    loss.backward()
    g = float(k_est.grad)
    opt.step()
    traj.append((float(loss), g, float(k_est)))
traj = np.array(traj)
save("g7_kappa_recovery", **mesh_arrays(mesh), f=f.numpy(), u_data=u_data.numpy(), traj=traj)

# --- G8: PhysicsLoss ------------------------------------------------------------------
mesh = FEMesh.line(10)
x = mesh.nodes[:, 0]
u_pred = torch.sin(math.pi * x) / 10
fm = float(PhysicsLoss(mesh, lambda x: torch.ones_like(x), "fem_match")(u_pred))
va = float(PhysicsLoss(mesh, lambda x: torch.ones_like(x), "variational")(u_pred))
save("g8_physics_loss", **mesh_arrays(mesh), u_pred=u_pred.numpy(), f=np.ones(11), fem_match=np.float64(fm),
     variational=np.float64(va))

# --- G9: batched semantics = loop of reference solves -----------------------------------
for name, mesh in (("g9_batch_1d_50", FEMesh.line(50, bc_left=0.5, bc_right=-0.5)),
                   ("g9_batch_2d_8", FEMesh.rectangle(8, 8, bc_value=0.25))):
    B = 8
    g = torch.Generator().manual_seed(99)
    kap = 0.5 + 1.5 * torch.rand(B, generator=g, dtype=T64)
    fs = 1.0 + 0.5 * torch.randn(B, mesh.n_nodes, generator=g, dtype=T64)
    us, dks, dfs, Ls = [], [], [], []
    for b in range(B):
        u, L, dk, df = solve_with_grads(mesh, float(kap[b]), fs[b], "sumsq")
        us.append(u), dks.append(dk), dfs.append(df), Ls.append(L)
    save(name, **mesh_arrays(mesh), kappa=kap.numpy(), f=fs.numpy(), u=np.stack(us), loss=np.array(Ls),
         dkappa=np.array(dks), df=np.stack(dfs), dkappa_sum=np.float64(np.sum(dks)))

with open(os.path.join(OUT, "MANIFEST.json"), "w") as fh:
    json.dump(dict(reference="danieleschmidt/DiffFE-Physics-Lab @ /root/reference",
                   torch=torch.__version__, numpy=np.__version__, cases=cases), fh, indent=1)
print(len(cases), "fixtures")
