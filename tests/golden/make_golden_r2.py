#!/usr/bin/env python3
"""Round-2 additions to the golden vectors, produced by IMPORTING THE REFERENCE (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py [g10] [g11] [g6] [g12]

  G10  BASELINE config 2 against the reference itself: FEMesh.line(10 000), kappa = 1, rows 0, 1 and 1023 of
       f = 1 + 0.5 randn((1024, n), seed 1234) -> u from the reference's dense assembly + torch.linalg.solve
       (reference solver.py:73-98, 153-183).  Forward only (the reference's backward is O(n_elem n^2)).
  G11  2D forward at the reference's ceiling: rectangle(64, 64), kappa = 1.3, random f (solver.py:104-147).
  G6   SHA-256 of the bench mesh rectangle(1024, 1024) (mesh.py:79-121), merged into g6_mesh_sha256.json.
  G13  1D gradients where the conditioning shows (cond = 0.4 N^2 = 1.6e6): line(2000), kappa = 1.5, random f,
       L = sum u^2 -> u, dL/dkappa, dL/df from the reference's autograd (its backward is O(N^3): ~1 min here).
  G12  RHS ensemble for PhysicsLoss: a loop of reference `PhysicsLoss(mesh, f_k, "fem_match")(u_pred)` calls
       (loss.py:78-83) over 6 forcing functions on line(40): the per-member values the batched call must return.

Outputs are data (inputs + the reference's outputs), never source.
"""
import hashlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

REF = os.environ.get("DIFFHE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from diffhe import FEMesh, DifferentiableFESolver, PhysicsLoss  # noqa: E402  (the reference)

OUT = os.path.dirname(os.path.abspath(__file__))
T64 = torch.float64
which = set(sys.argv[1:]) or {"g10", "g11", "g6", "g12", "g13"}
new_cases = []


def mesh_arrays(mesh):
    bc_nodes = np.array(list(mesh.dirichlet_nodes.keys()), dtype=np.int64)
    bc_vals = np.array(list(mesh.dirichlet_nodes.values()), dtype=np.float64)
    return dict(nodes=mesh.nodes.numpy().copy(), elements=mesh.elements.numpy().copy(),
                bc_nodes=bc_nodes, bc_vals=bc_vals)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    new_cases.append(name)
    print("wrote", name, flush=True)


if "g10" in which:
    N, B = 10_000, 1024
    mesh = FEMesh.line(N)
    gen = torch.Generator().manual_seed(1234)
    f_all = 1.0 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    rows = [0, 1, 1023]
    us = []
    for b in rows:
        t0 = time.time()
        with torch.no_grad():
            us.append(DifferentiableFESolver(mesh, 1.0)(f_all[b]).numpy().copy())
        print(f"  config 2 row {b}: reference solve {time.time() - t0:.1f} s", flush=True)
    # mesh arrays are not stored (line(10 000) is SHA-pinned by g6): keeps the fixture at ~0.5 MB
    save("g10_config2_1d_10000", n_elements=np.int64(N), batch=np.int64(B), seed=np.int64(1234),
         rows=np.array(rows, dtype=np.int64), kappa=np.float64(1.0), f=f_all[rows].numpy(), u=np.stack(us))

if "g11" in which:
    mesh = FEMesh.rectangle(64, 64)
    gen = torch.Generator().manual_seed(6464)
    f = 1.0 + 0.5 * torch.randn(mesh.n_nodes, generator=gen, dtype=T64)
    t0 = time.time()
    with torch.no_grad():
        u = DifferentiableFESolver(mesh, 1.3)(f)
    print(f"  2D 64x64: reference solve {time.time() - t0:.1f} s", flush=True)
    save("g11_2d_fwd_64", **mesh_arrays(mesh), kappa=np.float64(1.3), f=f.numpy(), u=u.numpy())

if "g6" in which:
    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    path = os.path.join(OUT, "g6_mesh_sha256.json")
    with open(path) as fh:
        pins = json.load(fh)
    t0 = time.time()
    mesh = FEMesh.rectangle(1024, 1024)
    print(f"  reference rectangle(1024, 1024): {time.time() - t0:.1f} s", flush=True)
    ma = mesh_arrays(mesh)
    pins["rect_1024_1024"] = dict(nodes=sha(ma["nodes"]), elements=sha(ma["elements"]), bc_nodes=sha(ma["bc_nodes"]),
                                  n_bc=int(len(ma["bc_nodes"])))
    with open(path, "w") as fh:
        json.dump(pins, fh, indent=1, sort_keys=True)
    print("updated g6_mesh_sha256.json", flush=True)

if "g12" in which:
    mesh = FEMesh.line(40, bc_left=0.0, bc_right=0.0)
    x = mesh.nodes[:, 0]
    u_pred = torch.sin(math.pi * x) / 10 + 0.01 * x * (1 - x)
    members = [("one", lambda x: torch.ones_like(x)),
               ("lin", lambda x: 1.0 + 2.0 * x),
               ("sin1", lambda x: (math.pi ** 2) * torch.sin(math.pi * x)),
               ("sin3", lambda x: torch.sin(3 * math.pi * x)),
               ("quad", lambda x: 4.0 * x * (1.0 - x)),
               ("exp", lambda x: torch.exp(-x))]
    vals, fs, us = [], [], []
    for _, fn in members:
        vals.append(float(PhysicsLoss(mesh, fn, "fem_match")(u_pred)))
        fs.append(fn(x).numpy().copy())
        with torch.no_grad():
            us.append(DifferentiableFESolver(mesh, 1.0)(fn(x)).numpy().copy())
    save("g12_physics_loss_ensemble", **mesh_arrays(mesh), u_pred=u_pred.numpy(), names=np.array([m[0] for m in members]),
         f=np.stack(fs), u_fem=np.stack(us), fem_match=np.array(vals))

if "g13" in which:
    mesh = FEMesh.line(2000)
    gen = torch.Generator().manual_seed(1313)
    f = (1.0 + 0.5 * torch.randn(mesh.n_nodes, generator=gen, dtype=T64)).requires_grad_(True)
    k = torch.tensor(1.5, dtype=T64, requires_grad=True)
    t0 = time.time()
    u = DifferentiableFESolver(mesh, k)(f)
    L = (u ** 2).sum()
    L.backward()
    print(f"  1D 2000 fwd+bwd: reference {time.time() - t0:.1f} s", flush=True)
    save("g13_1d_grad_2000", n_elements=np.int64(2000), kappa=np.float64(1.5), f=f.detach().numpy(),
         u=u.detach().numpy(), loss_kind=np.array("sumsq"), loss=np.float64(float(L)), dkappa=np.float64(float(k.grad)),
         df=f.grad.numpy())

mpath = os.path.join(OUT, "MANIFEST.json")
with open(mpath) as fh:
    man = json.load(fh)
man["cases"] = sorted(set(man["cases"]) | set(new_cases))
man["round2_script"] = "tests/golden/make_golden_r2.py"
with open(mpath, "w") as fh:
    json.dump(man, fh, indent=1)
print(len(new_cases), "new fixtures")
