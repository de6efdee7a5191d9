#!/usr/bin/env python3
"""Launch the dominant kernel of the bench workload (fine-level Jacobi sweep, 1024x1024 x 256
samples, shared unit matrix + kappa_b scale) and two calibration kernels a few times each, for
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (one counter per pass, as the guide
prescribes).  Calibration kernels move a KNOWN number of bytes with the same 8 B/lane access
shape (Jacobi from a zero guess: reads rhs once, writes x once) and with torch's wide copy."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import torch  # noqa: E402
from diffhe import FEMesh, _hip  # noqa: E402
from diffhe.plan import get_plan  # noqa: E402
from diffhe.solver import _Engine, K_SAMPLE  # noqa: E402

N, B, REPS = 1024, 256, 5
dev = torch.device("cuda", 0)
plan = get_plan(FEMesh.rectangle(N, N), dev)
n = plan.n
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
eng = _Engine(plan, 1e-12, 100, 1, "gather")
vals, Bv, scale, _, _ = eng.lattice_assemble(torch.rand(B, dtype=torch.float64) + 0.5, K_SAMPLE, B, B)
arr = eng.lattice_levels(vals)
x = torch.rand((n, B), dtype=torch.float64, device=dev)
r = torch.rand((n, B), dtype=torch.float64, device=dev)
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(REPS):   # dominant kernel: dia_strip_kernel<double,...,M_JACOBI>
    _hip.check(L.diffhe_lattice_smooth(arr, Bv, _hip.ptr(scale), _hip.ptr(r), _hip.ptr(x), _hip.ptr(y), 0.8, B, st), "s")
z32 = torch.rand((n, B), dtype=torch.float32, device=dev)
p32_in = torch.rand((n, B), dtype=torch.float32, device=dev)
p2 = torch.empty_like(p32_in)
Ap = torch.empty_like(x)
ab = torch.rand(2, B, dtype=torch.float64, device=dev)
part = torch.empty(L.diffhe_lattice_blocks(n, B) * B, dtype=torch.float64, device=dev)
for _ in range(REPS):   # the kernel with the largest time share: fused CG step dia_strip_kernel<M_APPLY,F_PUPD>
    _hip.check(L.diffhe_lattice_cg_step(arr, Bv, _hip.ptr(scale), _hip.ptr(z32), 1, _hip.ptr(p32_in), _hip.ptr(p2), _hip.ptr(x),
                                        _hip.ptr(ab[0]), _hip.ptr(ab[1]), 0, _hip.ptr(Ap), _hip.ptr(part), B, st), "cg")
for _ in range(REPS):   # calibration A: dia_jacobi_kernel (xin = NULL): read 8nB, write 8nB, 8 B/lane
    _hip.check(L.diffhe_lattice_smooth(arr, Bv, _hip.ptr(scale), _hip.ptr(r), None, _hip.ptr(y), 0.8, B, st), "s0")
for _ in range(REPS):   # calibration B: torch copy, wide loads
    y.copy_(x)
torch.cuda.synchronize()
print("pass_bytes", n * B * 8)
