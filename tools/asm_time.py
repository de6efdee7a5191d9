"""Time the per-sample lattice assembly (node-per-wave vs strip form) at the bench size: python tools/asm_time.py [nx ny B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "difffe-physics-lab_amd"))
import torch
from diffhe import FEMesh, _hip
from diffhe.plan import get_plan, _stream

nx, ny, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 1024, 256)
dev = torch.device("cuda:0")
plan = get_plan(FEMesh.rectangle(nx, ny, bc_value=0.3), dev)
lev = plan.levels[0]
L, st = _hip.lib(), _stream(dev)
kap = torch.rand(lev.m, B, dtype=torch.float64, device=dev) + 0.5
v = torch.empty((lev.nd, lev.n, B), dtype=torch.float64, device=dev)
lf = torch.empty((lev.n, B), dtype=torch.float64, device=dev)
tab = lev.compact("k0ref")
flag = 1
if tab is None: tab, flag = lev.k0ref(), 0
for strip in ("0", "1", "0", "1"):
    os.environ["DIFFHE_ASM_STRIP"] = strip
    ts = []
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _hip.check(L.diffhe_lattice_assemble_rows(_hip.ptr(tab), flag, _hip.ptr(kap), B, 1, _hip.ptr(lev.is_bc), _hip.ptr(plan.g),
                                                  _hip.ptr(v), _hip.ptr(lf), lev.nx, lev.ny, lev.nd, B, st), "assemble")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    byts = (lev.m + (lev.nd + 1) * lev.n) * B * 8
    print(f"strip={strip} {min(ts[1:]):.3f} ms  {byts / min(ts[1:]) / 1e6:.0f} GB/s algorithmic")
