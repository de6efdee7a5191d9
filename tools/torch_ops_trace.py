"""Which torch-side kernels run inside one differentiable solve, and which Python line launches them:
python tools/torch_ops_trace.py [sample|element] [nx B]   (torch.profiler with stacks; ops above 50 us of device time)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "difffe-physics-lab_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from diffhe import FEMesh, DifferentiableFESolver

kind = sys.argv[1] if len(sys.argv) > 1 else "element"
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
mesh = FEMesh.rectangle(nx, nx)
n, m = mesh.n_nodes, mesh.n_elements
g = torch.Generator(device=dev).manual_seed(2025)
if kind == "element":
    kap = torch.exp(0.3 * torch.randn(B, m, generator=g, dtype=torch.float64, device=dev)).t().contiguous().requires_grad_(True)
else:
    kap = (0.5 + 1.5 * torch.rand(B, generator=g, dtype=torch.float64, device=dev)).requires_grad_(True)
f = torch.ones(n, B, dtype=torch.float64, device=dev)
solver = DifferentiableFESolver(mesh, kap, device=dev)

def step():
    kap.grad = None
    u = solver(f, layout="node")
    (torch.linalg.vector_norm(u, dim=0).square().sum() / B).backward()

step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for ev in prof.events():
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    self_dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if self_dt >= 50 and ev.name.startswith("aten::"):
        st = [s for s in (ev.stack or []) if "diffhe" in s or "torch_ops_trace" in s or "bench" in s][:3]
        rows.append((self_dt, ev.name, [tuple(i) for i in (ev.input_shapes or [])][:3] if hasattr(ev, "input_shapes") else "", st))
rows.sort(key=lambda r: -r[0])
tot = 0.0
for dt, name, shp, st in rows:
    tot += dt
    print(f"{dt:9.1f} us  {name:28s} {' <- '.join(s.strip() for s in st)}")
print(f"total {tot / 1e3:.3f} ms of aten device time in one step")
