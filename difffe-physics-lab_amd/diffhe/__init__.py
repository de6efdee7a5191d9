"""diffhe -- MI355X-native differentiable P1-FEM solve path.

Drop-in for the public surface of danieleschmidt/DiffFE-Physics-Lab
(reference diffhe/__init__.py:6-12): same names, same call semantics; the solve
itself runs in hand-written HIP kernels (libdiffhe_hip.so, include/diffhe_hip.h).
Extras that have no reference counterpart live in submodules only
(`diffhe.distributed`: batch sharding over ranks; `diffhe.heat`: time stepping of the heat equation, the
reference's roadmap item; `diffhe._hip`: the ctypes binding).
"""
from . import loss as _loss, mesh as _mesh, neural as _neural, solver as _solver

FEMesh = _mesh.FEMesh
DifferentiableFESolver = _solver.DifferentiableFESolver
PhysicsLoss = _loss.PhysicsLoss
NeuralPDE = _neural.NeuralPDE

__all__ = ("FEMesh", "DifferentiableFESolver", "PhysicsLoss", "NeuralPDE")
__version__ = "0.1.0"          # tracks the reference release this surface mirrors
