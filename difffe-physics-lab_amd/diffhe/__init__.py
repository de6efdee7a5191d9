"""diffhe -- MI355X-native differentiable P1-FEM solve path.

Drop-in for the public surface of danieleschmidt/DiffFE-Physics-Lab
(reference diffhe/__init__.py:6-12): same names, same call semantics; the solve
itself runs in hand-written HIP kernels (libdiffhe_hip.so, include/diffhe_hip.h).
"""
from .mesh import FEMesh
from .solver import DifferentiableFESolver
from .loss import PhysicsLoss
from .neural import NeuralPDE

__version__ = "0.1.0"
__all__ = ["FEMesh", "DifferentiableFESolver", "PhysicsLoss", "NeuralPDE"]
