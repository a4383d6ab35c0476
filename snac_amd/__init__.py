"""snac_amd -- MI355X-native batched simulator for ai4ce/SNAC's mobile-construction grid worlds.

The product path is HIP only (libsnac_hip.so behind include/snac_hip.h); nothing here falls back to a CPU
implementation.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import SnacError, build  # noqa: F401
from . import plans  # noqa: F401

__all__ = ["BatchedDMPEnv", "VectorizedEnvWrapper", "ReplayRing", "NodePool2D", "SnacError", "build", "plans"]


def __getattr__(name):  # torch is imported lazily so that `import snac_amd` stays cheap
    if name == "BatchedDMPEnv":
        from .batched import BatchedDMPEnv

        return BatchedDMPEnv
    if name == "ReplayRing":
        from .replay import ReplayRing

        return ReplayRing
    if name == "NodePool2D":
        from .nodes import NodePool2D

        return NodePool2D
    if name == "VectorizedEnvWrapper":
        from .vector import VectorizedEnvWrapper

        return VectorizedEnvWrapper
    raise AttributeError(name)
