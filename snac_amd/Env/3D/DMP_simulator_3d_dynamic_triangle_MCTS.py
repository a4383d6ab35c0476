"""Import shim: the reference's module name for Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py (MCTS variant: state tuples + transition()), backed by the HIP path."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs_mcts import deep_mobile_printing_3d1r_MCTS_dynamic as deep_mobile_printing_3d1r  # noqa: E402,F401
