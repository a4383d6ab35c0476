"""Import shim: the reference's module name for Env/3D/DMP_simulator_3d_dynamic_triangle_hindsight_replay.py, backed by the HIP path."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs import deep_mobile_printing_3d1r_hindsight_dynamic as deep_mobile_printing_3d1r_hindsight  # noqa: E402,F401
