"""Import shim: the reference's module name for Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py, backed by the HIP path.

Scripts that do `sys.path.append('<...>/Env/3D/')` and `from DMP_simulator_3d_dynamic_triangle_usedata import deep_mobile_printing_3d1r` run unchanged when the path
points here (snac_amd/Env/3D/) instead of the reference tree."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs import deep_mobile_printing_3d1r_dynamic as deep_mobile_printing_3d1r  # noqa: E402,F401
