"""Import shim: the reference's module name for Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py, backed by the HIP path.
Parity unpinned (see the class docstring): the reference's reset() needs cv2 for a plan it discards."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs import deep_mobile_printing_2d1r_hindsight_dynamic as deep_mobile_printing_2d1r_hindsight  # noqa: E402,F401
