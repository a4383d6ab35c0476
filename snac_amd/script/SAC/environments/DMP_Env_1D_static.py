"""Import shim: the reference's module name for script/SAC/environments/DMP_Env_1D_static.py (flat observation, 3-tuple step), backed by the HIP path."""
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs_ppo import deep_mobile_printing_1d1r_sac_static as deep_mobile_printing_1d1r  # noqa: E402,F401
