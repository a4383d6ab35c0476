"""Import shim: the reference's module name for script/Rainbow/env/Env2D.py (Env2DStatic / Env2DDynamic), backed by the HIP path."""
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs_rainbow import Env2DDynamic, Env2DStatic  # noqa: E402,F401
