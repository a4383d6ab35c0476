"""Import shim: the reference's module name for script/Rainbow/env/Env3D.py (Env3DStatic / Env3DDynamic), backed by the HIP path."""
import os
import sys

_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)

from snac_amd.envs_rainbow import Env3DDynamic, Env3DStatic  # noqa: E402,F401
