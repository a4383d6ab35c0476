"""Import helper for the golden-capture scripts (THIS container only).

The reference (ai4ce/SNAC, mounted read-only at /root/reference) is pure Python but
needs `gym`, which is not installed.  The reference only uses `gym.Env` /
`gym.Wrapper` as base classes, so a ten-line stand-in module is enough
(SURVEY.md section 8c).  Nothing in here runs on the GPU box: the goldens it helps to
produce are committed as data under tests/golden/.
"""
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

REF = os.environ.get("SNAC_REFERENCE", "/root/reference")


def install_gym_stub():
    if "gym" in sys.modules:
        return
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env

    gym.Env = Env
    gym.Wrapper = Wrapper
    # the MCTS variants also do `from gym import spaces` and build spaces.Discrete(action_dim)
    spaces = types.ModuleType("gym.spaces")

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    spaces.Discrete = Discrete
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces


def install_cv2_stub():
    """Env/2D/DMP_ENV_2D_dynamic_MCTS.py imports cv2 at module level; only its unused create_plan() calls into it."""
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")


def ref_available():
    return os.path.isdir(os.path.join(REF, "Env", "2D"))


def load_ref_classes():
    """Return {(dim, dynamic): class} for the six canonical reference envs."""
    import matplotlib

    matplotlib.use("Agg")
    install_gym_stub()
    for d in ("1D", "2D", "3D"):
        p = os.path.join(REF, "Env", d)
        if p not in sys.path:
            sys.path.append(p)
    names = {
        (1, False): "DMP_Env_1D_static",
        (1, True): "DMP_Env_1D_dynamic_usedata_plan",
        (2, False): "DMP_Env_2D_static",
        (2, True): "DMP_Env_2D_dynamic_usedata_plan",
        (3, False): "DMP_simulator_3d_static_circle",
        (3, True): "DMP_simulator_3d_dynamic_triangle_usedata",
    }
    out = {}
    for (dim, dyn), mod in names.items():
        m = importlib.import_module(mod)
        out[(dim, dyn)] = getattr(m, "deep_mobile_printing_%dd1r" % dim)
    return out


def dataset_path(dim, density, split):
    if dim == 1:
        return os.path.join(REF, "Env/1D/data_1d_dynamic_sin_envplan_500_%s.pkl" % split)
    return os.path.join(REF, "Env/%dD/data_%dd_dynamic_%s_envplan_500_%s.pkl" % (dim, dim, density, split))
