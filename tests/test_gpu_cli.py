"""GPU: the counterpart of `python multiprocess.py --env ... --plan_type ... --num_envs N` (multiprocess.py:34-97):
prints the three shapes the reference prints (multiprocess.py:85-87), for every env name, plus the fused form."""
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu


def _run(*args):
    out = subprocess.run([sys.executable, "-m", "snac_amd.multiprocess"] + list(args), cwd=helpers.ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    return out.stdout


@pytest.mark.parametrize("name,D", [("1DStatic", 7), ("1DDynamic", 7), ("2DStatic", 51), ("2DDynamic", 51), ("3DStatic", 51), ("3DDynamic", 51)])
def test_driver_prints_reference_shapes(name, D):
    out = _run("--env", name, "--plan_type", "0", "--num_envs", "5")
    lines = [ln.strip() for ln in out.splitlines() if ln.startswith("(")]
    assert lines == ["(5, 1, %d)" % D, "(5,)", "(5,)"]          # observations / rewards / dones, as the reference prints


def test_driver_messages_and_fused_mode():
    assert "please choose an environment" in _run()
    assert "please choose a shape" in _run("--env", "2DDynamic")
    out = _run("--env", "2DDynamic", "--plan_type", "1", "--num_envs", "4096", "--fused")
    assert "(4096, 51)" in out and "env-steps/s" in out and "'episodes'" in out
    out = _run("--env", "1DStatic", "--plan_type", "2", "--num_envs", "3", "--reference-actions")
    assert "(3, 1, 7)" in out
