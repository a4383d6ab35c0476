"""CPU: the exhaustive check behind the division-free observation scalars of the pipelined 3D rollout kernel
(tests/native/recip_check.c): reciprocal + two fused multiply-adds == IEEE float64 division on the whole integer domain."""
import os
import subprocess

import helpers


def test_reciprocal_with_fma_correction_is_the_ieee_quotient(tmp_path):
    src = os.path.join(helpers.TESTS, "native", "recip_check.c")
    exe = str(tmp_path / "recip_check")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe, src, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    words = out.stdout.split()
    assert int(words[3]) == 0 and int(words[1]) > 0 and int(words[5]) == 32767 * 32768   # the correction is needed, and sufficient
