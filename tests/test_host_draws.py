"""CPU: the host-side randomness of the drop-in classes.  The reference draws np.random.randint(1, 4) on EVERY step()
(Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:87); the facades take the same words of numpy's global MT19937 straight from the bit
generator (snac_amd.envs._draw_step_size: 0.26 instead of 1.8 us per step) -- value for value and leaving the stream where randint
leaves it, whatever else the script draws in between."""
import numpy as np
import pytest


@pytest.mark.parametrize("seed", [0, 1, 12345, 2 ** 31 - 1])
def test_fast_step_size_draw_is_randint_1_4_on_the_global_stream(seed):
    from snac_amd.envs import _draw_step_size

    def script(draw):
        np.random.seed(seed)
        out = []
        for i in range(3000):
            out.append(int(draw()))
            if i % 7 == 0:
                out.append(float(np.random.uniform()))               # the agent's epsilon test (script/DQN/2d/DQN_2d_dynamic.py:129)
            if i % 211 == 0:
                out.append(int(np.random.randint(0, 400)))           # a dynamic reset's plan draw
            if i % 97 == 0:
                out.append(np.random.randint(5, size=3).tolist())    # multiprocess.py:83: a vector of actions
        return out

    assert script(_draw_step_size) == script(lambda: np.random.randint(1, 4))


def test_size_n_draws_are_n_scalar_draws_in_element_order():
    """VectorizedEnvWrapper draws its N step sizes as np.random.randint(1, 4, size=N); for a handful of envs it takes N scalar draws
    instead: the same words in the same order."""
    from snac_amd.envs import _draw_step_size

    np.random.seed(9)
    a = [np.random.randint(1, 4, size=n).tolist() for n in (1, 2, 3, 5, 8) * 200] + [float(np.random.uniform())]
    np.random.seed(9)
    b = [[_draw_step_size() for _ in range(n)] for n in (1, 2, 3, 5, 8) * 200] + [float(np.random.uniform())]
    assert a == b


def test_known_answers_of_the_reference_stream():
    """the first 20 randint(1, 4) after seed(0) and seed(1), as numpy gives them to the reference."""
    from snac_amd.envs import _draw_step_size

    np.random.seed(0)
    assert [_draw_step_size() for _ in range(20)] == [1, 2, 1, 2, 2, 3, 1, 3, 1, 1, 1, 3, 2, 3, 3, 1, 2, 2, 2, 2]
    np.random.seed(1)
    assert [_draw_step_size() for _ in range(20)] == [2, 1, 1, 2, 2, 1, 1, 2, 1, 2, 1, 3, 2, 3, 1, 3, 2, 3, 1, 1]


def test_fast_draw_follows_a_global_generator_swapped_after_import():
    """the fast path bound the bit generator at import.  A script that swaps the global generator afterwards
    (np.random.set_bit_generator, numpy >= 1.25, or a rebound mtrand._rand) must keep getting np.random.randint(1, 4)'s stream."""
    from snac_amd.envs import _draw_step_size

    mt = np.random.mtrand
    old = mt._rand
    try:
        if hasattr(np.random, "set_bit_generator"):
            np.random.set_bit_generator(np.random.MT19937(77))
            a = [_draw_step_size() for _ in range(200)]
            np.random.set_bit_generator(np.random.MT19937(77))
            b = [int(np.random.randint(1, 4)) for _ in range(200)]
            assert a == b
            np.random.set_bit_generator(np.random.PCG64(5))          # not an MT19937: the fast path steps aside, value for value
            a = [_draw_step_size() for _ in range(50)]
            np.random.set_bit_generator(np.random.PCG64(5))
            b = [int(np.random.randint(1, 4)) for _ in range(50)]
            assert a == b
            np.random.set_bit_generator(np.random.MT19937(3))
        np.random.seed(4)
        a = [_draw_step_size() for _ in range(100)]
        np.random.seed(4)
        assert a == [int(np.random.randint(1, 4)) for _ in range(100)]
    finally:
        if hasattr(np.random, "set_bit_generator"):
            np.random.set_bit_generator(old._bit_generator)
