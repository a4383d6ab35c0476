"""Pin the CPU oracle (oracle/snac_oracle.c) to the reference: replay every golden trajectory recorded
from the imported reference (tests/golden/make_golden.py) and require bit-for-bit equality."""
import hashlib

import numpy as np
import pytest

import helpers
import rng_spec


def _plan_tag(name):
    return name.split(".")[0]


@pytest.mark.parametrize("dim,dyn,name", helpers.case_ids(), ids=lambda v: str(v))
def test_oracle_replays_golden(dim, dyn, name):
    orc = helpers.oracle()
    rec = helpers.load_case(dim, dyn, name)
    table = helpers.plan_table(dim, dyn, _plan_tag(name))
    env = orc.OracleEnv(dim, dyn)
    starts = rec["ep_start"].tolist()
    ep = -1
    S = len(rec["actions"])
    H, W = (1, 34) if dim == 1 else (26, 26)
    for t in range(S):
        if ep + 1 < len(starts) and t == starts[ep + 1]:
            ep += 1
            idx = max(int(rec["ep_plan_idx"][ep]), 0)
            obs = env.reset(table[idx], idx)
            assert env.e.tb == rec["ep_total_brick"][ep]
            want = np.concatenate([rec["ep_reset_win"][ep].astype(np.float64), rec["ep_reset_sc"][ep]])
            assert obs.tobytes() == want.tobytes()
        obs, r, d = env.step(int(rec["actions"][t]), int(rec["step_size"][t]))
        want = helpers.obs_from_golden(rec, t, dim)
        assert obs.tobytes() == want.tobytes(), (name, t)
        assert r == rec["reward"][t] and d == bool(rec["done"][t]), (name, t)
        assert env.e.cb == rec["cb"][t] and env.e.cs == rec["cs"][t]
        if dim == 1:
            assert env.pos[0] == rec["pos"][t][0]
            if dyn:  # obs[0] of the 1D dynamic class carries the raw counters (D1:92-96)
                assert (float(env.e.cb), float(env.e.cs)) == tuple(rec["sc_raw"][t])
        else:
            assert env.pos == tuple(rec["pos"][t])
        if d or t == S - 1:
            assert np.array_equal(env.grid[:H * W], rec["ep_final_grid"][ep].astype(np.int32))
            assert np.float64(env.iou()).tobytes() == np.float64(rec["ep_iou"][ep]).tobytes(), (name, ep)
            assert t + 1 - starts[ep] == rec["ep_len"][ep]


@pytest.mark.parametrize("dim,dyn,name", helpers.case_ids(), ids=lambda v: str(v))
def test_mt19937_reproduces_reference_draws(dim, dyn, name):
    """np.random.seed(seed) + the env's randint calls, restated: plan index on every dynamic reset
    (random mode), one randint(1,4) per step."""
    orc = helpers.oracle()
    rec = helpers.load_case(dim, dyn, name)
    table = helpers.plan_table(dim, dyn, _plan_tag(name))
    mt = orc.MT19937(int(rec["seed"]))
    starts = set(rec["ep_start"].tolist())
    ep = -1
    for t in range(len(rec["actions"])):
        if t in starts:
            ep += 1
            if dyn and int(rec["random_choose"]):
                assert mt.randint(0, len(table)) == rec["ep_plan_idx"][ep]
            elif dyn:
                assert rec["ep_plan_idx"][ep] == ep % len(table)
        assert mt.randint(1, 4) == rec["step_size"][t], (name, t)


def test_mt19937_known_answers():
    orc = helpers.oracle()
    for s, d in helpers.digests()["mt19937"].items():
        mt = orc.MT19937(int(s))
        assert [mt.randint(1, 4) for _ in range(40)] == d["randint_1_4"]
        assert [mt.randint(0, 400) for _ in range(10)] == d["then_randint_0_400"]
        assert [mt.randint(0, 3) for _ in range(16)] == d["then_randint_3_size16"]
    # SURVEY.md section 8c known answers
    mt = orc.MT19937(0)
    assert [mt.randint(1, 4) for _ in range(20)] == [1, 2, 1, 2, 2, 3, 1, 3, 1, 1, 1, 3, 2, 3, 3, 1, 2, 2, 2, 2]


def test_static_plans_match_reference_capture():
    orc = helpers.oracle()
    z = helpers.static_plans_npz()
    for pc in (0, 1, 2):
        assert np.array_equal(orc.static_plan(1, pc), z["1d_p%d" % pc].astype(np.int32))
    for pc in (0, 1):
        assert np.array_equal(orc.static_plan(2, pc), z["2d_p%d" % pc].reshape(-1).astype(np.int32))
        assert np.array_equal(orc.static_plan(3, pc), z["3d_p%d" % pc].reshape(-1).astype(np.int32))
    assert {k: int(z[k]) for k in z.files if k.endswith("_tb")} == {
        "1d_p0_tb": 600, "1d_p1_tb": 590, "1d_p2_tb": 600, "2d_p0_tb": 148, "2d_p1_tb": 60, "3d_p0_tb": 888, "3d_p1_tb": 360}


def test_plan_dataset_digests():
    z = helpers.plans_npz()
    pins = {"2d_dense_train": "26e41f57d1ee", "3d_dense_train": "558bed850747", "1d_sin_train": "860de5942345"}
    for k, v in pins.items():
        assert hashlib.sha1(z[k].astype(np.int8).tobytes()).hexdigest().startswith(v)
        assert str(z["sha1_" + k]).startswith(v)


@pytest.mark.parametrize("entry", helpers.digests()["streams"], ids=lambda e: "%dd_%s" % (e["dim"], "dyn" if e["dynamic"] else "sta"))
def test_seed_driven_stream_digest(entry):
    """100k steps driven only by the seed: plan indices and step sizes from the MT19937 restatement,
    actions from the counter RNG; sha256 over (obs f64, reward f32, done u8) must equal the reference's."""
    orc = helpers.oracle()
    dim, dyn, seed, n = entry["dim"], entry["dynamic"], entry["seed"], entry["steps"]
    table = helpers.plan_table(dim, dyn, entry["plan"])
    actions = rng_spec.counter_actions(seed, 0, n, helpers.DIMS[dim]["A"])
    mt = orc.MT19937(seed)
    env = orc.OracleEnv(dim, dyn)
    h = hashlib.sha256()

    def reset():
        idx = mt.randint(0, len(table)) if dyn else 0
        env.reset(table[idx], idx)

    reset()
    episodes = 1
    for t in range(n):
        obs, r, d = env.step(int(actions[t]), mt.randint(1, 4))
        h.update(obs.astype("<f8").tobytes())
        h.update(np.float32(r).tobytes())
        h.update(b"\x01" if d else b"\x00")
        if d:
            reset()
            episodes += 1
    assert episodes == entry["episodes"]
    assert h.hexdigest() == entry["sha256"]


def test_counter_rng_matches_numpy_statement():
    orc = helpers.oracle()
    L = orc.lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(2))
        stream = int(rng.integers(0, 4))
        env = int(rng.integers(0, 2**40))
        t = int(rng.integers(0, 2**32))
        assert L.orc_rng_word(seed, stream, env, t) == int(rng_spec.words(seed, stream, env, t))


def test_batch_state_view_equals_the_structs_read_one_by_one():
    """OracleBatch.state() / set_total_step() go through ONE int32 view of the env structs (the per-env ctypes loops took seconds at the
    large batches of the GPU suite): the view's columns are the structs' fields."""
    orc_mod = helpers.oracle()
    for dim, dyn in ((1, False), (2, True), (3, True)):
        table = helpers.plan_table(dim, dyn, "dense_train" if dyn else "p1")
        b = orc_mod.OracleBatch(dim, dyn, 37, table, seed=5)
        b.set_total_step(23)
        b.reset()
        for t in range(40):
            b.step(t, None, None, auto_reset=True)
        st = b.state()
        envs = b.b.contents.envs
        cells = envs[0].H * envs[0].W
        for i in range(37):
            e = envs[i]
            assert e.total_step == 23
            assert list(st["grid"][i]) == list(e.grid[:cells]) and tuple(st["pos"][i]) == (e.pos[0], e.pos[1])
            assert (st["cb"][i], st["cs"][i], st["tb"][i], st["plan_idx"][i]) == (e.cb, e.cs, e.tb, e.plan_idx)
        cs0, grid0 = st["cs"].copy(), st["grid"].copy()
        b.step(40, None, None, auto_reset=True)                      # copies, not views: the next step does not change them
        assert np.array_equal(st["cs"], cs0) and np.array_equal(st["grid"], grid0)
        assert not np.array_equal(b.state()["cs"], cs0)
