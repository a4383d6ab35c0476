"""Trajectory memory (snac_traj_alloc / snac_traj_free, snac_amd/trajmem.py): blocks of the HIP virtual-memory API viewed as
torch tensors.  What is checked is that the memory behaves like any other device memory -- every kernel's output in it equals
the output in a torch.empty tensor bit for bit -- and that blocks are released."""
import ctypes as C
import gc

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_raw_abi_alloc_write_read_free():
    import torch
    from snac_amd import _lib

    L = _lib.lib()
    p = C.c_void_p()
    assert L.snac_traj_alloc(0, 0, C.byref(p)) == -1 and b"bytes" in L.snac_last_error()
    assert L.snac_traj_alloc(4096, 99, C.byref(p)) == -1 and b"device" in L.snac_last_error()
    assert L.snac_traj_alloc(4096, 0, None) == -1
    nbytes = (70 << 20) + 12345                                  # three 32 MB handles, the last one partly used
    assert L.snac_traj_alloc(nbytes, 0, C.byref(p)) == 0 and p.value and p.value % (2 << 20) == 0
    src = torch.arange(nbytes // 8, dtype=torch.int64, device="cuda")
    hip = C.CDLL("libamdhip64.so.7")                         # the SONAME: the copy torch has loaded, not a second runtime
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(p, C.c_void_p(src.data_ptr()), src.numel() * 8, 3) == 0      # device to device, across handle borders
    back = torch.empty_like(src)
    assert hip.hipMemcpy(C.c_void_p(back.data_ptr()), p, src.numel() * 8, 3) == 0
    torch.cuda.synchronize()
    assert torch.equal(src, back)
    assert L.snac_traj_free(C.c_void_p(src.data_ptr())) == -1 and b"snac_traj_alloc" in L.snac_last_error()
    assert L.snac_traj_free(p) == 0
    assert L.snac_traj_free(p) == -1                             # already gone
    assert L.snac_traj_free(None) == 0


def test_traj_empty_is_an_ordinary_tensor_and_is_released():
    import torch
    from snac_amd import trajmem

    free0 = torch.cuda.mem_get_info()[0]
    t = trajmem.traj_empty((3, 1000, 51), torch.float64, "cuda")
    assert tuple(t.shape) == (3, 1000, 51) and t.dtype == torch.float64 and t.is_contiguous() and t.device.type == "cuda"
    assert t.data_ptr() % (2 << 20) == 0
    t.copy_(torch.arange(t.numel(), dtype=torch.float64, device="cuda").view_as(t))
    assert float(t.sum().item()) == float(sum(range(t.numel())))
    v = t[1]                                                     # a view keeps the block alive
    del t
    gc.collect()
    assert float(v[0, 0].item()) == 51000.0
    assert trajmem.layout_of(v) == "one run" and trajmem.layout_of(torch.empty(4, device="cuda")) is None
    big = trajmem.traj_empty((1 << 30,), torch.uint8, "cuda")    # 1 GiB: visible in the driver's free-memory figure
    assert trajmem.layout_of(big) in ("measured: two slices in turn", "three runs 32 GiB apart")
    assert torch.cuda.mem_get_info()[0] <= free0 - (1 << 30) + (64 << 20)
    # a block of this size takes the measured layout (probe of a handle pool, near / far chunks in turn): every one of its 32
    # chunks must be its own memory -- a counter written through the whole block reads back intact
    words = big.view(torch.int64)
    words.copy_(torch.arange(words.numel(), dtype=torch.int64, device="cuda"))
    assert bool((words[1:] - words[:-1] == 1).all()) and int(words[-1].item()) == words.numel() - 1
    assert int(words[::4099].sum().item()) == sum(range(0, words.numel(), 4099))
    del big, v, words
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                     # the temporaries of the checks above (torch caches them)
    assert torch.cuda.mem_get_info()[0] >= free0 - (512 << 20)   # the 1 GiB block went back (slack: the driver's own bookkeeping)


@pytest.mark.parametrize("kind,n,T", [(2, 4096, 40), (3, 1024, 60), (1, 5000, 70)])
def test_rollout_into_trajectory_memory_equals_a_plain_tensor(kind, n, T):
    import torch
    from snac_amd import BatchedDMPEnv, trajmem

    a = BatchedDMPEnv(kind, True, n, seed=9)
    b = BatchedDMPEnv(kind, True, n, seed=9)
    a.reset(); b.reset()
    out = trajmem.traj_empty((T, n, a.obs_dim), torch.float64, a.device)
    oa, ra, da = a.rollout(T, out=out)
    ob, rb, db = b.rollout(T)
    assert oa.data_ptr() == out.data_ptr()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db)
    tiled = trajmem.traj_empty(((n + 63) // 64, T, 64, a.obs_dim), torch.float64, a.device)
    ta, _, _ = a.rollout(T, obs="tiled", out=tiled)
    tb, _, _ = b.rollout(T, obs="tiled")
    assert torch.equal(a.untile(ta), b.untile(tb))               # (rows of the last tile beyond N are never written)
    # the per-tick step() writes its rows there too
    row = trajmem.traj_empty((n, a.obs_dim), torch.float64, a.device)
    rew, done = torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
    sa = a.step(auto_reset=True, out=(row, rew, done))
    sb = b.step(auto_reset=True)
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and torch.equal(sa[2], sb[2])


def test_alloc_trajectory_uses_trajectory_memory_by_default():
    import torch
    from snac_amd import BatchedDMPEnv

    env = BatchedDMPEnv(2, True, 2048, seed=3)
    env.reset()
    out, rep = env.alloc_trajectory(30)
    assert rep["memory"] == "vmm" and out.data_ptr() % (2 << 20) == 0 and tuple(out.shape) == (30, 2048, 51)
    out2, rep2 = env.alloc_trajectory(30, memory="malloc")
    assert rep2["memory"] == "malloc"
    with pytest.raises(ValueError):
        env.alloc_trajectory(30, memory="host")
    ref = BatchedDMPEnv(2, True, 2048, seed=3)
    ref.reset()
    o1, _, _ = env.rollout(30, out=out)
    o2, _, _ = ref.rollout(30, out=out2)
    assert torch.equal(o1, o2)
    assert np.isfinite(o1.cpu().numpy()).all()


def test_blocks_allocated_and_freed_in_turn_always_read_back_what_was_written():
    """Address-range churn: blocks of changing sizes allocated, filled by a kernel, read back through a copy engine and freed, with
    hipMalloc blocks coming and going in between -- a fresh block must never serve a stale translation of an earlier one."""
    import torch
    from snac_amd import trajmem

    for i in range(60):
        n = (1 + (i * 7) % 35) << 20                             # 1 .. 35 M float16 = 2 .. 70 MB
        t = trajmem.traj_empty((n,), torch.float16, "cuda")
        t.fill_(float(i % 200))
        spare = torch.empty((3 + i % 5) << 20, dtype=torch.uint8, device="cuda")
        host = t.cpu()
        assert float(host.min()) == float(i % 200) == float(host.max()), i
        assert float(t[-1].item()) == float(i % 200)
        del t, spare, host
        if i % 4 == 0:
            torch.cuda.empty_cache()


def test_fixed_three_run_layout_when_the_measurement_is_switched_off(monkeypatch):
    """SNAC_TRAJ_PROBE=0: the fallback layout (runs created 32 GiB apart, chunk j -> run j % 3) is a block like any other."""
    import torch
    from snac_amd import BatchedDMPEnv, trajmem

    monkeypatch.setenv("SNAC_TRAJ_PROBE", "0")
    blk = trajmem.traj_empty(((1 << 30) + (40 << 20),), torch.uint8, "cuda")          # 34 chunks: runs of 12 / 11 / 11
    assert trajmem.layout_of(blk) == "three runs 32 GiB apart"
    words = blk[: (1 << 30)].view(torch.int64)
    words.copy_(torch.arange(words.numel(), dtype=torch.int64, device="cuda"))
    assert int(words[::4099].sum().item()) == sum(range(0, words.numel(), 4099))
    assert bool((words[1:] - words[:-1] == 1).all())
    a = BatchedDMPEnv(2, True, 4096, seed=4)
    b = BatchedDMPEnv(2, True, 4096, seed=4)
    a.reset(); b.reset()
    out = blk[: 60 * 4096 * 51 * 8].view(torch.float64).view(60, 4096, 51)           # 100 MB across four chunk borders
    oa, _, _ = a.rollout(60, out=out)
    ob, _, _ = b.rollout(60)
    assert torch.equal(oa, ob)


def test_thirty_two_large_blocks_allocated_checked_and_freed_in_turn():
    """Round 3: every block passes the library's own check (a pattern written by one kernel, read back by another and, one word per
    chunk, by a copy) before it is handed out; here 32 blocks of 1.0-1.3 GiB are built one after the other on a side stream, filled
    by torch, read back through a copy and through a kernel, and freed.  Every one has the measured layout or is an explicit
    fallback, none fails (how long a build takes is noted for tests/test_zz_gpu_perf.py; no clock decides this test)."""
    import time

    import torch
    from snac_amd import trajmem

    side = torch.cuda.Stream()
    layouts, secs = [], []
    with torch.cuda.stream(side):
        for i in range(32):
            n = (1 << 27) + (i % 5) * (1 << 23)                  # int64 words: 1.0 .. 1.25 GiB
            t0 = time.perf_counter()
            t = trajmem.traj_empty((n,), torch.int64, "cuda", pool_cap=16 << 30)
            secs.append(time.perf_counter() - t0)
            layouts.append(trajmem.layout_of(t))
            t.copy_(torch.arange(n, dtype=torch.int64, device="cuda") + i)
            assert int(t[::65537].sum().item()) == sum(range(i, n + i, 65537)), i
            host = t[n - 4096:].cpu()
            assert int(host[0]) == n - 4096 + i and int(host[-1]) == n - 1 + i, i
            del t, host
    side.synchronize()
    assert all(x in ("measured: two slices in turn", "three runs 32 GiB apart") for x in layouts), layouts
    assert layouts.count("measured: two slices in turn") >= 24, layouts
    helpers.perf_note("trajmem_32_blocks_build_s", {"median": sorted(secs)[len(secs) // 2], "max": max(secs)})   # judged in tests/test_zz_gpu_perf.py


def test_pool_cap_bounds_what_the_measurement_holds():
    """pool_cap: with 1 GiB of room beyond the block there is nothing worth probing -- the block takes the fixed layout at once and the
    free-memory figure never dips by more than the block and its gaps."""
    import torch
    from snac_amd import trajmem

    blk = trajmem.traj_empty(((1 << 30) + (64 << 20),), torch.uint8, "cuda", pool_cap=1 << 30)
    assert trajmem.layout_of(blk) == "three runs 32 GiB apart"
    blk.fill_(7)
    assert int(blk[-1].item()) == 7 and int(blk[::1 << 20].sum().item()) == 7 * ((blk.numel() + (1 << 20) - 1) >> 20)


def test_headline_sized_blocks_in_turn_hold_the_same_rollout():
    """Four headline-sized blocks (65 536 envs x 600 ticks, float64 rows: 16 GB) allocated, rolled into and freed in turn: every one
    describes itself (snac_traj_describe) and holds, row for row, what the same pass writes into the next block -- the memory is
    ordinary memory whatever its backing.  How FAST the pass is on such a block is judged in tests/test_zz_gpu_perf.py (last in the
    suite, reporting): no clock decides this test."""
    import torch
    from snac_amd import BatchedDMPEnv, trajmem

    n, T = 65536, 600
    env = BatchedDMPEnv(2, True, n, seed=1)
    digests = []
    for _ in range(4):
        buf = trajmem.traj_empty((T, n, env.obs_dim), torch.float64, "cuda")
        d = trajmem.describe(buf)
        assert d["layout"] in ("measured: two slices in turn", "three runs 32 GiB apart"), d
        e = BatchedDMPEnv(2, True, n, seed=1)
        e.reset()
        e.rollout(T, obs="all", out=buf, want_reward=False, want_done=False)
        w = buf.view(torch.int64)
        digests.append((int(w[::4099].sum().item()), int(w[T - 1].sum().item()), int(w[0].sum().item())))
        del buf, w, e
    assert len(set(digests)) == 1, digests
    assert trajmem.reserved_bytes() > 0


def test_cache_of_measured_blocks_recycles_instead_of_reallocating():
    """cached_empty(): 1000 allocate / release turns of a 1 GiB block build ONE block -- no probe launches, no new address range,
    snac_traj_reserved_bytes() does not move -- and what is written through one turn's tensor is what the next turn's tensor shows
    (it is the same memory).  cache_trim() unmaps it."""
    import torch
    from snac_amd import trajmem

    trajmem.cache_trim()
    before, dead0 = trajmem.cache_stats(), None
    ptr = None
    for i in range(1000):
        t = trajmem.cached_empty((1 << 27,), torch.float64, "cuda:0")        # 1 GiB
        if ptr is None:
            ptr = t.data_ptr()
            assert trajmem.layout_of(t) is not None
            dead0 = trajmem.reserved_bytes()                     # (the build's own probe ranges are dead by now)
        assert t.data_ptr() == ptr
        if i % 250 == 0:
            t[-5:] = float(i)
            assert t[-1].item() == float(i)
        del t
    after = trajmem.cache_stats()
    assert after["built"] - before["built"] == 1 and after["reused"] - before["reused"] == 999
    assert after["free_bytes"] == 1 << 30 and trajmem.reserved_bytes() == dead0
    # two tensors alive at once are two blocks; both come back
    a = trajmem.cached_empty((1 << 27,), torch.float64, "cuda:0")
    b = trajmem.cached_empty((1 << 27,), torch.float64, "cuda:0")
    assert a.data_ptr() != b.data_ptr() and a.data_ptr() == ptr
    dead1 = trajmem.reserved_bytes()
    del a, b
    gc.collect()
    assert trajmem.cache_stats()["free_bytes"] == 2 << 30 and trajmem.reserved_bytes() == dead1
    assert trajmem.cache_trim() == 2 << 30 and trajmem.cache_stats()["free_bytes"] == 0
    assert trajmem.reserved_bytes() >= dead1 + (2 << 30)                     # unmapped ranges stay reserved (never recycled)


def test_rollout_allocates_its_own_large_outputs_from_the_cache():
    """rollout(T) without out= writes rows of a GiB and more into a measured trajectory block (the fast
    memory is the default, not an opt-in), the block is recycled from call to call, and the rows equal those of an explicit
    torch.empty output."""
    import torch
    from snac_amd import BatchedDMPEnv, trajmem

    trajmem.cache_trim()
    n, T = 65536, 48                                             # 48 x 65536 x 51 x 8 B = 1.28 GB
    env = BatchedDMPEnv(2, True, n, seed=9)
    twin = BatchedDMPEnv(2, True, n, seed=9)
    env.reset(), twin.reset()
    built0 = trajmem.cache_stats()["built"]
    ptrs = set()
    for turn in range(4):
        o, r, d = env.rollout(T)
        assert trajmem.layout_of(o) is not None and o.shape == (T, n, 51) and o.is_contiguous()
        ptrs.add(o.data_ptr())
        ref = torch.empty((T, n, 51), dtype=torch.float64, device="cuda")
        o2, r2, d2 = twin.rollout(T, out=ref)
        assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2)
        del o, o2, ref
    assert len(ptrs) == 1 and trajmem.cache_stats()["built"] - built0 == 1
    small, _, _ = env.rollout(4)                                 # 107 MB: an ordinary tensor
    assert trajmem.layout_of(small) is None
    trajmem.cache_trim()
