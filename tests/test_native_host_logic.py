"""CPU: the HOST logic of the library under sanitizers.  snac_amd/csrc/snac_traj.hip (pools, handles,
windows, rebuilds, reserved ranges: 700 lines of virtual-memory bookkeeping) and snac_amd/csrc/mailbox_host.h (the host half of the
resident stepper's protocol) are compiled AS THEY ARE by gcc against a fake HIP layer (tests/native/fakehip: real virtual memory behind
the VMM calls, worker threads behind streams, a simulated device with HBM slices, failure injection at every call) with
-fsanitize=address,undefined -- the mailbox also with -fsanitize=thread -- and driven through: allocate / release turns, every HIP
call failing in turn, pool cap and device memory exhausted, rebuilds, foreign pointers; 20 000 mailbox steps, idle exits, four waves,
queued launches, withdrawn commands.  GPU sanitizers do not exist on this pool; these are the CPU ones."""
import os
import subprocess

import pytest

import helpers

NATIVE = os.path.join(helpers.TESTS, "native")
INC = ["-I", os.path.join(NATIVE, "fakehip"), "-I", os.path.join(helpers.ROOT, "snac_amd", "csrc"), "-I", os.path.join(helpers.ROOT, "include")]


def _build_and_run(tmp_path, src, san, timeout):
    exe = str(tmp_path / (os.path.splitext(src)[0] + "_" + san.replace(",", "_")))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-sanitize-recover=undefined"] + INC + \
          [os.path.join(NATIVE, src), os.path.join(NATIVE, "fakehip", "fakehip.cpp"), "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    env.pop("SNAC_MAILBOX_TIMEOUT_S", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "all checks passed" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    return r.stdout


def test_trajectory_memory_bookkeeping_under_asan_ubsan(tmp_path):
    out = _build_and_run(tmp_path, "traj_host_test.cpp", "address,undefined", 900)
    assert "ended in an error" in out and "snac_traj_reserved_bytes()" in out


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_mailbox_host_protocol_under_sanitizers(tmp_path, san):
    _build_and_run(tmp_path, "mailbox_host_test.cpp", san, 600)


def test_traj_empty_falls_back_to_torch_empty_only_for_an_exhausted_address_space(monkeypatch):
    """snac_amd/trajmem.py traj_empty: a process that has used up its address ranges (freed blocks keep theirs) gets an ordinary
    hipMalloc tensor and a warning; any other failure of the allocator is raised.  Injected: no GPU needed."""
    import contextlib
    import warnings

    import torch

    from snac_amd import _lib, trajmem

    made = []

    class Boom:
        def __init__(self, nbytes, index, pool_cap=0):
            raise _lib.SnacError(self.msg)

    monkeypatch.setattr(trajmem, "_Block", Boom)
    monkeypatch.setattr(trajmem, "reserved_bytes", lambda: 123 << 30)
    monkeypatch.setattr(torch.cuda, "device", lambda index: contextlib.nullcontext())
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **kw: (made.append((a, kw)), real_empty(*a, **{k: v for k, v in kw.items() if k != "device"}))[1])
    Boom.msg = "hipMemAddressReserve: out of memory"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t = trajmem.traj_empty((3, 5, 7), torch.float64, "cuda:0")
    assert tuple(t.shape) == (3, 5, 7) and t.dtype == torch.float64
    assert any("falling back to torch.empty" in str(x.message) and "123 GiB" in str(x.message) for x in w)
    assert made and made[-1][1]["device"] == torch.device("cuda", 0)
    Boom.msg = "hipMemCreate (out of device memory?): out of memory"
    with pytest.raises(_lib.SnacError):
        trajmem.traj_empty((3, 5, 7), torch.float64, "cuda:0")
    with pytest.raises(_lib.SnacError):
        trajmem.traj_empty((4,), torch.float32, "cpu")


def test_trajectory_cache_recycles_bounds_and_trims_without_a_gpu(monkeypatch):
    """snac_amd/trajmem.py cached_empty / _Lease / cache_trim (the recycling cache behind rollout()'s own outputs) with the block and the
    torch view injected: a released block is reused by the next request of its size, a request on another stream waits for the device
    first, the free list is bounded by SNAC_TRAJ_CACHE_BYTES (what does not fit is unmapped), cache_trim() unmaps the rest, sizes are
    rounded to whole 32 MB handles."""
    import contextlib

    import torch

    from snac_amd import trajmem

    built, freed, syncs = [], [], []

    class FakeBlock:
        def __init__(self, nbytes, index, pool_cap=0):
            self.ptr, self.nbytes, self.device_index = 0x7000000000 + len(built) * (1 << 34), int(nbytes), int(index)
            self.__cuda_array_interface__ = {}
            built.append(self)

        def free(self):
            freed.append(self)

    class View:                                                     # what _view() hands out: keeps its owner (a _Lease) alive
        def __init__(self, owner):
            self.owner = owner

    stream = [111]
    monkeypatch.setattr(trajmem, "_Block", FakeBlock)
    monkeypatch.setattr(trajmem, "_view", lambda owner, index, numel, shape, dtype: View(owner))
    monkeypatch.setattr(trajmem, "_free", {})
    monkeypatch.setattr(trajmem, "_stats", {"built": 0, "reused": 0, "returned": 0, "trimmed": 0})
    monkeypatch.setattr(torch.cuda, "device", lambda index: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda index=None: syncs.append(index))
    monkeypatch.setattr(torch._C, "_cuda_getCurrentRawStream", lambda index: stream[0], raising=False)
    monkeypatch.setenv("SNAC_TRAJ_CACHE_BYTES", str(3 << 30))
    monkeypatch.delenv("SNAC_TRAJ_CACHE", raising=False)
    G = 1 << 30
    a = trajmem.cached_empty((G + 5,), torch.uint8, "cuda:0")
    assert len(built) == 1 and built[0].nbytes == G + (32 << 20)      # whole 32 MB handles
    blk = a.owner.block
    a.owner.free()                                                   # the last tensor viewing the block died
    assert trajmem.cache_stats() == {"built": 1, "reused": 0, "returned": 1, "trimmed": 0, "free_bytes": G + (32 << 20)} and not freed
    b = trajmem.cached_empty((G + 9,), torch.uint8, "cuda:0")        # the same size class: recycled, same stream: no wait
    assert b.owner.block is blk and len(built) == 1 and not syncs and trajmem.cache_stats()["reused"] == 1
    b.owner.free()
    b.owner.free()                                                   # (idempotent)
    stream[0] = 222
    c = trajmem.cached_empty((G + 9,), torch.uint8, "cuda:0")        # another stream: the device goes idle before the block is handed out
    assert c.owner.block is blk and syncs == [0]
    d = trajmem.cached_empty((2 * G,), torch.uint8, "cuda:0")        # another size: its own block
    e = trajmem.cached_empty((2 * G,), torch.uint8, "cuda:0")
    assert len(built) == 3
    c.owner.free(); d.owner.free()                                   # 1.03 + 2 GiB would be 3.03: above the 3 GiB bound
    held = trajmem.cache_stats()["free_bytes"]
    assert held <= (3 << 30) and (held == G + (32 << 20) and freed == [built[1]])     # d did not fit: unmapped at once
    e.owner.free()                                                   # 1.03 + 2 = 3.03 GiB again: unmapped too
    assert freed == [built[1], built[2]] and trajmem.cache_stats()["trimmed"] == 2
    assert trajmem.cache_trim(1) == 0                                # another device: nothing
    assert trajmem.cache_trim() == G + (32 << 20) and freed[-1] is blk and trajmem.cache_stats()["free_bytes"] == 0
    monkeypatch.setenv("SNAC_TRAJ_CACHE", "0")                       # the cache switched off: a released block is unmapped
    f = trajmem.cached_empty((G,), torch.uint8, "cuda:0")
    f.owner.free()
    assert freed[-1] is built[3] and trajmem.cache_stats()["free_bytes"] == 0
    with pytest.raises(trajmem._lib.SnacError):
        trajmem.cached_empty((4,), torch.uint8, "cpu")
