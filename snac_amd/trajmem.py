"""Trajectory tensors backed by snac_traj_alloc (include/snac_hip.h): one contiguous virtual range over 32 MB chunks of physical
memory from different 32 GiB slices of the address space taking turns (HIP virtual-memory API; the slices are found by measurement).

On MI355X write streams confined to one slice reach ~5.7 TB/s, spread over two ~7.1 (tools/wr_blocks.hip);
torch.empty() hands out hipMalloc memory -- one physical run -- and PyTorch-ROCm's own virtual-memory mode (expandable segments) is
not available on this platform.  traj_empty() returns an ordinary torch tensor viewing one block; the block is unmapped and released
when the tensor's storage dies.  A block of 1 GiB or more takes 0.5-1.5 s to build (and is checked: written, read back).  PyTorch is plumbing here: it only learns the pointer.
"""
import ctypes as C

import torch

from . import _lib


class _Block:
    """Owner of one snac_traj_alloc block.  torch views it through __cuda_array_interface__ (and keeps this object alive for as
    long as any tensor shares the storage); DLPack is the second route if a build refuses the first."""

    def __init__(self, nbytes, device_index, pool_cap=0):
        L = _lib.lib()
        p = C.c_void_p()
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the probe and the block's check run on torch's current stream
        stream = raw(int(device_index)) if raw is not None else torch.cuda.current_stream(int(device_index)).cuda_stream
        _lib.check(L.snac_traj_alloc_ex(int(nbytes), int(device_index), int(pool_cap), C.c_void_p(stream), C.byref(p)))
        self.ptr, self.nbytes, self.device_index = p.value, int(nbytes), int(device_index)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2,
                                         "strides": None}

    def free(self):
        ptr, self.ptr = self.ptr, None
        if ptr:
            try:
                _lib.lib().snac_traj_free(C.c_void_p(ptr))       # waits for the device, then unmaps
            except Exception:                                   # interpreter shutdown: the process is going away anyway
                pass

    def __del__(self):
        self.free()


# ---- DLPack route (kDLROCM): torch takes the device from the capsule, not from a pointer query -------------------------------
class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int), ("device_id", C.c_int)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]
_live = {}                                                       # id -> (block, managed tensor, shape array): until torch's deleter runs


@_DELETER
def _dl_deleter(mt):
    key = C.addressof(mt.contents)
    ent = _live.pop(key, None)
    if ent is not None:
        ent[0].free()                                            # a _Block unmaps, a _Lease returns its block to the cache


def _via_dlpack(block):
    shape = (C.c_int64 * 1)(block.nbytes)
    mt = _DLManagedTensor()
    mt.dl_tensor.data = block.ptr
    mt.dl_tensor.device = _DLDevice(10, block.device_index)      # kDLROCM
    mt.dl_tensor.ndim = 1
    mt.dl_tensor.dtype = _DLDataType(1, 8, 1)                    # uint8
    mt.dl_tensor.shape = shape
    mt.dl_tensor.strides = None
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _dl_deleter
    _live[C.addressof(mt)] = (block, mt, shape)
    new = C.pythonapi.PyCapsule_New
    new.restype, new.argtypes = C.py_object, [C.c_void_p, C.c_char_p, C.c_void_p]
    cap = new(C.addressof(mt), b"dltensor", None)
    try:
        return torch.utils.dlpack.from_dlpack(cap)
    except Exception:
        _live.pop(C.addressof(mt), None)
        raise


LAYOUTS = {1: "one run", 2: "three runs 32 GiB apart", 3: "measured: two slices in turn"}


def layout_of(t):
    """How the snac_traj_alloc block under tensor `t` is backed (LAYOUTS), or None if `t` does not start at such a block."""
    rc = _lib.lib().snac_traj_layout(C.c_void_p(t.untyped_storage().data_ptr()))
    return LAYOUTS.get(rc)


def describe(t):
    """What snac_traj_alloc measured while it built the block under tensor `t` (snac_traj_describe), as a dict -- times in
    microseconds per GiB written under the rollout's store shape: `fast` / `slow` = the two-slice and the single-slice level of this
    box as the probe saw them, `block` = the finished block in one launch, `windows` = each of its 1 GiB windows -- or None if `t`
    does not start at such a block."""
    info = _lib.TrajInfo()
    if _lib.lib().snac_traj_describe(C.c_void_p(t.untyped_storage().data_ptr()), C.byref(info)) != 0:
        return None
    r = lambda x: round(float(x), 1)
    return {"layout": LAYOUTS.get(info.layout), "rebuilds": info.rebuilds, "pool_groups": info.pool_groups,
            "probe_launches": info.probe_launches, "windows_slow": info.windows_slow, "build_ms": r(info.build_ms),
            "us_per_gib": {"self": r(info.self_us_per_gib), "fast": r(info.fast_us_per_gib), "slow": r(info.slow_us_per_gib),
                           "block": r(info.block_us_per_gib), "window_mean": r(info.window_mean_us_per_gib),
                           "window_max": r(info.window_max_us_per_gib),
                           "windows": [r(info.window_us[i]) for i in range(min(info.windows, _lib.TRAJ_INFO_WINDOWS))]},
            "bytes": int(info.bytes)}


def reserved_bytes():
    """Address space (not memory) this process keeps reserved for ranges it has unmapped (snac_traj_free never recycles a range)."""
    return int(_lib.lib().snac_traj_reserved_bytes())


def _nbytes(shape, dtype):
    numel = 1
    for d in shape:
        numel *= int(d)
    return numel, max(1, numel * torch.empty((), dtype=dtype).element_size())


def _view(owner, index, numel, shape, dtype):
    """A torch tensor of `shape` viewing owner.ptr in place; torch keeps `owner` alive with the storage and owner.free() runs when
    the last tensor sharing it dies."""
    try:
        flat = torch.as_tensor(owner, device=torch.device("cuda", index))
        if flat.data_ptr() != owner.ptr:                         # a copy instead of a view: not what was asked for
            raise RuntimeError("__cuda_array_interface__ was copied")
    except Exception:
        flat = _via_dlpack(owner)
    if flat.data_ptr() != owner.ptr or flat.device.index != index:
        raise RuntimeError("this PyTorch build does not view a foreign device pointer in place")
    return flat[: numel * torch.empty((), dtype=dtype).element_size()].view(dtype).view(tuple(int(d) for d in shape))


def traj_empty(shape, dtype, device, pool_cap=0):
    """torch.empty(shape, dtype=dtype, device=device) on snac_traj_alloc memory (contiguous; it holds the pattern of the library's
    own check, not zeros).  pool_cap: bytes of device memory the slice measurement may hold beyond the block while it runs
    (0 = 160 GiB, never more than half of what is free nor the last 4 GiB; snac_traj_alloc_ex).  Raises SnacError when the block cannot be allocated
    or fails its check, and RuntimeError when this PyTorch build cannot view a foreign device pointer.  The block is unmapped when
    the last tensor viewing it dies -- snac_traj_free then waits for the whole device to go idle first, so that garbage collection
    is where the wait happens.  (cached_empty() is the recycling form: what rollout() uses for its own outputs.)"""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SnacError("trajectory memory lives on a ROCm GPU")
    index = device.index if device.index is not None else torch.cuda.current_device()
    numel, nbytes = _nbytes(shape, dtype)
    with torch.cuda.device(index):
        try:
            block = _Block(nbytes, index, pool_cap)
        except _lib.SnacError as e:
            # Address space: a freed block's range is never handed out again (snac_traj_free), so a process that has allocated and
            # freed thousands of large blocks can run out of ranges to reserve (reserved_bytes() says how much is held).  The memory is
            # still there: fall back to an ordinary hipMalloc tensor -- one physical run, the single-slice write rate -- and say so.
            if "hipMemAddressReserve" not in str(e):
                raise
            import warnings

            warnings.warn("snac_traj_alloc could not reserve an address range (%.0f GiB of dead ranges held): falling back to torch.empty"
                          % (reserved_bytes() / 2 ** 30))
            return torch.empty(tuple(int(d) for d in shape), dtype=dtype, device=torch.device("cuda", index))
        return _view(block, index, numel, shape, dtype)


# ---- a cache of measured blocks: rollout()'s own outputs ---------------------------------------------------------------------------
# A measured block costs 0.1-5 s to build and its address range for the life of the process, so blocks handed out by cached_empty()
# are RECYCLED: when the last tensor viewing a block dies, the block goes back to a free list (per device and size) instead of being
# unmapped, and the next request of that size takes it from there -- steady state: no allocation, no probe launch, no new address
# range.  The free list is bounded (SNAC_TRAJ_CACHE_BYTES, default 40 GiB per device; SNAC_TRAJ_CACHE=0 switches the cache off);
# what does not fit is unmapped as before.  Same-stream reuse is ordered by the stream; a block last used on another stream is
# handed out behind a device synchronisation.
_GRAIN = 32 << 20                                                # blocks are whole 32 MB handles
_free = {}                                                       # (device index, block bytes) -> [(block, stream it was last used on)]
_stats = {"built": 0, "reused": 0, "returned": 0, "trimmed": 0}


def _cache_limit():
    import os

    if os.environ.get("SNAC_TRAJ_CACHE", "1") == "0":
        return 0
    return int(float(os.environ.get("SNAC_TRAJ_CACHE_BYTES", str(40 << 30))))


def default_pool_cap():
    """What the slice measurement of a block that rollout() allocates by itself may hold beyond the block while it runs:
    SNAC_TRAJ_POOL_CAP_BYTES, default 64 GiB (the library never takes more than half of what is free, nor the last 4 GiB)."""
    import os

    return int(float(os.environ.get("SNAC_TRAJ_POOL_CAP_BYTES", str(64 << 30))))


class _Lease:
    """What a cached_empty() tensor keeps alive: the block goes back to the free list when the storage dies."""

    def __init__(self, block, stream):
        self.block, self.stream = block, stream
        self.ptr, self.nbytes, self.device_index = block.ptr, block.nbytes, block.device_index
        self.__cuda_array_interface__ = block.__cuda_array_interface__

    def free(self):
        block, self.block = self.block, None
        if block is None:
            return
        try:
            key = (block.device_index, block.nbytes)
            held = sum(k[1] * len(v) for k, v in _free.items() if k[0] == block.device_index)
            if held + block.nbytes <= _cache_limit():
                _free.setdefault(key, []).append((block, self.stream))
                _stats["returned"] += 1
                return
        except Exception:                                       # interpreter shutdown
            pass
        _stats["trimmed"] += 1
        block.free()

    def __del__(self):
        self.free()


def cached_empty(shape, dtype, device, pool_cap=0):
    """traj_empty() from the cache of measured blocks (see above): the first request of a size builds a block, later ones reuse it
    once its tensor has died.  Sizes are rounded up to 32 MB.  Raises like traj_empty (no torch.empty fallback: the caller decides)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SnacError("trajectory memory lives on a ROCm GPU")
    index = device.index if device.index is not None else torch.cuda.current_device()
    numel, nbytes = _nbytes(shape, dtype)
    nbytes = (nbytes + _GRAIN - 1) // _GRAIN * _GRAIN
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    stream = raw(index) if raw is not None else torch.cuda.current_stream(index).cuda_stream
    lst = _free.get((index, nbytes))
    with torch.cuda.device(index):
        if lst:
            block, last = lst.pop()
            if last != stream:
                torch.cuda.synchronize(index)
            _stats["reused"] += 1
        else:
            block = _Block(nbytes, index, pool_cap)
            _stats["built"] += 1
        return _view(_Lease(block, stream), index, numel, shape, dtype)


def cache_trim(device=None):
    """Unmap every block on the free list (of one device index, or all).  Returns the number of bytes released."""
    n = 0
    for key in list(_free):
        if device is None or key[0] == device:
            for block, _ in _free.pop(key):
                n += block.nbytes
                _stats["trimmed"] += 1
                block.free()
    return n


def cache_stats():
    """{"built", "reused", "returned", "trimmed", "free_bytes"}: what the cache of measured blocks has done in this process."""
    d = dict(_stats)
    d["free_bytes"] = sum(k[1] * len(v) for k, v in _free.items())
    return d
