"""Where a large streaming output lies in HBM decides how fast it can be written.

On MI355X the physical address space behaves as slices of 32 GiB: write streams that stay inside one slice reach ~5.7 TB/s, the
same streams spread over two or more ~7.1 (tools/wr_blocks.hip, tools/wr_vmm.hip).  A hipMalloc tensor of
16 GB is one physical run: slow unless it happens to straddle a slice boundary -- which is why the same rollout takes 2.70 to
2.97 ms in twelve tensors allocated one after the other, each reproducibly.  snac_traj_alloc (snac_amd/trajmem.py) backs one
virtual range with chunks from two slices taking turns, the slices found by a measurement of its own; blocks built that way differ
by 1-3 %.  Either way the last word is the workload: fastest_tensor() allocates a few candidates, runs the caller's own rollout
into each and keeps the fastest.  A trajectory buffer is allocated once and written millions of times; a few seconds at start-up
buy 15-20 % on every pass.
"""
import torch


def fastest_tensor(shape, dtype, device, run, candidates=6, reps=3, min_bytes=1 << 28, alloc=None):
    """Allocate `candidates` tensors of `shape` one after the other (each is held while the next is allocated, so they lie in
    different places), time `run(tensor)` -- the caller's workload writing into it, enqueued on the current stream -- `reps` times
    on each, keep the fastest and release the rest.  Returns (tensor, report) with report = {"candidates_ms": [...], "chosen": i}.
    Tensors below `min_bytes` (256 MB) are not probed: the regions are GBs wide, a small tensor is not bound by its place.
    alloc(shape, dtype, device) -> tensor replaces torch.empty as the source of the candidates (snac_amd.trajmem.traj_empty:
    one virtual range over chunks from two slices of physical memory)."""
    if alloc is None:
        def alloc(shape, dtype, device):
            return torch.empty(shape, dtype=dtype, device=device)
    numel = 1
    for d in shape:
        numel *= int(d)
    if numel * torch.empty((), dtype=dtype).element_size() < min_bytes:
        t = alloc(shape, dtype, device)
        return t, {"candidates_ms": [], "chosen": 0, "blocks": [_describe(t)]}
    held, times, notes = [], [], []
    for _ in range(max(1, int(candidates))):
        try:
            t = alloc(shape, dtype, device)
        except RuntimeError:                                       # out of memory (SnacError is one too): choose among what there is
            break
        held.append(t)
        run(t)                                                     # the first touch (page tables)
        # a GPU that comes from idle needs ~15-30 ms of launches to reach its sustained clocks, and every
        # candidate's allocation (up to 2 s for a snac_traj_alloc block) lets it idle again: without this warm-up a candidate is
        # timed on the ramp and an issue-bound workload looks 15 % slower than it is
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(t)
        b.record()
        b.synchronize()
        for _ in range(max(0, min(200, int(30.0 / max(a.elapsed_time(b), 1e-3))))):
            run(t)
        best = None
        for _ in range(max(1, int(reps))):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run(t)
            b.record()
            b.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None else min(best, ms)
        times.append(best)
        notes.append(_describe(t))
    if not held:
        raise RuntimeError("no candidate tensor of shape %s could be allocated" % (tuple(shape),))
    i = min(range(len(times)), key=times.__getitem__)
    chosen = held[i]
    del held, t
    torch.cuda.empty_cache()                                       # the other candidates go back to the driver
    # single_block_ms: what a caller who takes the FIRST allocation gets (candidates=1) -- the spread between this and the best is
    # what placement buys; `blocks`: every candidate's own description (snac_traj_describe: layout, the allocator's microseconds per
    # GiB for the whole block and its slowest window, rebuilds), so a slow candidate explains itself
    rep = {"candidates_ms": [round(x, 4) for x in times], "chosen": i, "single_block_ms": round(times[0], 4), "blocks": notes}
    if notes[i] and notes[i].get("layout"):
        rep["layout"] = notes[i]["layout"]
    return chosen, rep


def _describe(t):
    """The allocator's own account of the block under `t`, condensed (None for hipMalloc tensors)."""
    try:
        from . import trajmem

        d = trajmem.describe(t)
    except Exception:
        return None
    if not d:
        return None
    u = d["us_per_gib"]
    return {"layout": d["layout"], "rebuilds": d["rebuilds"], "pool_groups": d["pool_groups"], "probe_launches": d["probe_launches"],
            "build_ms": d["build_ms"], "windows_slow": d["windows_slow"], "us_per_gib": {k: u[k] for k in ("fast", "slow", "block", "window_max")}}
