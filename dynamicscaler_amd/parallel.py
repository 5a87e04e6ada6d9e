"""Multi-GPU sharding of one DDIM step's tiles: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference is sequential (Gauss-Seidel over tiles).  Tiles whose ring footprints do not overlap are
independent, so a step is cut into LEVELS of pairwise-disjoint windows (`plan_levels`), and the overlap graph of a
step falls into connected COMPONENTS (`plan_components`: with no W overlap every panorama column is one -- configs
2-5).  `run_step` shares the step out in one of two exact ways:

  components  (>= world components): a rank owns whole components, walks their levels locally (its own tiles go
              straight into its replica) and the step has ONE exchange at its end -- an ALL-GATHER of every rank's
              (x_prev, x0) tiles, after which each rank scatters the others' tiles.  SURVEY.md 8-e's column ownership.
  levels      (fewer components than ranks, e.g. W-overlapped rings = one chain): each rank takes a strided share of
              every level and the level's tiles are all-gathered before the next level starts.
  units       (a level with fewer tiles than ranks, e.g. config 2 -- 4 columns -- on 8 GPUs): the unit of work is one
              UNet EVALUATION, (tile, cond | uncond): the two forwards of a tile are independent
              (pipeline/t2v_sphere_panorama_pipeline.py:588-599), so they go to different ranks; every rank does the
              level's cheap tile ops (gather, re-noise) itself, evaluates its share of the units, the eps tensors are
              all-gathered and every rank finishes CFG + DDIM + scatter for all tiles of the level.

Either exchange is identical in result to an all-reduce(sum) of zero-filled panorama-sized accumulators with a 0/1
weight map (the "overlap accumulator" reading of north_star) at a fraction of the bytes.  Every rank scatters all
tiles into its own replica of the panorama in an order that keeps every overlapping pair in reference order, so
replicas stay bit-identical to the single-process panorama.
"""
import torch
import torch.distributed as dist


def _ring_overlap(a0, la, b0, lb, size):
    return ((b0 - a0) % size) < la or ((a0 - b0) % size) < lb


def windows_overlap(w1, w2, pano_fhw):
    """Windows (left, right, top, down, f_begin, f_end) on a ring of size (F, H, W)."""
    F, H, W = pano_fhw
    return (_ring_overlap(w1[0], w1[1] - w1[0], w2[0], w2[1] - w2[0], W)
            and _ring_overlap(w1[2], w1[3] - w1[2], w2[2], w2[3] - w2[2], H)
            and _ring_overlap(w1[4], w1[5] - w1[4], w2[4], w2[5] - w2[4], F))


def plan_levels(windows, pano_fhw):
    """Group the step's windows (reference order) into levels: level(j) = 1 + max level of any EARLIER window
    that overlaps j.  Windows of one level are pairwise disjoint, and every overlapping pair keeps its reference
    order across levels, so processing level by level reproduces the sequential result exactly.
    Returns a list of lists of window indices."""
    level = []
    for j, wj in enumerate(windows):
        lv = 0
        for k in range(j):
            if level[k] >= lv and windows_overlap(windows[k], wj, pano_fhw):
                lv = level[k] + 1
        level.append(lv)
    out = [[] for _ in range(max(level) + 1)] if level else []
    for j, lv in enumerate(level):
        out[lv].append(j)
    return out


def plan_components(windows, pano_fhw):
    """Connected components of the step's overlap graph (union-find), each a list of window indices in reference
    order; components ordered by their first window.  Windows of different components never touch the same panorama
    element within the step, so components can be processed in any order or concurrently."""
    n = len(windows)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for j in range(n):
        for k in range(j):
            if windows_overlap(windows[k], windows[j], pano_fhw):
                ra, rb = find(k), find(j)
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
    comps = {}
    for j in range(n):
        comps.setdefault(find(j), []).append(j)
    return [comps[k] for k in sorted(comps)]


def plan_owners(windows, pano_fhw, world):
    """Owner rank per window when the step has at least `world` components (component c -> rank c % world, so every
    rank's tile count differs by at most one component), else None (`run_step` then shares every level out)."""
    if world <= 1:
        return None
    comps = plan_components(windows, pano_fhw)
    if len(comps) < world:
        return None
    owner = [0] * len(windows)
    for c, members in enumerate(comps):
        for j in members:
            owner[j] = c % world
    return owner


def rank_share(items, rank, world):
    """Strided share of a level for one rank (items keep their order)."""
    return items[rank::world]


def share_counts(n_items, world):
    return [len(range(r, n_items, world)) for r in range(world)]


_HOST_STAGED = False
_FORCE_LEVELS = False      # share_mode("levels"): always share every level out (diagnostics / A-B runs)
_PROFILE = None            # profile_begin(): per-rank account of one step (bench.py's per_rank diagnostics)


def profile_begin():
    """Start accounting the steps that follow on THIS rank: tiles / evaluation units it computes, number of exchanges and the time
    it spends in them (device-synchronised around every exchange, so the figure includes the wait for the slowest rank -- the
    diagnostic: a rank with little compute and a long exchange is waiting for others).  Diagnostic runs only: the
    synchronisations serialise what a normal step overlaps."""
    global _PROFILE
    _PROFILE = {"tiles_owned": 0, "units_owned": 0, "exchanges": 0, "exchange_s": 0.0, "exchange_bytes": 0}
    return _PROFILE


def profile_end():
    global _PROFILE
    out, _PROFILE = _PROFILE, None
    return out


def _sync():
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()


def _account(kind, n):
    if _PROFILE is not None:
        _PROFILE[kind] += int(n)


def _exchange(fn, *tensors):
    """Run one collective exchange; under profile_begin() bracket it with device synchronisations and a host clock."""
    if _PROFILE is None:
        return fn()
    import time
    _sync()
    t0 = time.perf_counter()
    out = fn()
    _sync()
    _PROFILE["exchange_s"] += time.perf_counter() - t0
    _PROFILE["exchanges"] += 1
    _PROFILE["exchange_bytes"] += sum(int(t.numel()) * t.element_size() for t in tensors if torch.is_tensor(t))
    return out


def share_mode(mode):
    """'auto' (default): whole components per rank when there are enough of them, else levels shared out by tile -- or by
    evaluation where a level has fewer tiles than ranks; 'levels': always share every level out by tile (A/B runs)."""
    global _FORCE_LEVELS
    _FORCE_LEVELS = mode == "levels"


def host_staged_collectives(on=True):
    """Rehearsal mode (a backend without device collectives, e.g. gloo with several ranks on one GPU): the level
    all-gather goes through host buffers.  RCCL runs never enable it."""
    global _HOST_STAGED
    _HOST_STAGED = bool(on)


def all_gather_tiles(x_prev_local, x0_local, counts, group=None):
    """All-gather of per-rank tile lists of different lengths.  x_prev_local / x0_local: [counts[rank], C, tf, th, tw].
    Returns a list over ranks of (x_prev_r, x0_r) with counts[r] tiles each (this rank's own entry is the input)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cmax = max(counts)
    tile_shape = tuple(x_prev_local.shape[1:])
    nt = 1 if x0_local is None else 2          # x0_local None: only the x_prev tiles travel (run_step need_x0=False)
    send = torch.zeros((nt, cmax) + tile_shape, dtype=x_prev_local.dtype, device=x_prev_local.device)
    if counts[rank]:
        send[0, :counts[rank]] = x_prev_local
        if nt == 2:
            send[1, :counts[rank]] = x0_local
    recv = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if _HOST_STAGED and send.is_cuda:
        recv_h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(recv_h, send.cpu(), group=group)
        recv.copy_(recv_h)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world,) + tuple(send.shape))
    return [(x_prev_local, x0_local) if r == rank else (recv[r, 0, :counts[r]], recv[r, 1, :counts[r]] if nt == 2 else None) for r in range(world)]


class EvalUnits:
    """The three stages of `process` taken apart, for sharing a level out by UNet evaluation instead of by tile
    (run_step's "units" mode).  branches = 2 with classifier-free guidance ([cond, uncond]), else 1.
      prepare(ids)          -> ctx: the level's tiles gathered and re-noised (every rank, all tiles: HBM-bound tile ops)
      eps(ctx, units)       -> eps [len(units), C, tf, th, tw] of the units (k, b) = (position in ids, branch); a unit's
                               values must not depend on which other units share the call (a batch equals its separate forwards)
      finish(ctx, eps_all)  -> (x_prev, x0) tiles [len(ids), ...] from eps_all [len(ids), branches, C, tf, th, tw]"""

    def __init__(self, branches, prepare, eps, finish):
        self.branches, self.prepare, self.eps, self.finish = branches, prepare, eps, finish


def unit_share(n_tiles, branches, rank, world):
    """Units (k, b) of a level in unit order u = k * branches + b; rank r takes u = r, r + world, ..."""
    units = [(k, b) for k in range(n_tiles) for b in range(branches)]
    return units[rank::world]


def exchange_units(local, n_units, group=None):
    """All-gather of one tensor per unit: local [n_local, ...] holds this rank's strided share (units rank, rank + world, ...);
    returns [n_units, ...] in unit order on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = share_counts(n_units, world)
    cmax = max(counts)
    shape = tuple(local.shape[1:])
    send = torch.zeros((cmax,) + shape, dtype=local.dtype, device=local.device)
    if counts[rank]:
        send[:counts[rank]] = local
    recv = torch.empty((world * cmax,) + shape, dtype=send.dtype, device=send.device)
    if _HOST_STAGED and send.is_cuda:
        recv_h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(recv_h, send.cpu(), group=group)
        recv.copy_(recv_h)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world, cmax) + shape)
    out = torch.empty((n_units,) + shape, dtype=send.dtype, device=send.device)
    for r in range(world):
        if counts[r]:
            out[r::world] = recv[r, :counts[r]]
    return out


def windows_cover(windows, pano_fhw):
    """True when the union of the windows (left, right, top, down, f_begin, f_end; wrapped on the ring) is the whole panorama."""
    import numpy as np
    F, H, W = pano_fhw
    seen = np.zeros((F, H, W), dtype=bool)
    for (l, r, t, d, f0, f1) in windows:
        seen[np.ix_(np.arange(f0, f1) % F, np.arange(t, d) % H, np.arange(l, r) % W)] = True
    return bool(seen.all())


def run_step(windows, pano_fhw, rank, world, process, scatter, empty_tiles, group=None, units=None, need_x0=True):
    """One DDIM step over `windows` (reference order), shared over `world` ranks -- the scheduling both the HIP pipelines
    (pipelines._denoise_windows) and the CPU rehearsal (tests/test_parallel_gloo.py) run.
      process(ids) -> (x_prev, x0) tiles [len(ids), ...] of the windows `ids` (pairwise disjoint, in the given order)
      scatter(ids, x_prev, x0)       writes tiles into this rank's panorama replica; `ids` are pairwise disjoint (a batched
                                     scatter has no order among its windows)
      empty_tiles()                  -> a [0, ...] tile tensor (dtype / device of the tiles)
      units                          EvalUnits or None: lets a level with fewer tiles than ranks be shared out by evaluation
      need_x0                        False: the other ranks' pred-x0 tiles are neither exchanged nor scattered (scatter gets x0 = None for
                                     them) -- the pred-x0 panorama is only read when a loop ends, so a loop whose LAST step's
                                     windows cover the panorama (windows_cover) exchanges it on that step only: half the bytes per step
    Returns the mode used: "single", "components", "levels" or "units" (at least one level shared out by evaluation)."""
    levels = plan_levels(windows, pano_fhw)
    if world <= 1:
        for level in levels:
            _account("tiles_owned", len(level))
            xp, x0 = process(level)
            scatter(level, xp, x0)
        return "single"
    owner = None if _FORCE_LEVELS else plan_owners(windows, pano_fhw, world)
    if owner is not None:
        mine_xp, mine_x0 = [], []
        for level in levels:
            ids = [j for j in level if owner[j] == rank]
            if ids:
                _account("tiles_owned", len(ids))
                xp, x0 = process(ids)
                scatter(ids, xp, x0)                     # own tiles: visible to this rank's later levels at once
                mine_xp.append(xp)
                mine_x0.append(x0)
        # what every rank sends: its tiles level by level (the order it produced them in)
        by_level = [[[j for j in level if owner[j] == r] for level in levels] for r in range(world)]
        counts = [sum(len(ids) for ids in by_level[r]) for r in range(world)]
        xp_l = torch.cat(mine_xp, 0) if mine_xp else empty_tiles()
        x0_l = torch.cat(mine_x0, 0) if mine_x0 else empty_tiles()
        if not need_x0:
            x0_l = None
        parts = _exchange(lambda: all_gather_tiles(xp_l, x0_l, counts, group), xp_l, x0_l)    # the step's ONE exchange
        # the other ranks' tiles go into this replica LEVEL BY LEVEL: one scatter call only ever holds pairwise-disjoint
        # windows (a component's later level overwrites part of its earlier one, and a batched scatter has no order)
        for li in range(len(levels)):
            ids, xs, x0s = [], [], []
            for r in range(world):
                if r == rank or not by_level[r][li]:
                    continue
                off = sum(len(b) for b in by_level[r][:li])
                n = len(by_level[r][li])
                ids += by_level[r][li]
                xs.append(parts[r][0][off:off + n])
                if need_x0:
                    x0s.append(parts[r][1][off:off + n])
            if ids:
                scatter(ids, torch.cat(xs, 0), torch.cat(x0s, 0) if need_x0 else None)
        return "components"
    mode = "levels"
    for level in levels:
        if units is not None and not _FORCE_LEVELS and len(level) < world and units.branches > 1:
            # fewer tiles than ranks: share the level out by (tile, branch) evaluation -- the cross-rank CFG split
            mode = "units"
            n, nb = len(level), units.branches
            ctx = units.prepare(level)
            mine = unit_share(n, nb, rank, world)
            _account("units_owned", len(mine))
            e_local = units.eps(ctx, mine)
            e_all = _exchange(lambda: exchange_units(e_local, n * nb, group), e_local)
            xp_all, x0_all = units.finish(ctx, e_all.view((n, nb) + tuple(e_all.shape[1:])))
            scatter(level, xp_all, x0_all)
            continue
        ids = rank_share(level, rank, world)
        _account("tiles_owned", len(ids))
        xp, x0 = process(ids) if ids else (empty_tiles(), empty_tiles())
        if not need_x0:
            x0 = None
        xp_all, x0_all = _exchange(lambda: exchange_level(xp, x0, len(level), group), xp, x0)
        scatter(level, xp_all, x0_all)
    return mode


def exchange_level(x_prev_local, x0_local, n_items, group=None, force=False):
    """All-gather the tiles of one level.  x_prev_local / x0_local: [n_local, C, tf, th, tw] of this rank's
    share (strided assignment).  Returns (x_prev_all, x0_all) [n_items, ...] in level order on every rank.
    force=True runs the collective even in a one-rank group (tests: the RCCL call pattern on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return x_prev_local, x0_local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = share_counts(n_items, world)
    cmax = max(counts)
    tile_shape = tuple(x_prev_local.shape[1:])
    nt = 1 if x0_local is None else 2          # x0_local None: x_prev tiles only (returns x0_all = None)
    send = torch.zeros((nt, cmax) + tile_shape, dtype=x_prev_local.dtype, device=x_prev_local.device)
    n_local = counts[rank]
    if n_local:
        send[0, :n_local] = x_prev_local
        if nt == 2:
            send[1, :n_local] = x0_local
    # output = concatenation along dim 0 (the form both RCCL and gloo accept), viewed back as [world, 2, cmax, ...]
    recv = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if _HOST_STAGED and send.is_cuda:
        recv_h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(recv_h, send.cpu(), group=group)
        recv.copy_(recv_h)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view((world,) + tuple(send.shape))
    x_prev_all = torch.empty((n_items,) + tile_shape, dtype=send.dtype, device=send.device)
    x0_all = torch.empty_like(x_prev_all) if nt == 2 else None
    for r in range(world):
        if counts[r]:
            x_prev_all[r::world] = recv[r, 0, :counts[r]]
            if nt == 2:
                x0_all[r::world] = recv[r, 1, :counts[r]]
    return x_prev_all, x0_all


class StreamPool:
    """n worker threads, each bound to its own HIP stream: `map(fn, items)` runs fn(item) for up to n items
    concurrently (item k on stream k), ordered after everything already queued on the caller's stream, and makes the
    caller's stream wait for all of them.  Tensors returned by fn are handed over to the caller's stream
    (`record_stream`), so the caching allocator does not recycle them while it still reads them."""

    def __init__(self, device, n):
        from concurrent.futures import ThreadPoolExecutor
        self.device, self.n = device, n
        self.streams = [torch.cuda.Stream(device) for _ in range(n)]
        self.pool = ThreadPoolExecutor(max_workers=n)

    def map(self, fn, items, inline=False, on_slot=None):
        """inline=True: no worker threads -- the caller's thread enqueues item k on stream k itself (right when every
        fn(item) only enqueues a few launches, e.g. hipGraph replays); on_slot(k) is called before fn(item)."""
        outs = []
        for g in range(0, len(items), self.n):
            group = items[g:g + self.n]
            main = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)

            def work(k, item):
                torch.cuda.set_device(self.device)
                s = self.streams[k]
                s.wait_event(ready)
                with torch.no_grad(), torch.cuda.stream(s):
                    if on_slot is not None:
                        on_slot(k)
                    out = fn(item)
                done = torch.cuda.Event()
                done.record(s)
                return out, done

            if inline:
                results = [work(k, it) for k, it in enumerate(group)]
            else:
                results = [f.result() for f in [self.pool.submit(work, k, it) for k, it in enumerate(group)]]
            for out, done in results:
                main.wait_event(done)
                for t in (out if isinstance(out, (tuple, list)) else (out,)):
                    if torch.is_tensor(t):
                        t.record_stream(main)
                outs.append(out)
        if on_slot is not None:
            on_slot(0)
        return outs
