"""Multi-GPU sharding of one DDIM step's tiles: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference is sequential (Gauss-Seidel over tiles).  Tiles whose ring footprints do not overlap are
independent, so a step is cut into LEVELS of pairwise-disjoint windows (`plan_levels`); each rank takes a
strided share of a level, and after the level every rank needs the others' updated tiles because the next level
(and the next step's shifted grid) straddles ownership.  That is the path's one real exchange: an ALL-GATHER of
the (x_prev, x0) tiles of the level -- identical in result to an all-reduce(sum) of zero-filled panorama-sized
accumulators with a 0/1 weight map (the "overlap accumulator" reading), at 1/W of the bytes.  Every rank then
scatters all tiles into its own replica of the panorama, so replicas stay bit-identical.
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


def rank_share(items, rank, world):
    """Strided share of a level for one rank (items keep their order)."""
    return items[rank::world]


def share_counts(n_items, world):
    return [len(range(r, n_items, world)) for r in range(world)]


_HOST_STAGED = False


def host_staged_collectives(on=True):
    """Rehearsal mode (a backend without device collectives, e.g. gloo with several ranks on one GPU): the level
    all-gather goes through host buffers.  RCCL runs never enable it."""
    global _HOST_STAGED
    _HOST_STAGED = bool(on)


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
    send = torch.zeros((2, cmax) + tile_shape, dtype=x_prev_local.dtype, device=x_prev_local.device)
    n_local = counts[rank]
    if n_local:
        send[0, :n_local] = x_prev_local
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
    x0_all = torch.empty_like(x_prev_all)
    for r in range(world):
        if counts[r]:
            x_prev_all[r::world] = recv[r, 0, :counts[r]]
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
