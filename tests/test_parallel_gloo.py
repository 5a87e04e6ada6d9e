"""N>1 path on CPU: world_size 2, gloo.  The tile compute is replaced by a deterministic CPU stand-in (the HIP
kernels cannot run here); what is exercised is the product's level planning, strided sharing, the tile
all-gather (parallel.exchange_level) and that every rank's panorama replica equals the single-process result."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), "golden")


def _update(tile, j):
    return tile * 0.5 + (j + 1), tile * 0.25 - (j + 1)


def _run_step(pano, pano_x0, wins, pano_fhw, rank, world, with_units=False, need_x0=True):
    """parallel.run_step -- the scheduling the HIP pipelines run -- with a CPU stand-in for the tile compute.
    with_units: also hand over the per-evaluation stages (parallel.EvalUnits), the stand-in's two "branches" being the two
    halves of _update -- what lets a level with fewer tiles than ranks be shared out by evaluation (the CFG split)."""
    from dynamicscaler_amd import parallel
    from oracle import ring as oring
    shape = (0, pano.shape[1], wins[0][5] - wins[0][4], wins[0][3] - wins[0][2], wins[0][1] - wins[0][0])

    def u_prepare(ids):
        return ids, torch.cat([oring.ring_gather(pano, *wins[j]) for j in ids])

    def u_eps(ctx, units):
        ids, tiles = ctx
        if not units:
            return torch.empty(shape)
        return torch.cat([_update(tiles[k:k + 1], ids[k])[b] for k, b in units])

    def u_finish(ctx, e_all):
        return e_all[:, 0].contiguous(), e_all[:, 1].contiguous()

    units = parallel.EvalUnits(2, u_prepare, u_eps, u_finish) if with_units else None

    def process(ids):
        xp, x0 = [], []
        for j in ids:
            l, r, t, d, fb, fe = wins[j]
            a, b = _update(oring.ring_gather(pano, l, r, t, d, fb, fe), j)
            xp.append(a)
            x0.append(b)
        return torch.cat(xp), torch.cat(x0)

    def scatter(ids, xp, x0):
        # the HIP scatter writes all windows of a call concurrently: they must be pairwise disjoint
        assert not any(parallel.windows_overlap(wins[a], wins[b], pano_fhw) for k, a in enumerate(ids) for b in ids[k + 1:])
        for n, j in reversed(list(enumerate(ids))):          # ... so any order within a call gives the same panorama
            l, r, t, d, fb, fe = wins[j]
            oring.ring_scatter(pano, xp[n:n + 1], l, r, t, d, fb, fe)
            if x0 is not None:                                # None: another rank's tiles on a step that does not exchange pred-x0
                oring.ring_scatter(pano_x0, x0[n:n + 1], l, r, t, d, fb, fe)

    return parallel.run_step(wins, pano_fhw, rank, world, process, scatter, lambda: torch.empty(shape), units=units, need_x0=need_x0)


def _worker(rank, world, port, steps, geom_name, out, with_units=False, x0_last_only=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = json.load(open(os.path.join(G, "loop_traces.json")))[geom_name]
    g = rec["geom"]
    fhw = (g["frames"], g["total_h"] // 8, g["total_w"] // 8)
    torch.manual_seed(0)
    pano = torch.randn((1, 4) + fhw)
    pano_x0 = torch.zeros_like(pano)
    modes = set()
    from dynamicscaler_amd import parallel
    prof = parallel.profile_begin()           # the per-rank account bench.py reports (per_rank)
    for k, step in enumerate(rec["trace"][:steps]):
        wins = [tuple(w) for w in step["windows"]]
        need_x0 = True
        if x0_last_only and parallel.windows_cover([tuple(w) for w in rec["trace"][steps - 1]["windows"]], fhw):
            need_x0 = k == steps - 1           # the pipelines' rule: pred-x0 tiles travel on the last step only, if that step covers the panorama
        modes.add(_run_step(pano, pano_x0, wins, fhw, rank, world, with_units, need_x0=need_x0))
    parallel.profile_end()
    out[rank] = (pano, pano_x0, sorted(modes), dict(prof), len(rec["trace"][0]["windows"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("geom_name,steps,world,mode,with_units", [
    ("cfg3_4096x512", 3, 2, "components", False),     # 8 columns of 2 dependent tiles: 4 columns per rank, one exchange per step
    ("cfg2_2048x512", 2, 3, "components", True),      # 4 columns over 3 ranks: uneven tile counts (4 / 2 / 2) in the all-gather
    ("cfg3_overlap_nw10", 2, 2, "levels", False),     # W overlap: one chain -> a strided share of every level
    ("cfg2_2048x512", 2, 5, "units", True),           # 4 columns < 5 ranks: levels of 4 tiles shared out as 8 evaluations (2/2/2/1/1)
    ("cfg3_overlap_nw10", 1, 3, "units", True),       # one chain: levels of 1-2 tiles over 3 ranks, by evaluation
    ("cfg3_4096x512", 2, 8, "components", True),      # the bench's panorama on 8 ranks: one column (2 dependent tiles) per rank
    ("cfg2_2048x512", 2, 8, "units", True),           # config 2 on 8 ranks: 4 tiles per level = 8 evaluations, one per rank
])
def test_ranks_equal_single_process(geom_name, steps, world, mode, with_units):
    mgr = mp.Manager()
    single = mgr.dict()
    _worker(0, 1, 0, steps, geom_name, single)
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), steps, geom_name, out, with_units), nprocs=world, join=True)
    for r in range(world):
        assert torch.equal(out[r][0], single[0][0]) and torch.equal(out[r][1], single[0][1])
        assert out[r][2] == [mode]
    # the per-rank account: every tile (or evaluation unit) of every step is owned by exactly one rank; whole-column ownership has ONE
    # exchange per step, a single process none
    profs, ntiles = [out[r][3] for r in range(world)], out[0][4]
    assert single[0][3]["tiles_owned"] == steps * ntiles and single[0][3]["exchanges"] == 0
    if mode == "components":
        assert sum(p["tiles_owned"] for p in profs) == steps * ntiles and all(p["exchanges"] == steps for p in profs)
    elif mode == "levels":
        assert sum(p["tiles_owned"] for p in profs) == steps * ntiles and all(p["exchanges"] >= steps for p in profs)
    else:
        assert sum(2 * p["tiles_owned"] + p["units_owned"] for p in profs) == 2 * steps * ntiles
    assert all(p["exchange_s"] > 0 and p["exchange_bytes"] >= 0 for p in profs)


@pytest.mark.parametrize("geom_name,steps,world,with_units", [("cfg3_4096x512", 3, 2, False), ("cfg3_overlap_nw10", 3, 2, False), ("cfg2_2048x512", 3, 5, True)])
def test_pred_x0_exchanged_on_the_last_step_only(geom_name, steps, world, with_units):
    """need_x0=False on every step but the last (parallel.run_step; SURVEY 8-e: P_denoised is only needed when the loop ends): after the last
    step every rank's pred-x0 AND latent replicas equal the single-process panoramas, with fewer bytes exchanged than with x0 on every step."""
    mgr = mp.Manager()
    single = mgr.dict()
    _worker(0, 1, 0, steps, geom_name, single)
    every, last = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), steps, geom_name, every, with_units, False), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), steps, geom_name, last, with_units, True), nprocs=world, join=True)
    for r in range(world):
        assert torch.equal(last[r][0], single[0][0]) and torch.equal(last[r][1], single[0][1]), r
    b_every, b_last = sum(every[r][3]["exchange_bytes"] for r in range(world)), sum(last[r][3]["exchange_bytes"] for r in range(world))
    from dynamicscaler_amd import parallel
    rec = json.load(open(os.path.join(G, "loop_traces.json")))[geom_name]
    g = rec["geom"]
    covered = parallel.windows_cover([tuple(w) for w in rec["trace"][steps - 1]["windows"]], (g["frames"], g["total_h"] // 8, g["total_w"] // 8))
    assert covered == (geom_name != "cfg3_overlap_nw10")      # that grid leaves rows of the panorama untouched: pred-x0 travels on every step there
    if covered and "units" not in every[0][2]:   # (a level shared out by evaluation exchanges eps tensors, not tiles: nothing to save there)
        assert b_last < b_every, (b_last, b_every)
    elif not covered:
        assert b_last == b_every


def test_windows_cover():
    from dynamicscaler_amd import parallel
    assert parallel.windows_cover([(0, 8, 0, 4, 0, 2), (8, 16, 0, 4, 0, 2)], (2, 4, 16))
    assert parallel.windows_cover([(12, 20, 2, 6, 0, 2), (4, 12, 2, 6, 0, 2)], (2, 4, 16))          # wrapped in W and H
    assert not parallel.windows_cover([(0, 8, 0, 4, 0, 2), (8, 15, 0, 4, 0, 2)], (2, 4, 16))


def test_plan_components_and_owners():
    """Columns of the BASELINE geometries are the connected components of a step's overlap graph; with fewer components
    than ranks plan_owners declines (levels are shared out instead)."""
    from dynamicscaler_amd import parallel
    traces = json.load(open(os.path.join(G, "loop_traces.json")))
    for name, ncomp, per in (("cfg2_2048x512", 4, 2), ("cfg3_4096x512", 8, 2), ("cfg5_8192x1024x24", 16, 4), ("cfg3_overlap_nw10", 1, 20)):
        g = traces[name]["geom"]
        fhw = (g["frames"], g["total_h"] // 8, g["total_w"] // 8)
        for step in traces[name]["trace"][:3]:
            wins = [tuple(w) for w in step["windows"]]
            comps = parallel.plan_components(wins, fhw)
            assert len(comps) == ncomp and all(len(c) == per for c in comps), (name, [len(c) for c in comps])
            assert sorted(j for c in comps for j in c) == list(range(len(wins)))
            for a in range(len(comps)):            # no window of one component touches a window of another
                for b in range(a + 1, len(comps)):
                    assert not any(parallel.windows_overlap(wins[i], wins[j], fhw) for i in comps[a] for j in comps[b])
            owner = parallel.plan_owners(wins, fhw, 8)
            if ncomp >= 8:
                assert all(len({owner[j] for j in c}) == 1 for c in comps) and set(owner) == set(range(8))
            else:
                assert owner is None


def test_sequential_reference_order_equals_levels():
    """plan_levels + level-by-level processing == the reference's strictly sequential tile order."""
    from oracle import ring as oring
    rec = json.load(open(os.path.join(G, "loop_traces.json")))["cfg3_overlap_nw10"]
    fhw = (16, 64, 512)
    torch.manual_seed(1)
    p_seq = torch.randn((1, 4) + fhw)
    p_lvl, x_seq, x_lvl = p_seq.clone(), torch.zeros_like(p_seq), torch.zeros_like(p_seq)
    for step in rec["trace"][:2]:
        wins = [tuple(w) for w in step["windows"]]
        for j, (l, r, t, d, fb, fe) in enumerate(wins):
            a, b = _update(oring.ring_gather(p_seq, l, r, t, d, fb, fe), j)
            oring.ring_scatter(p_seq, a, l, r, t, d, fb, fe)
            oring.ring_scatter(x_seq, b, l, r, t, d, fb, fe)
        _run_step(p_lvl, x_lvl, wins, fhw, 0, 1)
    assert torch.equal(p_seq, p_lvl) and torch.equal(x_seq, x_lvl)
