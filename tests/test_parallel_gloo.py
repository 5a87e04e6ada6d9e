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


def _run_step(pano, pano_x0, wins, pano_fhw, rank, world):
    from dynamicscaler_amd import parallel
    from oracle import ring as oring
    for level in parallel.plan_levels(wins, pano_fhw):
        mine = parallel.rank_share(level, rank, world)
        xp, x0 = [], []
        for j in mine:
            l, r, t, d, fb, fe = wins[j]
            tile = oring.ring_gather(pano, l, r, t, d, fb, fe)
            a, b = _update(tile, j)
            xp.append(a)
            x0.append(b)
        shape = (0, pano.shape[1], wins[0][5] - wins[0][4], wins[0][3] - wins[0][2], wins[0][1] - wins[0][0])
        xp = torch.cat(xp) if xp else torch.empty(shape)
        x0 = torch.cat(x0) if x0 else torch.empty(shape)
        xp_all, x0_all = parallel.exchange_level(xp, x0, len(level))
        order = level if world > 1 else mine
        for n, j in enumerate(order):
            l, r, t, d, fb, fe = wins[j]
            oring.ring_scatter(pano, xp_all[n:n + 1], l, r, t, d, fb, fe)
            oring.ring_scatter(pano_x0, x0_all[n:n + 1], l, r, t, d, fb, fe)


def _worker(rank, world, port, steps, geom_name, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rec = json.load(open(os.path.join(G, "loop_traces.json")))[geom_name]
    g = rec["geom"]
    fhw = (g["frames"], g["total_h"] // 8, g["total_w"] // 8)
    torch.manual_seed(0)
    pano = torch.randn((1, 4) + fhw)
    pano_x0 = torch.zeros_like(pano)
    for step in rec["trace"][:steps]:
        _run_step(pano, pano_x0, [tuple(w) for w in step["windows"]], fhw, rank, world)
    out[rank] = (pano, pano_x0)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("geom_name,steps", [("cfg3_4096x512", 3), ("cfg3_overlap_nw10", 2)])
def test_two_ranks_equal_single_process(geom_name, steps):
    mgr = mp.Manager()
    single = mgr.dict()
    _worker(0, 1, 0, steps, geom_name, single)
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), steps, geom_name, out), nprocs=2, join=True)
    for r in (0, 1):
        assert torch.equal(out[r][0], single[0][0]) and torch.equal(out[r][1], single[0][1])


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
