#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (GPU box): data-flow probe for the concurrent-graph-replay hazard of round 1
(profiles/r1_notes.md "Withdrawn ..."): two hipGraphs of the UNet evaluation replaying concurrently on two streams once
gave panoramas that differed in the last fp16 bit from process to process.

    python tests/hazard_probe.py unet   [rounds]   two UNet graphs, every block's output kept (block taps) and compared
    python tests/hazard_probe.py poison            every kernel preceded by the LDS / register poison launch (launch hook)
    python tests/hazard_probe.py pipe   [runs]     the toy ring pipeline (2 streams x graphs) repeated in this process

DS_HIP_LIBRARY=dynamicscaler_amd/libdynscaler_hip_barebarrier.so selects the diagnostic build with round 1's bare K-step
barrier (python -m dynamicscaler_amd.build --variant barebarrier): the root cause found with this probe
(profiles/r2_notes.md).
Writes one JSON line per finding to stdout.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
G = os.path.join(REPO, "tests", "golden")

from dynamicscaler_amd import ops, _lib  # noqa: E402
from dynamicscaler_amd.synth import synth_state_dict, synth_normal  # noqa: E402
from dynamicscaler_amd.unet import UNetModel  # noqa: E402
from dynamicscaler_amd.unet_spec import param_shapes  # noqa: E402

d = torch.device("cuda:0")
def toy_unet():
    z = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(z["params_json"]).decode())
    m = UNetModel(**params)
    m.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
    m = m.to(d).eval()
    m.prepare(d)
    return m, params


class Recorder:
    """Block taps of the C launch program (ds_unet_set_hooks through UNetModel._tap): while `on`, a copy of every block's output rows
    is appended to `log` (under graph capture the copies become nodes of the graph: every replay refreshes them)."""

    def __init__(self, model):
        self.log, self.on, self.model = [], False, model
        model._tap = self._tap

    def _tap(self, name, rows, geo):
        if self.on:
            self.log.append((f"{len(self.log)}:{name} {tuple(rows.shape)}", rows))

    def restore(self):
        self.model._tap = None


def probe_unet(rounds):
    m, params = toy_unet()
    rec = Recorder(m)
    streams = [torch.cuda.Stream(d), torch.cuda.Stream(d)]
    n = 2   # tiles per batch: x = [tiles | tiles], cfg_pairs = n  (the multi-rank rehearsal's shape)
    findings = []
    for keep in (True, False):
        graphs = []
        for slot in range(2):
            tiles = synth_normal((n, 4, 4, 8, 16), 100 + slot).to(d, torch.float16)
            x = torch.cat([tiles, tiles], 0)
            ctx = torch.cat([synth_normal((1, 77, 64), 61)] * n + [synth_normal((1, 77, 64), 62)] * n, 0).to(d)
            ts = torch.full((2 * n,), 500 + slot, device=d, dtype=torch.long)
            with torch.cuda.stream(streams[slot]):
                m(x, ts, context=ctx, fps=8, cfg_pairs=n)                # warm
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            rec.log, rec.on = [], keep
            with torch.cuda.graph(g):
                out = m(x, ts, context=ctx, fps=8, cfg_pairs=n)
            rec.on = False
            kept = list(rec.log) + [("final:eps", out)]
            graphs.append((g, kept, (x, ctx, ts)))     # the static inputs stay alive with the graph
        # serial references
        refs = []
        for slot, (g, kept, _in) in enumerate(graphs):
            with torch.cuda.stream(streams[slot]):
                g.replay()
            torch.cuda.synchronize()
            refs.append([t.clone() for _, t in kept])
        # serial repeatability first
        for slot, (g, kept, _in) in enumerate(graphs):
            with torch.cuda.stream(streams[slot]):
                g.replay()
            torch.cuda.synchronize()
            bad = [nm for (nm, t), r in zip(kept, refs[slot]) if not torch.equal(t, r)]
            if bad:
                findings.append({"mode": "serial", "keep": keep, "slot": slot, "first": bad[0], "count": len(bad)})
        nbad = 0
        gen = torch.Generator().manual_seed(5)
        for r in range(rounds):
            reps = 1 + int(torch.randint(0, 3, (1,), generator=gen))
            delay = int(torch.randint(0, 200000, (1,), generator=gen))
            for slot in ((0, 1) if r % 2 == 0 else (1, 0)):
                with torch.cuda.stream(streams[slot]):
                    if slot == r % 2:
                        torch.cuda._sleep(delay)
                    for _ in range(reps):
                        graphs[slot][0].replay()
            torch.cuda.synchronize()
            for slot, (g, kept, _in) in enumerate(graphs):
                bad = [(i, nm) for i, ((nm, t), rf) in enumerate(zip(kept, refs[slot])) if not torch.equal(t, rf)]
                if bad:
                    nbad += 1
                    i, nm = bad[0]
                    t, rf = kept[i][1], refs[slot][i]
                    diff = (t.float() - rf.float()).abs()
                    findings.append({"mode": "concurrent", "keep": keep, "round": r, "slot": slot, "first": nm,
                                     "count": len(bad), "of": len(kept), "max_abs_diff": float(diff.max()),
                                     "frac_elems": float((diff > 0).float().mean())})
        print(json.dumps({"probe": "unet", "lib": os.path.basename(_lib.LIB_PATH), "keep_intermediates": keep,
                          "rounds": rounds, "rounds_with_mismatch": nbad, "blocks_per_graph": len(graphs[0][1])}), flush=True)
    for f in findings[:12]:
        print(json.dumps(f), flush=True)
    rec.restore()
    return len(findings)


def probe_poison():
    """Every launch of the UNet program preceded by ds_dbg_poison_cu_state: results must not change."""
    diag = _lib.load_diag()          # the poison launch lives outside the product library (csrc/diag.hip)

    def poison_in_front(phase, kernel, flops, info):
        if phase == 0:
            assert diag.ds_dbg_poison_cu_state(torch.cuda.current_stream().cuda_stream) == 0

    m, params = toy_unet()
    n = 2
    tiles = synth_normal((n, 4, 4, 8, 16), 100).to(d, torch.float16)
    x = torch.cat([tiles, tiles], 0)
    ctx = torch.cat([synth_normal((1, 77, 64), 61)] * n + [synth_normal((1, 77, 64), 62)] * n, 0).to(d)
    ts = torch.full((2 * n,), 500, device=d, dtype=torch.long)
    rec = Recorder(m)
    rec.on, rec.log = True, []
    m(x, ts, context=ctx, fps=8, cfg_pairs=n)
    torch.cuda.synchronize()
    clean = [(nm, t.clone()) for nm, t in rec.log]
    m.launch_hook = poison_in_front
    rec.log = []
    m(x, ts, context=ctx, fps=8, cfg_pairs=n)
    torch.cuda.synchronize()
    m.launch_hook = None
    rec.on = False
    bad = [(nm, float((t.float() - c.float()).abs().max()), bool(torch.isnan(t.float()).any()))
           for (nm, t), (_, c) in zip(rec.log, clean) if not torch.equal(t, c)]
    print(json.dumps({"probe": "poison", "lib": os.path.basename(_lib.LIB_PATH), "blocks": len(clean),
                      "blocks_changed_by_poison": len(bad), "first": bad[:5]}), flush=True)
    rec.restore()
    return len(bad)


def probe_pipe(runs):
    """The guard test's workload (toy ring pipeline, grid4x2, two streams x graph replays) several times in ONE process,
    each with fresh pipeline objects: hashes must agree with each other and with a serial eager run."""
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    cond, uncond = torch.from_numpy(z["cond"]), torch.from_numpy(z["uncond"])
    hashes = []
    for k in range(runs + 1):
        ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: uncond if p[0] == "" else cond)
        ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
        ld.temporal_length = 4
        ld = ld.to(d).eval()
        pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="reference"),
                                           {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
        if k == 0:
            pipe.use_graph, pipe.num_streams, pipe.max_tile_batch = False, 1, 2      # serial eager reference
        else:
            pipe.use_graph, pipe.num_streams, pipe.max_tile_batch = True, 2, 2
        torch.manual_seed(2333333)
        _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                       **meta["geoms"]["grid4x2"])
        torch.cuda.synchronize()
        hashes.append(hashlib.sha256(den.float().cpu().numpy().tobytes() +
                                     pipe.final_latent.float().cpu().numpy().tobytes()).hexdigest()[:12])
    print(json.dumps({"probe": "pipe", "lib": os.path.basename(_lib.LIB_PATH), "serial_eager": hashes[0],
                      "graph_2streams": hashes[1:], "distinct": len(set(hashes))}), flush=True)
    return len(set(hashes)) - 1


if __name__ == "__main__":
    what = sys.argv[1]
    arg = int(sys.argv[2]) if len(sys.argv) > 2 else None
    if what == "unet":
        probe_unet(arg or 40)
    elif what == "poison":
        probe_poison()
    elif what == "pipe":
        probe_pipe(arg or 4)
