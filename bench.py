#!/usr/bin/env python3
"""bench.py -- denoising-steps/sec of the tiled panoramic denoising loop on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself: the parent -- which never
touches the GPU -- runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py <same arguments>` as a CHILD process, relays rank 0's JSON line and exits with the child's return code.

Workload (BASELINE.json metric: 4096x512x16f): t2v overlapped-ring panorama, P = [1,4,16,64,512], 8x2 windows of
512x320x16f (16 tiles / DDIM step, shifted every step), CFG 7.5 (2 UNet evaluations per tile), 50-step DDIM schedule,
VideoCrafter2 t2v UNet (1.41 B parameters, synthetic fp16-representable weights), fp16 matrix-core operands and
activations with fp32 accumulation, fp32 panorama latent (4 MB; `--latents f16` for fp16 storage).  A "step" = one DDIM step over ALL tiles: ring gather -> re-noise/mix -> 32 UNet evaluations
-> CFG+DDIM -> scatter (+ the per-level tile all-gather when N > 1).  Inputs are resident in HBM before the timed
region; nothing is skipped.  N > 1 shards the tiles of the SAME panorama over the ranks (strong scaling).

Defaults: the tile batches of a dependency level run on two HIP streams (4 + 4 tiles at N = 1) as hipGraph replays of
the batched UNet evaluation (`--streams 1 --graph 0` = plain eager loop, same results bit for bit).

Operand policy (round 6): at set-up the pipeline measures, on this UNet, how far the guided eps of its own mode is from the wide operand
mode's fp32-level result and predicts each step's error from it; the schedule's first steps may then run on another rung (the strict
residual mode, +12 %: on these synthetic weights the first TWO of the 50).  Those leading steps are executed in their own modes before
the warm-up; the W warm-up + K timed steps are own-mode steps (48 of the 50); `sec_per_50_step_panorama` is one complete loop with
every step in the mode the policy gives it; `config.strict_step_ms / wide_step_ms / own_mode_step_eager_ms` time one step per rung.

stdout carries exactly ONE line, the JSON record (everything else goes to stderr), with the driver's fields plus
  "roofline":     the dominant kernel family (implicit-GEMM MFMA kernel): algorithmic FLOPs of every launch of one
                  step / sum of their HIP-event durations, against the dense fp16 MFMA peak (2.5 PFLOP/s)
  "cpu_baseline": the CPU oracle (oracle/, torch fp32) timed on this host on a bounded sample (one UNet evaluation
                  of one tile = 1/32 of a step + the tile ops), extrapolated to steps/s.
"""
import argparse
import json
import os
import sys
import time

import torch
import yaml

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_PEAK_F16 = 2500e12      # dense fp16 MFMA peak, MI355X_MICROARCH.md
# BASELINE.json configs 2-5 (config 1 is the CPU plumbing case: a parity test, tests/test_gpu_unet.py).  f_unet = FLOPs
# (2*MAC) of one UNet evaluation at the config's tile; geometry as in SURVEY.md 8-d / BASELINE.md 3.
CONFIGS = {
    "cfg2": dict(model="t2v", f_unet=12580.6e9, ctx_len=77, label="t2v_sphere_panorama 2048x512x16f, 4x2 shifted ring windows (8 tiles/step)",
                 size="2048x512x16f",
                 geom=dict(height=320, width=512, frames=16, total_w=2048, total_h=512, num_windows_w=4, num_windows_h=2,
                           num_windows_f=1, loop_step=8, num_inference_steps=50)),
    "cfg3": dict(model="t2v", f_unet=12580.6e9, ctx_len=77, label="t2v_sphere_panorama 4096x512x16f, 8x2 shifted ring windows (16 tiles/step)",
                 size="4096x512x16f",
                 geom=dict(height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=8, num_windows_h=2,
                           num_windows_f=1, loop_step=8, num_inference_steps=50)),
    "cfg4": dict(model="i2v", f_unet=12601.1e9, ctx_len=93, label="i2v_sphere_panorama 4096x512x16f from a synthetic panorama image, 8x2 shifted "
                 "ring windows (16 tiles/step), 77 text + 16 image tokens per window", size="4096x512x16f",
                 geom=dict(height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=8, num_windows_h=2,
                           num_windows_f=1, loop_step=8, num_inference_steps=50)),
    # diagnostic, not a BASELINE configuration: TWO columns of cfg3 (2 x 2 tiles per step); with --tile-batch 1 --streams 1 a step
    # is twice what a rank of an 8-GPU cfg3 run executes, exchange aside (profiles/r2_notes.md section 5)
    "col2": dict(model="t2v", f_unet=12580.6e9, ctx_len=77, label="diagnostic: two columns of cfg3 (1024x512x16f, 2x2 ring windows, 4 tiles/step)",
                 size="1024x512x16f",
                 geom=dict(height=320, width=512, frames=16, total_w=1024, total_h=512, num_windows_w=2, num_windows_h=2,
                           num_windows_f=1, loop_step=8, num_inference_steps=50)),
    "cfg5": dict(model="t2v", f_unet=18884.0e9, ctx_len=77, label="t2v_sphere_panorama 8192x1024x24f, 16x4 shifted ring windows (64 tiles/step), "
                 "UNet at T=24", size="8192x1024x24f",
                 geom=dict(height=320, width=512, frames=24, total_w=8192, total_h=1024, num_windows_w=16, num_windows_h=4,
                           num_windows_f=1, loop_step=8, num_inference_steps=50)),
}


def self_launch(argv, n, timeout_s):
    """Start the n ranks as a child `torch.distributed.run` and relay rank 0's line.  Called before anything in this
    process has touched the GPU (no torch.cuda call above): the parent only waits -- at most `timeout_s`: ranks that stall
    (a collective some rank never enters) are killed as a process group and the parent exits 124 instead of hanging."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)   # fresh children, own process group
    timed_out = []

    def kill_group():
        timed_out.append(True)
        sys.stderr.write(f"bench.py: the {n}-rank child has not finished within {timeout_s} s (--rank-timeout): killing its process group\n")
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass

    timer = threading.Timer(timeout_s, kill_group)
    timer.daemon = True
    timer.start()
    line = None
    for out in p.stdout:
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = p.wait()
    timer.cancel()
    if timed_out:
        raise SystemExit(124)
    if line is not None:
        print(line, file=sys.__stdout__, flush=True)
    if rc != 0 or line is None:
        sys.stderr.write(f"bench.py: the {n}-rank child exited with code {rc}" + ("" if line else " and printed no result line") + "\n")
        raise SystemExit(rc if rc != 0 else 1)
    raise SystemExit(0)


def usable_cpus():
    """Threads this process may really use: affinity mask, capped by the cgroup CPU quota (a 256-thread pool on a
    throttled container is far slower than a right-sized one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def main():
    # stdout carries ONE line, the JSON record: whatever the libraries underneath print goes to stderr
    real_stdout = sys.__stdout__
    sys.stdout = sys.stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3",
                    help="BASELINE.json configuration (default cfg3 = the 4096x512x16f panorama the metric is quoted on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-panorama", type=int, default=int(os.environ.get("DS_FULL_PANORAMA", "-1")),
                    help="also run one complete 50-step panorama and report its measured wall time (the metric's second "
                         "figure); default: yes for cfg2-cfg4, no for cfg5 and no under rocprofv3")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tile-batch", type=int, default=int(os.environ.get("DS_TILE_BATCH", "8")))
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DS_STREAMS", "2")))
    ap.add_argument("--graph", type=int, default=int(os.environ.get("DS_GRAPH", "1")), help="hipGraph replay of the UNet evaluation")
    ap.add_argument("--latents", choices=["f32", "f16"], default=os.environ.get("DS_LATENTS", "f32"),
                    help="storage type of the panorama latent / tiles (fp32 like the reference: the default; the matrix-core "
                         "operands are fp16 either way)")
    ap.add_argument("--residual", choices=["default", "f16", "f32outer", "f32"], default="default",
                    help="residual-stream storage of the UNet: default = the library's (f32outer: fp32 between the blocks, the cheapest "
                         "mode inside 1e-3 on every asserted bound); f16 = the fast mode; f32 = strict")
    ap.add_argument("--other-mode", type=int, default=int(os.environ.get("DS_BENCH_OTHER_MODE", "0")),
                    help="also time the same steps in the other residual mode (fast <-> default) and report both ms/step (off by "
                         "default: it repacks the weights twice and runs the timed steps a second time)")
    ap.add_argument("--other-configs", type=int, default=int(os.environ.get("DS_BENCH_OTHER_CONFIGS", "-1")),
                    help="after the headline measurement also time the other single-GPU-runnable BASELINE configurations (cfg2, cfg4, "
                         "cfg5: 1 warm-up + 2 timed steps each, same bracketing) and report them under other_configs; default: yes "
                         "for the plain `python bench.py` (cfg3, one GPU, not under a profiler)")
    ap.add_argument("--wide-step", type=int, default=int(os.environ.get("DS_BENCH_WIDE_STEP", "-1")),
                    help="also time ONE step of the panorama in the wide operand mode (config.wide_step_ms: what a step costs where "
                         "the operand policy selects it; none of the 50-step schedule's steps at CFG 7.5); default: yes at one GPU")
    ap.add_argument("--rank-timeout", type=float, default=float(os.environ.get("DS_BENCH_RANK_TIMEOUT", "2400")),
                    help="seconds after which a rank that has not finished aborts the job with exit code 124 (and says in which phase)")
    ap.add_argument("--share-cfg-prefix", type=int, default=int(os.environ.get("DS_SHARE_CFG", "1")),
                    help="evaluate the context-free UNet prefix once per [cond | uncond] pair (bit-identical result)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(sys.argv[1:], args.gpus, args.rank_timeout + 60)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    cfg = CONFIGS[args.config]
    GEOM, F_UNET = cfg["geom"], cfg["f_unet"]
    # a rank that stalls (e.g. in a collective another rank never enters) must not hang the job: after --rank-timeout seconds it
    # says where it is and leaves with 124 (under torch.distributed.run the agent then stops the other ranks)
    import threading
    phase = ["init"]

    def _stalled():
        sys.stderr.write(f"bench.py: rank {rank} stalled in phase '{phase[0]}' for more than {args.rank_timeout:.0f} s: aborting\n")
        sys.stderr.flush()
        os._exit(124)

    watchdog = threading.Timer(args.rank_timeout, _stalled)
    watchdog.daemon = True
    watchdog.start()
    if os.environ.get("DS_BENCH_FAULT") in (f"stall:{rank}", "stall:*"):      # fault injection for the tests of the stall path: this rank never arrives
        phase[0] = "injected stall (DS_BENCH_FAULT)"
        time.sleep(10 ** 6)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one rank per GPU over RCCL.  DS_BENCH_DEVICE / DS_DIST_BACKEND exist for the single-GPU rehearsal of the N > 1 path
    # (tests/test_gpu_multirank.py: two ranks share cuda:0 and exchange through gloo); the driver never sets them.
    dev_index = int(os.environ.get("DS_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("DS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
            from dynamicscaler_amd import parallel
            parallel.host_staged_collectives(True)

    from dynamicscaler_amd import ops
    from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.unet_spec import param_shapes
    from dynamicscaler_amd.synth import synth_state_dict, synth_normal

    MODES = {"f16": (torch.float16, "full"), "f32outer": (torch.float32, "outer"), "f32": (torch.float32, "full")}
    hosts = {}                 # model kind -> (ld, params, sd | None): built once, shared by the configurations that use it

    def host_for(model, keep_sd=False):
        if model in hosts:
            return hosts[model]
        yaml_name = {"t2v": "t2v_512_v2_unet.yaml", "i2v": "i2v_512_v1_unet.yaml"}[model]
        params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", yaml_name)))
        sd = synth_state_dict(param_shapes(params), seed=0)
        ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"]))
        u = ld.model.diffusion_model
        u.load_state_dict(sd, strict=True)
        if args.residual != "default":
            u.residual_dtype, u.residual_scope = MODES[args.residual]
        u.prepare(dev)                     # fp16 repack straight to HBM
        if model == "i2v":
            # config 4: input/pano_surfing_1.png is absent from the reference tree, so the panorama image is
            # synthetic; every window takes the 16 image tokens of the crop under it through the REAL conditioning path
            # (LatentVisualDiffusion.get_image_embeds, ddpm3d.py:689-693): the HIP OpenCLIP ViT-H/14 image tower
            # (encoders.FrozenOpenCLIPImageEmbedderV2: preprocess + 32 blocks, all 257 tokens) and the Resampler, synthetic
            # weights, cached per crop position by the pipeline like the product does.
            from dynamicscaler_amd.encoders import FrozenOpenCLIPImageEmbedderV2, Resampler
            from dynamicscaler_amd.encoder_spec import CLIP_VIT_H_14, RESAMPLER_I2V, clip_vision_param_shapes, resampler_param_shapes
            from dynamicscaler_amd.synth import synth_encoder_state_dict
            ld.embedder = FrozenOpenCLIPImageEmbedderV2()
            ld.embedder.load_state_dict(synth_encoder_state_dict(clip_vision_param_shapes(CLIP_VIT_H_14["vision"]), 71))
            ld.image_proj_model = Resampler(**RESAMPLER_I2V)
            ld.image_proj_model.load_state_dict(synth_encoder_state_dict(resampler_param_shapes(**RESAMPLER_I2V), 72))
            ld.to(dev)
        hosts[model] = (ld, params, sd if keep_sd else None)
        return hosts[model]

    lat_dt = {"f32": torch.float32, "f16": torch.float16}[args.latents]

    def make_pipe(name):
        """The pipeline of BASELINE configuration `name` on this rank, and begin() -> a fresh ring state on the synthetic panorama."""
        c = CONFIGS[name]
        g = c["geom"]
        ld_, params_, _ = host_for(c["model"])
        sched_ = lvdm_DDIM_Scheduler(ld_, rng_mode="device")   # Philox noise in-kernel: no host RNG in the timed loop
        shape = (1, 4, g["frames"] * g["num_windows_f"], g["total_h"] // 8, g["total_w"] // 8)
        init_ = synth_normal(shape, 2333333).to(dev)
        extra_ = {}
        if c["model"] == "i2v":
            from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano as Pipe
            extra_ = dict(pano_image_tensor=synth_normal((3, g["total_h"], g["total_w"]), 77).clamp(-1, 1),
                          overlap_ratio_list_f=[0.0] * g["num_inference_steps"])
        else:
            from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano as Pipe
        pp = Pipe(ld_, sched_, {"params": {"unet_config": {"params": params_}}})
        pp.to(dev, lat_dt)
        pp.max_tile_batch = args.tile_batch
        pp.num_streams = args.streams
        pp.use_graph = bool(args.graph)
        pp.share_cfg_prefix = bool(args.share_cfg_prefix)

        def begin():
            return pp.ring_begin(prompt="a synthetic prompt", fps=8, guidance_scale=7.5, init_panorama_latent=init_, **g, **extra_)
        return pp, begin, shape

    phase[0] = "setup (weights, repack)"
    t0 = time.time()
    ld, params, sd = host_for(cfg["model"], keep_sd=(rank == 0 and not args.no_cpu_baseline and world == 1))
    unet = ld.model.diffusion_model

    def mode_name():
        return "f16" if unet.residual_dtype == torch.float16 else ("f32outer" if unet.residual_scope == "outer" else "f32")

    setup_s = time.time() - t0
    pipe, begin_state, pano_shape = make_pipe(args.config)
    sched = pipe.scheduler
    st = begin_state()
    wide_steps_of_schedule = pipe.wide_steps_of(GEOM["num_inference_steps"], 7.5)
    strict_steps_of_schedule = pipe.strict_steps_of(GEOM["num_inference_steps"], 7.5)
    # The operand policy (calibrated on this UNet by ring_begin) may put the schedule's FIRST steps on another rung (strict: +12 %, wide:
    # 3.7x).  The timed region measures the own-mode steps -- the other 48-50 of the schedule -- and starts behind them; the complete
    # 50-step panorama below is timed with every step in the mode the policy gives it, and one step per rung is timed separately.
    NSTEPS_ = GEOM["num_inference_steps"]
    lead_steps = 0
    while lead_steps < NSTEPS_ - 1 and pipe.precision_for(NSTEPS_ - 1 - lead_steps, 7.5) is not None:
        lead_steps += 1

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # who took part: one record per rank, gathered through the process group the data path uses (RCCL when --gpus > 1)
    ranks_seen = [{"rank": 0, "device": dev_index, "name": torch.cuda.get_device_name(dev), "backend": "none (single process)"}]
    if world > 1:
        import torch.distributed as dist
        try:
            ver = ".".join(map(str, torch.cuda.nccl.version()))
        except Exception:
            ver = "?"
        props = torch.cuda.get_device_properties(dev)
        pci = getattr(props, "pci_bus_id", None)
        if pci is not None:
            try:
                pci = f"{int(getattr(props, 'pci_domain_id', 0)):04x}:{int(pci):02x}:{int(getattr(props, 'pci_device_id', 0)):02x}"
            except (TypeError, ValueError):
                pci = str(pci)
        mine = {"rank": rank, "device": dev_index, "name": torch.cuda.get_device_name(dev), "backend": dist.get_backend(),
                "rccl": ver, "pci": pci, "uuid": str(getattr(props, "uuid", "")) or None}
        ranks_seen = [None] * world
        phase[0] = "rank census (all_gather_object + all_reduce)"
        dist.all_gather_object(ranks_seen, mine)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                       # a device collective over the same group: must count every rank
        assert int(probe.item()) == world, f"all_reduce saw {int(probe.item())} ranks, expected {world}"
        # one rank per GPU: N distinct devices (by PCI address where torch reports it, else by device index).  The single-GPU
        # rehearsal of the N > 1 path (DS_BENCH_DEVICE: every rank on one GPU, tests/test_gpu_multirank.py) is the one exception.
        if "DS_BENCH_DEVICE" not in os.environ:
            ids = [r.get("pci") or r.get("uuid") or f"index {r['device']}" for r in ranks_seen]
            assert len(set(ids)) == world, f"{world} ranks on {len(set(ids))} distinct GPUs: {ids}"

    import hashlib as _hl

    def max_over_ranks(x):
        if world > 1:
            tt = torch.tensor([x], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return float(tt.item())
        return x

    def timed_steps(pp, state, first, warmup, steps, nsched_):
        """W untimed + K timed steps from step index `first`, bracketed by barrier + synchronize, max over ranks -> (seconds, next index).
        A run longer than the schedule wraps around inside [first, nsched_): never back into the leading steps the operand policy gives
        another rung (they lie in front of `first`)."""
        span = max(1, nsched_ - first)
        wrap = lambda k_: first + (k_ - first) % span
        k = first
        for _ in range(warmup):
            pp.ring_step(state, wrap(k))
            k += 1
        barrier()
        t_ = time.perf_counter()
        for _ in range(steps):
            pp.ring_step(state, wrap(k))
            k += 1
        barrier()
        return max_over_ranks(time.perf_counter() - t_), k

    nsched = GEOM["num_inference_steps"] - 1      # steps 0..48 re-noise the overlaps; the last step of a schedule does not,
    phase[0] = "warm-up + timed steps"            # so a run longer than one panorama wraps around before it
    for k in range(lead_steps):                  # the schedule's leading steps, each in the mode the policy gives it (untimed here: timed
        pipe.ring_step(st, k)                    # per rung further down, and inside the complete 50-step panorama)
    own_step = lambda k_: lead_steps + (k_ - lead_steps) % max(1, nsched - lead_steps)     # a later step index, kept among the own-mode steps
    elapsed, step_idx = timed_steps(pipe, st, lead_steps, args.warmup, args.steps, nsched)
    assert bool(torch.isfinite(st.pano.float()).all()), "non-finite latent after the timed steps"
    if pipe.operand_policy == "auto":         # every warm-up / timed step ran in the model's own mode
        assert all(pipe.precision_for(NSTEPS_ - 1 - k, 7.5) is None for k in range(lead_steps, min(step_idx, nsched))), (lead_steps, step_idx)
    # what the job computed, so that runs can be compared: the panorama latent after warmup + timed steps is a function of
    # (config, warmup, steps, latents, residual mode) only -- not of --gpus, --tile-batch, --streams or --graph (rank sharding
    # and batching are bit-exact: tests/test_gpu_fullsize.py, test_gpu_multirank.py)
    digests = {"latent_after_timed_steps": _hl.sha256(st.pano.float().cpu().numpy().tobytes()).hexdigest()[:16]}
    ms_per_step = 1e3 * elapsed / args.steps
    steps_per_s = args.steps / elapsed

    # ---- per-rank account of ONE more (untimed) step: what each rank computed, how long it worked before it entered the step's
    #      exchange(s) and how long it sat in them (device-synchronised around every exchange: wait for the slowest rank included).
    #      A first real N-GPU run reads straight off this which rank is slow and whether compute or the exchange is. ----
    from dynamicscaler_amd import parallel
    phase[0] = "per-rank instrumented step"
    barrier()
    prof = parallel.profile_begin()
    t_ = time.perf_counter()
    pipe.ring_step(st, own_step(step_idx))
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t_
    parallel.profile_end()
    step_idx += 1
    mine_rank = {"rank": rank, "step_ms": round(1e3 * t_rank, 2), "compute_ms": round(1e3 * (t_rank - prof["exchange_s"]), 2),
                 "exchange_ms": round(1e3 * prof["exchange_s"], 2), "exchanges": prof["exchanges"], "exchange_bytes_sent": prof["exchange_bytes"],
                 "tiles_owned": prof["tiles_owned"], "eval_units_owned": prof["units_owned"], "share_mode": getattr(st, "share_mode", None)}
    per_rank = [mine_rank]
    if world > 1:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine_rank)

    # ---- the metric's second figure, measured: one complete 50-step panorama (steps 0..49, the last one without re-noise),
    #      fresh state, same mode as the timed steps; bracketed like the timed region, max over ranks ----
    full_s = None
    # under rocprofv3 the default is off: a trace of 50 more steps (a quarter of a million kernel records) has crashed the
    # profiler's own tool thread; --full-panorama 1 still forces it
    profiled = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ)
    if args.full_panorama > 0 or (args.full_panorama < 0 and args.config != "cfg5" and not profiled):
        phase[0] = "complete 50-step panorama"
        st2 = begin_state()
        barrier()
        t0 = time.perf_counter()
        for i in range(GEOM["num_inference_steps"]):
            pipe.ring_step(st2, i)
        barrier()
        full_s = max_over_ranks(time.perf_counter() - t0)
        assert bool(torch.isfinite(st2.pano_x0.float()).all()), "non-finite pred_x0 panorama after the 50-step run"
        digests["pred_x0_of_the_50_step_panorama"] = _hl.sha256(st2.pano_x0.float().cpu().numpy().tobytes()).hexdigest()[:16]
        del st2
    tiles_per_step = GEOM["num_windows_w"] * GEOM["num_windows_h"]
    flops_per_step = tiles_per_step * 2 * F_UNET

    # ---- the same steps in the other residual mode (outside the reported region): both ms/step on one line ----
    timed_mode = mode_name()
    mode_ms = {timed_mode: round(ms_per_step, 2)}
    if args.other_mode and not profiled and timed_mode in ("f16", "f32outer"):
        other = "f16" if timed_mode == "f32outer" else "f32outer"
        unet.residual_dtype, unet.residual_scope = MODES[other]
        phase[0] = "other residual mode"
        st3 = begin_state()
        t_other, _ = timed_steps(pipe, st3, 0, max(1, args.warmup), args.steps, nsched)
        mode_ms[other] = round(1e3 * t_other / args.steps, 2)
        del st3
        unet.residual_dtype, unet.residual_scope = MODES[timed_mode]

    # ---- roofline of the dominant kernel: per-launch HIP events on one extra (untimed) step, this rank's share ----
    roofline = None
    if not args.no_roofline:
        events = []

        def hook(name, flops, launch, info=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            events.append((name, flops, e0, e1, info))

        # the tile ops go through the kernel-level front end (ops.set_timing_hook wraps each call); the UNet's launches are made by the
        # C program, which reports each one to a host callback right before and right after it is enqueued (ds_unet_set_hooks)
        pending = {}

        def unet_hook(phase, kernel, flops, info):
            if phase == 0:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                pending[kernel] = e0
            else:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                events.append((kernel, flops, pending.pop(kernel), e1, info))

        ops.set_timing_hook(hook)
        unet.launch_hook = unet_hook
        pipe.num_streams = 1               # per-launch durations are taken with one kernel on the GPU at a time
        pipe.use_graph = False             # ... and launch by launch
        pipe.ring_step(st, own_step(step_idx))
        torch.cuda.synchronize()
        ops.set_timing_hook(None)
        unet.launch_hook = None
        agg = {}
        shapes = {}
        for name, flops, e0, e1, info in events:
            dt = e0.elapsed_time(e1) * 1e-3
            nbytes = 0.0
            if name == "gemm" and info:   # A + W + residual + out once each; fp16 operands, residual / out rows fp16 or fp32 as the
                _, M_, N_, K_, epi_, res_ = info      # launch's epilogue says (GEGLU writes N/2 columns)
                kin = K_ // 9 if info[0] == 1 else (K_ // 3 if info[0] == 2 else K_)
                nbytes = 2.0 * (M_ * kin + N_ * K_) + (4.0 if epi_ & 4 else 2.0) * M_ * (N_ // 2 if epi_ & 1 else N_) + \
                    res_ * (4.0 if epi_ & 8 else 2.0) * M_ * N_
            for a in (agg.setdefault(name, [0, 0.0, 0.0, 0.0]), shapes.setdefault((name,) + tuple(info or ()), [0, 0.0, 0.0, 0.0])):
                a[0] += 1
                a[1] += flops
                a[2] += dt
                a[3] += nbytes
        if os.environ.get("DS_BENCH_BREAKDOWN") and rank == 0:   # per-shape table (diagnostics, profiles/)
            with open(os.environ["DS_BENCH_BREAKDOWN"], "w") as f:
                f.write("kernel,shape,launches,ms_per_step,tflops\n")
                for k, a in sorted(shapes.items(), key=lambda kv: -kv[1][2]):
                    f.write(f"{k[0]},{'x'.join(map(str, k[1:]))},{a[0]},{a[2] * 1e3:.3f},{a[1] / a[2] / 1e12:.1f}\n")
        # second roofline of SURVEY 8-d: the tile ops are HBM-bound byte movers -- algorithmic bytes / HIP-event time / 8 TB/s
        HBM_PEAK = 8.0e12
        tile_ops = {}
        for (name, flops, e0, e1, info) in events:
            if info and info[0] == "bytes":
                a = tile_ops.setdefault(name, [0, 0, 0.0])
                a[0] += 1
                a[1] += info[1]
                a[2] += e0.elapsed_time(e1) * 1e-3
        tile_roof = {k: {"launches_per_step": a[0], "bytes_per_launch": int(a[1] / a[0]), "avg_launch_us": round(1e6 * a[2] / a[0], 2),
                         "achieved": round(a[1] / a[2] / 1e9, 1), "frac": round(a[1] / a[2] / HBM_PEAK, 4)} for k, a in tile_ops.items()}
        # the same launches by rocprofv3's kernel durations (committed capture of this workload): the HIP-event figure above is mostly
        # the gap in front of a 9 us kernel, this one is the kernel
        try:
            import csv
            import glob
            stats = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench_kernel_stats_cfg3_1stream.csv")))[-1]
            with open(stats) as f:
                krows = list(csv.DictReader(f))
            for k, rec in tile_roof.items():
                hit = [r for r in krows if f"{k}_kernel" in r["Name"]]
                if hit:
                    us = sum(float(r["TotalDurationNs"]) for r in hit) / sum(int(r["Calls"]) for r in hit) * 1e-3
                    rec["rocprof_kernel_us"] = round(us, 2)
                    rec["rocprof_achieved"] = round(rec["bytes_per_launch"] / (us * 1e-6) / 1e9, 1)
                    rec["rocprof_frac"] = round(rec["bytes_per_launch"] / (us * 1e-6) / HBM_PEAK, 4)
                    rec["rocprof_source"] = os.path.relpath(stats, REPO)
        except (OSError, IndexError, KeyError, ValueError, ZeroDivisionError):
            pass
        if tile_ops:
            tb, tt_ = sum(a[1] for a in tile_ops.values()), sum(a[2] for a in tile_ops.values())
            tile_roof["all"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9, "achieved": round(tb / tt_ / 1e9, 1),
                                "frac": round(tb / tt_ / HBM_PEAK, 4), "bytes_per_tile_per_step": int(tb / tiles_per_step * world),
                                "ms_per_step": round(1e3 * tt_, 3),
                                "note": "launch-latency bound at these sizes (a few MB per launch); HIP-event bracketing adds a few us each"}
        g = agg["gemm"]
        achieved = g[1] / g[2] / 1e12
        # context for the vendor peak: what the vendor GEMM library (hipBLASLt through torch.matmul) reaches on THIS box on a
        # large square fp16 GEMM (random data) -- the chip lowers its clock under sustained MFMA load, so
        # 2.5 PFLOP/s is not reachable; measurement only, not part of the product path
        lib_tf = None
        try:
            n_ = 8192
            a_ = (torch.randn(n_, n_, device=dev) * 0.5).half()
            b_ = (torch.randn(n_, n_, device=dev) * 0.5).half()
            for _ in range(3):
                torch.matmul(a_, b_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                torch.matmul(a_, b_)
            e1.record()
            torch.cuda.synchronize()
            lib_tf = round(20 * 2.0 * n_ ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
            del a_, b_
        except Exception:
            pass
        # HBM bytes per GEMM launch: PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/pmc_summary.py) cannot
        # run inside this process; the committed summary of the same workload is quoted, with its source
        # `traffic` is only reported when that summary was captured on THIS kernel source (it records the SHA-256 of
        # csrc/gemm.hip); a summary of an older build is named under quoted_from_profile, marked stale, and traffic is null.
        import glob
        import hashlib
        import re
        cur = hashlib.sha256(open(os.path.join(REPO, "dynamicscaler_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            def _rv(f):                    # (round, version): r3_pmc_hbm_traffic.json, r2_pmc_hbm_traffic_v14.json
                m = re.search(r"r(\d+)_pmc_hbm_traffic(?:_v(\d+))?\.json$", f)
                return (int(m.group(1)), int(m.group(2) or 0))
            pat = re.compile(r"r\d+_pmc_hbm_traffic(?:_v\d+)?\.json$")       # (not the other configs' captures, e.g. ..._cfg5.json)
            src = max((f for f in glob.glob(os.path.join(REPO, "profiles", "r*_pmc_hbm_traffic*.json")) if pat.search(f)), key=_rv)
            summ = json.load(open(src))
            traffic_src = os.path.relpath(src, REPO)
            traffic_stale = summ.get("gemm_hip_sha256") != cur
            if not traffic_stale and summ.get("residual_mode", "f16") == timed_mode:     # (fp32 rows move more bytes: same mode only)
                traffic = summ["gemm_f16_kernel(all)"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        # the same figure from the committed rocprofv3 --kernel-trace --stats summary of this command (one stream, so that a
        # kernel's duration is its own): algorithmic FLOPs of this run / the profile's GEMM time per step
        rocprof = None
        try:
            cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_rocprof_step_summary_cfg3_1stream.json")))
            src = cands[-1] if cands else ""
            summ = json.load(open(src)) if (args.config == "cfg3" and os.path.exists(src)) else None
            if summ is not None and summ.get("residual_mode", "f16") == timed_mode:
                fam = summ["families"]["gemm"]
                rp = g[1] / (fam["ms_per_step"] * 1e-3) / 1e12
                rocprof = {"avg_launch_us": round(fam["avg_launch_us"], 1), "achieved": round(rp, 1),
                           "frac": round(rp * 1e12 / MFMA_PEAK_F16, 4), "source": os.path.relpath(src, REPO),
                           "stale": summ.get("gemm_hip_sha256") != cur,
                           "note": "HIP-event bracketing adds a few us per launch; the two figures agree within run-to-run "
                                   "variation of the box (the chip's clock under load differs from box to box)"}
        except Exception:
            pass
        roofline = {
            "bound": "mfma", "kernel": "gemm_f16_kernel (implicit GEMM: linear / conv3x3 / temporal conv)",
            "achieved": round(achieved, 2), "peak": MFMA_PEAK_F16 / 1e12, "unit": "TFLOP/s",
            "frac": round(achieved * 1e12 / MFMA_PEAK_F16, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
            "algorithmic_bytes_per_launch": round(g[3] / g[0]),
            "measured_with": "per-launch HIP events on one extra step, one stream, eager launches, tile batch %d" % args.tile_batch,
            "vendor_library_gemm_tflops_on_this_box": lib_tf,
            # figures that do NOT come from this run: committed rocprofv3 / PMC summaries of the same command (PMC passes cannot run
            # inside this process); "stale" = captured on a different csrc/gemm.hip than the one that just ran
            "quoted_from_profile": {"traffic_source": traffic_src, "traffic_stale": traffic_stale, "rocprof": rocprof,
                                    "pmc_summary": "profiles/*_pmc_mfma_util.json (MFMA busy, wait / issue shares, VALU:MFMA, LDS bank "
                                                   "conflicts, clock and HBM bytes of the top shapes)"},
            "tile_ops": tile_roof,
            "launches_per_step": g[0], "algorithmic_tflop_per_step": round(g[1] / 1e12, 2),
            "avg_launch_us": round(1e6 * g[2] / g[0], 2), "gemm_time_share_of_step": round(g[2] / (elapsed / args.steps), 3),
            "attention_tflops": round(agg["attention"][1] / agg["attention"][2] / 1e12, 2) if "attention" in agg else None,
            "whole_step_tflops": round(flops_per_step / (elapsed / args.steps) / 1e12, 2),
            "whole_step_frac": round(flops_per_step / (elapsed / args.steps) / MFMA_PEAK_F16, 4),
            "whole_step_flops_basis": "32 full UNet evaluations per step (the reference's algorithm); with cfg_prefix_shared "
                                      "the context-free prefix of each cond/uncond pair is executed once, so the executed "
                                      "FLOPs are lower -- `achieved` above counts executed GEMM FLOPs only",
        }

    # ---- one step of the same panorama in the WIDE operand mode (outside the reported region): what a step costs where the operand
    #      policy selects it -- none of the 50-step schedule at CFG 7.5 (config.wide_steps_of_schedule), config 1's first three ----
    wide_step_ms, strict_step_ms, own_step_eager_ms = None, None, None
    if (args.wide_step > 0 or (args.wide_step < 0 and world == 1 and not profiled)) and args.config != "cfg5":
        pipe.num_streams, pipe.use_graph = args.streams, False
        pol = pipe.operand_policy
        rung_ms = {}
        for rung in ("f16", "strict", "wide"):      # eager launches for all three, so that the ratios compare like with like
            phase[0] = f"{rung}-rung step"
            pipe.operand_policy = rung
            try:
                stw = begin_state()
                pipe.ring_step(stw, 0)             # packs the twin's image (strict: a second fp16 image; wide: hi + lo planes), loads the kernels
                barrier()
                t_ = time.perf_counter()
                pipe.ring_step(stw, 1)
                barrier()
                rung_ms[rung] = round(1e3 * max_over_ranks(time.perf_counter() - t_), 1)
                assert bool(torch.isfinite(stw.pano.float()).all())
                del stw
            finally:
                pipe.operand_policy = pol
        own_step_eager_ms, strict_step_ms, wide_step_ms = rung_ms["f16"], rung_ms["strict"], rung_ms["wide"]
        unet._twins.clear()                    # the twins' packed images (2.6 + 5.3 GB) are not needed any more
        torch.cuda.empty_cache()

    # ---- the other BASELINE configurations one GPU can run (cfg2 / cfg4 / cfg5): W = 1 + K = 2 steps each, same bracketing, so
    #      that the driver's record holds a figure for every configuration, not only the headline's ----
    other_configs = None
    if args.other_configs > 0 or (args.other_configs < 0 and world == 1 and args.config == "cfg3" and not profiled):
        other_configs = {}
        pipe.num_streams, pipe.use_graph = args.streams, bool(args.graph)
        pipe._graphs.clear()
        torch.cuda.empty_cache()
        for name in ("cfg2", "cfg4", "cfg5"):
            if name == args.config:
                continue
            phase[0] = f"other configuration {name}"
            t_setup = time.time()
            pp, begin_o, _shape = make_pipe(name)
            so = begin_o()
            g_ = CONFIGS[name]["geom"]
            lead_o = 0                              # (own-mode steps, like the headline's timed region)
            while lead_o < g_["num_inference_steps"] - 1 and pp.precision_for(g_["num_inference_steps"] - 1 - lead_o, 7.5) is not None:
                lead_o += 1
            t_o, _k = timed_steps(pp, so, lead_o, 1, 2, g_["num_inference_steps"] - 1)
            assert bool(torch.isfinite(so.pano.float()).all()), name
            other_configs[name] = {"workload": CONFIGS[name]["label"], "ms_per_step": round(1e3 * t_o / 2, 2), "value": round(2 / t_o, 4),
                                   "unit": "denoising-steps/sec", "steps": 2, "warmup": 1, "tiles_per_step": g_["num_windows_w"] * g_["num_windows_h"],
                                   "residual_mode": timed_mode, "timed_region_starts_at_step": lead_o, "setup_s": round(time.time() - t_setup - t_o, 1),
                                   "result_sha256": _hl.sha256(so.pano.float().cpu().numpy().tobytes()).hexdigest()[:16]}
            del pp, so, begin_o
            torch.cuda.empty_cache()
        # what gen_pano_360.py runs FIRST is not a ring loop but the i2v sphere loop (gen_pano_360.py:227-260, 400-478): 44 perspective
        # views of 512 x 320 x 16f per step on a 2048 x 1024 equirect (fov 120, theta offset walking over 10 steps), per-view image tokens,
        # paste_on_static, merge-prev, the 48-step schedule.  1 warm-up + 2 timed steps, device-synchronised after every step.
        phase[0] = "other configuration sphere_stage1"
        try:
            from dynamicscaler_amd.sphere import VC2_Pipeline_I2V_SpherePano as SpherePipe
            t_setup = time.perf_counter()
            ld_s, params_s, _ = host_for("i2v")
            sp = SpherePipe(ld_s, lvdm_DDIM_Scheduler(ld_s, rng_mode="device"), {"params": {"unet_config": {"params": params_s}}})
            sp.to(dev, lat_dt)
            sp.max_tile_batch, sp.use_graph = args.tile_batch, bool(args.graph)
            ring_ = [360 * t // 6 for t in range(6)]
            stamps = []

            def sphere_cb(i, t, views, p, p0):
                torch.cuda.synchronize()
                stamps.append((time.perf_counter(), len(views)))
            t_s0 = time.perf_counter()
            fin_s, _den_s = sp.basic_sample_shift_shpere_panorama(
                prompt="a synthetic prompt", fps=8, guidance_scale=7.5, output_type="latent", height=320, width=512, frames=16, total_f=16,
                equirect_width=2048, equirect_height=1024, view_fov=120, loop_step_theta=10, num_inference_steps=48, denoise_to_step=5,
                phi_theta_dict={90: [0], -90: [0], 75: ring_, -75: ring_, 60: ring_, -60: ring_, 45: ring_, -45: ring_, 0: ring_},
                merge_renoised_overlap_latent_ratio=1, overlap_ratio_list_f=[0.75] * 24 + [0.5] * 24, loop_step_frame=8, paste_on_static=True,
                merge_prev_denoised_ratio_list=[0.5 * (1 - t / 10) for t in range(10)] + [0] * 38,
                init_sphere_latent=synth_normal((1, 4, 16, 128, 256), 2333333), pano_image_tensor=synth_normal((3, 1024, 2048), 77).clamp(-1, 1),
                static_frame_latent=synth_normal((1, 4, 1, 128, 256), 78), step_callback=sphere_cb)
            # steps 0-2 untimed (the operand policy puts the 48-step schedule's first two on the strict rung), steps 3 and 4 timed
            t_timed = stamps[4][0] - stamps[2][0]
            assert sp.precision_for(48 - 1 - 3, 7.5) is None and sp.precision_for(48 - 1 - 4, 7.5) is None
            assert bool(torch.isfinite(fin_s.float()).all()), "sphere_stage1"
            other_configs["sphere_stage1"] = {
                "workload": "gen_pano_360.py stage 1: i2v sphere loop, 2048x1024 equirect, 44 views of 512x320x16f per step (fov 120), per-view "
                            "image tokens, paste_on_static, merge-prev, CFG 7.5, 48-step schedule",
                "ms_per_step": round(1e3 * t_timed / 2, 2), "value": round(2 / t_timed, 4), "unit": "denoising-steps/sec", "steps": 2, "warmup": 3,
                "strict_steps_of_schedule": sp.strict_steps_of(48, 7.5), "strict_step_ms": round(1e3 * (stamps[1][0] - stamps[0][0]), 1),
                "views_per_step": stamps[0][1], "unet_evals_per_step": 2 * stamps[0][1], "first_step_ms": round(1e3 * (stamps[0][0] - t_s0), 1),
                "residual_mode": timed_mode, "setup_s": round(t_s0 - t_setup, 1),
                "note": "the views of a step overlap: 31 dependency levels of 1-2 views, so an evaluation runs at the small-batch rate",
                "result_sha256": _hl.sha256(fin_s.float().cpu().numpy().tobytes()).hexdigest()[:16]}
            del sp, fin_s, _den_s
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001 -- a diagnostic entry must not cost the headline line
            other_configs["sphere_stage1"] = {"error": f"{type(e).__name__}: {e}"[:300]}

        # ---- the reference's real entry point as ONE figure: gen_pano_360.py's default run (main(): 227-384; defaults :47-69, :404-455) is
        #      15 steps of the i2v sphere loop on the 2048 x 1024 equirect (44 views a step: sphere_stage1 above), then -- resumed with
        #      use_skip_time at step 15 of 48 -- 33 steps of the i2v ring loop on 1024 x 512 (2 x 2 windows) and, after a bicubic x2 and a
        #      re-noise, 33 steps on 2048 x 1024 (4 x 4 windows), then the seam-safe decode of 16 frames: 2 640 UNet evaluations = 33 PFLOP
        #      (BASELINE.md section 1).  Each stage: 1 warm-up + 2 timed steps, EXTRAPOLATED to its step count; the hand-offs (nearest
        #      resize, bicubic resize + re-noise) and the decode tail are timed once, whole. ----
        phase[0] = "other configuration gen_pano_360_default"
        try:
            if "ms_per_step" not in other_configs.get("sphere_stage1", {}):
                raise RuntimeError("sphere_stage1 was not measured")
            import numpy as _np
            from dynamicscaler_amd.pipelines_i2v import VC2_Pipeline_I2V_SpherePano as RingI2V
            from dynamicscaler_amd.tensor_utils import resize_video_latent
            from dynamicscaler_amd.vae import AutoencoderKL
            from dynamicscaler_amd.vae_spec import vae_param_shapes
            NSCHED, D2S = 48, 15
            ld_s, params_s, _ = host_for("i2v")
            merge_list = [0.5 * (1 - t / 20) for t in range(20)] + [0] * (NSCHED - 20)       # gen_pano_360.py:491-493 (merge_denoised defaults)
            ov_f = [0.75] * (NSCHED // 2) + [0.5] * (NSCHED // 2)

            def ring_stage(tw, th, nw, nh, init_lat, img_seed):
                pp = RingI2V(ld_s, lvdm_DDIM_Scheduler(ld_s, rng_mode="device"), {"params": {"unet_config": {"params": params_s}}})
                pp.to(dev, lat_dt)
                pp.max_tile_batch, pp.num_streams, pp.use_graph, pp.share_cfg_prefix = args.tile_batch, args.streams, bool(args.graph), bool(args.share_cfg_prefix)
                st_ = pp.ring_begin(prompt="a synthetic prompt", fps=8, guidance_scale=7.5, init_panorama_latent=init_lat, height=320, width=512, frames=16,
                                    total_w=tw, total_h=th, total_f=16, num_windows_w=nw, num_windows_h=nh, num_windows_f=1, loop_step=16, dock_at_f=True,
                                    loop_step_frame=8, overlap_ratio_list_f=ov_f, merge_prev_denoised_ratio_list=merge_list, num_inference_steps=NSCHED,
                                    use_skip_time=True, skip_time_step_idx=D2S, pano_image_tensor=synth_normal((3, th, tw), img_seed).clamp(-1, 1))
                t_, _k = timed_steps(pp, st_, 0, 1, 2, NSCHED - D2S - 1)
                assert bool(torch.isfinite(st_.pano.float()).all())
                return t_ / 2, pp, st_

            sphere_lat = synth_normal((1, 4, 16, 128, 256), 2333341).to(dev)              # stands in for stage 1's panorama latent
            torch.cuda.synchronize(); t_h = time.perf_counter()
            lat1 = resize_video_latent(sphere_lat.clone(), target_height=64, target_width=128, mode="nearest")
            torch.cuda.synchronize(); t_resize1 = time.perf_counter() - t_h
            s2_step, pp2, st2_ = ring_stage(1024, 512, 2, 2, lat1, 91)
            lat2 = st2_.pano_x0.clone()
            del pp2, st2_
            torch.cuda.synchronize(); t_h = time.perf_counter()
            up = resize_video_latent(lat2.clone(), target_height=128, target_width=256, mode="bicubic")
            sch_ = lvdm_DDIM_Scheduler(ld_s, rng_mode="device")
            sch_.make_schedule(NSCHED, verbose=False)
            mixed = sch_.re_noise(up, 0, NSCHED - D2S)
            torch.cuda.synchronize(); t_resize2 = time.perf_counter() - t_h
            s3_step, pp3, st3_ = ring_stage(2048, 1024, 4, 4, mixed, 92)
            # decode tail (seam-safe: W padded with wrapped 1/16 chunks, per-frame decode, crop): the real first-stage config, synthetic weights
            dd = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                      attn_resolutions=[], dropout=0.0)
            vae = AutoencoderKL(dd, 4)
            vae.load_state_dict(synth_state_dict(vae_param_shapes(dd, 4), seed=24))
            ld_s.first_stage_model, sf_keep = vae.to(dev), getattr(ld_s, "scale_factor", 1.0)
            ld_s.scale_factor = 0.18215
            try:
                for timed in (False, True):                                        # first pass: repack + warm caches
                    torch.cuda.synchronize(); t_h = time.perf_counter()
                    videos, _lat = pp3.ring_finish(st3_, output_type="tensor")
                    torch.cuda.synchronize(); t_decode = time.perf_counter() - t_h
                assert tuple(videos.shape) == (1, 3, 16, 1024, 2048) and bool(torch.isfinite(videos).all())
            finally:
                ld_s.first_stage_model, ld_s.scale_factor = None, sf_keep
            del pp3, st3_, videos, vae
            torch.cuda.empty_cache()
            s1_step = other_configs["sphere_stage1"]["ms_per_step"] * 1e-3
            n1, n2, n3 = D2S, NSCHED - D2S, NSCHED - D2S
            total_s = n1 * s1_step + n2 * s2_step + n3 * s3_step + t_resize1 + t_resize2 + t_decode
            evals = 2 * (n1 * 44 + n2 * 4 + n3 * 16)
            f_total = evals * CONFIGS["cfg4"]["f_unet"]
            other_configs["gen_pano_360_default"] = {
                "workload": "gen_pano_360.py default run: 48-step schedule, denoise_to_step 15 -- 15 i2v sphere steps x 44 views (2048x1024 equirect) + 33 i2v ring "
                            "steps x 4 windows (1024x512) + 33 x 16 windows (2048x1024), CFG 7.5, + hand-offs + seam-safe decode of 16 frames",
                "sec_per_run": round(total_s, 2), "sec_per_run_is": "EXTRAPOLATED: 2 timed steps per stage (1 warm-up) x the stage's step count; hand-offs and decode timed once",
                "unet_evals": evals, "pflop": round(f_total / 1e15, 2), "tflops": round(f_total / total_s / 1e12, 1),
                "stages": {"sphere_2048x1024_44_views": {"steps": n1, "ms_per_step": round(1e3 * s1_step, 1)},
                           "ring_1024x512_2x2": {"steps": n2, "ms_per_step": round(1e3 * s2_step, 1)},
                           "ring_2048x1024_4x4": {"steps": n3, "ms_per_step": round(1e3 * s3_step, 1)},
                           "resize_nearest_ms": round(1e3 * t_resize1, 2), "resize_bicubic_renoise_ms": round(1e3 * t_resize2, 2),
                           "decode_16_frames_1024x2048_ms": round(1e3 * t_decode, 1)},
                "note": "stage inputs are synthetic latents of the right shape and noise level (the stages are timed, not chained); image tokens through the HIP "
                        "CLIP image tower per crop; the reference's own CPU path for this run: 2 640 evaluations x 52.9 s (BASELINE.md section 2) = 38.8 h",
                "residual_mode": timed_mode}
        except Exception as e:      # noqa: BLE001 -- a diagnostic entry must not cost the headline line
            other_configs["gen_pano_360_default"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- CPU baseline: the oracle on this host, bounded sample ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        phase[0] = "cpu baseline"
        from oracle.unet import unet_forward
        from oracle import ring as oring, ddim as oddim
        ncpu = usable_cpus()
        torch.set_num_threads(ncpu)
        T_ = GEOM["frames"]
        x = synth_normal((1, 4, T_, 40, 64), 7)
        ctx = synth_normal((1, cfg["ctx_len"], 1024), 1)
        tc = time.perf_counter()
        e = unet_forward(sd, params, x, torch.tensor([499]), ctx, fps=8)
        t_fwd = time.perf_counter() - tc
        pano_c = synth_normal(pano_shape, 3)
        osched = oddim.DDIMSchedule(oddim.DiffusionTables(), 50)
        tc = time.perf_counter()
        for _ in range(3):
            w = oring.ring_gather(pano_c, 8, 72, 3, 43, 0, T_)
            m = oring.ring_gather(torch.zeros_like(pano_c), 8, 72, 3, 43, 0, T_)
            w = oddim.mix_latents_with_mask(w, oddim.re_noise(osched, w, 24, 25), m[0, 0, [0]], 1)
            xp, x0 = oddim.ddim_step(osched, w, oddim.cfg_combine(e, e, 7.5), [25] * T_)
            for dst in (pano_c, pano_c, pano_c):
                oring.ring_scatter(dst, xp, 8, 72, 3, 43, 0, T_)
        t_ops = (time.perf_counter() - tc) / 3
        t_step = tiles_per_step * (2 * t_fwd + t_ops)
        cpu_baseline = {
            "value": 1.0 / t_step, "unit": "denoising-steps/sec", "cores": ncpu, "kind": "port",
            "sample": f"1 fp32 UNet evaluation of one 512x320x{T_}f tile ({t_fwd:.1f} s) = 1/{2 * tiles_per_step} of a step + the "
                      f"tile ops of one tile ({t_ops * 1e3:.1f} ms); step time extrapolated as {tiles_per_step} tiles x "
                      f"(2 x UNet + tile ops)",
            "sec_per_step": round(t_step, 1),
        }

    # what the operand policy decided, on which (calibrated) figures; and what a 25- / 40-step schedule costs with the strict rung tried
    # before the wide one against round 5's wide-only ladder (from the three eager single-step figures above)
    operand_report = pipe.operand_report(NSTEPS_, 7.5)
    short_schedule_cost = None
    if wide_step_ms is not None:
        short_schedule_cost = {}
        keep_rungs = pipe.operand_rungs
        for n_ in (25, 40):
            sched.make_schedule(n_, verbose=False)
            row = {}
            for name_, rungs_ in (("strict_first", ("strict", "wide")), ("wide_only", ("wide",))):
                pipe.operand_rungs = rungs_
                ns_, nw_ = len(pipe.strict_steps_of(n_, 7.5)), len(pipe.wide_steps_of(n_, 7.5))
                row[name_] = {"strict_steps": ns_, "wide_steps": nw_,
                              "sec_per_panorama": round(1e-3 * (ns_ * strict_step_ms + nw_ * wide_step_ms + (n_ - ns_ - nw_) * own_step_eager_ms), 2)}
            short_schedule_cost[f"{n_}_steps"] = row
        pipe.operand_rungs = keep_rungs
        sched.make_schedule(NSTEPS_, verbose=False)
    if rank == 0:
        line = {
            "metric": f"denoising-steps/sec (whole node), {cfg['size']} panorama", "value": steps_per_s,
            "unit": "denoising-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['label']}, CFG 7.5, VideoCrafter {cfg['model']} UNet "
                                   f"({'1.41' if cfg['model'] == 't2v' else '1.44'} B parameters), DDIM 50-step schedule",
                       "baseline_config": args.config,
                       "tiles_per_step": tiles_per_step, "unet_evals_per_step": 2 * tiles_per_step,
                       "tile_batch": args.tile_batch, "streams": args.streams, "hipgraph": bool(args.graph), "cfg_prefix_shared": bool(args.share_cfg_prefix), "parallelism": f"tiles sharded over {world} GPU(s)",
                       "rng": "philox in-kernel (perf mode)",
                       "latents": "fp32 (as the reference)" if args.latents == "f32" else "fp16",
                       "residual_mode": timed_mode, "ms_per_step_by_residual_mode": mode_ms,
                       # which steps of the schedule the operand policy evaluates in the wide mode (fp32 storage, split-fp16 products:
                       # csrc/wide.hip) -- by schedule index, per DDIM step; [] = every step on single fp16 operands -- and what one
                       # such step of this panorama costs (measured outside the reported region)
                       "operand_policy": operand_report, "wide_steps_of_schedule": wide_steps_of_schedule,
                       "strict_steps_of_schedule": strict_steps_of_schedule,
                       "operand_mode_per_step": (f"strict rung (fp32 residual stream, fp16 operands) at schedule indices {strict_steps_of_schedule}, "
                                                 if strict_steps_of_schedule else "") +
                                                (f"wide at schedule indices {wide_steps_of_schedule}, " if wide_steps_of_schedule else "") +
                                                "the model's own mode (fp16 operands) at every other step of the 50-step schedule",
                       "timed_region_starts_at_step": lead_steps,
                       "wide_step_ms": wide_step_ms, "strict_step_ms": strict_step_ms, "own_mode_step_eager_ms": own_step_eager_ms,
                       "short_schedule_cost": short_schedule_cost,
                       "residual_stream": ("fp16 (matrix-core operands are fp16 in every mode)" if unet.residual_dtype == torch.float16 else
                                           "fp32 between the blocks, fp16 inside the transformers (DS_RESIDUAL_DTYPE=f32outer)"
                                           if unet.residual_scope == "outer" else "fp32 everywhere (strict precision mode, DS_RESIDUAL_DTYPE=f32)"),
                       "unet_program": "ds_unet_forward (block program and launch loop in C++; one C call per evaluation)",
                       "bit_repeatable": "yes, in every mode (streams x hipGraph included): the cause of round 1's run-to-run "
                                         "differences under concurrent graph replays is fixed (profiles/r2_notes.md section 1)"},
            "sec_per_50_step_panorama": full_s if full_s is not None else 50 * elapsed / args.steps,
            "sec_per_50_step_panorama_is": "measured: one complete 50-step loop" if full_s is not None else "50 x the timed steps' mean",
            "speedup_vs_cpu_baseline": (steps_per_s / cpu_baseline["value"]) if cpu_baseline else None,
            "setup_s": round(setup_s, 1),
            # sha-256 prefixes of the latents this run produced: equal for every --gpus / --tile-batch / --streams at the same
            # --config / --warmup / --steps (the N-GPU job computes the 1-GPU panorama, bit for bit)
            "result_sha256": digests,
            "ranks_seen": ranks_seen,
            # one instrumented step after the timed region, per rank: compute_ms = its own work up to (and between) the exchanges,
            # exchange_ms = time inside the collectives including the wait for the slowest rank
            "per_rank": per_rank,
            "other_configs": other_configs,
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), file=real_stdout, flush=True)
    phase[0] = "teardown"
    if world > 1:
        torch.distributed.destroy_process_group()
    watchdog.cancel()


if __name__ == "__main__":
    main()
