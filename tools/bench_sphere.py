"""The sphere loops at gen_pano_360.py's stage-1 geometry (what the reference's entry point runs first: gen_pano_360.py:227-260,
400-478): equirect 2048 x 1024 (latent 128 x 256), 44 perspective views of 512 x 320 at fov 120 (phi 90 / -90: one view; phi +-75, +-60,
+-45, 0: phi_num = 6 each), theta offset walking with loop_step_theta = 10, 16 frames, CFG 7.5, the 48-step schedule.  The i2v loop
(VC2_Pipeline_I2V_SpherePano.basic_sample_shift_shpere_panorama: per-view image tokens, paste_on_static on a given static latent,
merge-prev) and the t2v loop, real UNet configs with synthetic weights and embeddings.  Prints ms per step (after one warm-up step)
and, with --check, that the level-batched execution equals the one-view-at-a-time one bit for bit."""
import argparse, json, os, sys, time
import numpy as np, torch, yaml
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
from dynamicscaler_amd.sphere import VC2_Pipeline_T2V_SpherePano, VC2_Pipeline_I2V_SpherePano
from dynamicscaler_amd.unet_spec import param_shapes
from dynamicscaler_amd.synth import synth_state_dict, synth_normal

ap = argparse.ArgumentParser()
ap.add_argument("--model", choices=["i2v", "t2v"], default="i2v")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--tile-batch", type=int, default=8)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--check", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
phi_num = 6
ring = [360 * t // phi_num for t in range(phi_num)]
phi_theta = {90: [0], -90: [0], 75: ring, -75: ring, 60: ring, -60: ring, 45: ring, -45: ring, 0: ring}
geom = dict(height=320, width=512, frames=16, equirect_width=2048, equirect_height=1024, view_fov=120, loop_step_theta=10,
            phi_theta_dict=phi_theta, merge_renoised_overlap_latent_ratio=1, num_inference_steps=48, denoise_to_step=args.steps)
name = {"t2v": "t2v_512_v2_unet.yaml", "i2v": "i2v_512_v1_unet.yaml"}[args.model]
params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", name)))
ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"]))
ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 0), strict=True)
if args.model == "i2v":
    from helpers import synth_image_embedder
    ld.get_image_embeds = synth_image_embedder(params["context_dim"])
    ld.embedder = object()
ld = ld.to(dev)
init = synth_normal((1, 4, 16, 128, 256), 2333333)
extra = {}
if args.model == "i2v":
    extra = dict(total_f=16, overlap_ratio_list_f=[0.75] * 24 + [0.5] * 24, loop_step_frame=8, paste_on_static=True,
                 pano_image_tensor=synth_normal((3, 1024, 2048), 77).clamp(-1, 1), static_frame_latent=synth_normal((1, 4, 1, 128, 256), 78),
                 merge_prev_denoised_ratio_list=[0.5 * (1 - t / 10) for t in range(10)] + [0] * 38)


def run(tile_batch):
    Pipe = VC2_Pipeline_I2V_SpherePano if args.model == "i2v" else VC2_Pipeline_T2V_SpherePano
    pipe = Pipe(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), {"params": {"unet_config": {"params": params}}})
    pipe.to(dev, torch.float32)
    pipe.max_tile_batch, pipe.use_graph = tile_batch, bool(args.graph)
    stamps, nviews = [], []

    def cb(i, t, views, p, p0):
        torch.cuda.synchronize()
        stamps.append(time.time())
        nviews.append(len(views))
    torch.manual_seed(2333333)
    torch.cuda.synchronize()
    t0 = time.time()
    final, den = pipe.basic_sample_shift_shpere_panorama(prompt="a synthetic prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                         init_sphere_latent=init, step_callback=cb, **geom, **extra)
    torch.cuda.synchronize()
    steps = [1e3 * (b - a) for a, b in zip([t0] + stamps[:-1], stamps)]
    return final, den, steps, nviews


final, den, steps, nviews = run(args.tile_batch)
line = dict(workload=f"gen_pano_360 stage 1: {args.model} sphere loop, equirect 2048x1024, {nviews[0]} views of 512x320x16f per step, CFG 7.5",
            unet_evals_per_step=2 * nviews[0], tile_batch=args.tile_batch, hipgraph=bool(args.graph), ms_per_step=[round(s, 1) for s in steps],
            ms_per_step_after_warmup=round(float(np.mean(steps[1:])), 1) if len(steps) > 1 else None,
            finite=bool(torch.isfinite(den).all()))
if args.check:
    f1, d1, _, _ = run(1)
    line["level_batches_equal_one_view_at_a_time"] = bool(torch.equal(final, f1) and torch.equal(den, d1))
print(json.dumps(line))
