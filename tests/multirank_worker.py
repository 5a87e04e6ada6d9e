"""Worker for tests/test_gpu_multirank.py (launched by torch.distributed.run, 2 ranks sharing cuda:0, gloo):
runs the t2v ring pipeline with the toy UNet rank-sharded and writes each rank's panorama replica."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
G = os.path.join(REPO, "tests", "golden")


def main():
    out_dir, geom_name, rng_mode = sys.argv[1], sys.argv[2], sys.argv[3]
    from dynamicscaler_amd import parallel
    from dynamicscaler_amd.host_model import LatentDiffusionHost
    from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
    from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
    from dynamicscaler_amd.synth import synth_state_dict
    from dynamicscaler_amd.unet_spec import param_shapes
    d = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group("gloo")
        parallel.host_staged_collectives(True)
    z = np.load(os.path.join(G, "loops_small.npz"))
    meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
    zt = np.load(os.path.join(G, "unet_tiny_t2v.npz"))
    params = json.loads(bytes(zt["params_json"]).decode())
    cond, uncond = torch.from_numpy(z["cond"]), torch.from_numpy(z["uncond"])
    ld = LatentDiffusionHost({"params": params}, conditioner=lambda p: uncond if p[0] == "" else cond)
    ld.model.diffusion_model.load_state_dict(synth_state_dict(param_shapes(params), 5), strict=True)
    ld.temporal_length = 4
    ld = ld.to(d).eval()
    pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode=rng_mode),
                                       {"params": {"unet_config": {"params": params}}}).to(d, torch.float16)
    pipe.use_graph, pipe.num_streams, pipe.max_tile_batch = True, 2, 2
    if os.environ.get("DS_SHARE_MODE"):
        parallel.share_mode(os.environ["DS_SHARE_MODE"])
    if os.environ.get("DS_WORKER_EAGER"):
        pipe.use_graph, pipe.num_streams = False, 1
    steps = []

    share_modes = []

    def on_step(i, t, wins, pano, pano_x0):
        import hashlib
        share_modes.append(getattr(pipe, "last_share_mode", None))
        steps.append(hashlib.sha256(pano.float().cpu().numpy().tobytes()).hexdigest()[:10])

    torch.manual_seed(2333333)
    _, den = pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent",
                                                   step_callback=on_step, **meta["geoms"][geom_name])
    print("STEP_HASHES", dist.get_rank() if dist.is_initialized() else 0, steps, flush=True)
    print("SHARE_MODE", share_modes[-1] if share_modes else None, flush=True)
    rank = dist.get_rank() if dist.is_initialized() else 0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), den=den.float().cpu().numpy(),
             final=pipe.final_latent.float().cpu().numpy(), share_mode=np.array(str(share_modes[-1] if share_modes else None)))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
