#!/usr/bin/env python3
"""Full-size repeatability check (GPU box): the bench's configuration (cfg3 panorama, hipGraph replays on two streams,
in-kernel Philox noise) run for 3 steps in N fresh processes must give the same panorama bits.
    python tools/check_repeatable_fullsize.py [N=3]"""
import hashlib, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib, yaml, torch
sys.path.insert(0, %r)
from dynamicscaler_amd.host_model import LatentDiffusionHost, SyntheticConditioner
from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler
from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
from dynamicscaler_amd.unet_spec import param_shapes
from dynamicscaler_amd.synth import synth_state_dict, synth_normal
dev = torch.device("cuda:0")
params = yaml.safe_load(open(os.path.join(%r, "dynamicscaler_amd", "configs", "t2v_512_v2_unet.yaml")))
ld = LatentDiffusionHost({"params": params}, conditioner=SyntheticConditioner(77, params["context_dim"]))
unet = ld.model.diffusion_model
unet.load_state_dict(synth_state_dict(param_shapes(params), seed=0), strict=True)
unet.prepare(dev)
pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld, rng_mode="device"), {"params": {"unet_config": {"params": params}}})
pipe.to(dev, torch.float16)
pipe.max_tile_batch, pipe.num_streams, pipe.use_graph = 8, 2, True
st = pipe.ring_begin(prompt="p", fps=8, guidance_scale=7.5, init_panorama_latent=synth_normal((1, 4, 16, 64, 512), 2333333).to(dev),
                     height=320, width=512, frames=16, total_w=4096, total_h=512, num_windows_w=8, num_windows_h=2,
                     num_windows_f=1, loop_step=8, num_inference_steps=50)
for i in range(3):
    pipe.ring_step(st, i)
torch.cuda.synchronize()
print("HASH", hashlib.sha256(st.pano.cpu().numpy().tobytes() + st.pano_x0.cpu().numpy().tobytes()).hexdigest()[:16])
''' % (REPO, REPO)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hs = []
for _ in range(n):
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    hs.append([ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
print(hs, "REPEATABLE" if len(set(hs)) == 1 else "DIFFERENT")
sys.exit(0 if len(set(hs)) == 1 else 1)
