"""debug: product ring loop vs oracle loop with the fake eps model, step by step (GPU box)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loops as oloops, ddim as oddim
from dynamicscaler_amd.scheduler import lvdm_DDIM_Scheduler, DiffusionTables
from dynamicscaler_amd.pipelines import VC2_Pipeline_T2V_SpherePano
G = "tests/golden"
z = np.load(os.path.join(G, "loops_small.npz"))
meta = json.load(open(os.path.join(G, "loops_small_traces.json")))
cond, uncond = torch.from_numpy(z["cond"]), torch.from_numpy(z["uncond"])
d = torch.device("cuda:0")

class FakeModel(torch.nn.Module):
    diffusion_model = None
    def forward(self, x, t, c_crossattn=None, fps=None, **kw):
        ctx = torch.cat(c_crossattn, 1)
        m = torch.stack([0.01 * c[None].float().cpu().mean() for c in ctx]).to(x.device)
        return 0.1 * x.float() + m.reshape(-1, 1, 1, 1, 1)

class Host: pass
tables = DiffusionTables()
ld = Host(); ld.model = FakeModel()
for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "num_timesteps", "use_scale"):
    setattr(ld, k, getattr(tables, k))
ld.uncond_type, ld.temporal_length, ld.device = "empty_seq", 4, d
ld.get_learned_conditioning = lambda p: uncond if p[0] == "" else cond
geom = meta["geoms"]["grid4x2"]
ref_steps = []
torch.manual_seed(2333333)
snap = {}
def on_tile(i, win, pano, pano_x0):
    snap[i] = (pano.clone(), pano_x0.clone())
oloops.t2v_ring_sample(lambda x, ts, ctx: 0.1 * x + 0.01 * ctx.mean(), oddim.DiffusionTables(), cond, uncond,
                       guidance_scale=7.5, on_tile=on_tile, **geom)
pipe = VC2_Pipeline_T2V_SpherePano(ld, lvdm_DDIM_Scheduler(ld), {"params": {"unet_config": {"params": {"in_channels": 4}}}})
pipe.to(d, torch.float32)
torch.manual_seed(2333333)
def cb(i, t, wins, p, p0):
    a, b = snap[i]
    print(i, t, "pano maxdiff", float((p.cpu() - a).abs().max()), "x0 maxdiff", float((p0.cpu() - b).abs().max()),
          "nonequal", int((p.cpu() != a).sum()), int((p0.cpu() != b).sum()))
pipe.basic_sample_shift_multi_windows(prompt="a prompt", fps=8, guidance_scale=7.5, output_type="latent", step_callback=cb, **geom)
# isolate the model
x = torch.randn(2, 4, 4, 8, 16)
e = ld.model(x.to(d), None, c_crossattn=[torch.cat([cond, uncond]).to(d)])
r0 = 0.1 * x[:1] + 0.01 * cond.mean(); r1 = 0.1 * x[1:] + 0.01 * uncond.mean()
print("model diff", float((e[:1].cpu() - r0).abs().max()), float((e[1:].cpu() - r1).abs().max()))
