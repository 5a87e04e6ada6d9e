#!/usr/bin/env python3
"""One [cond | uncond] evaluation of the full t2v UNet at the real tile in the library's default mode and in the wide operand mode:
ms per evaluation pair (HIP events), and -- under `rocprofv3 --kernel-trace --stats -- python3 tools/bench_wide.py` -- where a wide
evaluation spends its time.    python tools/bench_wide.py [reps]"""
import os
import sys
import time

import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dynamicscaler_amd.unet import UNetModel  # noqa: E402
from dynamicscaler_amd.unet_spec import param_shapes  # noqa: E402
from dynamicscaler_amd.synth import synth_state_dict, synth_normal  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = torch.device("cuda:0")
params = yaml.safe_load(open(os.path.join(REPO, "dynamicscaler_amd", "configs", "t2v_512_v2_unet.yaml")))
m = UNetModel(**params)
m.load_state_dict(synth_state_dict(param_shapes(params), 0))
m = m.to(d)
x = synth_normal((1, 4, 16, 40, 64), 3).to(d)
xx = torch.cat([x, x])
ctx = torch.cat([synth_normal((1, 77, 1024), 1), synth_normal((1, 77, 1024), 2)]).to(d)
t = torch.tensor([999, 999], device=d)
for prec in (None, "wide"):
    m(xx, t, context=ctx, fps=8, cfg_pairs=1, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        m(xx, t, context=ctx, fps=8, cfg_pairs=1, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    print(f"{prec or 'default (f32outer)'}: {e0.elapsed_time(e1) / reps:.1f} ms per [cond | uncond] evaluation pair of one 512x320x16f tile", flush=True)
