#!/usr/bin/env python3
"""Wall time of the conditioning producers at the reference's sizes (one prompt / one 320x512 crop), synthetic weights."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamicscaler_amd.encoders import FrozenOpenCLIPEmbedder, FrozenOpenCLIPImageEmbedderV2, Resampler
from dynamicscaler_amd.encoder_spec import *
from dynamicscaler_amd.synth import synth_encoder_state_dict, synth_normal

d = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


txt = FrozenOpenCLIPEmbedder(layer="penultimate")
txt.load_state_dict(synth_encoder_state_dict(clip_text_param_shapes(CLIP_VIT_H_14["text"]), 1)); txt.to(d)
tok = torch.randint(0, 49408, (1, 77), device=d)
print(f"text tower (23 blocks, 77 tokens): {timeit(lambda: txt.encode(tok)):.2f} ms")
vis = FrozenOpenCLIPImageEmbedderV2()
vis.load_state_dict(synth_encoder_state_dict(clip_vision_param_shapes(CLIP_VIT_H_14["vision"]), 2)); vis.to(d)
res = Resampler(**RESAMPLER_I2V)
res.load_state_dict(synth_encoder_state_dict(resampler_param_shapes(**RESAMPLER_I2V), 3)); res.to(d)
img = synth_normal((1, 3, 320, 512), 4, 0.5).clamp(-1, 1).to(d)
print(f"preprocess 320x512 -> 224: {timeit(lambda: vis.preprocess(img)):.2f} ms")
pix = vis.preprocess(img)
print(f"image tower (32 blocks, 257 tokens): {timeit(lambda: vis.encode_pixels(pix)):.2f} ms")
t = vis.encode_pixels(pix)
print(f"resampler (4 layers, 273 keys): {timeit(lambda: res(t)):.2f} ms")
