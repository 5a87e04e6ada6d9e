"""Decode-tail micro-benchmark (GPU box): the real first-stage decoder on T frames of a 40x64 latent (320x512 px tile).
    python tools/bench_vae.py [frames]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops
from dynamicscaler_amd.vae import AutoencoderKLDecoder
from dynamicscaler_amd.vae_spec import decoder_param_shapes, decoder_blocks
from dynamicscaler_amd.synth import synth_state_dict, synth_normal

dd = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
          attn_resolutions=[], dropout=0.0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = torch.device("cuda:0")
m = AutoencoderKLDecoder(dd, 4)
m.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=22))
m.prepare(d)
z = synth_normal((1, 4, T, 40, 64), 5).to(d)
# algorithmic FLOPs per frame (2*MAC)
h, w, fl = 40, 64, 0.0
for kind, p, cin, cout in decoder_blocks(dd):
    px = h * w
    if kind == "conv_in":
        fl += 2 * px * cout * 9 * cin
    elif kind == "res":
        fl += 2 * px * (cout * 9 * cin + cout * 9 * cout + (cin * cout if cin != cout else 0))
    elif kind == "attn":
        fl += 2 * px * 4 * cin * cin + 4 * px * px * cin
    elif kind == "up":
        h, w = 2 * h, 2 * w
        fl += 2 * h * w * cin * 9 * cin
    elif kind == "conv_out":
        fl += 2 * px * cout * 9 * cin
for fpc in (4, 8, 16):
    m.frames_per_chunk = fpc
    m.decode_frames(z, 1.0 / 0.18215)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = m.decode_frames(z, 1.0 / 0.18215)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"decode {T} frames 40x64 -> {tuple(out.shape)}, frames_per_chunk {fpc}: {dt*1e3:.1f} ms ({dt/T*1e3:.2f} ms/frame), "
          f"{fl*T/dt/1e12:.0f} TFLOP/s algorithmic ({fl/1e12:.2f} TFLOP/frame), peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
