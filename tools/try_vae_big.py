"""First-stage decode of one panorama frame at a given latent size (default: cfg5's 128 x 1024 -> 1024 x 8192 pixels), real config,
synthetic weights: does it run, how long, what does a banded evaluation differ by."""
import json, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dynamicscaler_amd.vae import AutoencoderKLDecoder
from dynamicscaler_amd.vae_spec import decoder_param_shapes
from dynamicscaler_amd.synth import synth_state_dict, synth_normal

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 1024)
mode = sys.argv[3] if len(sys.argv) > 3 else "f16"
zf = np.load(os.path.join(REPO, "tests", "golden", "vae_full.npz"))
dd = json.loads(bytes(zf["full_dd_json"]).decode())
m = AutoencoderKLDecoder(dd, 4)
m.operand_mode = mode
m.load_state_dict(synth_state_dict(decoder_param_shapes(dd, 4), seed=22))
z = synth_normal((1, 4, 1, h, w), 5).cuda()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    try:
        out = m.decode_frames(z, in_scale=1.0 / 0.18215)
    except Exception as e:      # noqa: BLE001
        print("FAILED:", type(e).__name__, str(e)[:300]); sys.exit(0)
    torch.cuda.synchronize()
    print(f"decode {h}x{w} [{mode}] -> {tuple(out.shape)} in {time.time() - t0:.2f} s, finite {bool(torch.isfinite(out).all())}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
if len(sys.argv) > 4:       # banded / blocked evaluation against the single-launch one (must be bit-identical)
    m.operand_limit = int(float(sys.argv[4]))
    out2 = m.decode_frames(z, in_scale=1.0 / 0.18215)
    print(f"operand_limit {m.operand_limit}: bit-identical to the unbanded decode: {bool(torch.equal(out, out2))}, "
          f"max abs diff {float((out - out2).abs().max()):.3e}")
if os.environ.get("TRY_VAE_CROSS"):     # the other operand mode at the same size: independent kernels, agreement to fp16 accuracy
    m.operand_mode = "wide" if mode == "f16" else "f16"
    m.operand_limit = 3 << 29
    t0 = time.time()
    out3 = m.decode_frames(z, in_scale=1.0 / 0.18215)
    torch.cuda.synchronize()
    e = float((out.double() - out3.double()).norm() / out3.double().norm())
    print(f"{m.operand_mode} decode of the same latent in {time.time() - t0:.2f} s: rel-L2 between the two modes {e:.3e}, peak mem "
          f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
