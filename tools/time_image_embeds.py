import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd.encoders import FrozenOpenCLIPImageEmbedderV2, Resampler
from dynamicscaler_amd.encoder_spec import CLIP_VIT_H_14, RESAMPLER_I2V, clip_vision_param_shapes, resampler_param_shapes
from dynamicscaler_amd.synth import synth_encoder_state_dict, synth_normal
d = torch.device("cuda:0")
e = FrozenOpenCLIPImageEmbedderV2(); e.load_state_dict(synth_encoder_state_dict(clip_vision_param_shapes(CLIP_VIT_H_14["vision"]), 71))
r = Resampler(**RESAMPLER_I2V); r.load_state_dict(synth_encoder_state_dict(resampler_param_shapes(**RESAMPLER_I2V), 72))
e.to(d); r.to(d)
for k in range(6):
    img = synth_normal((1, 3, 320, 512), 100 + k).clamp(-1, 1).to(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tok = r(e(img))
    torch.cuda.synchronize(); print(k, "get_image_embeds %.1f ms" % ((time.perf_counter() - t0) * 1e3), tuple(tok.shape), flush=True)
