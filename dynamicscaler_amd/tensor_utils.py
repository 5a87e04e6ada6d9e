"""Drop-in for utils/tensor_utils.py:mix_latents_with_mask (:19-39) on GPU tensors."""
import torch

from . import ops


def mix_latents_with_mask(latent_1, latent_to_add, mask, mix_ratio):
    """out = l1*(1-m) + (l1*(1-r) + add*r)*m, computed by ds_renoise_mix with c=0, s=1, noise=latent_to_add
    (0*x + 1*add == add exactly), same fp32 op order as the reference."""
    if len(mask.shape) == 3:
        frame0 = True
        m = (mask != 0).to(torch.uint8).reshape(1, 1, mask.shape[-2], mask.shape[-1])
        m = m.expand(latent_1.shape[0], latent_1.shape[2], -1, -1).contiguous()
    elif len(mask.shape) == 5:
        frame0 = False
        # the kernel's mask is one byte per (f,y,x); a 5-D mask is uniform over channels in every reference call site
        m = (mask[:, 0] != 0).to(torch.uint8).contiguous()
    else:
        print("noise shape should be [1, H, W] or [B, N, C, H, W]")
        raise NotImplementedError
    out = latent_1.contiguous().clone()
    pano_shape = (1,) + tuple(out.shape[1:])
    ops.renoise_mix_(out, m, pano_shape, 0.0, 1.0, mix_ratio, noise=latent_to_add.to(out.dtype).contiguous(),
                     mask_frame0=frame0)
    return out


def resize_video_latent(input_latent, target_height, target_width, mode="bilinear", align_corners=False):
    """utils/diffusion_utils.py:21-33 on the GPU (stage hand-off of gen_pano_360.py:287-289 'nearest', :345-347
    'bicubic').  Only the two modes the driver uses are implemented."""
    return ops.resize_latent(input_latent.contiguous(), target_height, target_width, mode)
