"""Make the reference's import paths resolve to this build, so `gen_pano_360.py`-style drivers and the yaml
`target:` strings keep working:

    import dynamicscaler_amd.dropin as dropin; dropin.install()
    from pipeline.t2v_sphere_panorama_pipeline import VC2_Pipeline_T2V_SpherePano
    from lvdm.modules.networks.openaimodel3d import UNetModel
"""
import sys
import types


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        if parent:
            setattr(_mod(parent), child, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def install():
    from . import unet, scheduler, pipelines, pipelines_i2v, ring, tensor_utils, host_model, sphere, vae, encoders
    from . import panorama_tensors as pt
    _mod("lvdm.modules.networks.openaimodel3d", UNetModel=unet.UNetModel)
    _mod("lvdm.models.ddpm3d", DiffusionWrapper=unet.DiffusionWrapper, LatentDiffusion=host_model.LatentDiffusionHost,
         LatentVisualDiffusion=host_model.LatentDiffusionHost)
    _mod("lvdm.models.autoencoder", AutoencoderKL=vae.AutoencoderKL)
    _mod("lvdm.modules.encoders.condition", FrozenOpenCLIPEmbedder=encoders.FrozenOpenCLIPEmbedder,
         FrozenOpenCLIPImageEmbedderV2=encoders.FrozenOpenCLIPImageEmbedderV2)
    _mod("lvdm.modules.encoders.ip_resampler", Resampler=encoders.Resampler)
    _mod("pipeline.scheduler", lvdm_DDIM_Scheduler=scheduler.lvdm_DDIM_Scheduler)
    _mod("pipeline.t2v_normal_pipeline", VC2_Pipeline_T2V=pipelines.VC2_Pipeline_T2V)
    _mod("pipeline.t2v_sphere_panorama_pipeline", VC2_Pipeline_T2V_SpherePano=sphere.VC2_Pipeline_T2V_SpherePano)
    _mod("utils.panorama_tensor_utils", PanoramaLatentProxy=sphere.PanoramaLatentProxy, PanoramaTensor=pt.PanoramaTensor)
    _mod("utils.ring_panorama_tensor_utils", RingPanoramaTensor=pt.RingPanoramaTensor, RingLatentProxy=pt.RingLatentProxy,
         RingPanoramaLatentProxy=pt.RingPanoramaLatentProxy)
    _mod("pipeline.i2v_normal_pipeline", VC2_Pipeline_I2V=pipelines_i2v.VC2_Pipeline_I2V)
    _mod("pipeline.i2v_sphere_panorama_pipeline", VC2_Pipeline_I2V_SpherePano=sphere.VC2_Pipeline_I2V_SpherePano)
    _mod("utils.shift_window_utils", RingLatent=ring.RingLatent, RingImageTensor=pipelines_i2v.RingImageTensor,
         get_dimension_slices_and_sizes=ring.get_dimension_slices_and_sizes)
    _mod("utils.tensor_utils", mix_latents_with_mask=tensor_utils.mix_latents_with_mask)
    _mod("utils.diffusion_utils", resize_video_latent=tensor_utils.resize_video_latent)
    _mod("utils.multi_prompt_utils",
         select_prompt_from_multi_prompt_dict_by_factor=pipelines.select_prompt_from_multi_prompt_dict_by_factor)
    _mod("utils.utils", instantiate_from_config=host_model.instantiate_from_config)
    _mod("scripts.evaluation.funcs", load_model_checkpoint=host_model.load_model_checkpoint)
