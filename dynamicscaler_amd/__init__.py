"""dynamicscaler_amd -- MI355X-native tiled panoramic denoising hot path of DynamicScaler.

HIP kernels (csrc/, C ABI in include/dynscaler_hip.h) + the Python host mirror of the reference's pipeline /
scheduler / ring-latent / UNet call surfaces.  No CPU fallback: importing works anywhere, running needs gfx950.
"""
__all__ = ["build", "ops", "unet", "scheduler", "ring", "pipelines", "parallel", "host_model", "dropin", "synth"]
