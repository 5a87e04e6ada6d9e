"""TEST INFRASTRUCTURE ONLY (oracle side) -- never imported by the product path.

Imports the *reference* (sh-Lin/DynamicScaler, mounted read-only at /root/reference)
in THIS container so that `tests/golden/make_golden.py` can run the reference's own
Python on CPU and record golden input/output vectors.  The reference never travels to
the GPU box; only the vectors under tests/golden/ do.

The reference imports a handful of third-party modules that are absent from this
image (cv2, pytorch_lightning, torchvision, diffusers, imageio, omegaconf, decord,
open_clip, kornia).  None of them is used by the arithmetic of the hot path, so they
are replaced with the minimal inert stand-ins below.
"""
import os
import sys
import types
import contextlib

REFERENCE_ROOT = os.environ.get("DS_REFERENCE_ROOT", "/root/reference")


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch.nn as nn

    if "cv2" not in sys.modules:
        _module("cv2", INTER_LINEAR=1)
    if "imageio" not in sys.modules:
        _module("imageio")
    if "pytorch_lightning" not in sys.modules:
        def seed_everything(seed):
            import torch, random
            import numpy as np
            random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        class LightningModule(nn.Module):
            """nn.Module plus the two LightningModule properties the reference reads (AutoencoderKL.decode: self.dtype)."""
            @property
            def dtype(self):
                return next(self.parameters()).dtype

            @property
            def device(self):
                return next(self.parameters()).device

        _module("pytorch_lightning", LightningModule=LightningModule, seed_everything=seed_everything)
    if "torchvision" not in sys.modules:
        tv = _module("torchvision")
        tv.utils = _module("torchvision.utils", make_grid=lambda *a, **k: None)
        tv.transforms = _module("torchvision.transforms")
    if "diffusers" not in sys.modules:
        class _Logger:
            def __getattr__(self, name):
                return lambda *a, **k: None

        class DiffusionPipeline:
            def __init__(self):
                pass

            def register_modules(self, **kw):
                for k, v in kw.items():
                    setattr(self, k, v)

            @contextlib.contextmanager
            def progress_bar(self, total=None):
                class _Bar:
                    def update(self, *a):
                        pass
                yield _Bar()

            @property
            def _execution_device(self):
                import torch
                return torch.device("cpu")

            def to(self, *a, **k):
                return self

        class _Mixin:  # ConfigMixin placeholder (pipeline/d_scheduler.py, never instantiated)
            pass

        class _Mixin2:  # SchedulerMixin placeholder
            pass

        d = _module("diffusers", DiffusionPipeline=DiffusionPipeline, ConfigMixin=_Mixin,
                    SchedulerMixin=_Mixin2)
        d.logging = _module("diffusers.logging", get_logger=lambda name=None: _Logger())
        cu = _module("diffusers.configuration_utils", ConfigMixin=_Mixin,
                     register_to_config=lambda f: f)
        d.configuration_utils = cu
        su = _module("diffusers.schedulers")
        su.scheduling_utils = _module("diffusers.schedulers.scheduling_utils", SchedulerMixin=_Mixin2)
        d.schedulers = su
        du = _module("diffusers.utils", BaseOutput=object, logging=d.logging)
        du.torch_utils = _module("diffusers.utils.torch_utils", randn_tensor=None)
        d.utils = du


def import_reference():
    """Put /root/reference on sys.path (front) with the stubs installed."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not present at {REFERENCE_ROOT} (only exists in the build container)")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
