"""dmme_amd: MI355X-native DDPM/DDIM/IDDPM denoiser path behind the reference's `dmme` surface.

Importable as `dmme_amd` (the directory name `diffusion-models-made-easy_amd` is not a
valid Python identifier; `dmme_amd/__init__.py` at the repository root aliases it)."""

__version__ = "0.1.0"

from . import _lib  # noqa: F401
from .common.noise import gaussian, gaussian_like, uniform_int, pad  # noqa: F401
from .common.norm import denorm, norm  # noqa: F401
from . import models, diffusion_models, equations, lit_modules, lr_scheduler, data_modules  # noqa: F401
from .data_modules import CIFAR10  # noqa: F401
from .diffusion_models import DDPM, DDIM, IDDPM  # noqa: F401
from .lit_modules import LitDDPM, LitDDIM, LitIDDPM  # noqa: F401
from .models.ddpm import UNet  # noqa: F401

__all__ = ["gaussian", "gaussian_like", "uniform_int", "pad", "denorm", "norm", "UNet", "DDPM", "DDIM", "LitDDPM", "LitDDIM", "IDDPM", "LitIDDPM", "CIFAR10"]
