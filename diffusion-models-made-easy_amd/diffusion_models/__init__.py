from .ddpm import DDPM  # noqa: F401
from .ddim import DDIM  # noqa: F401
