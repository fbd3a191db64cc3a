from .ddpm import DDPM  # noqa: F401
from .ddim import DDIM  # noqa: F401
from .iddpm import IDDPM  # noqa: F401
