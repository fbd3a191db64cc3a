from .ddpm import LitDDPM  # noqa: F401
from .ddim import LitDDIM  # noqa: F401
from .iddpm import LitIDDPM  # noqa: F401
