from . import ddpm, ddim  # noqa: F401
