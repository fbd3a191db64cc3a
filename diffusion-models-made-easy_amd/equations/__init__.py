from . import ddpm, ddim, iddpm  # noqa: F401
