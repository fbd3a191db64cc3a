from . import ddpm  # noqa: F401
