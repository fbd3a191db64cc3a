from . import ddpm, iddpm  # noqa: F401
