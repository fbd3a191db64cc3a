from .warmup import WarmupLR  # noqa: F401
