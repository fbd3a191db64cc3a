from .noise import gaussian, gaussian_like, uniform_int, pad  # noqa: F401
from .norm import norm, denorm  # noqa: F401
