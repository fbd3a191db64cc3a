from .ddpm import linear_schedule, sampling_coefficients  # noqa: F401
