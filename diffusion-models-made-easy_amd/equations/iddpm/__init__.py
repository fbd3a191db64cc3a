from .iddpm import cosine_schedule, interpolate_variance, process_coefficients  # noqa: F401
