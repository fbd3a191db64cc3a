from .ddim import linear_tau, quadratic_tau  # noqa: F401
