"""[0,1] <-> [-1,1] helpers with the reference's names (src/dmme/common/norm.py:4-11)."""

import torch


def norm(x):
    r"""[0, 1] -> [-1, 1]"""
    return 2 * (x - 0.5)


def denorm(x):
    r"""[-1, 1] -> [0, 1], clipped"""
    return torch.clip(0.5 * (x + 1), 0, 1)
