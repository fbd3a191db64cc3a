"""CIFAR10 scaled to [-1, 1], resident in HBM.

Drop-in for the reference's `dmme.CIFAR10` data module (src/dmme/data_modules/cifar10.py:11-50: `data_dir`, `batch_size`,
`augs`; random horizontal flip by default; ToTensor then `norm`).  torchvision is not part of this image, so the
python-pickle batches torchvision would download (`cifar-10-batches-py/data_batch_{1..5}`) are read directly; there is
no network here either, so `prepare_data()` only checks that they exist.  `synthetic=True` (not in the reference) fills
the set with uniform random bytes of the same shape for benchmarking without the files."""

from __future__ import annotations

import os
import pickle
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from .data_module import DataModule, GpuBatchLoader

CIFAR_DIR = "cifar-10-batches-py"
TRAIN_FILES = tuple(f"data_batch_{i}" for i in range(1, 6))


class RandomHorizontalFlip:
    """stands in for `torchvision.transforms.RandomHorizontalFlip` in the YAML `augs` list: the flip itself happens on the
    GPU inside `dmme_image_batch` (one bit per image)"""

    def __init__(self, p: float = 0.5):
        self.p = float(p)


def read_cifar10_batches(data_dir: str, files=TRAIN_FILES) -> Tuple[np.ndarray, np.ndarray]:
    """(N, 3, 32, 32) uint8 images and (N,) int64 labels from the python-pickle batch files: each holds `data`
    (10000 x 3072 uint8, row = R plane, G plane, B plane, row-major 32 x 32) and `labels`."""
    xs, ys = [], []
    for name in files:
        path = os.path.join(data_dir, CIFAR_DIR, name)
        with open(path, "rb") as f:
            entry = pickle.load(f, encoding="latin1")
        xs.append(np.asarray(entry["data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(entry["labels"] if "labels" in entry else entry["fine_labels"], dtype=np.int64))
    return np.concatenate(xs), np.concatenate(ys)


class CIFAR10(DataModule):
    def __init__(self, data_dir: str = ".", batch_size: int = 128, augs: Optional[List[Callable]] = None, synthetic: bool = False, device="cuda"):
        super().__init__(batch_size)
        self.data_dir = data_dir
        if augs is None:
            augs = [RandomHorizontalFlip()]
        for a in augs:
            if type(a).__name__ != "RandomHorizontalFlip":
                raise NotImplementedError(f"augmentation {type(a).__name__} is PIL-side in the reference; only RandomHorizontalFlip runs on the GPU here")
        self.augs = augs
        self.synthetic = synthetic
        self.device = device
        self._cache = None

    def prepare_data(self):
        """the reference downloads through torchvision (cifar10.py:36-37); there is no network here: only check"""
        if self.synthetic:
            return
        missing = [n for n in TRAIN_FILES if not os.path.exists(os.path.join(self.data_dir, CIFAR_DIR, n))]
        if missing:
            raise FileNotFoundError(f"{os.path.join(self.data_dir, CIFAR_DIR)}: missing {missing}; place the extracted CIFAR10 python batches there or pass synthetic=True")

    def _resident(self):
        if self._cache is None:
            if self.synthetic:
                g = torch.Generator().manual_seed(1337)
                x = torch.randint(0, 256, (50000, 3, 32, 32), generator=g, dtype=torch.uint8)
                y = torch.randint(0, 10, (50000,), generator=g)
            else:
                self.prepare_data()
                xn, yn = read_cifar10_batches(self.data_dir)
                x, y = torch.from_numpy(xn), torch.from_numpy(yn)
            self._cache = (x.to(self.device), y.to(self.device))
        return self._cache

    def _loader(self, flip_p: float, shuffle: bool):
        import torch.distributed as dist

        x, y = self._resident()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        return GpuBatchLoader(x, y, self.batch_size, shuffle=shuffle, flip_p=flip_p, rank=rank, world=world, seed=torch.initial_seed() & 0x7FFFFFFF)

    def setup_train(self):
        p = 0.0
        for a in self.augs:
            p = a.p  # a list of flips composes to a flip with the last probability only in the degenerate single-entry case the YAML uses
        return self._loader(p, shuffle=True)

    def setup_test(self):
        return self._loader(0.0, shuffle=False)
