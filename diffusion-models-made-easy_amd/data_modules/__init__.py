from .data_module import DataModule, GpuBatchLoader  # noqa: F401
from .cifar10 import CIFAR10, RandomHorizontalFlip, read_cifar10_batches  # noqa: F401
