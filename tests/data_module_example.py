"""A caller-side DataLoader factory for `main.py --data-module data_module_example:loaders` (tests/test_parity_gpu.py):
stands where the reference builds its torchvision CIFAR-10 loaders inline (/root/reference/main.py:169-230).  CPU tensors
from a plain torch DataLoader, as a real pipeline would hand them over."""
import torch
from torch.utils.data import DataLoader, TensorDataset


def loaders(batch_size, img_size, classes, rank, world, seed):
    g = torch.Generator().manual_seed(seed + rank)
    def make(n):
        x = torch.randn(n, 3, img_size, img_size, generator=g)
        y = torch.randint(0, classes, (n,), generator=g)
        return DataLoader(TensorDataset(x, y), batch_size=batch_size, shuffle=False, drop_last=True)
    return make(4 * batch_size), make(2 * batch_size)
