from abc import ABC, abstractmethod

import torch
import torch.nn as nn


class BasePatchEmbedding(nn.Module, ABC):
    """Image [B, C, H, W] -> token sequence [B, N, D]
    (interface of src/tokenizers/base_patch_embedding.py:6-21)."""

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ...
