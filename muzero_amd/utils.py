"""Host-side mirrors of the reference's value-range helpers (utils.py:37-76, 138-148).

The device keeps one (min, max) pair per tree inside the HIP pool; these classes exist so that code written
against the reference (`MinMaxStats(known_bounds)`, `KnownBounds(minv, maxv).endpoints()`) keeps working and
so that a pool's statistics can be read back into the familiar object.
"""
import math
import random
import string
from typing import Optional, Tuple

import torch

INFINITY = float('inf')


def random_id() -> str:
    """utils.py:25-26"""
    return ''.join(random.choice(string.ascii_letters) for _ in range(8))


class KnownBounds:
    """A closed value range; minv=None means the range is symmetric, [-maxv, maxv] (utils.py:37-49)."""

    def __init__(self, minv: Optional[float], maxv: float) -> None:
        self.minv = minv
        self.maxv = maxv

    def endpoints(self) -> Tuple[float, float]:
        return (-self.maxv, self.maxv) if self.minv is None else (self.minv, self.maxv)


class MinMaxStats:
    """Running min/max of the values seen in one tree (utils.py:52-76)."""

    def __init__(self, known_bounds: Optional[KnownBounds]) -> None:
        self.known_bounds = known_bounds
        self.maximum = -INFINITY
        self.minimum = INFINITY

    def update(self, value: float) -> None:
        if self.known_bounds is not None:
            return
        if value > self.maximum:
            self.maximum = value
        if value < self.minimum:
            self.minimum = value

    def normalize(self, value: float) -> float:
        if self.known_bounds is not None:
            low, high = self.known_bounds.endpoints()
        elif self.maximum == self.minimum:
            return value
        else:
            low, high = self.minimum, self.maximum
        return (value - low) / (high - low)


def scalar_to_support(x_scalar: torch.Tensor, support_size: int) -> torch.Tensor:
    """utils.py:138-143 in torch (fp32 in, fp32 out): sqrt transform, two-hot between floor and ceil."""
    y = support_size * (torch.sqrt(x_scalar + 1) - 1) / (math.sqrt(2.) - 1)
    low = torch.floor(y)
    high = torch.ceil(y).long()
    p = (y - low).unsqueeze(-1)
    eye = torch.eye(support_size + 1, dtype=x_scalar.dtype, device=x_scalar.device)
    return p * eye[high] + (1 - p) * eye[low.long()]


def support_to_scalar(x_conv: torch.Tensor, support_size: int) -> torch.Tensor:
    """utils.py:146-148 in torch."""
    bins = torch.arange(support_size + 1, dtype=x_conv.dtype, device=x_conv.device)
    y = (x_conv * bins).sum(-1) / support_size
    return ((math.sqrt(2.) - 1) * y + 1) ** 2 - 1
