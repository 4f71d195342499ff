"""Device-side counterparts of the reference's input-pipeline helpers, same names and argument meaning as
medseg/common_utils/basic_operations.py so the training script can swap the import; tensors stay on the GPU."""
from __future__ import annotations

import torch

from . import ops


def rescale_intensity(data: torch.Tensor, new_min=0, new_max=1, eps=1e-20) -> torch.Tensor:
    """basic_operations.py:232-245: min-max rescale of every (n, c) plane of an N*C*H*W batch."""
    return ops.rescale_intensity(data, float(new_min), float(new_max), float(eps))


def crop_or_pad(image: torch.Tensor, crop_size, label: torch.Tensor = None):
    """basic_operations.py:173-220.  Returns what upstream returns: (image, label) when nothing has to change, otherwise
    (image, label, h_s, w_s, h, w) with the crop offsets after padding and the padded size."""
    h, w = image.shape[-2], image.shape[-1]
    new_h, new_w = int(crop_size[0]), int(crop_size[1])
    if (new_h, new_w) == (h, w):
        return image, label
    img, lab = ops.crop_or_pad(image, (new_h, new_w), label)
    ph, pw = max(h, new_h), max(w, new_w)                      # size after the padding stage
    return img, lab, (ph - new_h) // 2, (pw - new_w) // 2, ph, pw


def add_input_noise(clean_image: torch.Tensor, sigma: float = 0.05, seed: int = 0, noise: torch.Tensor = None) -> torch.Tensor:
    """train_adv_supervised_segmentation_triplet.py:185-187: clamp(clean + 0.05 * N(0,1), 0, 1) in one kernel (noise drawn on
    device from `seed`, or injected)."""
    return ops.noise_clamp(clean_image, noise=noise, sigma=sigma, lo=0.0, hi=1.0, seed=seed)


def set_seed(seed):
    """Seeds the three host generators the training script seeds (basic_operations.py:22-34).  The engine draws the seeds of
    its device-side counter hashes (dropout patterns, soft-mask noise, input noise) from torch's host generator, so a seeded run
    repeats bit for bit (`tools/check_two_streams.py`); there is no autotuner whose choice could differ between runs."""
    import random

    import numpy as np
    if seed is not None:
        np.random.seed(seed)
        random.seed(seed)
        torch.manual_seed(seed)
