import random

import numpy as np
import torch


def set_seed(seed=None, deterministic: bool = False) -> int:
    seed = int(seed) if seed is not None else 42
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    return seed
