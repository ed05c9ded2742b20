import numpy as np
import torch


def to_tensor(pic):
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
