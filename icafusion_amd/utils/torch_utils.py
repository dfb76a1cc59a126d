"""Device / timing helpers with the reference's names (utils/torch_utils.py)."""
import time

import torch

from ..models.yolo import fuse_conv_and_bn  # noqa: F401


def select_device(device="", batch_size=None):
    """'cpu' is rejected: this implementation has no CPU path (reference utils/torch_utils.py:63-86)."""
    if str(device).lower() == "cpu":
        raise RuntimeError("icafusion_amd runs on MI355X GPUs only")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible")
    idx = int(str(device).split(",")[0]) if str(device) not in ("", "cuda") else 0
    return torch.device("cuda", idx)


def time_synchronized():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()
