"""`from utils.datasets import LoadImages, letterbox, create_dataloader_rgb_ir` — reference module path."""
from icafusion_amd.utils.datasets import *  # noqa: F401,F403
from icafusion_amd.utils.datasets import (LoadImages, PairedValSet, create_dataloader_rgb_ir, imread_bgr,  # noqa: F401
                                          imwrite_bgr, img2label_paths, letterbox, resize_area, resize_bilinear)
