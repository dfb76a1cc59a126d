"""`from models.yolo_test import Model` — the module path train.py / test.py / pickled checkpoints use."""
from icafusion_amd.models.yolo import Model, Detect, parse_model, fuse_conv_and_bn, check_anchor_order  # noqa: F401
from icafusion_amd.models.common import *  # noqa: F401,F403
