"""`from models.yolo import Model` — two-stream Model with the reference's yolo_test semantics."""
from icafusion_amd.models.yolo import Model, Detect, parse_model, fuse_conv_and_bn, check_anchor_order  # noqa: F401
