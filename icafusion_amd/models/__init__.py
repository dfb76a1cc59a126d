from .common import *   # noqa: F401,F403
from .yolo import Model, parse_model, Detect  # noqa: F401
