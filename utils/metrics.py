from icafusion_amd.utils.metrics import *  # noqa: F401,F403
