from icafusion_amd.utils.general import *  # noqa: F401,F403
