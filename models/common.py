"""Alias of icafusion_amd.models.common under the reference's module path (so pickles / imports resolve)."""
from icafusion_amd.models.common import *  # noqa: F401,F403
from icafusion_amd.models.common import (Add, NiNfusion, Conv, Bottleneck, C3, SPPF, Concat, LearnableCoefficient,  # noqa: F401
                                         LearnableWeights, AdaptivePool2d, CrossAttention, CrossTransformerBlock,
                                         TransformerFusionBlock, autopad)
