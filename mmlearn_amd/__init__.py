"""mmlearn_amd -- MI355X-native contrastive / I-JEPA hot path for VectorInstitute/mmlearn.

Only the hot path lives here: the ContrastiveLoss boundary, the
ContrastivePretraining / IJEPA task surface around it and the HIP kernels (``csrc/``) behind a
C ABI (``include/mmlearn_hip.h``).  ``mmlearn_amd.fused`` / ``mmlearn_amd.attention`` hold the opt-in
encoder-side kernels of the first widening (section 8(f1)): ``fused.accelerate_encoder(module, ...)``.
Importing the package does not load the HIP library; the first op does, and fails loudly if it is missing.
"""

from .losses import ContrastiveLoss, LossPairSpec, find_matching_indices  # noqa: F401
from .modalities import Modalities  # noqa: F401
from .ema import ExponentialMovingAverage  # noqa: F401
from .layers import L2Norm, LearnableLogitScaling  # noqa: F401
from .masking import IJEPAMaskGenerator  # noqa: F401
from .ops import IndexedMasks, apply_masks, ijepa_loss, ijepa_target, l2_normalize, predictor_assemble, repeat_interleave_batch  # noqa: F401

from .strategy import TowerDDPStrategy  # noqa: F401  (registers "tower_ddp" in Lightning's StrategyRegistry when Lightning is there)

__version__ = "0.1.0"
