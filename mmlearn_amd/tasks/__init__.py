"""Task (LightningModule) surface of the hot path."""

from .base import EvaluationHooks, TrainingTask  # noqa: F401
from .contrastive_pretraining import (  # noqa: F401
    AuxiliaryTaskSpec,
    ContrastivePretraining,
    EvaluationSpec,
    LossPairSpec,
    ModuleKeySpec,
)
from .ijepa import IJEPA  # noqa: F401
