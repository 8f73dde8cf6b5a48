from .optim import FlatGradBuffer, FusedAdamW, WarmupCosine, WarmupCosineScheduler  # noqa: F401
from .train import GraphedTrainStep, SoftTargetCrossEntropy, mixup_soft_targets, train_step  # noqa: F401
from .distributed import GradReducer  # noqa: F401
