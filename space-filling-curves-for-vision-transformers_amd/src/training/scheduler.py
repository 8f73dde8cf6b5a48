from sfcvit.training.optim import WarmupCosineScheduler  # noqa: F401
