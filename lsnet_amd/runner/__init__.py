from .checkpoint import load_checkpoint, save_checkpoint
from .hooks import (HOOKS, CheckpointHook, DistEvalHook, DistSamplerSeedHook, EvalHook, Hook, IterTimerHook, OptimizerHook, StepLrUpdaterHook, TextLoggerHook)
from .runner import EpochBasedRunner, build_optimizer

__all__ = ['EpochBasedRunner', 'build_optimizer', 'load_checkpoint', 'save_checkpoint', 'HOOKS', 'Hook',
           'StepLrUpdaterHook', 'OptimizerHook', 'IterTimerHook', 'TextLoggerHook', 'CheckpointHook', 'DistSamplerSeedHook', 'EvalHook', 'DistEvalHook']
