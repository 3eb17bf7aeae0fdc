"""`train_detector` (mmdet/apis/train.py:33-128): wrap the model for data-parallel training, build
the optimizer and the runner, register the training hooks, run."""
import torch

from ..parallel import DataParallelModel
from ..runner import DistEvalHook, DistSamplerSeedHook, EpochBasedRunner, EvalHook, build_optimizer


def train_detector(model, data_loaders, cfg, distributed=False, validate=False, timestamp=None, meta=None,
                   logger=None, channels_last=True):
    data_loaders = data_loaders if isinstance(data_loaders, (list, tuple)) else [data_loaders]
    if torch.cuda.is_available():
        model = model.cuda()
        if channels_last:
            model = model.to(memory_format=torch.channels_last)
    # One process per GPU; the wrapper also is the gradient arena of a single process (bucket views as `p.grad` and as the
    # kernels' gradient sinks, parallel/reducer.py), so it is used on the device whether or not there are peers.
    if distributed or next(model.parameters()).is_cuda:
        model = DataParallelModel(model)
    optimizer = build_optimizer(model, cfg.optimizer)
    runner = EpochBasedRunner(model, optimizer=optimizer, work_dir=cfg.get('work_dir'), logger=logger, meta=meta)
    runner.register_training_hooks(cfg.lr_config, cfg.optimizer_config, cfg.get('checkpoint_config'),
                                   cfg.get('log_config'))
    if distributed:
        runner.register_hook(DistSamplerSeedHook())
    if validate:                                   # mmdet/apis/train.py:112-122
        from ..data import build_dataloader, build_dataset
        val = build_dataset(cfg.data.val, dict(test_mode=True))
        loader = build_dataloader(val, samples_per_gpu=1, workers_per_gpu=cfg.data.get('workers_per_gpu', 0),
                                  dist=distributed, shuffle=False)
        hook = DistEvalHook if distributed else EvalHook
        runner.register_hook(hook(loader, **dict(cfg.get('evaluation', {}))))
    if cfg.get('resume_from'):
        runner.resume(cfg.resume_from)
    elif cfg.get('load_from'):
        runner.load_checkpoint(cfg.load_from)
    runner.run(list(data_loaders), cfg.workflow, cfg.total_epochs)
    return runner
