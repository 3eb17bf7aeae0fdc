"""EpochBasedRunner with the reference's API (mmcv/runner/{base_runner,epoch_based_runner}.py):
`runner.run(data_loaders, workflow, max_epochs)`, `register_training_hooks(lr_config,
optimizer_config, checkpoint_config, log_config)`, `resume`, `save_checkpoint`; each iteration is
`model.train_step(data_batch, optimizer)` followed by the hooks."""
import os
from collections import OrderedDict

import torch
import torch.distributed as dist

from ..parallel.data_container import DataContainer, scatter
from .checkpoint import load_checkpoint, save_checkpoint
from .hooks import CheckpointHook, Hook, IterTimerHook, OptimizerHook, StepLrUpdaterHook, build_hook


def build_optimizer(model, cfg):
    """No paramwise config in the LSNet schedules: every parameter (norm scales and biases included)
    gets the same lr / weight decay (mmcv/runner/optimizer/default_constructor.py:197-198)."""
    cfg = dict(cfg)
    kind = cfg.pop('type')
    cfg.pop('paramwise_cfg', None)
    target = model.module if hasattr(model, 'module') else model
    params = list(target.parameters())
    if kind == 'SGD' and 'fused' not in cfg and 'foreach' not in cfg and params and all(p.is_cuda for p in params):
        cfg['fused'] = True    # weight decay + momentum + update in ONE multi-tensor launch (same arithmetic as the
                               # foreach form, which takes four launches over the 160 parameter tensors)
    return getattr(torch.optim, kind)(params, **cfg)


class EpochBasedRunner:

    def __init__(self, model, batch_processor=None, optimizer=None, work_dir=None, logger=None, meta=None):
        assert hasattr(model, 'train_step') or hasattr(getattr(model, 'module', None), 'train_step')
        self.model, self.optimizer, self.work_dir, self.meta = model, optimizer, work_dir, meta
        self.logger = logger.info if hasattr(logger, 'info') else (logger or print)
        if work_dir:
            os.makedirs(work_dir, exist_ok=True)
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.hooks = []
        self.epoch = self.iter = self.inner_iter = 0
        self.max_epochs = self.max_iters = None
        self.epoch_len = 0
        self.outputs, self.last_log = None, {}
        self.graph_step = None     # GraphedForwardBackward when enable_hip_graph() was called
        self._buf = OrderedDict()   # name -> [sum (tensor or float), count]

    # ---- log buffer (device-resident sums; averaged on demand) -----------------------------
    def log_buffer_update(self, vars_, count=1):
        for k, v in vars_.items():
            cur = self._buf.get(k)
            val = v * count
            if cur is None:
                self._buf[k] = [val, count]
            else:
                cur[0] = cur[0] + val
                cur[1] += count

    def log_buffer_average(self):
        out = OrderedDict()
        for k, (s, n) in self._buf.items():
            out[k] = float(s) / n
        self._buf.clear()
        return out

    def enable_hip_graph(self, warmup=3, bucket_mb=64.0):
        """Replay forward + backward of every iteration from one captured hipGraph (runner/graph_step.py).
        Needs static batch shapes; the data-parallel wrapper's backward hooks are bypassed (its buckets are
        replaced by the graph step's gradient-view buckets)."""
        from .graph_step import GraphedForwardBackward
        if hasattr(self.model, 'reducer'):
            for h in getattr(self.model.reducer, '_handles', []):
                h.remove()
        self.graph_step = GraphedForwardBackward(self.model, bucket_mb=bucket_mb, warmup=warmup)
        return self.graph_step

    def run_iter(self, batch):
        """forward (+ backward when graphed) of one iteration; the hooks do the rest.  A batch collated from the
        data pipeline still holds DataContainers: unwrap it onto this process's device first (what the reference's
        MMDistributedDataParallel.train_step does with `scatter`, mmcv/parallel/distributed.py:21-35)."""
        if isinstance(batch, dict) and any(isinstance(v, DataContainer) for v in batch.values()):
            p = next(self.model.parameters())
            batch = scatter(batch, p.device, channels_last=p.device.type == 'cuda')
        if self.graph_step is not None:
            return self.graph_step(batch)
        return self.model.train_step(batch, self.optimizer)

    # ---- hooks -------------------------------------------------------------------------------
    def register_hook(self, hook):
        assert isinstance(hook, Hook)
        self.hooks.append(hook)
        self.hooks.sort(key=lambda h: h.priority)

    def register_training_hooks(self, lr_config, optimizer_config=None, checkpoint_config=None, log_config=None):
        lr = dict(lr_config)
        policy = lr.pop('policy', 'step')
        assert policy == 'step', 'LSNet schedules use the step policy'
        self.register_hook(StepLrUpdaterHook(**lr))
        if optimizer_config is not None:
            self.register_hook(optimizer_config if isinstance(optimizer_config, Hook)
                               else OptimizerHook(**dict(optimizer_config)))
        if checkpoint_config is not None:
            self.register_hook(CheckpointHook(**dict(checkpoint_config)))
        self.register_hook(IterTimerHook())
        if log_config is not None:
            interval = log_config.get('interval', 50)
            for h in log_config.get('hooks', [dict(type='TextLoggerHook')]):
                h = dict(h)
                h.setdefault('interval', interval)
                self.register_hook(build_hook(h))

    def call_hook(self, name):
        for h in self.hooks:
            getattr(h, name)(self)

    # ---- checkpoints -------------------------------------------------------------------------
    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth', save_optimizer=True, meta=None):
        meta = dict(meta or {}, epoch=self.epoch + 1, iter=self.iter)
        path = os.path.join(out_dir, filename_tmpl.format(self.epoch + 1))
        save_checkpoint(self.model, path, optimizer=self.optimizer if save_optimizer else None, meta=meta)
        latest = os.path.join(out_dir, 'latest.pth')
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        return load_checkpoint(self.model, filename, map_location, strict)

    def resume(self, checkpoint, resume_optimizer=True, map_location='cpu'):
        ckpt = self.load_checkpoint(checkpoint, map_location=map_location)
        self.epoch, self.iter = ckpt['meta']['epoch'], ckpt['meta']['iter']
        if 'optimizer' in ckpt and resume_optimizer:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        self.logger(f'resumed epoch {self.epoch}, iter {self.iter}')

    # ---- main loop ---------------------------------------------------------------------------
    def train(self, data_loader):
        self.model.train()
        self.data_loader = data_loader
        self.epoch_len = len(data_loader)
        self.call_hook('before_train_epoch')
        for i, batch in enumerate(data_loader):
            self.inner_iter = i
            self.call_hook('before_train_iter')
            self.outputs = self.run_iter(batch)
            if not isinstance(self.outputs, dict):
                raise TypeError('model.train_step() must return a dict')
            if 'log_vars' in self.outputs:
                self.log_buffer_update(self.outputs['log_vars'], self.outputs['num_samples'])
            self.call_hook('after_train_iter')
            self.iter += 1
        self.call_hook('after_train_epoch')
        self.epoch += 1

    def run(self, data_loaders, workflow, max_epochs, **kwargs):
        assert isinstance(data_loaders, list) and len(data_loaders) == len(workflow)
        self.max_epochs = max_epochs
        for (mode, epochs), loader in zip(workflow, data_loaders):
            if mode == 'train':
                self.max_iters = max_epochs * len(loader)
        self.call_hook('before_run')
        while self.epoch < max_epochs:
            for (mode, epochs), loader in zip(workflow, data_loaders):
                if mode != 'train':
                    raise ValueError('only the train workflow is on the LSNet hot path')
                for _ in range(epochs):
                    if self.epoch >= max_epochs:
                        break
                    self.train(loader)
        self.call_hook('after_run')
