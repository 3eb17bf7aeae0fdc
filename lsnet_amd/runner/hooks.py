"""Training hooks with the reference's semantics (mmcv/runner/hooks/*.py): step LR schedule with
linear warm-up, the optimizer hook (backward -> gradient averaging -> clip -> SGD step), iteration
timer, text logger and epoch checkpoints.

By design nothing here forces a device->host synchronisation per iteration: the gradient norm and
the loss scalars stay on the device and are only read when the logger prints (every `interval`
iterations); the reference reads them every iteration (hooks/optimizer.py:22-27, base.py:202-207)."""
import time

import torch

from ..utils.registry import Registry, build_from_cfg

HOOKS = Registry('hook')


class Hook:
    priority = 50

    def before_run(self, runner): pass
    def after_run(self, runner): pass
    def before_train_epoch(self, runner): pass
    def after_train_epoch(self, runner): pass
    def before_train_iter(self, runner): pass
    def after_train_iter(self, runner): pass

    @staticmethod
    def every_n_iters(runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False

    @staticmethod
    def every_n_epochs(runner, n):
        return (runner.epoch + 1) % n == 0 if n > 0 else False


@HOOKS.register_module()
class DistSamplerSeedHook(Hook):
    """mmcv/runner/hooks/sampler_seed.py:6-10: a new shuffle of the distributed sampler every epoch."""

    def before_train_epoch(self, runner):
        sampler = getattr(getattr(runner, 'data_loader', None), 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(runner.epoch)


@HOOKS.register_module()
class StepLrUpdaterHook(Hook):
    """lr = base * gamma^(#passed steps); linear warm-up from warmup_ratio*lr over warmup_iters
    (hooks/lr_updater.py:83-92, 123-141, 153-181)."""
    priority = 10

    def __init__(self, step, gamma=0.1, by_epoch=True, warmup=None, warmup_iters=0, warmup_ratio=0.1, **kwargs):
        assert warmup in (None, 'constant', 'linear', 'exp')
        self.step = [step] if isinstance(step, int) else list(step)
        self.gamma, self.by_epoch = gamma, by_epoch
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.base_lr = []

    def _regular(self, runner):
        progress = runner.epoch if self.by_epoch else runner.iter
        exp = sum(progress >= s for s in self.step)
        return [lr * self.gamma ** exp for lr in self.base_lr]

    def _warm(self, cur_iter, regular):
        if self.warmup == 'constant':
            return [lr * self.warmup_ratio for lr in regular]
        if self.warmup == 'linear':
            k = (1 - cur_iter / self.warmup_iters) * (1 - self.warmup_ratio)
            return [lr * (1 - k) for lr in regular]
        k = self.warmup_ratio ** (1 - cur_iter / self.warmup_iters)
        return [lr * k for lr in regular]

    @staticmethod
    def _set(runner, lrs):
        for group, lr in zip(runner.optimizer.param_groups, lrs):
            group['lr'] = lr

    def before_run(self, runner):
        for g in runner.optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in runner.optimizer.param_groups]

    def before_train_epoch(self, runner):
        if self.by_epoch:
            self.regular_lr = self._regular(runner)
            self._set(runner, self.regular_lr)

    def before_train_iter(self, runner):
        cur = runner.iter
        if not self.by_epoch:
            self.regular_lr = self._regular(runner)
        if self.warmup is None or cur > self.warmup_iters:
            if not self.by_epoch:
                self._set(runner, self.regular_lr)
        elif cur == self.warmup_iters:
            self._set(runner, self.regular_lr)
        else:
            self._set(runner, self._warm(cur, self.regular_lr))


@HOOKS.register_module()
class OptimizerHook(Hook):
    """zero_grad -> loss.backward() (bucketed RCCL all-reduce overlaps here) -> finish the
    reduction -> clip_grad_norm_(max_norm, norm_type) -> optimizer.step()
    (hooks/optimizer.py:8-28)."""
    priority = 30

    def __init__(self, grad_clip=None):
        self.grad_clip = grad_clip

    def clip_grads(self, params):
        params = [p for p in params if p.requires_grad and p.grad is not None]
        if params:
            return torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)   # device scalar, no .item()

    def after_train_iter(self, runner):
        if not runner.outputs.get('backward_done'):   # a GraphedForwardBackward step has done all of this
            if hasattr(runner.model, 'zero_grad_buckets'):
                runner.model.zero_grad_buckets()     # data parallel: gradients live in the all-reduce buckets
            else:
                runner.optimizer.zero_grad(set_to_none=True)
            runner.outputs['loss'].backward()
            if hasattr(runner.model, 'reduce_gradients'):
                runner.model.reduce_gradients()
        fused = self._fused_step(runner)
        if fused is not None:      # clip + SGD in the library (runner/fused_sgd.py): same operations, three launches
            norm = fused.step()
            if norm is not None:
                runner.log_buffer_update({'grad_norm': norm.clone()}, runner.outputs['num_samples'])
            return
        if self.grad_clip is not None:
            norm = self.clip_grads(runner.model.parameters())
            if norm is not None:
                runner.log_buffer_update({'grad_norm': norm.detach()}, runner.outputs['num_samples'])
        runner.optimizer.step()

    def _fused_step(self, runner):
        """The library's clip + SGD plan for this runner, or None (not SGD with plain momentum, gradients that are not stable
        views of the all-reduce buckets, CPU)."""
        if not hasattr(runner.model, 'zero_grad_buckets') or not torch.cuda.is_available():
            return None
        plan = getattr(runner, '_clip_sgd', None)
        from .fused_sgd import ClipSGD, _signature
        if plan is None or plan.opt is not runner.optimizer or (plan.ok and not plan.still_valid()) or \
                (not plan.ok and plan.signature != _signature(runner.optimizer)):   # a declined plan: retried when what it saw changed
            plan = runner._clip_sgd = ClipSGD(runner.optimizer, self.grad_clip)
        return plan if plan.ok else None


@HOOKS.register_module()
class IterTimerHook(Hook):
    priority = 80

    def before_train_epoch(self, runner):
        self.t = time.time()

    def before_train_iter(self, runner):
        runner.log_buffer_update({'data_time': time.time() - self.t}, 1)

    def after_train_iter(self, runner):
        runner.log_buffer_update({'time': time.time() - self.t}, 1)
        self.t = time.time()


@HOOKS.register_module()
class TextLoggerHook(Hook):
    """Averages the buffered scalars every `interval` iterations and prints one line
    (hooks/logger/text.py:44-130); the only place where loss values reach the host."""
    priority = 90

    def __init__(self, interval=50, by_epoch=True, **kwargs):
        self.interval = interval

    def after_train_epoch(self, runner):
        runner.log_buffer_average()           # windows do not span epochs (hooks/logger/base.py:158-163)

    def after_train_iter(self, runner):
        if (runner.inner_iter + 1) % self.interval == 0:      # every_n_inner_iters, hooks/logger/base.py:142-150
            avg = runner.log_buffer_average()
            if getattr(runner, 'rank', 0) != 0:                # only rank 0 prints (text.py:100-130 via master_only)
                runner.last_log = avg
                return
            lr = runner.optimizer.param_groups[0]['lr']
            msg = ', '.join(f'{k}: {v:.4f}' for k, v in avg.items())
            line = f'Epoch [{runner.epoch + 1}][{runner.inner_iter + 1}/{runner.epoch_len}]\tlr: {lr:.5f}, {msg}'
            if torch.cuda.is_available():
                line += f', memory: {int(torch.cuda.max_memory_allocated() / (1024 * 1024))}'
            runner.logger(line)
            runner.last_log = avg


@HOOKS.register_module()
class EvalHook(Hook):
    """Evaluate on the validation loader every `interval` epochs and log the metrics
    (mmdet/core/evaluation/eval_hooks.py:7-42).  The metrics of the last evaluation stay in `runner.eval_results`."""
    priority = 85

    def __init__(self, dataloader, interval=1, bbox_head=None, **eval_kwargs):
        from torch.utils.data import DataLoader
        if not isinstance(dataloader, DataLoader):
            raise TypeError(f'dataloader must be a pytorch DataLoader, but got {type(dataloader)}')
        self.dataloader, self.interval, self.eval_kwargs = dataloader, interval, eval_kwargs

    def _results(self, runner):
        from ..apis.test import single_gpu_test
        return single_gpu_test(runner.model, self.dataloader)

    def after_train_epoch(self, runner):
        if not self.every_n_epochs(runner, self.interval):
            return
        results = self._results(runner)
        runner.model.train()
        if results is None:                       # not rank 0 of a distributed evaluation
            return
        res = self.dataloader.dataset.evaluate(results, logger=runner.logger, **self.eval_kwargs)
        runner.eval_results = res
        runner.logger('Epoch(val) [%d]\t' % (runner.epoch + 1) +
                      ', '.join(f'{k}: {v}' for k, v in res.items() if not isinstance(v, dict)))


@HOOKS.register_module()
class DistEvalHook(EvalHook):
    """eval_hooks.py:45-98: every rank tests its share (DistributedSampler without shuffle), rank 0 gathers and scores."""

    def __init__(self, dataloader, interval=1, bbox_head=None, gpu_collect=False, **eval_kwargs):
        super().__init__(dataloader, interval, bbox_head, **eval_kwargs)

    def _results(self, runner):
        from ..apis.test import multi_gpu_test
        return multi_gpu_test(runner.model, self.dataloader)


@HOOKS.register_module()
class CheckpointHook(Hook):
    """Rank-0 checkpoint every `interval` epochs: epoch_N.pth + latest.pth
    (hooks/checkpoint.py:43-52)."""
    priority = 70

    def __init__(self, interval=1, by_epoch=True, save_optimizer=True, out_dir=None, **kwargs):
        self.interval, self.save_optimizer, self.out_dir = interval, save_optimizer, out_dir

    def after_train_epoch(self, runner):
        if not self.every_n_epochs(runner, self.interval) or runner.rank != 0:
            return
        out = self.out_dir or runner.work_dir
        if out:
            runner.save_checkpoint(out, save_optimizer=self.save_optimizer)


def build_hook(cfg, default_type=None):
    if isinstance(cfg, Hook):
        return cfg
    cfg = dict(cfg)
    if default_type and 'type' not in cfg:
        cfg['type'] = default_type
    return build_from_cfg(cfg, HOOKS)
