"""hipGraph replay of forward+backward vs eager launches on the full LSNet R-50-FPN model (small images)."""
import copy

import pytest
import torch

from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner import EpochBasedRunner, build_optimizer


def _train(model, cfg, batches, graphed):
    opt = build_optimizer(model, cfg.optimizer)
    r = EpochBasedRunner(model, optimizer=opt, logger=lambda s: None)
    r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
    gs = r.enable_hip_graph(warmup=1) if graphed else None
    losses = []

    class Spy(type(r.hooks[0]).__mro__[1]):
        priority = 95

        def after_train_iter(self, runner):
            losses.append(runner.outputs['log_vars']['loss'].detach().clone())
    r.register_hook(Spy())
    r.run([batches], [('train', 1)], 1)
    return torch.stack(losses).cpu(), gs


@pytest.mark.gpu
def test_graph_replay_trains_like_eager():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    twin = copy.deepcopy(model)
    # six different batches of one shape: the graph step must rebind its inputs every iteration
    batches = [synthetic_batch("bbox", 2, 384, 480, seed=40 + i, device=dev) for i in range(6)]
    eager, _ = _train(model, cfg, batches, graphed=False)
    graph, gs = _train(twin, cfg, batches, graphed=True)
    assert gs.graph is not None and gs.calls == 6          # 1 eager warm-up call, 1 capture, 4 replays
    assert torch.isfinite(graph).all()
    # same kernels, same order; only the fp32 atomics of the DCN backward differ run to run
    assert torch.allclose(eager, graph, rtol=5e-3), (eager, graph)
    for (k, a), (_, b) in zip(model.state_dict().items(), twin.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-1, atol=5e-5), k
    with pytest.raises(RuntimeError):
        gs(synthetic_batch("bbox", 2, 384, 512, seed=1, device=dev))   # other shapes are not this graph


@pytest.mark.gpu
def test_capture_with_fresh_weight_images_still_follows_the_optimizer():
    """the prepared weight images are refreshed from Python when an optimizer step made them stale.  A capture
    that happens with FRESH images (no optimizer step between the last warm-up call and the capture: a custom loop,
    skipped / accumulated steps) used to record no rebuild launch -- every replay then ran forward and data gradient on
    the capture-time weights.  _capture now invalidates the images first, so the rebuild is always in the graph: after
    an optimizer step the replayed forward must see the new weights."""
    from lsnet_amd.runner.graph_step import GraphedForwardBackward
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    data = synthetic_batch('bbox', 2, 384, 480, seed=3, device=dev)
    gs = GraphedForwardBackward(model, warmup=2)
    for _ in range(2):
        gs(data)                       # warm-up calls: eager, NO optimizer step in between
    out = gs(data)                     # capture + first replay, images fresh at capture time
    assert gs.graph is not None
    loss0 = float(out['loss'])
    with torch.no_grad():              # the parameters move (a small clipped gradient step on the replay's gradients)
        params = [p for p in model.parameters() if p.grad is not None]
        norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in params]))
        for p in params:
            p.add_(p.grad, alpha=-float(0.5 / norm.clamp_min(1e-6)))
    loss1 = float(gs(data)['loss'])    # replay: must run on the NEW weights
    with torch.no_grad():
        eager = float(model.train_step(data, None)['loss'])
    assert abs(loss1 - eager) <= 1e-3 * abs(eager), (loss0, loss1, eager)
    assert abs(loss1 - loss0) > 1e-3 * abs(loss0), 'the replay still computes with the capture-time weights'
