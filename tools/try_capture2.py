"""Debug helper: full training iterations (runner hooks) eager vs graphed, loss per step."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd.data import synthetic_batch
from lsnet_amd.model_zoo import build_lsnet
from lsnet_amd.runner import EpochBasedRunner, build_optimizer

dev = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 480)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
torch.manual_seed(0)
model, cfg = build_lsnet('bbox', 'r50')
model = model.to(dev).to(memory_format=torch.channels_last).train()
twin = copy.deepcopy(model)
data = synthetic_batch('bbox', 2, h, w, seed=40, device=dev)
import os as _os
if _os.environ.get('LR0'):
    cfg.optimizer['lr'] = 0.0
modes = (('graph', twin, True),) if _os.environ.get('GRAPH_ONLY') else (('eager', model, False), ('graph', twin, True))
for name, m, graphed in modes:
    opt = build_optimizer(m, cfg.optimizer)
    r = EpochBasedRunner(m, optimizer=opt, logger=lambda s: None)
    r.register_training_hooks(cfg.lr_config, cfg.optimizer_config, None, dict(interval=10 ** 9, hooks=[]))
    if graphed:
        r.enable_hip_graph(warmup=2)
    r.epoch_len = 10 ** 9
    r.call_hook('before_run'); r.call_hook('before_train_epoch')
    for i in range(n):
        r.call_hook('before_train_iter')
        r.outputs = r.run_iter(data)
        r.log_buffer_update(r.outputs['log_vars'], 2)
        r.call_hook('after_train_iter')
        r.iter += 1
        torch.cuda.synchronize()
        gn = r._buf.get('grad_norm')
        print(name, i, 'loss %.6f' % float(r.outputs['log_vars']['loss']), 'lr %.6g' % opt.param_groups[0]['lr'], flush=True)
