"""SURVEY 8(d)(i) / BASELINE.md 3.1: the REFERENCE's own Python (mmdet LSDetector + mmcv optimizer, imported through
bootstrap.py) with the CPU oracle behind its native-op stubs, timed in the build container: BASELINE config 1
(R-50-FPN bbox, 2 images 3x800x800), forward + backward + clip + SGD, one warm-up iteration excluded.

    python oracle/ref_harness/time_reference_cpu.py [iterations]

Test / measurement infrastructure (needs /root/reference); prints one JSON line."""
import copy
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle.ref_harness import bootstrap  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    bootstrap.load_reference()
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    import mmcv
    from mmdet.models import build_detector
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet, lsnet_config
    torch.manual_seed(0)
    own, _ = build_lsnet('bbox', 'r50')
    cfg = lsnet_config('bbox', 'r50')
    model_cfg = mmcv.Config(copy.deepcopy(cfg.model.to_dict() if hasattr(cfg.model, 'to_dict') else dict(cfg.model)))._cfg_dict
    model = build_detector(model_cfg, train_cfg=mmcv.Config(dict(cfg.train_cfg)), test_cfg=mmcv.Config(dict(cfg.test_cfg)))
    model.load_state_dict(own.state_dict(), strict=True)
    model.train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4)
    data = synthetic_batch('bbox', 2, 800, 800, seed=1234, device='cpu', channels_last=False)
    times, loss = [], None
    for it in range(iters + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        out = model.train_step(data, opt)
        out['loss'].backward()
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad and p.grad is not None], 35, 2)
        opt.step()
        dt = time.perf_counter() - t0
        loss = float(out['loss'])
        if it:
            times.append(dt)
    s = sum(times) / len(times)
    print(json.dumps({'what': 'reference Python (mmdet + mmcv) + CPU oracle ops, R-50-FPN bbox, 2 x 3x800x800, fwd+bwd+clip+SGD',
                      'seconds_per_iteration': round(s, 2), 'img_per_s': round(2 / s, 4), 'threads': threads, 'iterations': iters,
                      'last_loss': loss}))


if __name__ == '__main__':
    main()
