"""Device-side image preparation: time of lsn_image_prep_u8 for the benchmark's image size (800 x 1344 slot from a
640 x 480 COCO-sized source) and the HBM roofline fraction of its 12 B / output-pixel write stream; the host library
beside it.   python tools/bench_image_prep.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsnet_amd.data.device_prep import prepare_batch  # noqa: E402

mean, std = np.array([123.675, 116.28, 103.53], np.float32), np.array([58.395, 57.12, 57.375], np.float32)
rng = np.random.RandomState(0)
imgs = [rng.randint(0, 256, (480, 640, 3)).astype(np.uint8) for _ in range(2)]
meta = dict(img_shape=(800, 1067, 3), pad_shape=(800, 1344, 3), flip=False, flip_direction='horizontal',
            img_norm_cfg=dict(mean=mean, std=std, to_rgb=True))
dev_imgs = [torch.from_numpy(i).cuda() for i in imgs]                      # resident: kernel time only
for _ in range(3):
    out = prepare_batch(dev_imgs, [meta, meta], 'cuda:0')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    out = prepare_batch(dev_imgs, [meta, meta], 'cuda:0')
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
t = time.time()
for _ in range(n):
    prepare_batch([torch.from_numpy(i) for i in imgs], [meta, meta], 'cuda:0')      # pinned upload + kernel, end to end
torch.cuda.synchronize()
e2e = (time.time() - t) / n * 1e3
t = time.time()
host = prepare_batch(imgs, [meta, meta], 'cpu')
host_ms = (time.time() - t) * 1e3
bytes_out = out.numel() * 4
print(json.dumps(dict(batch=list(out.shape), kernel_ms_per_batch=ms, with_upload_ms=e2e, host_ms=host_ms,
                      write_GBps=bytes_out / ms / 1e6, hbm_frac=bytes_out / ms / 1e6 / 8000.0,
                      equal_to_host=bool(torch.equal(out.cpu(), host)))))
