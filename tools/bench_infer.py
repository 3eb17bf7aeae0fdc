"""Secondary metric of BASELINE.json (config #5): inference latency of LSNet R-50-FPN with the pose head
(17 keypoints), 1333x800 padded to 800x1344, batch 4, one MI355X; also the bbox head for reference.
Random-init weights, synthetic images resident on the device; decode + NMS included, no host conversion."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsnet_amd import _lib
from lsnet_amd.model_zoo import build_lsnet

dev = torch.device('cuda:0')
B, H, W = 4, 800, 1344
out = {}
for task in ('pose_kbox', 'bbox'):
    torch.manual_seed(0)
    model, cfg = build_lsnet(task, 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    img = torch.randn(B, 3, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    metas = [dict(pad_shape=(H, W, 3), img_shape=(H, W, 3), scale_factor=1.0, ori_shape=(H, W, 3), flip=False)] * B
    for math in ('bf16x3', 'fp32'):
        _lib.set_math_mode(math)
        with torch.no_grad():
            for _ in range(3):
                dets = model.simple_test_batch(img, metas)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                dets = model.simple_test_batch(img, metas)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        out[f'{task}/{math}'] = dict(ms_per_batch=round(dt * 1e3, 2), img_per_s=round(B / dt, 1),
                                     dets_img0=int(dets[0][0].shape[0]))
print(json.dumps(out))
