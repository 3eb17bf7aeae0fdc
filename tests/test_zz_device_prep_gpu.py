"""Device-side image preparation (csrc/image.hip, lsn_image_prep_u8) against the host pipeline: the float batch made on
the MI355X from uploaded 8-bit images equals, bit for bit, what Resize / RandomFlip / Normalize / Pad / collate make on
the CPU (the host library, itself pinned to the numpy statement in tests/test_data_pipeline.py)."""
import numpy as np
import pytest
import torch

from lsnet_amd.data.device_prep import prepare_batch

pytestmark = pytest.mark.gpu
NORM = dict(mean=np.array([123.675, 116.28, 103.53], np.float32), std=np.array([58.395, 57.12, 57.375], np.float32))


def _meta(size_wh, pad_hw, flip=False, direction='horizontal', to_rgb=True):
    w, h = size_wh
    return dict(img_shape=(h, w, 3), pad_shape=pad_hw + (3,), flip=flip, flip_direction=direction,
                img_norm_cfg=dict(mean=NORM['mean'], std=NORM['std'], to_rgb=to_rgb))


def _pad32(v):
    return int(np.ceil(v / 32)) * 32


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_device_batch_equals_host_batch(seed):
    assert torch.cuda.is_available()
    rng = np.random.RandomState(seed)
    images, metas = [], []
    for i in range(6):
        sh, sw = int(rng.randint(17, 300)), int(rng.randint(17, 300))
        dh, dw = (sh, sw) if i == 0 else (int(rng.randint(9, 500)), int(rng.randint(9, 500)))     # first: no resize
        images.append(rng.randint(0, 256, (sh, sw, 3)).astype(np.uint8))
        metas.append(_meta((dw, dh), (_pad32(dh), _pad32(dw)), flip=bool(i % 3), direction=['horizontal', 'vertical'][i % 2],
                           to_rgb=bool(i % 2)))
    host = prepare_batch(images, metas, 'cpu')
    dev = prepare_batch([torch.from_numpy(im) for im in images], metas, 'cuda:0')
    torch.cuda.synchronize()
    assert dev.is_cuda and dev.shape == host.shape and dev.is_contiguous(memory_format=torch.channels_last)
    same = torch.equal(dev.cpu(), host)
    if not same:
        diff = (dev.cpu() - host).abs()
        bad = [int((diff[b] > 0).sum()) for b in range(len(images))]
        raise AssertionError(f'device and host batches differ: elements per image {bad}, worst {float(diff.max())}')


def test_training_size_image_and_errors():
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    meta = _meta((1067, 800), (800, 1088), flip=True)
    host = prepare_batch([img, img], [meta, _meta((1067, 800), (800, 1088))], 'cpu')
    dev = prepare_batch([img, img], [meta, _meta((1067, 800), (800, 1088))], 'cuda:0')
    assert torch.equal(dev.cpu(), host) and (dev[:, :, :, 1067:] == 0).all()
    from lsnet_amd import _lib
    import ctypes
    z = torch.zeros(4, device='cuda:0')
    rc = _lib.load().lsn_image_prep_u8(ctypes.c_void_p(z.data_ptr()), 4, 4, 3, 8, 8, 0, 0, None, None, 0, ctypes.c_float(0),
                                       ctypes.c_void_p(z.data_ptr()), 8, 8, None)
    assert rc != 0 and b'null' in _lib.load().lsn_last_error()
