"""The golden-vector parity cases on the MI355X: LSHead / losses / assigners / decode / NMS with the
native ops going through liblsnet_hip.so (both memory formats)."""
import pytest
import torch

from tests import golden_cases as gc

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
@pytest.mark.parametrize('task', ['bbox', 'segm', 'pose_bbox', 'pose_kbox'])
def test_head_forward_loss_backward_decode(task, channels_last):
    from tests import golden_util as gu
    gu.STATS.clear()
    try:
        gc.head_case(task, _dev(), channels_last)
    finally:
        print(task, 'channels_last' if channels_last else 'contiguous', gu.stats_report())


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
@pytest.mark.parametrize('task', ['bbox', 'segm', 'pose_bbox', 'pose_kbox'])
def test_decode_is_exact_on_the_device(task, channels_last):
    """labels, top-k / NMS keep order and the max_per_img cut of get_bboxes equal the reference's EXACTLY
    (np.array_equal) on the MI355X -- on-device top-k, decode and lsn_nms on bit-identical inputs (fixture decode.npz)."""
    gc.decode_case(task, _dev(), channels_last)


def test_assigners_exact():
    gc.assign_case(_dev())


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
def test_head_at_256_channels(channels_last):
    """The bbox head at the width every LSNet config uses (256 channels, GN32) against the reference's own head run at
    that width (fixture head_bbox_256.npz, oracle/ref_harness/make_golden.py head_bbox_256): forward, losses, feature
    and parameter gradients, decode -- and the split kernels against exact fp32 on the same device."""
    gc.head_case('bbox', _dev(), channels_last, channels=256)


def test_cross_iou_loss():
    gc.cross_iou_case(_dev())


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
def test_backbone_fpn(channels_last):
    gc.backbone_case(_dev(), channels_last)


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
def test_res2net_dcn_backbone(channels_last):
    gc.res2net_case(_dev(), channels_last)


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
@pytest.mark.parametrize('name', ['r101-dcn', 'x101-dcn'])
def test_dcn_backbones_of_configs_3_and_4(name, channels_last):
    from tests import golden_util as gu
    gu.STATS.clear()
    try:
        # measured (profiles/r2_gpu_tests.log): R-101-DCN 4.9e-6 of the range in every sampled element; X-101-64x4d-DCN
        # 1.8e-3 (0.3 % of the input-gradient samples beyond 1e-3: ReLU kinks of a 101-layer grouped network)
        gc.backbone_dcn_case(name, _dev(), channels_last, grad_rtol=1e-4 if name == 'r101-dcn' else 3e-3, outliers=0.0)
    finally:
        print(name, 'channels_last' if channels_last else 'contiguous', gu.stats_report())


def test_multiclass_nms_lsvr():
    gc.nms_lsvr_case(_dev())


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x3', 'fp32'])
def test_training_curve_follows_reference_runner(math):
    """20 SGD iterations (round 5; 12 until then) on the device against the curve of the reference's detector + mmcv runner on
    CPU (a 20-iteration curve with a 10-iteration linear warm-up to lr 0.01, steps at 14 and 18).  History of the
    12-iteration fixture:  Measured on the MI355X, per iteration (profiles/r2_gpu_tests.log): iterations 1-6 (warm-up, lr still
    small) <= 1.6e-4 (bf16x6), 2.5e-4 (bf16x3), 9.5e-5 (exact fp32 MFMA) relative; iterations 7-12 up to 2.9e-2 / 6.3e-3 /
    3.3e-2 in that run and 9e-2 in an earlier one -- the trajectory amplifies rounding-level differences once the
    learning rate is up, the exact-fp32 kernels no less than the split ones, and fp32 atomics in the weight gradients make
    the late values vary from run to run.  Tolerances = 3 x the worst measured: early 7.5e-4 for the fp32-equivalent
    and exact modes and 5e-3 for the 3-product mode (1.4e-3 at iteration 3 in one run).  Round 3 (new dense-conv kernels, frozen stage-1
    BatchNorm folded into its convolutions, fused SGD; profiles/r3_gpu_tests.log): iterations 1-6 <= 2.0e-4 / 6.9e-4 / 1.3e-4,
    iterations 7-12 <= 2.8e-2 / 1.2e-2 / 4.7e-2 in one run, 3.4e-2 / 3.9e-2 / 3.2e-1 (loss_cls of the exact mode at
    iteration 8, where the loss has fallen to 1.25) in another with different deformable kernels: the late half of THIS
    fixture measures chaos, not kernels, and only has to stay within a factor of two (tolerance 1.0; 0.5 left the 0.32 of
    that run a margin of 1.5 in a suite the driver runs with -x).  What holds all twelve iterations tight is
    `test_training_curve_low_learning_rate` below.
    Round 4: the fixture was regenerated with the tie-free parameter fill (golden_util.fill_params) and is no longer the
    475 -> 1 collapse: the loss goes 2.41 -> 1.29 and the trajectory stays tame.  Measured (profiles/r4_gpu_tests.log,
    fp32-equivalent mode): total loss <= 9.1e-5 over iterations 1 - 6 and <= 6.7e-3 over 7 - 12; the refine term (where one
    re-assigned point shows) 9.3e-4 / 1.8e-2.  Tolerances: early 3e-3 (5e-3 in the 3-product mode), late 0.1 (round 3: 1.0)."""
    from lsnet_amd import _lib
    before = _lib.get_math_mode()
    _lib.set_math_mode(math)
    try:
        # Round 5, the 20-iteration fixture (442 -> 3.4 by iteration 9, then a plateau at 3.4 - 4.7; profiles/r5_gpu_tests.log):
        # iterations 1 - 8 <= 4.0e-4 (fp32-equivalent), 9.5e-3 (3-product), 5.2e-4 (exact); iterations 9 - 20 <= 1.1e-1 / 5.3e-2 /
        # 3.6e-2 -- the collapse amplifies rounding-level differences, the exact kernels' no less than the split ones'.  Early
        # tolerance 3e-3 for the default mode, 3e-2 for the 3-product mode and for the exact mode (9.9e-3 at iteration 8 in a second run:
        # its deformable kernels scatter with fp32 atomics and vary from run to run).  Late: after the collapse the loss hops between 1 and 3
        # from one iteration to the next and the exact mode is not reproducible: 0.31 at iteration 20 in one run of five
        # (loss_cls 1.70 against 2.48) -- these iterations only have to stay within a factor of two, as in round 3 (late 1.0 for the
        # exact and 3-product modes; the default mode is bit-reproducible and keeps 0.3).  What holds all twenty iterations tight is
        # the low-lr test.
        # Round 6: the exact mode no longer scatters with atomics (bit-reproducible like the default mode) and takes the default
        # mode's LATE bound (measured 6.0e-2, profiles/r6_gpu_tests.log); its early bound stays 3e-2 -- 1.0e-2 at iteration 8 of
        # this collapsing fixture, where any other summation order lands as far from the reference as from this one.  Only the
        # 3-product mode keeps "within a factor of two".  What is held tight in BOTH modes is the benchmark-model curve below.
        worst = gc.train_curve_case(_dev(), early_tol=3e-3 if math == 'bf16x6' else 3e-2, late_tol=1.0 if math == 'bf16x3' else 0.3,
                                    rtol_weight=5e-2, channels_last=True)
    finally:
        _lib.set_math_mode(before)
    print(math, f'worst relative loss deviation {worst:.2e}')


@pytest.mark.parametrize('math', ['bf16x6', 'fp32'])
def test_training_curve_low_learning_rate(math):
    """The same SGD iterations (detector, runner, clip 35, warm-up + step schedule) at a tenth of the learning rate against the
    reference's run of that schedule (fixture train_curve_lowlr.npz, oracle/ref_harness/make_golden.py train_curve_lowlr).
    Round 5: TWENTY iterations, the loss falling 442 -> 3.5, ALL of them compared; measured on the MI355X in the
    fp32-equivalent mode (profiles/r5_gpu_tests.log): total loss <= 2.3e-4 over iterations 1 - 18 and 1.0e-3 at 19 - 20, classification
    <= 3.1e-4 / 3.8e-3, regression terms <= 2.2e-3 / 6.8e-3.  History of the 12-iteration fixture: the loss fell 480 -> 31 without the
    collapse that makes the other fixture chaotic, so EVERY
    of the first eleven iterations was held tight in the fp32-equivalent and exact modes ("remove the chaos"): total loss and
    classification loss to 1e-3 relative, the two regression terms (2 % of the total; one re-assigned point moves them by
    up to 8e-3 -- measured between this package's host path and the reference on the SAME CPU: 2.8e-4 / 4.7e-5 / 5.0e-4 /
    8.0e-3) to 2e-2.  Measured on the MI355X (profiles/r3_gpu_tests.log): total / cls <= 6.1e-5, regression terms <=
    3.6e-4 / 3.2e-3 in both modes over iterations 1 - 11.  The twelfth iteration (the loss drops 77 -> 31 there) is left to
    the other fixture: it flipped by 1.4e-2 in one exact-mode run of three (profiles/r3_gpu_tests.log)."""
    from lsnet_amd import _lib
    before = _lib.get_math_mode()
    _lib.set_math_mode(math)
    try:
        tol = dict(loss=1e-3, loss_cls=1e-3, loss_bbox_init=2e-2, loss_bbox_refine=2e-2)
        # exact mode: the round-1 deformable kernels behind it accumulate with fp32 atomics, so a point re-assigned in one
        # run and not in the next is possible in the second half (seen once, at iteration 12); the default mode's step is
        # bit-reproducible and keeps the tight bound throughout
        # round 5: ALL twenty iterations (the loss falls 442 -> 3.5).  Iterations 1 - 8 to `tol`; 9 - 20: 5e-3 for the total and
        # the classification loss in the fp32-equivalent mode (measured 1.0e-3), 3e-2 for the regression terms and for the exact
        # mode (measured 1.6e-2: its round-1 deformable kernels scatter with fp32 atomics)
        # (round 6: the exact mode is bit-reproducible now -- atomic-free gather, ordered weight-gradient partials -- and is held
        # to the default mode's bounds; rounds 3 - 5 gave it 3e-2 / 5e-2 because its atomics re-assigned a point in some runs)
        late = dict(tol, loss=5e-3, loss_cls=5e-3, loss_bbox_init=3e-2, loss_bbox_refine=3e-2)
        worst = gc.train_curve_case(_dev(), early_tol=tol, late_tol=late, rtol_weight=1e-2, channels_last=True,
                                    fixture='train_curve_lowlr', lr=0.001, iters=LOWLR_ITERS)
    finally:
        _lib.set_math_mode(before)
    print(math, f'low-lr curve: worst relative loss deviation {worst:.2e}')


LOWLR_ITERS = 20


@pytest.mark.parametrize('math', ['bf16x6', 'fp32'])
def test_training_curve_of_the_benchmark_model(math):
    """Round 6: the curve the reference's schedule_1x.py:5-10 actually produces -- the UNTOUCHED
    seed-0 init_weights model (the model `python bench.py` trains), 2 images 3 x 384 x 512 per iteration, lr 0.01 behind a
    10-iteration linear warm-up, clip 35, SGD momentum 0.9 -- against the reference's detector + mmcv runner on the same weights
    and batches (fixture train_curve_init0.npz, make_golden.py train_curve_init0): the loss goes 5.02 -> 3.22 smoothly, no
    collapse, so ALL twenty iterations are held to 5e-3 (total, classification and init loss; 1e-2 for the refine term, where
    one re-assigned point shows) in both modes -- the 0.3 / 1.0 late tolerances of the parameter-fill fixture above
    are what a 442 -> 3.4 collapse needs, not what the kernels need."""
    from lsnet_amd import _lib
    before = _lib.get_math_mode()
    _lib.set_math_mode(math)
    try:
        # measured (profiles/r6_gpu_tests.log): total 1.3e-3, classification 4.8e-4, init 5.2e-5, refine 2.7e-3 (both modes alike)
        tol = dict(loss=5e-3, loss_cls=5e-3, loss_bbox_init=5e-3, loss_bbox_refine=1e-2)
        worst = gc.train_curve_case(_dev(), early_tol=tol, late_tol=tol, rtol_weight=2e-2, channels_last=True,
                                    fixture='train_curve_init0', init0=True)
    finally:
        _lib.set_math_mode(before)
    print(math, f'benchmark-model curve: worst relative loss deviation over twenty iterations {worst:.2e}')


def test_iteration0_losses_at_the_benchmark_shape():
    """SURVEY 8(d) "Loss parity": the untouched seed-0 model on the bench batch (2 x 3 x 800 x 1344, seed 1234) -- what
    `python bench.py` starts from -- against the REFERENCE's detector on the same weights and batch (fixture
    bench_iter0.npz: one CPU forward in the build container, make_golden.py::golden_bench_iter0): the loss triplet and the
    per-level terms within 1e-3."""
    import numpy as np
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    ref = gc.load('bench_iter0')
    dev = _dev()
    torch.manual_seed(0)
    model, _ = build_lsnet('bbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
    with torch.no_grad():
        losses = model(**data)
        _, lv = model._parse_losses(losses)
    for k in ('loss_cls', 'loss_bbox_init', 'loss_bbox_refine', 'loss'):
        assert abs(float(lv[k]) - float(ref[k])) <= 1e-3 * abs(float(ref[k])), (k, float(lv[k]), float(ref[k]))
        got = np.array([float(v) for v in losses[k]]) if k != 'loss' else None
        if got is not None:
            assert np.allclose(got, ref[f'per_level/{k}'], rtol=1e-3, atol=1e-6), (k, got, ref[f'per_level/{k}'])


def test_instances_vote_on_the_device():
    """(f-3) the per-class instance vote of multi-scale testing on device tensors against the reference's
    (lsnet.py:229-299, fixture vote.npz)."""
    import numpy as np
    from lsnet_amd.core.vote import instances_vote
    ref = gc.load('vote')
    case = 0
    while f'{case}/boxes' in ref.files:
        b, v, s = (torch.from_numpy(ref[f'{case}/{k}']).to(_dev()) for k in ('boxes', 'vectors', 'scores'))
        ob, ov, os_ = instances_vote(b, v, s)
        assert ob.is_cuda and ob.shape[0] == ref[f'{case}/out_boxes'].shape[0], case
        assert np.allclose(ob.cpu().numpy(), ref[f'{case}/out_boxes'], rtol=1e-5, atol=1e-3), case
        assert np.allclose(ov.cpu().numpy(), ref[f'{case}/out_vectors'], rtol=1e-5, atol=1e-3), case
        assert np.allclose(os_.cpu().numpy(), ref[f'{case}/out_scores'], rtol=1e-5, atol=1e-6), case
        case += 1
    assert case == 4


def test_aug_test_vote_on_the_device():
    """(f-3) LSDetector.aug_test (lsnet.py:301-401) on the MI355X: two scales x {plain, flipped} of one image through the
    HIP forward, mapped back and merged by the instance vote; per-class results of the reference's shapes, and the
    single-view subset of the multi-view result equals what simple_test gives for that view."""
    import numpy as np
    from lsnet_amd.model_zoo import build_lsnet
    dev = _dev()
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    model.test_cfg = cfg.test_cfg
    model.test_cfg.update(method='vote', scale_ranges=[[0, 10000], [0, 10000]], score_thr=0.0, nms_pre=50, max_per_img=50)
    model.bbox_head.test_cfg = model.test_cfg
    base = torch.randn(1, 3, 288, 352, device=dev)
    imgs, metas = [], []
    for s in (1.0, 1.5):
        h, w = int(288 * s) // 32 * 32, int(352 * s) // 32 * 32
        view = torch.nn.functional.interpolate(base, size=(h, w), mode='bilinear', align_corners=False)
        for flip in (False, True):
            imgs.append((view.flip(-1) if flip else view).contiguous(memory_format=torch.channels_last))
            metas.append([dict(img_shape=(h, w, 3), pad_shape=(h, w, 3), ori_shape=(288, 352, 3), flip=flip,
                               scale_factor=np.array([w / 352, h / 288, w / 352, h / 288], dtype=np.float32))])
    with torch.no_grad():
        boxes, vectors = model(imgs, metas, return_loss=False, rescale=True)
    assert len(boxes) == len(vectors) == 80
    n = sum(b.shape[0] for b in boxes)
    assert n > 0 and all(b.shape[1] == 5 for b in boxes) and all(v.shape[1] == 8 for v in vectors)
    allb = np.concatenate([np.asarray(b) for b in boxes])
    # (random weights: a "box" of the untrained head may have its corners in any order; what is checked is that every
    # view's detections came back in the ORIGINAL image's frame -- inside 352 x 288 after rescaling and un-flipping)
    assert np.isfinite(allb).all()
    assert allb[:, :4].min() >= -1e-2 and allb[:, [0, 2]].max() <= 352 + 1e-2 and allb[:, [1, 3]].max() <= 288 + 1e-2
