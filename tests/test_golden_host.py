"""Host-side parity against the reference's golden vectors, on CPU with the oracle registered as
the 'cpu' backend of the native ops (test infrastructure).  The same cases run on the MI355X
through the HIP path in tests/test_golden_gpu.py."""
import pytest
import torch

from tests import golden_cases as gc

CPU = torch.device('cpu')


@pytest.mark.parametrize('task', ['bbox', 'segm', 'pose_bbox', 'pose_kbox'])
def test_head_forward_loss_backward_decode(task, cpu_oracle_backend):
    gc.head_case(task, CPU)


@pytest.mark.parametrize('task', ['bbox', 'segm', 'pose_bbox', 'pose_kbox'])
def test_decode_is_exact(task, cpu_oracle_backend):
    gc.decode_case(task, CPU)


def test_cpv_decode_is_exact(cpu_oracle_backend):
    gc.cpv_decode_case(CPU)


def test_assigners_exact():
    gc.assign_case(CPU)


def test_cross_iou_loss():
    gc.cross_iou_case(CPU)


def test_backbone_fpn():
    gc.backbone_case(CPU)


def test_res2net_dcn_backbone(cpu_oracle_backend):
    gc.res2net_case(CPU)


@pytest.mark.parametrize('name', ['r101-dcn', 'x101-dcn'])
def test_dcn_backbones_of_configs_3_and_4(name, cpu_oracle_backend):
    gc.backbone_dcn_case(name, CPU)


def test_multiclass_nms_lsvr(cpu_oracle_backend):
    gc.nms_lsvr_case(CPU)


def test_cpv_head_forward_loss_backward_decode(cpu_oracle_backend):
    gc.cpv_head_case(CPU)


def test_training_curve_follows_reference_runner(cpu_oracle_backend):
    torch.set_num_threads(8)
    # (measured against the reference on this CPU: <= 3.1e-5 over iterations 1 - 7, 5.4e-4 at iteration 8, where the loss falls
    # 31 -> 8; the MI355X test runs all 20 iterations)
    worst = gc.train_curve_case(CPU, early_tol=2e-3, late_tol=0.15, rtol_weight=5e-2, iters=8)
    print(f'worst relative loss deviation over 8 iterations: {worst:.2e}')


def test_training_curve_low_learning_rate(cpu_oracle_backend):
    """A prefix of the non-chaotic fixture (train_curve_lowlr.npz) on the host; the MI355X test runs all twelve iterations."""
    torch.set_num_threads(8)
    worst = gc.train_curve_case(CPU, early_tol=1e-4, late_tol=1e-4, rtol_weight=5e-2, iters=5, fixture='train_curve_lowlr',
                                lr=0.001)
    print(f'low-lr curve, worst relative loss deviation over 5 iterations: {worst:.2e}')


def test_training_curve_of_the_benchmark_model(cpu_oracle_backend):
    """A prefix of fixture train_curve_init0.npz (the seed-0 init_weights model, 2 x 3 x 384 x 512) on the host; the MI355X test
    runs all twenty iterations."""
    torch.set_num_threads(8)
    worst = gc.train_curve_case(CPU, early_tol=1e-4, late_tol=1e-4, rtol_weight=5e-2, iters=1, fixture='train_curve_init0', init0=True)
    print(f'benchmark-model curve, worst relative loss deviation of the first iteration: {worst:.2e}')
