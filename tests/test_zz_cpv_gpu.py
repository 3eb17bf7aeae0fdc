"""LSCPVHead on the MI355X against the reference fixture (head_cpv.npz): the `feat_channels + 6`-channel pyramid
gathers, corner pooling, corner verification at decode.  (File name sorts last: this widening row runs after every
hot-path test.)"""
import pytest
import torch

from tests import golden_cases as gc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
def test_cpv_head_forward_loss_backward_decode(channels_last):
    assert torch.cuda.is_available()
    from tests import golden_util as gu
    gu.STATS.clear()
    try:
        worst = gc.cpv_head_case(torch.device('cuda:0'), channels_last)
        print('cpv', 'channels_last' if channels_last else 'contiguous', f'worst sample err {worst:.2e}')
    finally:
        print('cpv', 'channels_last' if channels_last else 'contiguous', gu.stats_report())


@pytest.mark.parametrize('channels_last', [False, True], ids=['nchw', 'nhwc'])
def test_cpv_decode_is_exact_on_the_device(channels_last):
    """Round 6: LSCPVHead.get_bboxes against the reference's on bit-identical inputs (decode_cpv.npz): the
    corner-verified candidate boxes, labels, keep order and the max_per_img cut np.array_equal on the MI355X."""
    assert torch.cuda.is_available()
    gc.cpv_decode_case(torch.device('cuda:0'), channels_last)
