"""ctypes binding of liblsnet_hip.so (C ABI declared in include/lsnet_hip.h).

The library is built in-tree by lsnet_amd/csrc/build.py.  If it is missing or fails to load the
ops FAIL LOUDLY -- there is no CPU or eager fallback in the product path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get('LSNET_HIP_SO') or os.path.join(_HERE, 'csrc', 'liblsnet_hip.so')   # env: diagnostic builds

c_float_p = ctypes.POINTER(ctypes.c_float)
c_i64_p = ctypes.POINTER(ctypes.c_int64)


class Strides4(ctypes.Structure):
    _fields_ = [('b', ctypes.c_int64), ('c', ctypes.c_int64), ('h', ctypes.c_int64), ('w', ctypes.c_int64)]


class DcnShape(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int), ('C', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int),
                ('Co', ctypes.c_int), ('Ho', ctypes.c_int), ('Wo', ctypes.c_int),
                ('kh', ctypes.c_int), ('kw', ctypes.c_int), ('stride', ctypes.c_int), ('pad', ctypes.c_int),
                ('dil', ctypes.c_int), ('groups', ctypes.c_int), ('deformable_groups', ctypes.c_int),
                ('scale_h', ctypes.c_float), ('scale_w', ctypes.c_float), ('mask_is_logit', ctypes.c_int),
                ('workspace', ctypes.c_void_p), ('gather_workspace', ctypes.c_void_p),
                ('gather_workspace_bytes', ctypes.c_int64), ('accumulate_param_grads', ctypes.c_int),
                ('out_pitch', ctypes.c_int), ('weights_prepared', ctypes.c_int)]


class DcnLevel(ctypes.Structure):
    _fields_ = [('input', ctypes.c_void_p), ('offset', ctypes.c_void_p), ('mask', ctypes.c_void_p),
                ('output', ctypes.c_void_p), ('grad_output', ctypes.c_void_p), ('grad_input', ctypes.c_void_p),
                ('grad_offset', ctypes.c_void_p), ('grad_mask', ctypes.c_void_p),
                ('off_st', Strides4), ('mask_st', Strides4),
                ('B', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('Ho', ctypes.c_int),
                ('Wo', ctypes.c_int), ('scale_h', ctypes.c_float), ('scale_w', ctypes.c_float)]


class OffsetChainLevel(ctypes.Structure):
    _fields_ = [('off', ctypes.c_void_p), ('out', ctypes.c_void_p * 3), ('gout', ctypes.c_void_p * 6), ('goff', ctypes.c_void_p),
                ('images', ctypes.c_int64), ('per_image', ctypes.c_int64), ('off_image_pitch', ctypes.c_int64),
                ('mh', ctypes.c_float * 3), ('mw', ctypes.c_float * 3)]


class WgradBnJob(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('g', ctypes.c_void_p), ('w', ctypes.c_void_p), ('bn_gamma', ctypes.c_void_p),
                ('bn_mean', ctypes.c_void_p), ('bn_var', ctypes.c_void_p), ('bn_eps', ctypes.c_float),
                ('grad_w', ctypes.c_void_p), ('grad_gamma', ctypes.c_void_p), ('grad_beta', ctypes.c_void_p)]


class SgdTensor(ctypes.Structure):
    _fields_ = [('param', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('momentum_buf', ctypes.c_void_p),
                ('numel', ctypes.c_int64), ('first_chunk', ctypes.c_int64), ('group', ctypes.c_int)]


class SgdGroup(ctypes.Structure):
    _fields_ = [('lr', ctypes.c_float), ('momentum', ctypes.c_float), ('weight_decay', ctypes.c_float)]


class ConvLevel(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('out', ctypes.c_void_p), ('grad_out', ctypes.c_void_p),
                ('B', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('residual', ctypes.c_void_p),
                ('gate', ctypes.c_void_p)]


class ConvWprep(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int), ('w', ctypes.c_void_p), ('prepared', ctypes.c_void_p), ('C', ctypes.c_int),
                ('Co', ctypes.c_int), ('kh', ctypes.c_int), ('kw', ctypes.c_int), ('stride', ctypes.c_int),
                ('pad', ctypes.c_int), ('dil', ctypes.c_int),
                ('bn_gamma', ctypes.c_void_p), ('bn_var', ctypes.c_void_p), ('bn_beta', ctypes.c_void_p),
                ('bn_mean', ctypes.c_void_p), ('shift_out', ctypes.c_void_p), ('bn_eps', ctypes.c_float)]


class GnLevel(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('y', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dx', ctypes.c_void_p),
                ('B', ctypes.c_int), ('HW', ctypes.c_int), ('y_batch_stride', ctypes.c_longlong),
                ('dy_batch_stride', ctypes.c_longlong)]


class GateJob(ctypes.Structure):     # lsn_gate_job
    _fields_ = [('grad_y', ctypes.c_void_p), ('y', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('B', ctypes.c_int),
                ('per_image', ctypes.c_int64), ('gy_batch_stride', ctypes.c_int64)]


class ProfEntry(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 48), ('launches', ctypes.c_longlong), ('total_ms', ctypes.c_double),
                ('flops', ctypes.c_double), ('bytes', ctypes.c_double)]


class ProfLaunch(ctypes.Structure):     # lsn_prof_launch
    _fields_ = [(k, ctypes.c_int) for k in ('kind', 'C', 'Co', 'kh', 'kw', 'stride', 'pad', 'dil', 'relu', 'xpitch', 'n_levels',
                                            'has_residual', 'has_gate')] + \
               [('B', ctypes.c_int * 16), ('H', ctypes.c_int * 16), ('W', ctypes.c_int * 16), ('ms', ctypes.c_float)]


# every symbol include/lsnet_hip.h declares (checked by tests/test_capi.py without a GPU)
EXPORTS = [
    'lsn_last_error', 'lsn_version', 'lsn_dcn_forward', 'lsn_dcn_backward', 'lsn_dcn_backward_workspace_bytes', 'lsn_dcn_pitched_ok', 'lsn_dcn_prepared_ok',
    'lsn_deform_conv_forward', 'lsn_deform_conv_backward_input', 'lsn_deform_conv_backward_parameters',
    'lsn_modulated_deform_conv_forward', 'lsn_modulated_deform_conv_backward',
    'lsn_pyramid_deform_conv_forward', 'lsn_pyramid_deform_conv_backward_input',
    'lsn_pyramid_deform_conv_backward_parameters',
    'lsn_sigmoid_focal_loss_forward', 'lsn_sigmoid_focal_loss_backward', 'lsn_sigmoid_focal_loss_sum',
    'lsn_sigmoid_focal_loss_backward_weighted', 'lsn_sigmoid_focal_loss_level_sums', 'lsn_sigmoid_focal_loss_backward_levels',
    'lsn_level_sums', 'lsn_level_expand',
    'lsn_nms_workspace_bytes', 'lsn_nms', 'lsn_topk_columns', 'lsn_offset_chain_forward', 'lsn_offset_chain_backward', 'lsn_clip_sgd_workspace_bytes', 'lsn_clip_sgd_step', 'lsn_selftest_mfma', 'lsn_debug_phase_clocks',
    'lsn_prof_enable', 'lsn_prof_read', 'lsn_prof_launch_log', 'lsn_prof_read_launches', 'lsn_scratch_stats', 'lsn_wgrad_defer', 'lsn_wgrad_flush',
    'lsn_group_norm_workspace_bytes', 'lsn_group_norm_forward', 'lsn_group_norm_backward',
    'lsn_set_math_mode', 'lsn_get_math_mode', 'lsn_conv2d_forward', 'lsn_conv2d_forward_pitched', 'lsn_conv2d_backward_data',
    'lsn_conv2d_prepared_bytes', 'lsn_conv2d_prepare_weights', 'lsn_conv2d_prepare_weights_multi',
    'lsn_conv2d_prepare_weights_item',
    'lsn_conv2d_forward_prepared',
    'lsn_conv2d_backward_data_prepared',
    'lsn_conv2d_forward_multi', 'lsn_conv2d_backward_data_multi', 'lsn_conv2d_backward_weight_multi', 'lsn_conv2d_backward_weight',
    'lsn_grouped_conv2d_forward', 'lsn_grouped_conv2d_backward_data', 'lsn_grouped_conv2d_backward_weight',
    'lsn_bn_eval_act_forward', 'lsn_bn_eval_act_backward', 'lsn_bn_eval_act_workspace_bytes',
    'lsn_relu_gate', 'lsn_relu_gate_multi', 'lsn_conv2d_backward_weight_bn', 'lsn_conv2d_backward_weight_bn_jobs',
    'lsn_image_prep_u8', 'lsn_cross_iou_bbox_forward', 'lsn_cross_iou_bbox_backward',
    'lsn_cross_iou_bbox_stage_forward', 'lsn_cross_iou_bbox_stage_backward',
    'lsn_cross_iou_rows_forward', 'lsn_cross_iou_rows_backward',
]

_lib = None


def load():
    """Returns the loaded library; raises RuntimeError when it is absent (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f'{SO_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). lsnet_amd has no fallback path.')
    lib = ctypes.CDLL(SO_PATH)
    lib.lsn_last_error.restype = ctypes.c_char_p
    lib.lsn_nms_workspace_bytes.restype = ctypes.c_int64
    lib.lsn_group_norm_workspace_bytes.restype = ctypes.c_int64
    lib.lsn_bn_eval_act_workspace_bytes.restype = ctypes.c_int64
    lib.lsn_dcn_backward_workspace_bytes.restype = ctypes.c_int64
    lib.lsn_clip_sgd_workspace_bytes.restype = ctypes.c_int64
    lib.lsn_conv2d_prepared_bytes.restype = ctypes.c_int64
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here means header and library disagree
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().lsn_last_error().decode('utf-8', 'replace')
        if rc == -2:
            raise NotImplementedError(msg)
        raise RuntimeError(msg)


PROF_FAMILIES = ('dcn_fwd', 'dcn_bwd_data', 'dcn_wgrad', 'conv_fwd', 'conv_bwd_data', 'conv_wgrad', 'norm', 'gconv')


def prof_enable(on, families=None):
    """Start (and clear) / stop the library's per-kernel event log (lsn_prof_enable); `families`: names of the only
    families to record (default: all)."""
    mask = 0
    if on:
        mask = 1 if families is None else sum(1 << PROF_FAMILIES.index(f) for f in families)
        if mask == 1 and families is not None:
            mask |= 1 << 31    # (the value 1 means "all": keep a single-family mask of bit 0 distinct)
    check(load().lsn_prof_enable(mask))


def prof_read():
    """{family: dict(launches, total_ms, flops, bytes)} of the launches logged since prof_enable(True)."""
    arr = (ProfEntry * 16)()
    n = load().lsn_prof_read(arr, 16)
    if n < 0:
        check(n)
    return {arr[i].name.decode(): dict(launches=int(arr[i].launches), total_ms=arr[i].total_ms,
                                       flops=arr[i].flops, bytes=arr[i].bytes) for i in range(n)}


def wgrad_defer(max_mbytes, stream=None):
    """lsn_wgrad_defer on `stream` (default: torch's current stream): max_mbytes > 0 starts deferring the reduces of accumulating
    weight-gradient calls, 0 flushes and ends the mode."""
    import ctypes
    import torch
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream if stream is None else stream)
    check(load().lsn_wgrad_defer(int(max_mbytes), st))


def wgrad_flush(stream=None):
    import ctypes
    import torch
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream if stream is None else stream)
    check(load().lsn_wgrad_flush(st))


def scratch_stats():
    """What the library has asked of the HIP runtime outside launches (lsn_scratch_stats): dict(mallocs, held_bytes,
    blocking_syncs, pool_allocs)."""
    import ctypes
    out = (ctypes.c_longlong * 4)()
    check(load().lsn_scratch_stats(out))
    return dict(mallocs=int(out[0]), held_bytes=int(out[1]), blocking_syncs=int(out[2]), pool_allocs=int(out[3]))


MATH_FP32, MATH_BF16X3, MATH_BF16X6 = 0, 1, 2
_MODES = {'fp32': MATH_FP32, 'bf16x3': MATH_BF16X3, 'bf16x6': MATH_BF16X6}


def set_math_mode(mode):
    """'bf16x6' (fp32-equivalent split products, the default), 'bf16x3' or 'fp32' (exact fp32 MFMA);
    see include/lsnet_hip.h."""
    check(load().lsn_set_math_mode(_MODES[mode] if isinstance(mode, str) else mode))


def get_math_mode():
    m = load().lsn_get_math_mode()
    return {v: k for k, v in _MODES.items()}[m]


def split_math():
    """True when the contractions run as split-bf16 products on the matrix pipe (either split mode)."""
    return get_math_mode() != 'fp32'
