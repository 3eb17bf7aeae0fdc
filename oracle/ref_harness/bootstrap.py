"""Import the reference (mmdet 2.2 + vendored mmcv 0.6.2 under /root/reference/code) on CPU.

TEST INFRASTRUCTURE ONLY -- runs in the build container, never on the GPU box (the reference does
not travel).  Used by oracle/ref_harness/make_golden.py to generate tests/golden/*.npz and by
tests that are skipped when /root/reference is absent.

What it does:
  * never writes bytecode into the read-only reference tree;
  * provides tiny stand-ins for third-party PYTHON packages the reference imports but this image
    lacks (addict, yapf, cv2, terminaltables, torchvision, pycocotools) -- none of them is on the
    hot path;
  * stubs the reference's compiled extension modules; `deform_conv_ext` and
    `sigmoid_focal_loss_ext` are then backed by the CPU oracle (oracle/lsnet_oracle.c) so the
    reference's own Python (LSHead.forward / loss / get_bboxes) runs end to end on CPU;
  * patches the two CUDA-only guards (`if not input.is_cuda: raise`) and the `device='cuda'`
    defaults of PointGenerator.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')

REF_ROOT = '/root/reference/code'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, 'mmdet'))


class _RaisingModule(types.ModuleType):
    """Stand-in for a compiled extension: attribute access yields a function that raises."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)

        def _missing(*a, **k):
            raise RuntimeError(f'{self.__name__}.{name} is not available in the oracle harness')

        return _missing


def _install_third_party_shims():
    # addict.Dict -- attribute dict used by mmcv.Config
    if 'addict' not in sys.modules:
        m = types.ModuleType('addict')

        class Dict(dict):
            def __init__(self, *args, **kwargs):
                super().__init__()
                for a in args:
                    if a:
                        for k, v in dict(a).items():
                            self[k] = self._hook(v)
                for k, v in kwargs.items():
                    self[k] = self._hook(v)

            @classmethod
            def _hook(cls, v):
                if isinstance(v, dict) and not isinstance(v, cls):
                    return cls(v)
                if isinstance(v, (list, tuple)):
                    return type(v)(cls._hook(i) for i in v)
                return v

            def __setattr__(self, k, v):
                self[k] = v

            def __setitem__(self, k, v):
                super().__setitem__(k, self._hook(v))

            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    return self.__missing__(k)

            def __missing__(self, k):
                raise KeyError(k)

            def __deepcopy__(self, memo):
                import copy
                return type(self)({k: copy.deepcopy(v, memo) for k, v in self.items()})

            def to_dict(self):
                out = {}
                for k, v in self.items():
                    out[k] = v.to_dict() if isinstance(v, Dict) else v
                return out

        m.Dict = Dict
        sys.modules['addict'] = m

    for name in ('yapf', 'yapf.yapflib', 'yapf.yapflib.yapf_api'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['yapf.yapflib.yapf_api'].FormatCode = lambda text, **kw: (text, True)

    if 'cv2' not in sys.modules:
        cv2 = _RaisingModule('cv2')
        for i, n in enumerate(['INTER_NEAREST', 'INTER_LINEAR', 'INTER_CUBIC', 'INTER_AREA',
                               'INTER_LANCZOS4', 'IMREAD_COLOR', 'IMREAD_GRAYSCALE',
                               'IMREAD_UNCHANGED', 'BORDER_CONSTANT', 'BORDER_REPLICATE',
                               'BORDER_REFLECT', 'BORDER_REFLECT_101', 'COLOR_BGR2RGB']):
            setattr(cv2, n, i)
        cv2.__version__ = '0.0'
        cv2.CAP_PROP_FRAME_WIDTH = 3
        cv2.CAP_PROP_FRAME_HEIGHT = 4
        cv2.CAP_PROP_FPS = 5
        cv2.CAP_PROP_FRAME_COUNT = 7
        cv2.CAP_PROP_FOURCC = 6
        cv2.CAP_PROP_POS_FRAMES = 1
        cv2.IMWRITE_JPEG_QUALITY = 1
        sys.modules['cv2'] = cv2

    if 'terminaltables' not in sys.modules:
        tt = types.ModuleType('terminaltables')
        tt.AsciiTable = type('AsciiTable', (), {'__init__': lambda self, *a, **k: None})
        sys.modules['terminaltables'] = tt

    if 'shapely' not in sys.modules:   # only LinearRing.is_ccw is used (loading.py:403-404): signed shoelace area > 0
        import numpy as _np
        sh, geo = types.ModuleType('shapely'), types.ModuleType('shapely.geometry')

        class _Ring:
            def __init__(self, pts):
                self.pts = _np.asarray(pts, dtype=_np.float64).reshape(-1, 2)

            @property
            def is_ccw(self):
                x, y = self.pts[:, 0], self.pts[:, 1]
                return float(_np.dot(x, _np.roll(y, -1)) - _np.dot(y, _np.roll(x, -1))) > 0

        class Polygon:
            def __init__(self, pts):
                self.exterior = _Ring(pts)
        geo.Polygon = Polygon
        sh.geometry = geo
        sys.modules['shapely'], sys.modules['shapely.geometry'] = sh, geo

    for name in ('torchvision', 'torchvision.models'):
        if name not in sys.modules:
            sys.modules[name] = _RaisingModule(name)
    _install_pycocotools()
    if 'lvis' not in sys.modules:
        sys.modules['lvis'] = _RaisingModule('lvis')


def _install_pycocotools():
    """The reference vendors the COCO api (code/cocoapi/pycocotools): its pure-Python modules are imported from
    there; its Cython module `_mask` is replaced by the ctypes binding of the reference's own maskApi.c
    (pycoco_mask.py over oracle/_ref/maskapi.so).  Without that library (or the vendored tree) stubs remain."""
    if 'pycocotools' in sys.modules:
        return
    pkg_dir = os.path.join(REF_ROOT, 'cocoapi', 'pycocotools', 'pycocotools')
    from oracle import build_ref
    so = build_ref.build_maskapi() if os.path.isdir(pkg_dir) else None
    if so is None:
        for name in ('pycocotools', 'pycocotools.mask', 'pycocotools.coco', 'pycocotools.cocoeval'):
            sys.modules[name] = _RaisingModule(name)
        sys.modules['pycocotools.coco'].COCO = type('COCO', (), {})
        sys.modules['pycocotools.cocoeval'].COCOeval = type('COCOeval', (), {})
        return
    import numpy as np
    if not hasattr(np, 'float'):          # cocoeval.py:317 uses the alias numpy removed in 1.24
        np.float = float
    pkg = types.ModuleType('pycocotools')
    pkg.__path__ = [pkg_dir]
    sys.modules['pycocotools'] = pkg
    from oracle.ref_harness import pycoco_mask
    sys.modules['pycocotools._mask'] = pycoco_mask
    pkg._mask = pycoco_mask


_EXT_STUBS = [
    'mmcv._ext', 'mmcv._flow_warp_ext',
    'mmdet.ops.dcn.deform_conv_ext', 'mmdet.ops.dcn.deform_pool_ext', 'mmdet.ops.nms.nms_ext',
    'mmdet.ops.roi_align.roi_align_ext', 'mmdet.ops.roi_pool.roi_pool_ext',
    'mmdet.ops.sigmoid_focal_loss.sigmoid_focal_loss_ext',
    'mmdet.ops.masked_conv.masked_conv2d_ext', 'mmdet.ops.carafe.carafe_ext',
    'mmdet.ops.carafe.carafe_naive_ext', 'mmdet.ops.corner_pool.corner_pool_ext',
    'mmdet.ops.utils.compiling_info', 'mmdet.ops.chamfer_2d.chamfer_2d',
]


def _oracle_dcn_ext():
    """A `deform_conv_ext` look-alike whose 8 entry points call the CPU oracle.

    Signatures follow mmdet/ops/dcn/src/deform_conv_ext.cpp:74-224 (note W-before-H order for
    v1/pyramid and H-before-W for the modulated op)."""
    import torch
    sys.path.insert(0, REPO) if REPO not in sys.path else None
    from oracle import oracle_py as orc

    ext = types.ModuleType('mmdet.ops.dcn.deform_conv_ext')

    def _fwd(input, weight, offset, output, bias, mask, kH, kW, dH, pH, dilH, group, dg, sH, sW):
        out = orc.deform_conv_forward(input, weight, bias, offset, mask, stride=dH, pad=pH, dil=dilH,
                                      groups=group, dg=dg, scale_h=sH, scale_w=sW,
                                      out_hw=(output.shape[2], output.shape[3]))
        output.copy_(out)

    def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH,
                            dilW, dilH, group, dg, im2col_step):
        _fwd(input, weight, offset, output, None, None, kH, kW, dH, padH, dilH, group, dg, 1.0, 1.0)
        return 1

    def pyramid_deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW,
                                    padH, dilW, dilH, scaleW, scaleH, group, dg, im2col_step):
        _fwd(input, weight, offset, output, None, None, kH, kW, dH, padH, dilH, group, dg, scaleH, scaleW)
        return 1

    def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kh, kw,
                                      sh, sw, ph, pw, dh, dw, group, dg, with_bias):
        _fwd(input, weight, offset, output, bias if with_bias else None, mask, kh, kw, sh, ph, dh, group,
             dg, 1.0, 1.0)

    def _bwd(input, weight, offset, mask, gout, dH, pH, dilH, group, dg, sH, sW, want):
        return orc.deform_conv_backward(input, weight, offset, mask, gout, stride=dH, pad=pH, dil=dilH,
                                        groups=group, dg=dg, scale_h=sH, scale_w=sW, want=want)

    def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns,
                                   kW, kH, dW, dH, padW, padH, dilW, dilH, group, dg, im2col_step):
        g = _bwd(input, weight, offset, None, gradOutput, dH, padH, dilH, group, dg, 1.0, 1.0,
                 ('gx', 'goff'))
        gradInput.copy_(g['gx']); gradOffset.copy_(g['goff'])
        return 1

    def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW,
                                        dH, padW, padH, dilW, dilH, group, dg, scale, im2col_step):
        g = _bwd(input, gradWeight.new_zeros(gradWeight.shape), offset, None, gradOutput, dH, padH, dilH,
                 group, dg, 1.0, 1.0, ('gw',))
        gradWeight.add_(g['gw'] * scale)
        return 1

    def pyramid_deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight,
                                           columns, kW, kH, dW, dH, padW, padH, dilW, dilH, scaleW,
                                           scaleH, group, dg, im2col_step):
        g = _bwd(input, weight, offset, None, gradOutput, dH, padH, dilH, group, dg, scaleH, scaleW,
                 ('gx', 'goff'))
        gradInput.copy_(g['gx']); gradOffset.copy_(g['goff'])
        return 1

    def pyramid_deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW,
                                                kH, dW, dH, padW, padH, dilW, dilH, scaleW, scaleH,
                                                group, dg, scale, im2col_step):
        g = _bwd(input, gradWeight.new_zeros(gradWeight.shape), offset, None, gradOutput, dH, padH, dilH,
                 group, dg, scaleH, scaleW, ('gw',))
        gradWeight.add_(g['gw'] * scale)
        return 1

    def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input,
                                       grad_weight, grad_bias, grad_offset, grad_mask, grad_output, kh,
                                       kw, sh, sw, ph, pw, dh, dw, group, dg, with_bias):
        g = _bwd(input, weight, offset, mask, grad_output, sh, ph, dh, group, dg, 1.0, 1.0,
                 ('gx', 'goff', 'gmask', 'gw', 'gb'))
        grad_input.copy_(g['gx']); grad_offset.copy_(g['goff']); grad_mask.copy_(g['gmask'])
        grad_weight.add_(g['gw'])
        if with_bias:
            grad_bias.add_(g['gb'])

    for f in (deform_conv_forward, deform_conv_backward_input, deform_conv_backward_parameters,
              pyramid_deform_conv_forward, pyramid_deform_conv_backward_input,
              pyramid_deform_conv_backward_parameters, modulated_deform_conv_forward,
              modulated_deform_conv_backward):
        setattr(ext, f.__name__, f)
    return ext


def _oracle_focal_ext():
    from oracle import oracle_py as orc
    ext = types.ModuleType('mmdet.ops.sigmoid_focal_loss.sigmoid_focal_loss_ext')
    ext.forward = lambda logits, targets, num_classes, gamma, alpha: orc.sigmoid_focal_loss_forward(
        logits, targets, gamma, alpha)
    ext.backward = lambda logits, targets, d_losses, num_classes, gamma, alpha: \
        orc.sigmoid_focal_loss_backward(logits, targets, d_losses, gamma, alpha)
    return ext


def _oracle_nms_ext():
    import torch
    from oracle import oracle_py as orc
    ext = _RaisingModule('mmdet.ops.nms.nms_ext')
    ext.nms = lambda dets, thr: orc.nms(dets, thr)
    return ext


def load_reference():
    """Returns the imported `mmdet` package of the reference (CPU, oracle-backed native ops)."""
    if not reference_available():
        raise RuntimeError('/root/reference is not present (it only exists in the build container)')
    if 'mmdet' in sys.modules and getattr(sys.modules['mmdet'], '__file__', '').startswith(REF_ROOT):
        return sys.modules['mmdet']
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    _install_third_party_shims()
    for name in _EXT_STUBS:
        sys.modules[name] = _RaisingModule(name)
    sys.modules['mmdet.ops.dcn.deform_conv_ext'] = _oracle_dcn_ext()
    sys.modules['mmdet.ops.sigmoid_focal_loss.sigmoid_focal_loss_ext'] = _oracle_focal_ext()
    sys.modules['mmdet.ops.nms.nms_ext'] = _oracle_nms_ext()
    ver = types.ModuleType('mmdet.version')
    ver.__version__ = '2.2.0'
    ver.short_version = '2.2.0'
    sys.modules['mmdet.version'] = ver
    for p in (os.path.join(REF_ROOT, 'mmcv'), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    # torch._six was removed from modern torch; mmcv 0.6.2 imports it in a few places.
    if 'torch._six' not in sys.modules:
        six = types.ModuleType('torch._six')
        import collections.abc as cabc
        six.container_abcs = cabc
        six.string_classes = (str, bytes)
        six.int_classes = (int,)
        six.inf = float('inf')
        sys.modules['torch._six'] = six
    sys.meta_path.insert(0, _PatchedDeformConvFinder())
    mmdet = importlib.import_module('mmdet')
    pg = importlib.import_module('mmdet.core.anchor.point_generator')
    pg.PointGenerator.grid_points.__defaults__ = (16, 'cpu')
    pg.PointGenerator.valid_flags.__defaults__ = ('cpu',)
    return mmdet


class _PatchedDeformConvFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Loads mmdet/ops/dcn/deform_conv.py with its six `if not x.is_cuda: raise` guards
    (deform_conv.py:46,67,136,154,221,242) neutralised, so CPU tensors reach the oracle-backed
    `deform_conv_ext`.  The file is read from the reference tree at import time; nothing is
    copied or written."""
    TARGET = 'mmdet.ops.dcn.deform_conv'
    FILE = os.path.join(REF_ROOT, 'mmdet/ops/dcn/deform_conv.py')

    def find_spec(self, fullname, path, target=None):
        if fullname != self.TARGET:
            return None
        return importlib.util.spec_from_loader(fullname, self, origin=self.FILE)

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        with open(self.FILE) as f:
            src = f.read()
        src = src.replace('not input.is_cuda', 'False').replace('not grad_output.is_cuda', 'False')
        module.__file__ = self.FILE
        exec(compile(src, self.FILE, 'exec'), module.__dict__)
