"""TEST INFRASTRUCTURE.  Builds the part of the reference's native code that compiles here from its own
sources, where they lie under /root/reference, into oracle/_ref/ (git-ignored; it still travels to the GPU
box with the snapshot).  Nothing is copied into the repository.

Buildable: the CPU NMS extension (mmdet/ops/nms/src/nms_ext.cpp + src/cpu/nms_cpu.cpp, no WITH_CUDA).
Unbuildable here (recorded in DESIGN.md): the DCN / pyramid-DCN / focal-loss extensions -- CUDA-only
sources that include THC headers removed from current torch, no CPU branch, no nvcc.

Also buildable: the vendored COCO mask API (cocoapi/pycocotools/common/maskApi.c, one C file) -> maskapi.so, bound
with ctypes by oracle/ref_harness/pycoco_mask.py so that the reference's own pycocotools (coco.py / cocoeval.py, pure
Python) runs here as the checker of lsnet_amd/evaluation (its Cython binding `_mask.pyx` is not built: that would
mean running the reference's build system).

Used by tests/test_oracle.py to pin oracle `orc_nms` against the reference's own greedy NMS, and by
tests/test_evaluation.py / make_golden.py for the evaluation fixtures.
"""
import importlib.util
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REF_NMS = '/root/reference/code/mmdet/ops/nms/src'
SO = os.path.join(OUT, 'nms_ext.so')


def available():
    return os.path.exists(SO)


def build(force=False):
    """Compile the reference NMS extension with g++ directly (no build system of the reference is run)."""
    srcs = [os.path.join(REF_NMS, 'nms_ext.cpp'), os.path.join(REF_NMS, 'cpu', 'nms_cpu.cpp')]
    if not all(os.path.exists(s) for s in srcs):
        return SO if available() else None   # GPU box: only a prebuilt file can exist
    if available() and not force and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return SO
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    inc = [f'-I{p}' for p in ce.include_paths()] + [f'-I{sysconfig.get_paths()["include"]}']
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-DTORCH_EXTENSION_NAME=nms_ext',
           '-DTORCH_API_INCLUDE_EXTENSION_H', f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
           '-w'] + inc + srcs + [f'-L{libdir}', '-ltorch', '-ltorch_cpu', '-lc10', '-ltorch_python',
                                 f'-Wl,-rpath,{libdir}', '-o', SO]
    subprocess.check_call(cmd)
    return SO


REF_MASK = '/root/reference/code/cocoapi/pycocotools/common'
MASK_SO = os.path.join(OUT, 'maskapi.so')


def build_maskapi(force=False):
    """gcc on the reference's maskApi.c where it lies -> oracle/_ref/maskapi.so (plain C ABI, no Python)."""
    src = os.path.join(REF_MASK, 'maskApi.c')
    if not os.path.exists(src):
        return MASK_SO if os.path.exists(MASK_SO) else None
    if os.path.exists(MASK_SO) and not force and os.path.getmtime(MASK_SO) >= os.path.getmtime(src):
        return MASK_SO
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-std=c99', '-fPIC', '-shared', '-w', f'-I{REF_MASK}', src, '-lm', '-o', MASK_SO])
    return MASK_SO


def load():
    """The reference's compiled `nms_ext` module (nms / soft_nms / nms_match), or None."""
    if not available():
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location('nms_ext', SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print(build_maskapi(force='--force' in sys.argv))
