"""Builds liblsnet_hip.so (gfx950) in-tree with hipcc, and liblsnet_host.so (host-side evaluation helpers, g++).
No torch dependency: plain HIP / C++ behind C ABIs (include/lsnet_hip.h, include/lsnet_host.h)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'liblsnet_hip.so')
SOURCES = ['dcn.hip', 'misc.hip', 'norm.hip', 'conv.hip', 'gconv.hip', 'image.hip', 'loss.hip']
HEADERS = ['common.h', 'dcn_kernels.h', 'conv_kernels.h', 'conv_wgrad_kernels.h', 'dcn_mm_kernels.h', 'prof.h', 'cross_iou_row.h', os.path.join('..', '..', 'include', 'lsnet_hip.h')]
# -fno-slp-vectorize: hipcc (ROCm 7.2) packs adjacent scalar fp32 adds / fmas into v_pk_add_f32 / v_pk_fma_f32.  In the
# backward-data kernels the HIGH dword of such packed accumulators came back wrong for the last 16 lanes of a wave in
# 0.7 % of the cases, differently on every run, whenever two workgroups shared a CU (tools/dbg_goff.py on the MI355X:
# 100x168 maps; exact with one workgroup per CU, exact without SLP packing, never caught by the small parity cases).
# Packed fp32 VALU beside MFMAs is also slower than the scalar form (cdna_hip_programming.md, price of one filler).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast',
         '-fno-slp-vectorize', '-Wno-unused-result']


def _hipcc():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: liblsnet_hip.so cannot be built')


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [_hipcc()] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ['-o', SO]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=HERE)
    return SO


HOST_SO = os.path.join(HERE, 'liblsnet_host.so')
HOST_SOURCES = [os.path.join('host', 'rle.cpp'), os.path.join('host', 'coco_match.cpp'),
                os.path.join('host', 'image.cpp'), os.path.join('host', 'nms_host.cpp')]
HOST_HEADERS = [os.path.join('..', '..', 'include', 'lsnet_host.h')]


def build_host(force=False, verbose=False):
    deps = [os.path.join(HERE, f) for f in HOST_SOURCES + HOST_HEADERS]
    if not force and os.path.exists(HOST_SO) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_SO) for d in deps):
        return HOST_SO
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        raise RuntimeError('g++ not found: liblsnet_host.so cannot be built')
    cmd = [cxx, '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-ffp-contract=off'] + \
        [os.path.join(HERE, s) for s in HOST_SOURCES] + ['-o', HOST_SO]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=HERE)
    return HOST_SO


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_host(force='--force' in sys.argv, verbose=True))
