"""Builds liblsnet_hip.so (gfx950) in-tree with hipcc, and liblsnet_host.so (host-side evaluation helpers, g++).
No torch dependency: plain HIP / C++ behind C ABIs (include/lsnet_hip.h, include/lsnet_host.h)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'liblsnet_hip.so')
SOURCES = ['dcn.hip', 'misc.hip', 'norm.hip', 'conv.hip', 'gconv.hip', 'image.hip', 'loss.hip']
HEADERS = ['common.h', 'dcn_kernels.h', 'dcn_gather_kernels.h', 'dcn_grouped_kernels.h', 'conv_kernels.h', 'conv_wgrad_kernels.h', 'dcn_mm_kernels.h', 'prof.h', 'cross_iou_row.h', os.path.join('..', '..', 'include', 'lsnet_hip.h')]
# -fno-slp-vectorize: hipcc (ROCm 7.2) packs adjacent scalar fp32 adds / fmas into v_pk_add_f32 / v_pk_fma_f32.  In the
# backward-data kernels the HIGH dword of such packed accumulators came back wrong for the last 16 lanes of a wave in
# 0.7 % of the cases, differently on every run, whenever two workgroups shared a CU (tools/dbg_goff.py on the MI355X:
# 100x168 maps; exact with one workgroup per CU, exact without SLP packing, never caught by the small parity cases).
# Packed fp32 VALU beside MFMAs is also slower than the scalar form (cdna_hip_programming.md, price of one filler).
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast',
         '-fno-slp-vectorize', '-Wno-unused-result']


def kernel_signature():
    """sha-256 (16 hex digits) over the sources of the deformable / dense kernel families: the counter files under profiles/
    carry the signature of the library they were measured on, and bench.py reports `roofline.traffic` only while it still
    matches (a counter reading of other kernels is not a measurement of these)."""
    import hashlib
    h = hashlib.sha256()
    for f in ('common.h', 'conv_kernels.h', 'conv_wgrad_kernels.h', 'conv.hip', 'dcn_kernels.h', 'dcn_gather_kernels.h', 'dcn_mm_kernels.h', 'dcn_grouped_kernels.h', 'dcn.hip'):
        with open(os.path.join(HERE, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: liblsnet_hip.so cannot be built')


OBJ_DIR = os.path.join(HERE, 'build')
# headers each translation unit includes (an object is rebuilt when its source or one of these is newer)
UNIT_HEADERS = {
    'dcn.hip': ['common.h', 'dcn_kernels.h', 'dcn_gather_kernels.h', 'dcn_mm_kernels.h', 'dcn_grouped_kernels.h', 'conv_kernels.h', 'prof.h'],
    'conv.hip': ['common.h', 'conv_kernels.h', 'conv_wgrad_kernels.h', 'prof.h'],
    'misc.hip': ['common.h', 'prof.h'], 'norm.hip': ['common.h', 'prof.h'], 'gconv.hip': ['common.h', 'prof.h'],
    'image.hip': ['common.h'], 'loss.hip': ['common.h', 'cross_iou_row.h'],
}
API_HEADER = os.path.join('..', '..', 'include', 'lsnet_hip.h')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _unit_deps(src):
    return [os.path.join(HERE, f) for f in [src] + UNIT_HEADERS.get(src, HEADERS) + [API_HEADER]] + [os.path.abspath(__file__)]


def needs_build(so=None, obj_dir=None):
    so = so or SO
    obj_dir = obj_dir or OBJ_DIR
    if not os.path.exists(so):
        return True
    return any(_stale(os.path.join(obj_dir, s + '.o'), _unit_deps(s)) for s in SOURCES) or \
        _stale(so, [os.path.join(obj_dir, s + '.o') for s in SOURCES])


def build(force=False, verbose=False, defines=(), so=None):
    """One object per translation unit (compiled in parallel, only the stale ones), then the link.  `defines` / `so`: an
    A/B variant of the library beside the product one (tools/README.md), with its own object directory."""
    so = so or SO
    obj_dir = OBJ_DIR if so == SO else so + '.build'
    if not force and not needs_build(so, obj_dir):
        return so
    os.makedirs(obj_dir, exist_ok=True)
    cc = _hipcc()
    cflags = [f for f in FLAGS if f != '-shared'] + ['-D' + d for d in defines]
    jobs = []
    for s in SOURCES:
        obj = os.path.join(obj_dir, s + '.o')
        if force or _stale(obj, _unit_deps(s)):
            cmd = [cc] + cflags + ['-c', os.path.join(HERE, s), '-o', obj]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            jobs.append((s, subprocess.Popen(cmd, cwd=HERE)))
    failed = [s for s, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError('hipcc failed for ' + ', '.join(failed))
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [os.path.join(obj_dir, s + '.o') for s in SOURCES] + ['-o', so]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=HERE)
    return so


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
    if '--ab' in sys.argv:   # the A/B variant: python build.py --ab [-DNAME ...]
        print(build(force='--force' in sys.argv, verbose=True, defines=['LSNET_AB=1'] + [a[2:] for a in sys.argv if a.startswith('-D')],
                    so=os.path.join(HERE, 'liblsnet_hip_ab.so')))
        sys.exit(0)
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_host(force='--force' in sys.argv, verbose=True))
