"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/step_shapes.py -> HBM bytes per launch of each deformable-conv
kernel family of the benchmark step, written as JSON for bench.py (profiles/r2_hbm_traffic.json).

step_shapes.py replays (tower launch, pyramid launch) pairs, so the dispatches of every kernel alternate tower, pyramid,
tower, ...; a step has 6 tower and 2 pyramid launches of each family: mean launch = (6 tower + 2 pyramid) / 8.
Units and corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is DOUBLED; WRITE_SIZE is taken as reported."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsnet_amd.csrc.build import kernel_signature  # noqa: E402

fetch_dir, write_dir, out_json = sys.argv[1:4]
math = sys.argv[4] if len(sys.argv) > 4 else 'bf16x6'
FAMILY = [('dcn_fwd_', 'dcn_fwd'), ('dcn_prepare_w_kernel', 'dcn_fwd'), ('wfrag#fwd', 'dcn_fwd'),
          ('dcn_wgrad_', 'dcn_wgrad'), ('dcn_gout_frag', 'dcn_wgrad'), ('dcn_chunk_meta', 'dcn_wgrad'),
          ('conv_wgrad_reduce', 'dcn_wgrad'),
          ('dcn_bwd_data', 'dcn_bwd_data'), ('dcn_gather', 'dcn_bwd_data'), ('dcn_bin', 'dcn_bwd_data'),
          ('dcn_fill', 'dcn_bwd_data'), ('dcn_sort_lists', 'dcn_bwd_data'), ('dcn_prepare_wt', 'dcn_bwd_data'),
          ('rocprim', 'dcn_bwd_data'), ('conv_mm_kernel', 'dcn_bwd_data'), ('dcn_anchor', 'dcn_bwd_data'),
          ('dcn_offgrad', 'dcn_bwd_data'), ('conv_splitk_reduce', 'dcn_bwd_data'), ('wfrag#bwd', 'dcn_bwd_data')]
# (round 3: the replay contains no dense convolution, so every conv_mm_kernel dispatch is the backward-data GEMM of a
# deformable call; each call builds one weight image with conv_wfrag_kernel: forward, backward, forward, ...)


def family(name):
    for frag, fam in FAMILY:
        if frag in name:
            return fam
    return None


def read(d, counter):
    """[(dispatch id, kernel name, value)] of one pass, in dispatch order"""
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    return sorted(rows)


def by_launch(rows):
    """{kernel: {'tower': [values], 'pyramid': [values]}}.  step_shapes.py replays (tower, pyramid) pairs and every
    deformable-conv call starts with dcn_prepare_w_kernel (forward): the n-th such dispatch opens launch group n; even
    groups are tower launches, odd ones pyramid launches.  (Kernel names no longer appear in both kinds of launch --
    the gather has a short-list and a long-list variant -- so alternation by name is not enough.)"""
    out = defaultdict(lambda: {'tower': [], 'pyramid': []})
    group, nfrag = -1, 0
    for _, name, v in rows:
        if 'conv_wfrag' in name:   # image i: call i // 2 (a forward and a backward per launch group), forward first
            g, name = nfrag // 2, 'conv_wfrag_kernel wfrag#' + ('fwd' if nfrag % 2 == 0 else 'bwd')
            nfrag += 1
            out[name]['tower' if g % 2 == 0 else 'pyramid'].append(v)
            continue
        if 'dcn_prepare_w_kernel' in name:   # (round-2 forward: its plane preparation precedes the forward kernel)
            out[name]['tower' if (group + 1) % 2 == 0 else 'pyramid'].append(v)
            continue
        if 'dcn_fwd_' in name:
            group += 1
        if group >= 0:
            out[name]['tower' if group % 2 == 0 else 'pyramid'].append(v)
    return out


fetch, write = by_launch(read(fetch_dir, 'FETCH_SIZE')), by_launch(read(write_dir, 'WRITE_SIZE'))
if not fetch or not write:
    print('missing counter data', len(fetch), len(write))
    sys.exit(1)
KIB = 1024 / 1e9
fam = defaultdict(lambda: {'tower': [0.0, 0.0], 'pyramid': [0.0, 0.0]})
print(f'{"kernel":64s} {"n":>3s} | tower: FETCHx2 + WRITE (MB) | pyramid: FETCHx2 + WRITE (MB)   (per launch of that kind)')
# per launch of a kind = sum over the kernel's dispatches of that kind / number of launches of that kind
# (round 6: the forward kernel has two instantiations -- whole tiles for the tower launch, stream-K pieces for the pyramid
# launch -- so the launches of a kind are counted over every forward kernel name)
n_launch = {k: max(sum(len(v[k]) for n, v in fetch.items() if 'dcn_fwd_' in n), 1) for k in ('tower', 'pyramid')}
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, {'tower': [], 'pyramid': []}), write.get(name, {'tower': [], 'pyramid': []})
    per = lambda v, k: sum(v[k]) / n_launch[k]
    ft, fp, wt, wp = 2 * per(f, 'tower'), 2 * per(f, 'pyramid'), per(w, 'tower'), per(w, 'pyramid')
    n = len(f['tower']) + len(f['pyramid'])
    print(f'{name[:64]:64s} {n:3d} | {ft * KIB * 1e3:9.1f} + {wt * KIB * 1e3:9.1f} | {fp * KIB * 1e3:9.1f} + {wp * KIB * 1e3:9.1f}')
    fa = family(name)
    if fa:
        fam[fa]['tower'][0] += ft * KIB
        fam[fa]['tower'][1] += wt * KIB
        fam[fa]['pyramid'][0] += fp * KIB
        fam[fa]['pyramid'][1] += wp * KIB
res = {'math': math, 'kernel_signature': kernel_signature(), 'source': 'tools/pmc_step_shapes.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over '
                               'tools/step_shapes.py; FETCH_SIZE x 2 (gfx950), KiB -> bytes', 'kernels': {}}
for fa, d in fam.items():
    t, p = sum(d['tower']), sum(d['pyramid'])
    mean = (6 * t + 2 * p) / 8
    res['kernels'][fa] = {
        'gbytes_per_mean_launch': round(mean, 4), 'tower_launch_gb': round(t, 4), 'pyramid_launch_gb': round(p, 4),
        'tower_fetch_write_gb': [round(v, 4) for v in d['tower']], 'pyramid_fetch_write_gb': [round(v, 4) for v in d['pyramid']],
        'note': f'rocprofv3 FETCH_SIZE x2 + WRITE_SIZE over all kernels of the family; tower launch (5 levels, 52.8 GFLOP) '
                f'{t:.3f} GB, pyramid launch (15 pairs, 158.5 GFLOP) {p:.3f} GB, step mean (6 tower + 2 pyramid) / 8; '
                f'profiles/{os.path.basename(out_json).split("_")[0]}_pmc_hbm.txt'}
    print(f'{fa}: tower {t:.3f} GB (fetch {d["tower"][0]:.3f} + write {d["tower"][1]:.3f}), pyramid {p:.3f} GB '
          f'(fetch {d["pyramid"][0]:.3f} + write {d["pyramid"][1]:.3f}); mean launch of the step {mean:.3f} GB')
json.dump(res, open(out_json, 'w'), indent=1)
