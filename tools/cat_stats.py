"""Category totals of a prof_summary kernel-stats file."""
import sys
rows = []
for l in open(sys.argv[1]):
    if l.startswith('#'):
        print(l.strip()); continue
    if l.strip().startswith('ms/step'):
        continue
    p = l.split(None, 6)
    if len(p) < 7:
        continue
    rows.append((float(p[0]), float(p[2]), p[6].strip()))


def c(n):
    if 'lsn::dcn' in n: return 'dcn'
    if 'lsn::conv' in n: return 'conv (own x3)'
    if 'lsn::gn' in n or 'lsn::bn' in n: return 'norm (hip)'
    if 'lsn::' in n: return 'lsn other'
    if 'igemm' in n or 'ck::' in n or 'Conv' in n or 'SubTensorOp' in n: return 'conv (MIOpen)'
    if 'batch_norm' in n or 'BatchNorm' in n: return 'batchnorm (ATen)'
    if 'direct_copy' in n or 'copyBuffer' in n: return 'copy'
    if 'fillBuffer' in n or 'FillFunctor' in n: return 'fill'
    if 'multi_tensor' in n: return 'optimizer/foreach'
    if 'reduce_kernel' in n: return 'reduce'
    if 'elementwise' in n: return 'elementwise'
    return 'other'


cat = {}
for ms, calls, n in rows:
    a = cat.setdefault(c(n), [0, 0]); a[0] += ms; a[1] += calls
for k, (ms, calls) in sorted(cat.items(), key=lambda x: -x[1][0]):
    print(f'{k:22s} {ms:8.2f} ms/step {calls:8.0f} calls/step')
