"""Summary of tools/pmc_stream_step.sh: per HBM-bound kernel of the benchmark step, launches per step, mean duration (from the
--kernel-trace pass), HBM bytes per step from the counter passes (FETCH_SIZE x 2 + WRITE_SIZE; rocprofv3 reports KiB, and on
gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes -- MI355X_MICROARCH.md, HBM section) and
the ALGORITHMIC bytes per step where the step's shapes give them (LSNet R-50 bbox, 2 x 3x800x1344: 44 800 points in five
levels, 256 channels, 80 classes, 38.6 M parameters), every operand once.
    python tools/pmc_stream_summary.py <fetch dir> <write dir> <trace dir> <steps>"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsnet_amd.csrc.build import kernel_signature  # noqa: E402

fetch_dir, write_dir, trace_dir, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
MB = 1e6
P, C, NCLS = 44800, 256, 80
X_HEAD = P * C * 4                                     # one pass over a five-level 256-channel head tensor
X_FPN = (33600 + 8400 + 2100) * C * 4 + X_HEAD         # laterals P3..P5 + outputs P3..P7 (single-level calls)
X_GN = 8 * X_HEAD + X_FPN                              # per step: 8 head GroupNorms (2 towers x 3 + 2 pyramid) + 8 in the neck
PARAMS = 38.577e6 * 4
ALG = {   # kernel-name fragment -> algorithmic bytes per STEP (None: shape-dependent glue, counters only)
    'gn_stats_kernel': X_GN, 'gn_apply_kernel': 2 * X_GN, 'gn_bwd_reduce_kernel': 2 * X_GN, 'gn_bwd_apply_kernel': 3 * X_GN,
    'focal_sum_kernel': P * NCLS * 4 + P * 12, 'focal_bwd_w_kernel': 2 * P * NCLS * 4 + P * 12,
    'sgd_sqnorm_kernel': PARAMS, 'sgd_step_kernel': 5 * PARAMS,
    'cross_iou_bbox_stage_kernel<false>': 2 * P * (20 + 10 + 3 + 4 + 1 + 1) * 4,       # forward: raw 20 + gt 10 + anchor 3 + box 4 + weight 1 in, 1 loss out; two stages
    'cross_iou_bbox_stage_kernel<true>': 2 * P * (20 + 10 + 3 + 4 + 1 + 1 + 20) * 4,   # backward: the same + 1 upstream gradient in, 20 grad raw out
    'topk_cols_kernel': 4 * 22400 * 7 * 4,
}


def read_counter(d, counter):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    return sorted(rows)


def timed(rows):
    """rows between the last two marker dispatches (bench.py brackets its timed steps with lsn::selftest32_kernel)"""
    marks = [i for i, r in enumerate(rows) if 'selftest32' in r[1]]
    return rows[marks[-2] + 1:marks[-1]] if len(marks) >= 2 else rows


def short(name):
    n = name.split('(')[0].replace('void ', '').replace('lsn::', '')
    return n


fetch = defaultdict(float)
write = defaultdict(float)
count = defaultdict(int)
for _, name, v in timed(read_counter(fetch_dir, 'FETCH_SIZE')):
    fetch[short(name)] += v * 1024 * 2
    count[short(name)] += 1
for _, name, v in timed(read_counter(write_dir, 'WRITE_SIZE')):
    write[short(name)] += v * 1024
dur = defaultdict(float)
dcount = defaultdict(int)
files = glob.glob(trace_dir + '/**/*kernel_trace.csv', recursive=True)
if files:
    rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r['Start_Timestamp']))
    rows = [(0, r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
    for _, name, us in timed(rows):
        dur[short(name)] += us
        dcount[short(name)] += 1
print(f'# HBM counters of the HBM-bound kernel classes of the benchmark step (tools/pmc_stream_step.sh), {steps} timed steps;')
print(f'# kernel_signature {kernel_signature()} (csrc sources of the dense / deformable families; norm.hip / misc.hip / loss.hip are not part of it)')
print('# counted = FETCH_SIZE x 2 + WRITE_SIZE (KiB -> bytes), per step; algorithmic = every operand once, per step; GB/s over the')
print('# kernel-trace durations of the same launches (HBM peak 8 000 GB/s, MI355X_MICROARCH.md)')
print(f'{"kernel":<44} {"launches":>8} {"us/step":>9} {"counted MB":>11} {"fetch MB":>9} {"write MB":>9} {"alg MB":>9} {"cnt/alg":>8} {"alg GB/s":>9} {"cnt GB/s":>9}')
names = sorted(set(fetch) | set(write), key=lambda n: -(fetch[n] + write[n]))
tot_c = tot_t = 0.0
for n in names:
    c = (fetch[n] + write[n]) / steps
    us = dur.get(n, 0.0) / steps
    alg = next((v for k, v in ALG.items() if k in n), None)
    tot_c += c
    tot_t += us
    gbs = lambda b: f'{b / us / 1e3:9.0f}' if us > 0 and b else f'{"-":>9}'
    print(f'{n[:44]:<44} {count[n] / steps:8.1f} {us:9.1f} {c / MB:11.1f} {fetch[n] / steps / MB:9.1f} {write[n] / steps / MB:9.1f} '
          f'{(alg / MB if alg else float("nan")):9.1f} {(c / alg if alg else float("nan")):8.2f} {gbs(alg)} {gbs(c)}')
print(f'{"all of the above":<44} {"":>8} {tot_t:9.1f} {tot_c / MB:11.1f}' + (f'  -> {tot_c / tot_t / 1e3:.0f} GB/s counted over their summed durations' if tot_t else ''))
