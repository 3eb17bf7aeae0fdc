"""Host-side checks of the measurement tooling: the HBM-counter aggregation that feeds bench.py's roofline.traffic."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pass(d, counter, scale):
    """A synthetic rocprofv3 counter_collection.csv: two replays of (tower launch, pyramid launch); kernel names as the
    library's, the gather with its two variants (short lists in tower launches, long lists in pyramid launches)."""
    names = ['void lsn::dcn_prepare_w_kernel<3>(x)', 'void lsn::dcn_fwd_xn_kernel<true, 6>(a)',
             'void lsn::dcn_prepare_wt_kernel<3>(y)', 'lsn::dcn_bin_kernel(a)', 'void lsn::dcn_bwd_data_xn_kernel<6, true>(a)',
             'void lsn::dcn_gather_kernel<%d>(g)', 'void lsn::dcn_wgrad_xn_kernel<false, 6, 256>(a)']
    os.makedirs(os.path.join(d, 'x'), exist_ok=True)
    with open(os.path.join(d, 'x', 'ops_counter_collection.csv'), 'w') as f:
        w = csv.writer(f)
        w.writerow(['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        i = 0
        for _ in range(2):
            for kind in (0, 1):
                for n in names:
                    i += 1
                    w.writerow([i, n % (1 if kind == 0 else 4) if '%d' in n else n, counter,
                                scale * (1000 if kind == 0 else 3000)])     # KiB


def test_pmc_traffic_splits_by_launch_kind(tmp_path):
    fd, wd, out = tmp_path / 'f', tmp_path / 'w', tmp_path / 't.json'
    _write_pass(str(fd), 'FETCH_SIZE', 1.0)
    _write_pass(str(wd), 'WRITE_SIZE', 0.5)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_traffic.py'), str(fd), str(wd), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    res = json.load(open(out))
    kib = 1024 / 1e9
    fam = res['kernels']
    # backward-data family = prepare_wt + bin + GEMM + gather (one variant per launch kind): 4 kernels per launch;
    # per kernel FETCH x 2 + WRITE = 2 * 1000 + 500 KiB (tower) resp. 3 x that (pyramid)
    assert abs(fam['dcn_bwd_data']['tower_launch_gb'] - 4 * 2500 * kib) < 1e-4
    assert abs(fam['dcn_bwd_data']['pyramid_launch_gb'] - 4 * 7500 * kib) < 1e-4
    assert abs(fam['dcn_fwd']['tower_launch_gb'] - 2 * 2500 * kib) < 1e-4          # prepare_w + forward
    assert abs(fam['dcn_wgrad']['pyramid_launch_gb'] - 7500 * kib) < 1e-4
    mean = (6 * fam['dcn_bwd_data']['tower_launch_gb'] + 2 * fam['dcn_bwd_data']['pyramid_launch_gb']) / 8
    assert abs(fam['dcn_bwd_data']['gbytes_per_mean_launch'] - mean) < 1e-3
    assert res['math'] == 'bf16x6'


def test_bench_reads_the_committed_traffic_file():
    """bench.py takes roofline.traffic from profiles/r3_hbm_traffic.json: the committed file has the fields it reads."""
    sys.path.insert(0, ROOT)
    import bench
    with open(bench.TRAFFIC_FILE) as f:
        tf = json.load(f)
    assert tf['math'] == 'bf16x6'
    for fam in ('dcn_fwd', 'dcn_bwd_data', 'dcn_wgrad'):
        e = tf['kernels'][fam]
        assert e['gbytes_per_mean_launch'] > 0 and isinstance(e['note'], str)
        assert abs(e['gbytes_per_mean_launch'] - (6 * e['tower_launch_gb'] + 2 * e['pyramid_launch_gb']) / 8) < 2e-3
