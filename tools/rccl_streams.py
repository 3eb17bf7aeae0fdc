"""does the order in which a process creates its streams -- the step's second stream (ops/streams.py), the
library's own (csrc/dcn.hip: anchor lists beside the backward GEMM) and RCCL's -- change the speed of the step?  The HIP runtime maps
streams onto a few hardware queues in creation order, and a stream that shares the queue of the step's stream serialises with it
(profiles/r5_stream_queues.txt: 50 ms steps).  Every order needs a fresh process:

    python tools/rccl_streams.py            # driver: one child per order, prints the table (profiles/r6_rccl_streams.txt)
    python tools/rccl_streams.py <order>    # child

orders: plain (no process group), and with a ONE-rank RCCL group whose bucket all-reduces run in every step (LSNET_FORCE_COLLECTIVES=1):
  side,lib,rccl   what DataParallelModel arranges (side_stream(), warm_library_streams(), then the first collective)
  rccl,side,lib   communicator first
  side,rccl,lib   communicator between the two
  rccl,side       communicator first, the library's stream created lazily by the first deformable backward of the step
  lazy            nothing arranged: streams appear where the code first needs them (first collective = first bucket of step 0)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORDERS = ['plain', 'side,lib,rccl', 'rccl,side,lib', 'side,rccl,lib', 'rccl,side', 'lazy',
          # `side!`: the second stream SUBMITS something the moment it is created (a stream seems to get its hardware queue with its
          # first submission, not at creation)
          'side!,lib,rccl', 'rccl,side!,lib', 'side!,rccl,lib', 'lib,rccl,side!', 'product']


def child(order):
    os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if order != 'plain':
        os.environ['LSNET_FORCE_COLLECTIVES'] = '1'
    import torch
    import torch.distributed as dist
    sys.argv = sys.argv[:1]
    import bench
    from lsnet_amd.data import synthetic_batch
    from lsnet_amd.model_zoo import build_lsnet
    from lsnet_amd.ops import streams
    from lsnet_amd import parallel
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    dummy = torch.zeros(1024, device=dev)

    def make(what):
        if what in ('side', 'side!'):
            st = streams._side.get(0)
            if st is None:
                st = streams._side[0] = torch.cuda.Stream(dev)     # (the module's own constructor may submit by now)
            if what == 'side!':
                with torch.cuda.stream(st):
                    dummy.add_(1.0)
                torch.cuda.synchronize()
        elif what == 'lib':
            streams.warm_library_streams(dev)
        elif what == 'rccl':
            dist.all_reduce(dummy)          # the communicator (and RCCL's streams) are created by the first collective
            torch.cuda.synchronize()

    if order != 'plain':
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    seq = [] if order in ('plain', 'lazy', 'product') else order.split(',')
    if order != 'product':
        streams.CLAIM_QUEUE = False     # (the arms place the second stream's first submission themselves; `product` = the package as it is)
    if order not in ('plain', 'product') and 'lib' not in seq:
        streams.warm_library_streams = lambda device: None       # DataParallelModel must not arrange anything itself
    if order == 'lazy':
        streams._side.clear()
    for what in seq:
        make(what)
    torch.manual_seed(0)
    model, cfg = build_lsnet('bbox', 'r50')
    model = parallel.DataParallelModel(model.to(dev).to(memory_format=torch.channels_last).train())
    assert model.reducer.collective == (order != 'plain')
    step, _ = bench.build_step(model, cfg)
    data = synthetic_batch('bbox', 2, 800, 1344, seed=1234, device=dev, channels_last=True)
    ts = []
    bench.timed_steps(step, data, 1, 4)
    for _ in range(3):
        ts.append(bench.timed_steps(step, data, 6, 0) * 1e3)
    side = [hex(s.cuda_stream) for s in streams._side.values()]
    print(f'RESULT {order:16s} {min(ts):7.2f} ms/step (three runs of 6 steps: {" ".join(f"{t:.2f}" for t in ts)}); '
          f'buckets {len(model.reducer.buckets)}, collectives {"on" if model.reducer.collective else "off"}; second stream {side}', flush=True)
    if order != 'plain':
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    print('# LSNet R-50 bbox step (2 x 3x800x1344) by stream creation order; one fresh process per row; RCCL rows all-reduce the 7 gradient '
          'buckets of every step through a one-rank group, launched from the hooks during backward')
    # (profiles/r6_rccl_streams.txt was measured before the package asked for two queues itself, with None = the runtime's default in
    # the first block; bench.py / lsnet_amd now set GPU_MAX_HW_QUEUES=2 unless the environment says otherwise, so every arm is explicit)
    for queues in ('4', '8', '2'):
        print(f'## GPU_MAX_HW_QUEUES = {queues or "(runtime default)"}', flush=True)
        env = dict(os.environ)
        env.pop('GPU_MAX_HW_QUEUES', None)
        if queues:
            env['GPU_MAX_HW_QUEUES'] = queues
        for order in ORDERS:
            t0 = time.time()
            p = subprocess.run([sys.executable, os.path.abspath(__file__), order], capture_output=True, text=True, timeout=600, env=env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith('RESULT')]
            print(lines[-1][7:] if lines else f'{order}: FAILED rc {p.returncode}: ' + (p.stderr.strip().splitlines() or ['?'])[-1][:300], flush=True)
            print(f'#   ({time.time() - t0:.0f} s)', flush=True)


if __name__ == '__main__':
    main()
