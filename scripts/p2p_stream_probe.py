"""One GPU, single-rank RCCL group: what a plane exchange costs on the critical path of a stream, by the way it is posted.
  A  dist.batch_isend_irecv + wait()                      (torch's internal RCCL stream: two cross-stream hops)
  B  the same P2P ops inside _coalescing_manager(async_ops=False)   (current stream, if the backend honours it for P2P)
  C  a device copy of the same bytes                     (what the loopback communicator does)
usage: python scripts/p2p_stream_probe.py [hi]       hi: TORCH_NCCL_HIGH_PRIORITY=1"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from torch.distributed import distributed_c10d as c10d


def main():
    if 'hi' in sys.argv[1:]:
        os.environ['TORCH_NCCL_HIGH_PRIORITY'] = '1'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    n = 512 * 512
    a, b = torch.ones(n, device=dev, dtype=torch.float64), torch.full((n,), 2.0, device=dev, dtype=torch.float64)
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    big = torch.zeros(64 << 20, device=dev)

    def ops():
        return [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, ra, 0), dist.P2POp(dist.isend, b, 0),
                dist.P2POp(dist.irecv, rb, 0)]

    def post_a():
        for r in dist.batch_isend_irecv(ops()):
            r.wait()

    def post_b():
        g = c10d._get_default_group()
        with c10d._coalescing_manager(g, dev, async_ops=False):
            for o in ops():
                o.op(o.tensor, o.peer, o.group, o.tag)

    def post_c():
        ra.copy_(a); rb.copy_(b)

    for name, post in (('A batch_isend_irecv', post_a), ('B coalesced, async_ops=False', post_b), ('C device copies', post_c)):
        try:
            for _ in range(5):
                post()
            torch.cuda.synchronize()
            assert float(ra[7]) == 1.0 and float(rb[7]) == 2.0
            tot = 0.0
            for _ in range(50):
                big.add_(1.0)                                      # the stream is busy when the exchange is posted
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); post(); big[:1024].add_(1.0); e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            print(f'{name:32s} {tot / 50 * 1e3:8.1f} us from the end of the kernel before to the end of the kernel after', flush=True)
        except Exception as e:                                     # noqa: BLE001 -- a probe: report and go on
            print(f'{name:32s} failed: {type(e).__name__}: {e}', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
