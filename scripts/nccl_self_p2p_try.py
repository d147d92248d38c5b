"""Can RCCL do batched send/recv to SELF in a single-rank group?  If so, SlabStepper can be run over the real
torch.distributed P2P path (side stream, batch_isend_irecv, events) on one GPU."""
import os, sys
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
a = torch.arange(1024, dtype=torch.float64, device='cuda'); b = torch.zeros_like(a)
c = torch.arange(1024, dtype=torch.float64, device='cuda') * 2; d = torch.zeros_like(a)
try:
    ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, c, 0), dist.P2POp(dist.irecv, d, 0)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    torch.cuda.synchronize()
    print('self p2p ok:', bool(torch.equal(a, b)), bool(torch.equal(c, d)))
except Exception as e:
    print('self p2p failed:', type(e).__name__, str(e)[:300])
dist.destroy_process_group()
