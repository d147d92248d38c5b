"""Stand-in for one rank of bench.py, so that bench.launch_ranks (the plain `python bench.py --gpus N` start) can be
driven without a GPU: a gloo process group, one all_reduce, and on rank 0 some stdout noise around ONE JSON line that
carries the contract's `metric` key.  argv[1]: ok | fail (rank 1 dies with exit code 5) | hang (never returns) |
mute (no JSON line)."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'ok'
    rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
    assert os.environ['MASTER_ADDR'] == '127.0.0.1'
    if mode == 'fail' and rank == 1:
        print('stub rank 1: deliberate failure', file=sys.stderr, flush=True)
        sys.exit(5)
    if mode == 'hang':
        time.sleep(600)
    dist.init_process_group('gloo')
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print('banner that is not JSON', flush=True)
        print('{"not_the_line": 1}', flush=True)
        if mode != 'mute':
            print(json.dumps(dict(metric='stub', value=float(t.item()), n_gpus=world, args=sys.argv[2:])), flush=True)
        print('trailing noise', flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
