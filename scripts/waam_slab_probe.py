"""BASELINE.json configs[4] on slabs: the layer-birth loop of waam_from_stl_v7_mm.py:515-550 on ONE rank of a 4-rank slab
decomposition of the synthetic 256 x 256 x 320 head (rank 1 of 4, loopback communicator: per-rank time without wire time),
against the single-domain device loop.  theta = 1 (D9).
    python scripts/waam_slab_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adi_thermal_fields_amd.adi3d_hip_coeff as hip
from adi_thermal_fields_amd import dist_slab, waam

STEEL = (7800.0, 490.0, 54.0)
shape = (256, 256, 320)
mask = waam.synthetic_head_mask(*shape)
layers = waam.plan_layers(mask, 2)
dx = 1e-3
times = waam.birth_times(mask, layers, dx, bead_width=4e-3, scan_speed=0.02)
outs = [times[-1]]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    T1, n1 = waam.run_layer_birth(hip, mask, dx, STEEL, 40.0, 20.0, 1000.0, 1.0, 2000.0, layers, times, outs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('single domain, device loop: %d births, %d steps: %.3f s (%.2f ms per birth + segment)' % (len(layers), n1, t1 - t0, (t1 - t0) / len(layers) * 1e3), flush=True)
world, rank = 4, 1
i0, i1 = rank * 64, (rank + 1) * 64
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Tl, n2 = waam.run_layer_birth_slab(dist_slab.LoopbackComm(world, rank), i0, i1, mask, dx, hip.Material(*STEEL), hip.Params,
                                       40.0, 20.0, 1000.0, 1.0, 2000.0, layers, times, outs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('rank %d of %d (64 planes), slab loop: %d births, %d steps: %.3f s (%.2f ms per birth + segment)' % (rank, world, len(layers), n2, t1 - t0, (t1 - t0) / len(layers) * 1e3), flush=True)
