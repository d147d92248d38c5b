"""oracle/waam_oracle.py -- TEST INFRASTRUCTURE, never shipped: the layer planning of the reference's WAAM driver restated
step by step (waam_from_stl_v7_mm.py:436-456 layer slicing, :458-476 birth times).  The driver script cannot be imported
here (it imports trimesh at module level, which is absent), so parity of adi_thermal_fields_amd.waam.plan_layers /
birth_times is pinned against this restatement on gapped masks (tests/test_waam_harness.py).  Parity unpinned against
the running reference for these two functions only."""
import numpy as np


def plan_layers(mask_full, n_per_layer):
    """waam_from_stl_v7_mm.py:436-456"""
    k_indices = np.where(mask_full.any(axis=(0, 1)))[0]
    if k_indices.size == 0:
        raise RuntimeError("empty voxel model")
    kmin, kmax = int(k_indices.min()), int(k_indices.max())
    n_per_layer = max(1, int(n_per_layer))
    layers = []
    ks = kmin
    while ks <= kmax:
        while ks <= kmax and not mask_full[:, :, ks].any():      # :443-444 skip empty planes
            ks += 1
        if ks > kmax:
            break
        ke = min(kmax, ks + n_per_layer - 1)
        while ke >= ks and not mask_full[:, :, ke].any():        # :449-450 trim trailing empty planes
            ke -= 1
        if ke < ks:
            ks += 1
            continue
        layers.append((ks, ke))
        ks = ke + 1
    return layers


def birth_times(mask_full, layers, dx, bead_width, scan_speed, eta_fill=1.0):
    """waam_from_stl_v7_mm.py:458-471"""
    times, t = [], 0.0
    for ks, ke in layers:
        areas = [float(mask_full[:, :, k].sum()) * dx * dx for k in range(ks, ke + 1)]
        A = float(np.mean(areas)) if areas else 0.0
        L = (A / max(bead_width, 1e-12)) * max(eta_fill, 1.0)
        t += float(L / max(scan_speed, 1e-12))
        times.append(t)
    return times
