"""waam -- the layer-birth / moving-deposit time loops of the reference's WAAM drivers, backend-agnostic.

The loops are restated from waam_from_stl_v7_mm.py (layers :436-456, birth times :458-471, activate_layer
:487-495, event loop :515-550) and single_track_on_plate.py:150-177 (column-by-column deposit).  They drive ANY
module with the reference's operator surface (`Grid3D`, `Material`, `Params`,
`precompute_coeff_packs_unified`, `adi_step_numba_coeff`): the HIP backend in production, the CPU oracle in
the parity tests.  STL loading / voxelisation (trimesh) is out of scope; `synthetic_head_mask` supplies the
formula-generated stand-in for the missing `11091_FemaleHead_v4.stl`.

With the HIP backend the temperature stays in HBM for the whole run: births are a masked fill on the
device, the mask/pack rebuild is one upload (1 B/cell) plus two kernels, frames download only at output times.
"""
import math

import numpy as np

__all__ = ['synthetic_head_mask', 'plan_layers', 'birth_times', 'layer_birth_schedule', 'run_layer_birth', 'run_single_track',
           'run_layer_birth_slab', 'run_single_track_slab']


def synthetic_head_mask(nx, ny, nz):
    """Union of an ellipsoid (semi-axes 0.35/0.42/0.45 of the box, centred at 0.5/0.5/0.55) and a neck
    cylinder (radius 0.16 of the box, lower 40 %), by formula."""
    x = (np.arange(nx) + 0.5) / nx - 0.5
    y = (np.arange(ny) + 0.5) / ny - 0.5
    z = (np.arange(nz) + 0.5) / nz
    X, Y, Z = np.meshgrid(x, y, z, indexing='ij')
    ell = (X / 0.35) ** 2 + (Y / 0.42) ** 2 + ((Z - 0.55) / 0.45) ** 2 <= 1.0
    neck = (X ** 2 + Y ** 2 <= 0.16 ** 2) & (Z <= 0.4)
    return ell | neck


def plan_layers(mask_full, n_per_layer):
    """Plane ranges (ks, ke) along axis 2, the layer list of waam_from_stl_v7_mm.py:436-456: a layer starts at the next
    occupied plane, spans at most `n_per_layer` planes and ends on an occupied plane; empty planes belong to no layer.
    Computed on the sorted indices of the occupied planes (two binary searches per layer) instead of the reference's
    plane-by-plane scan; the list is the same (tests/test_waam_harness.py pins it against the reference's on gapped masks)."""
    occ = np.flatnonzero(np.asarray(mask_full).any(axis=(0, 1)))
    if occ.size == 0:
        raise RuntimeError("empty voxel model")
    n = max(1, int(n_per_layer))
    layers, p = [], 0
    while p < occ.size:
        ks = int(occ[p])
        q = int(np.searchsorted(occ, ks + n - 1, side='right')) - 1     # last occupied plane inside the span
        layers.append((ks, int(occ[q])))
        p = q + 1
    return layers


def birth_times(mask_full, layers, dx, bead_width, scan_speed, eta_fill=1.0):
    """Cumulative deposition time per layer (waam_from_stl_v7_mm.py:458-471): the layer's mean cross-section divided by
    the bead width gives the track length, the scan speed its duration.  One reduction over the mask for all plane
    areas; the floating-point operations per layer are the reference's (count*dx*dx, mean, running sum), so the times
    are bit-identical."""
    areas = np.asarray(mask_full).sum(axis=(0, 1)).astype(np.float64) * dx * dx
    bw, v, eta = max(bead_width, 1e-12), max(scan_speed, 1e-12), max(eta_fill, 1.0)
    dur = [float(np.mean(areas[ks:ke + 1])) / bw * eta / v for ks, ke in layers]
    return [float(t) for t in np.cumsum(np.asarray(dur, dtype=np.float64))] if dur else []


def layer_birth_schedule(times_birth, times_out):
    """The clock of the reference's event loop (waam_from_stl_v7_mm.py:515-550) as a stream of actions, in its order:
         ('advance', seconds)   integrate the field over this interval (the consumer splits it into sub-steps under its
                                time-step cap and skips it while nothing is active, :524-528)
         ('birth', layer)       activate layer number `layer` (:487-495) and rebuild the packs (:534)
         ('frame', t)           an output time has been reached (:540-548)
    Between two consecutive event times every birth that is due is taken first, each preceded by the time that has passed
    since the previous action; intervals of at most 1e-15 s are dropped, output times are matched to 1e-12 s -- the
    reference's tolerances.  One schedule drives the single-domain loop and the per-rank slab loop alike."""
    nb = len(times_birth)
    nxt, t_now = 0, 0.0
    for te in sorted(set(times_out) | set(times_birth)):
        while nxt < nb and times_birth[nxt] <= te + 1e-15:
            t_b = times_birth[nxt]
            if t_b - t_now > 1e-15:
                yield ('advance', t_b - t_now)
            t_now = t_b
            yield ('birth', nxt)
            nxt += 1
        if te - t_now > 1e-15:
            yield ('advance', te - t_now)
        t_now = te
        if any(abs(te - to) <= 1e-12 for to in times_out):
            yield ('frame', te)


GRAPH_MIN_NSUB = 16      # segments at least this long run through StagedStepper.run (graph capture costs about a step)


def _is_device_backend(backend):
    return hasattr(backend, 'to_device')


def _step(backend, T, grid, mat, params, packs, Tinf):
    fn = getattr(backend, 'adi_step_hip_coeff', None) or backend.adi_step_numba_coeff
    return fn(T, grid, mat, params, packs, Tinf=Tinf)


def _birth(backend, T, grid, newborn, Ts):
    """T[newborn] = Ts (waam_from_stl_v7_mm.py:489-493)"""
    if _is_device_backend(backend) and hasattr(T, 'fill_where'):
        import torch
        T.fill_where(grid.layout.to_layout(newborn, torch.uint8), Ts)
    else:
        T[newborn] = Ts
    return T


def run_layer_birth(backend, mask_full, dx, mat_args, h, Tinf, Ts, theta, cfl, layers, times_birth, times_out,
                    on_frame=None, device_resident=True, device_loop=True):
    """The event loop of waam_from_stl_v7_mm.py:515-550.  Returns (T_final as NumPy, number of ADI steps)."""
    nx, ny, nz = mask_full.shape
    mask_act = np.zeros_like(mask_full, dtype=bool)
    grid = backend.Grid3D(nx, ny, nz, dx, mask_act)
    mat = backend.Material(*mat_args)
    params = backend.Params(dt=1e-3, theta=theta)
    alpha = mat_args[2] / (mat_args[0] * mat_args[1])
    dt_cap = cfl * dx * dx / alpha
    T = np.full((nx, ny, nz), float(Tinf), dtype=np.float64)
    if device_resident and _is_device_backend(backend):
        T = backend.to_device(T)
    robin = {f: h for f in ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')}

    def build_packs():
        return backend.precompute_coeff_packs_unified(grid, mat, dir_mask=None, dir_value=None, neumann=None,
                                                      robin_h=robin, robin_Tinf=Tinf)
    packs = None
    nsteps = 0

    def advance(seg):
        nonlocal T, nsteps
        nsub = max(1, int(math.ceil(seg / dt_cap)))
        params.dt = max(seg / nsub, 1e-15)
        if nsub >= GRAPH_MIN_NSUB and hasattr(backend, 'StagedStepper') and hasattr(T, 'fill_where'):
            # a long segment on the device backend: the nsub launches of this segment replayed from a HIP graph
            T = backend.StagedStepper(grid, mat, params, packs, Tinf).run(T, nsub)
        else:
            for _ in range(nsub):
                T = _step(backend, T, grid, mat, params, packs, Tinf)
        nsteps += nsub

    # device loop: the full mask, the active mask, the field and the packs live in HBM.  A birth
    # is ONE kernel on the layer's planes (newborn = full & ~active, T[newborn] = Ts, active |= full: adi_birth_planes),
    # then the flags and the six coefficient arrays are rebuilt on the planes whose exposure can have changed (the layer
    # and one plane either side) -- in place, no allocation, no host synchronisation: nothing crosses PCIe between
    # output times and nothing waits for the device between births
    dev_loop = device_loop and device_resident and hasattr(backend, 'BirthPacks') and hasattr(T, 'fill_where')
    if dev_loop:
        import torch
        d_full = grid.layout.to_layout(mask_full, torch.uint8)
        d_act = grid.layout.empty(torch.uint8, zero=True)
        grid.set_mask_device(d_act, all_solid=False)
        bpacks = backend.BirthPacks(grid, mat, robin_h=robin)
        packs = bpacks.packs
        plane_cells = np.asarray(mask_full).sum(axis=(0, 1)).astype(np.int64)      # newborn cells per plane, host-known
        plane_born = np.zeros(nz, dtype=bool)
    else:
        packs = build_packs()
    n_active = 0

    def birth(ks, ke):
        nonlocal T, packs, n_active
        if dev_loop:
            backend.birth_planes(T, d_act, d_full, grid, ks, ke + 1, Ts)          # :489-493 + mask_act |= born
            fresh = ~plane_born[ks:ke + 1]
            n_active += int(plane_cells[ks:ke + 1][fresh].sum())
            plane_born[ks:ke + 1] = True
            grid.set_mask_device(d_act, ks - 1 if ks > 0 else 0, min(nz, ke + 2), all_solid=False)   # :494-495
            packs = bpacks.update(ks - 1, ke + 2)                                 # :534, the planes that changed
            return
        born = np.zeros_like(mask_full, dtype=bool)
        born[:, :, ks:ke + 1] = mask_full[:, :, ks:ke + 1]
        newborn = born & (~mask_act)
        if newborn.any():
            T = _birth(backend, T, grid, newborn, Ts)
        mask_act[...] = mask_act | born
        n_active = int(mask_act.sum())
        grid.mask = mask_act                           # :494-495
        packs = build_packs()                          # :534

    for what, arg in layer_birth_schedule(times_birth, times_out):
        if what == 'advance':
            if n_active > 0:                           # no ADI steps while nothing is active (:524)
                advance(arg)
        elif what == 'birth':
            birth(*layers[arg])
        elif on_frame is not None:
            on_frame(arg, np.asarray(T), d_act.cpu().contiguous().numpy().astype(bool) if dev_loop else mask_act.copy())
    return np.asarray(T), nsteps


def run_single_track(backend, plate_mask, track_box, dx, mat_args, h, Tinf, T_track, theta, dt, t_step,
                     device_resident=True):
    """single_track_on_plate.py:150-177: the deposit advances one column per t_step along axis 1; packs are
    rebuilt after every column.  track_box = (x0, x1, z0, z1, n_columns)."""
    x0, x1, z0, z1, ncol = track_box
    nx, ny, nz = plate_mask.shape
    mask = plate_mask.copy()
    grid = backend.Grid3D(nx, ny, nz, dx, mask)
    mat = backend.Material(*mat_args)
    params = backend.Params(dt=dt, theta=theta)
    T = np.full((nx, ny, nz), float(Tinf), dtype=np.float64)
    if device_resident and _is_device_backend(backend):
        T = backend.to_device(T)
    robin = {f: h for f in ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')}
    dev_loop = device_resident and hasattr(grid, 'set_mask_device') and hasattr(T, 'fill_where')
    if dev_loop:                                    # the mask lives in HBM: a new column is two slice assignments
        import torch
        d_mask = grid.layout.to_layout(mask, torch.uint8)
    for yi in range(ncol):
        if dev_loop:
            d_mask[x0:x1, yi:yi + 1, z0:z1] = 1
            grid.set_mask_device(d_mask)
        else:
            mask[x0:x1, yi:yi + 1, z0:z1] = True
            grid.mask = mask
        packs = backend.precompute_coeff_packs_unified(grid, mat, robin_h=robin, robin_Tinf=Tinf)
        T[x0:x1, yi:yi + 1, z0:z1] = T_track
        n_sub = max(1, int(math.ceil(t_step / dt)))
        dt_orig = params.dt
        params.dt = t_step / n_sub
        for _ in range(n_sub):
            T = _step(backend, T, grid, mat, params, packs, Tinf)
        params.dt = dt_orig
    return np.asarray(T)


def run_layer_birth_slab(comm, i0, i1, mask_full, dx, mat, params_cls, h, Tinf, Ts, theta, cfl, layers, times_birth,
                         times_out, engine=None, on_frame=None):
    """The same event loop on ONE RANK of a slab decomposition (planes [i0, i1) of memory axis 0; BASELINE.json
    configs[4] runs it on 4 GPUs).  Every rank knows the full host mask (bookkeeping only, 1 bit of information per
    cell) and owns the temperature of its slab on its GPU; a birth is a masked fill of the local slab followed by
    SlabStepper.set_mask (mask halo exchange + flags + pack rebuild).  Returns (local field as NumPy, steps)."""
    from . import dist_slab
    nx, ny, nz = mask_full.shape
    mask_act = np.zeros_like(mask_full, dtype=bool)
    params = params_cls(dt=1e-3, theta=theta)
    alpha = mat.k / (mat.rho * mat.cp)
    dt_cap = cfl * dx * dx / alpha
    robin = {f: h for f in ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')}
    st = dist_slab.SlabStepper(mask_act[i0:i1], dx, mat, params, Tinf, robin_h=robin, comm=comm, engine=engine)
    T = np.full((i1 - i0, ny, nz), float(Tinf), dtype=np.float64)
    on_device = getattr(st.engine.device, 'type', 'cpu') == 'cuda'
    if on_device:                                   # the slab's field stays in HBM: step() hands back device tensors
        import torch
        T = torch.from_numpy(T).to(st.engine.device)
    # device loop (as run_layer_birth's): the slab's part of the full mask and its active mask live in HBM; a birth is one
    # kernel on the layer's planes of the slab + SlabStepper.set_mask_device (mask halo planes, flags and packs rebuilt in
    # place on those planes).  The host keeps only per-plane bookkeeping: which planes are born, how many cells are active.
    dev_loop = on_device and st.device_births_supported()
    if dev_loop:
        E = st.engine
        m_ext = np.zeros((i1 - i0 + 2, st.ny, st.nz), dtype=np.bool_)          # (the slab's planes may be padded: st._pad)
        m_ext[1:-1] = st._pad(np.asarray(mask_full[i0:i1], dtype=np.bool_), False)
        d_full_int = st.Lext.to_layout(m_ext, torch.uint8)[1:-1]
        count = torch.zeros(1, dtype=torch.int64, device=E.device)
        plane_cells = np.asarray(mask_full).sum(axis=(0, 1)).astype(np.int64)
        plane_born = np.zeros(nz, dtype=bool)
    n_active = 0
    nsteps = 0

    def advance(seg):
        nonlocal T, nsteps
        nsub = max(1, int(math.ceil(seg / dt_cap)))
        params.dt = max(seg / nsub, 1e-15)
        for s in range(nsub):
            T = st.step(T, prefetch_halo=(s + 1 < nsub))
        nsteps += nsub

    def birth(ks, ke):
        nonlocal T, n_active
        if dev_loop:
            if not isinstance(T, torch.Tensor) or T.data_ptr() != dist_slab._interior(st._ext_bufs[st._cur]).data_ptr():
                T = st._logical(dist_slab._interior(st._load_state(T)))   # the state buffer itself (nothing stepped yet)
            E.birth_planes(st.Lint, T, dist_slab._interior(st.d_mask_ext), d_full_int, ks, ke + 1, Ts, count)
            fresh = ~plane_born[ks:ke + 1]
            n_active += int(plane_cells[ks:ke + 1][fresh].sum())
            plane_born[ks:ke + 1] = True
            st.set_mask_device(ks, ke + 1)
            return
        born = np.zeros_like(mask_full, dtype=bool)
        born[:, :, ks:ke + 1] = mask_full[:, :, ks:ke + 1]
        newborn = (born & (~mask_act))[i0:i1]
        if newborn.any():
            if on_device:
                T[torch.from_numpy(newborn).to(T.device)] = Ts      # in place (the last sub-step sent no halo ahead)
            else:
                Tl = np.array(st.local_numpy(T))
                Tl[newborn] = Ts
                T = Tl
        mask_act[...] = mask_act | born
        n_active = int(mask_act.sum())
        st.set_mask(mask_act[i0:i1])

    for what, arg in layer_birth_schedule(times_birth, times_out):
        if what == 'advance':
            if n_active > 0:
                advance(arg)
        elif what == 'birth':
            birth(*layers[arg])
        elif on_frame is not None:
            act = st._logical(dist_slab._interior(st.d_mask_ext)).cpu().contiguous().numpy().astype(bool) if dev_loop \
                else mask_act[i0:i1].copy()
            on_frame(arg, np.array(st.local_numpy(T)), act)
    return np.array(st.local_numpy(T)), nsteps


def run_single_track_slab(comm, i0, i1, plate_mask, track_box, dx, mat, params_cls, h, Tinf, T_track, theta, dt, t_step,
                          engine=None):
    """The moving deposit of single_track_on_plate.py:150-177 on ONE RANK of a slab decomposition (planes [i0, i1) of
    memory axis 0; BASELINE.json configs[4]: "layer-birth + moving source, 4 GPUs").  The track box spans planes
    [x0, x1) of the SHARDED axis, so a new column lands on every rank whose slab meets that range: those ranks switch
    the column's cells on in their part of the mask and set them to T_track (:159-160, :166), every rank then rebuilds
    flags and packs for its slab -- SlabStepper.set_mask: mask halo exchange (a column next to a slab boundary changes the
    neighbour's halo coupling bits), flags, packs (:163) -- and takes the n_sub sub-steps of the column (:168-176).
    Every rank keeps the full host mask for bookkeeping (1 bit of information per cell), as run_layer_birth_slab does.
    Returns the local field as NumPy."""
    from . import dist_slab
    x0, x1, z0, z1, ncol = track_box
    nx, ny, nz = plate_mask.shape
    mask = np.array(plate_mask, dtype=bool)
    params = params_cls(dt=dt, theta=theta)
    robin = {f: h for f in ('x-', 'x+', 'y-', 'y+', 'z-', 'z+')}
    st = dist_slab.SlabStepper(mask[i0:i1], dx, mat, params, Tinf, robin_h=robin, comm=comm, engine=engine)
    T = np.full((i1 - i0, ny, nz), float(Tinf), dtype=np.float64)
    on_device = getattr(st.engine.device, 'type', 'cpu') == 'cuda'
    if on_device:
        import torch
        T = torch.from_numpy(T).to(st.engine.device)
    lx0, lx1 = max(x0, i0) - i0, min(x1, i1) - i0            # the track box inside this slab (empty when lx0 >= lx1)
    dev_loop = on_device and st.device_births_supported()      # the mask stays in HBM: a column is a slice assignment
    for yi in range(ncol):
        if dev_loop:
            if lx0 < lx1:
                dist_slab._interior(st.d_mask_ext)[lx0:lx1, yi:yi + 1, z0:z1] = 1
            st.set_mask_device(z0, z1)                         # collective: halo planes of the mask, flags and packs in place
        else:
            mask[x0:x1, yi:yi + 1, z0:z1] = True
            st.set_mask(mask[i0:i1])                           # collective: every rank, whether the column touches it or not
        if lx0 < lx1:
            if on_device:
                T[lx0:lx1, yi:yi + 1, z0:z1] = T_track         # in place: the last sub-step sent no halo ahead
            else:
                Tl = np.array(st.local_numpy(T))
                Tl[lx0:lx1, yi:yi + 1, z0:z1] = T_track
                T = Tl
        n_sub = max(1, int(math.ceil(t_step / dt)))
        params.dt = t_step / n_sub
        for s_ in range(n_sub):
            T = st.step(T, prefetch_halo=(s_ + 1 < n_sub))
        params.dt = dt
    return np.array(st.local_numpy(T))
