"""adi_thermal_fields_amd -- MI355X-native ADI heat-equation time-stepper.

Drop-in for the hot path of Matemusi/ADI_thermal_fields:

    import adi_thermal_fields_amd.adi3d_hip_coeff as adi      # replaces adi3d_numba_coeff / adi3d_gpu_coeff
    import adi_thermal_fields_amd.adi3d_hip_cyl as cyl        # replaces adi3d_cyl_phi_v3 (scheme="be")

Python host code over a ctypes C ABI (include/adi_hip.h) over hand-written HIP kernels for gfx950.
Importing a backend module without the built library raises ImportError (no CPU fallback).
"""
__version__ = "0.1.0"
