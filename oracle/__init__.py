"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's ADI hot path).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
The product package adi_thermal_fields_amd never imports anything from here.
"""
