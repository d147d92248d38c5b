"""oracle/perimeter_oracle.py -- TEST INFRASTRUCTURE, never shipped: the exposed-face count of the reference's lateral
perimeter correction restated cell by cell (quick_compare_layer_birth_robin_v3.py:97-108).  Pinned against the
imported reference function's behaviour by construction (the same four tests per cell); small masks only (Python loops)."""


def count_exposed_faces(mask2d):
    """quick_compare_layer_birth_robin_v3.py:97-108"""
    nx, ny = mask2d.shape
    cnt = 0
    for i in range(nx):
        for j in range(ny):
            if not mask2d[i, j]:
                continue
            if i - 1 < 0 or not mask2d[i - 1, j]:
                cnt += 1
            if i + 1 >= nx or not mask2d[i + 1, j]:
                cnt += 1
            if j - 1 < 0 or not mask2d[i, j - 1]:
                cnt += 1
            if j + 1 >= ny or not mask2d[i, j + 1]:
                cnt += 1
    return cnt
