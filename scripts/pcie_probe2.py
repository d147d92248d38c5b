"""host <-> device transfer options for the NumPy-in / NumPy-out path at 512^3 (1 GiB fields)"""
import time, numpy as np, torch
n = 512
a = np.random.default_rng(0).uniform(20, 1000, (n, n, n))
dev = torch.device('cuda')
d = torch.empty((n, n, n), dtype=torch.float64, device=dev)
def t(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print('pageable H2D  %.1f ms' % t(lambda: d.copy_(torch.from_numpy(a))))
pin = torch.empty((n, n, n), dtype=torch.float64, pin_memory=True)
print('cpu copy numpy -> pinned (torch copy_) %.1f ms, threads %d' % (t(lambda: pin.copy_(torch.from_numpy(a))), torch.get_num_threads()))
print('pinned H2D    %.1f ms' % t(lambda: d.copy_(pin, non_blocking=True)))
print('pinned D2H    %.1f ms' % t(lambda: pin.copy_(d, non_blocking=True)))
out = np.empty_like(a)
print('pageable D2H  %.1f ms' % t(lambda: torch.from_numpy(out).copy_(d)))
rt = torch.cuda.cudart()
b = np.random.default_rng(1).uniform(20, 1000, (n, n, n))
t0 = time.perf_counter(); rc = rt.cudaHostRegister(b.ctypes.data, b.nbytes, 0); t1 = time.perf_counter()
print('hostRegister 1 GiB rc=%s %.1f ms' % (rc, (t1 - t0) * 1e3))
tb = torch.from_numpy(b)
print('registered H2D %.1f ms (is_pinned %s)' % (t(lambda: d.copy_(tb, non_blocking=True)), tb.is_pinned()))
t0 = time.perf_counter(); rt.cudaHostUnregister(b.ctypes.data); print('unregister %.1f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); p2 = torch.empty((n, n, n), dtype=torch.float64, pin_memory=True); print('alloc pinned 1 GiB %.1f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); del p2; p3 = torch.empty((n, n, n), dtype=torch.float64, pin_memory=True); print('re-alloc pinned (cached) %.1f ms' % ((time.perf_counter() - t0) * 1e3))
# chunked double-buffered staging
ch = 16
bufs = [torch.empty((ch, n, n), dtype=torch.float64, pin_memory=True) for _ in range(2)]
evs = [torch.cuda.Event() for _ in range(2)]
cs = torch.cuda.Stream()
ta = torch.from_numpy(a)
def staged():
    with torch.cuda.stream(cs):
        for c in range(n // ch):
            b_ = bufs[c & 1]; evs[c & 1].synchronize()
            b_.copy_(ta[c * ch:(c + 1) * ch])
            d[c * ch:(c + 1) * ch].copy_(b_, non_blocking=True)
            evs[c & 1].record(cs)
    cs.synchronize()
print('staged double-buffered H2D %.1f ms' % t(staged))
