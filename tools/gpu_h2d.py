import time, numpy as np, torch
n = 1492992000   # 4K x60 uint8 RGB
a = np.random.randint(0, 255, n, dtype=np.uint8)
t = torch.from_numpy(a)
torch.cuda.synchronize()
for name in ("pageable", "pageable2"):
    t0 = time.perf_counter(); d = t.to("cuda"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-10s %.1f ms  %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9)); del d
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(t.data_ptr(), n, 0); dt = time.perf_counter() - t0
print("hostRegister rc", rc, "%.1f ms" % (dt * 1e3))
for name in ("registered", "registered2"):
    t0 = time.perf_counter(); d = t.to("cuda"); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-10s %.1f ms  %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9)); del d
d = torch.empty(n, dtype=torch.uint8, device="cuda")
t0 = time.perf_counter(); d.copy_(t, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("registered non_blocking copy_ %.1f ms  %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
t0 = time.perf_counter(); rt.cudaHostUnregister(t.data_ptr()); print("unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
p = torch.empty(n, dtype=torch.uint8).pin_memory()
t0 = time.perf_counter(); p.copy_(t); dt = time.perf_counter() - t0
print("host memcpy into pinned %.1f ms %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
t0 = time.perf_counter(); d.copy_(p, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pinned -> device %.1f ms  %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
