import torch, time
x = torch.empty(2 * 1024**3, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): y.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("torch copy 8 GiB: %.2f ms -> %.2f TB/s (read+write)" % (dt * 1e3, 2 * x.numel() * 4 / dt / 1e12))
for _ in range(2): s = x.sum()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): s = x.sum()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("torch sum 8 GiB: %.2f ms -> %.2f TB/s (read)" % (dt * 1e3, x.numel() * 4 / dt / 1e12))
