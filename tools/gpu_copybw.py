import torch, time
def bench(name, fn, nbytes, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("%-34s %.2f ms -> %.2f TB/s" % (name, dt * 1e3, nbytes / dt / 1e12))
n = 2 * 1024**3
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
b = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
bench("copy fp32 8+8 GiB (r+w)", lambda: y.copy_(x), 2 * n * 4)
bench("fill fp32 8 GiB (w only)", lambda: y.fill_(1.5), n * 4)
bench("sum fp32 8 GiB (r only)", lambda: x.sum(), n * 4)
bench("u8->fp32 convert 2r+8w GiB", lambda: y.copy_(b), n * 5)
bench("mul fp32 in place (r+w same)", lambda: x.mul_(1.0001), 2 * n * 4)
