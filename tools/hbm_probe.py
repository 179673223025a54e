import torch, time
dev = torch.device("cuda:0")
n = 1 << 30   # 1 GiB of bf16 = 2 GiB
a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
b = torch.empty_like(a)
c = torch.empty_like(a)
def timeit(f, it=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
t = timeit(lambda: b.copy_(a)); print(f"copy 2 GiB -> 2 GiB: {t*1e3:.0f} us, {2 * a.numel() * 2 / t / 1e9:.2f} TB/s (R+W)")
t = timeit(lambda: torch.add(a, b, out=c)); print(f"add (2R + 1W): {t*1e3:.0f} us, {3 * a.numel() * 2 / t / 1e9:.2f} TB/s")
t = timeit(lambda: a.sum()); print(f"sum (1R): {t*1e3:.0f} us, {a.numel() * 2 / t / 1e9:.2f} TB/s")
t = timeit(lambda: b.fill_(1.0)); print(f"fill (1W): {t*1e3:.0f} us, {a.numel() * 2 / t / 1e9:.2f} TB/s")
