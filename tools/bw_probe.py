import torch, sys
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (32, 134, 268, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    us = t(lambda: y.copy_(x))
    us2 = t(lambda: x.mul_(1.0001))
    us3 = t(lambda: x.sum())
    print(f"{mb:5d} MB: copy {us:8.1f} us = {2*mb*1.048576/us*1e3/1e3:6.2f} TB/s | inplace mul {us2:8.1f} us = {2*mb*1.048576/us2:6.2f} TB/s | sum(read) {us3:8.1f} us = {mb*1.048576/us3:6.2f} TB/s")
