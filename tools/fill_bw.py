import torch, time
n = 692060160
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("fill (write only)", lambda: x.zero_()), ("copy (read+write)", lambda: y.copy_(x))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(name, round(ms, 4), "ms", round(n / ms / 1e9, 3), "TB/s written")
