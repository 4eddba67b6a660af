import torch, time
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
a.random_(0, 255)
for n in (1 << 30, 1 << 28):
    x, y = a[:n], b[:n]
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): y.copy_(x)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print("torch device copy %d MiB: %.3f ms, read+write %.2f TB/s" % (n >> 20, ms, 2 * n / ms / 1e9))
