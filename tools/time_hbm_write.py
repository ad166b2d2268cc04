import torch, time
x = torch.empty(2_000_000_000, dtype=torch.float64, device="cuda")
for name, fn in (("fill_", lambda: x.fill_(1.0)), ("zero_", lambda: x.zero_()), ("copy_", None)):
    if fn is None:
        y = torch.empty(1_000_000_000, dtype=torch.float64, device="cuda"); fn = lambda: x[:1_000_000_000].copy_(y)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(name, round(ms, 3), "ms", round(16e9 / ms / 1e9, 2) if name != "copy_" else round(16e9 / ms / 1e9, 2), "TB/s (bytes moved 16 GB)")
