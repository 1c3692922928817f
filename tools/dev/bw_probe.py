import torch, time
for mb in (291, 1164):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): y.copy_(x)
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"copy {mb} MB: {ms*1e3:.1f} us -> {2*mb*1.048576/ms:.0f} GB/s (read+write)")
    a.record()
    for _ in range(20): torch.relu_(x)
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"in-place relu {mb} MB: {ms*1e3:.1f} us -> {2*mb*1.048576/ms:.0f} GB/s")
