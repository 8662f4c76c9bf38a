import torch, sys
torch.cuda.set_device(0)
for mb in (82, 164, 328, 655, 1310):
    n = mb * 1000 * 1000 // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    ts = []
    for _ in range(10):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts)//2]
    print(f"torch copy {mb} MB read + {mb} MB write: {t*1e3:.1f} us -> {2*n*4/t/1e9:.2f} TB/s")
