import torch, time
x = torch.empty(256*1024*1024, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for n in (8<<20, 64<<20, 256<<20):
    torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): d[:n].copy_(x[:n], non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    t=time.perf_counter()
    for _ in range(5): x[:n].copy_(d[:n], non_blocking=True)
    torch.cuda.synchronize(); dt2=(time.perf_counter()-t)/5
    print(n>>20, "MB: H2D %.1f GB/s  D2H %.1f GB/s" % (n/dt/1e9, n/dt2/1e9))
