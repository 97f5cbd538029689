import torch, time
dev = torch.device("cuda", 0)
for mb in (10, 40, 160):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"H2D pinned {mb} MiB: {dt*1e3:.2f} ms  {mb/1024/dt:.1f} GiB/s")
    hp = torch.empty(mb << 20, dtype=torch.uint8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): d.copy_(hp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"H2D pageable {mb} MiB: {dt*1e3:.2f} ms  {mb/1024/dt:.1f} GiB/s")
