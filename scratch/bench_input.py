import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.data import preprocess_u8, DevicePrefetcher
dev = torch.device("cuda", 0)
for B in (64, 256):
    x = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device=dev); out = torch.empty(B, 3, 224, 224, device=dev)
    fl = torch.randint(0, 4, (B,), dtype=torch.uint8, device=dev)
    for _ in range(3): preprocess_u8(x, fl, out=out)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): preprocess_u8(x, fl, out=out)
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) * 1e3 / 50
    by = B * 224 * 224 * 15
    print(f"preprocess_u8 B={B}: {us:6.1f} us  {by/us/1e6:5.2f} TB/s (3 B read + 12 B written per pixel)")
# host -> device rate through the prefetcher (pinned staging + copy stream), 20 batches of 64
host = [(torch.randint(0, 256, (64, 224, 224, 3), dtype=torch.uint8), torch.zeros(64)) for _ in range(4)]
pf = DevicePrefetcher(host * 5, dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for imgs, lab in pf: n += imgs.shape[0]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"DevicePrefetcher: {n/dt:8.0f} img/s host->device incl. pinned staging copy ({n*224*224*3/dt/1e9:.1f} GB/s of uint8)")

host = [(h[0].pin_memory(), h[1]) for h in host]
pf = DevicePrefetcher(host * 5, dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for imgs, lab in pf: n += imgs.shape[0]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"DevicePrefetcher, loader yields pinned frames: {n/dt:8.0f} img/s ({n*224*224*3/dt/1e9:.1f} GB/s of uint8)")
