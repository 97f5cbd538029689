import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4polyp_amd.data import DevicePrefetcher
dev = torch.device("cuda", 0)
host = [(torch.randint(0, 256, (64, 224, 224, 3), dtype=torch.uint8).pin_memory(), torch.zeros(64).pin_memory()) for _ in range(4)]
for _ in DevicePrefetcher(host, dev): pass
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for imgs, lab in DevicePrefetcher(host * 5, dev, flip_p=0.5): pass
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
