"""Can the host feed one MI355X?  Throughput of the reference's train transform after JPEG decoding, (a) on the host with the
library routines the reference uses (Pillow for resize / jitter / rotation; the 25-tap blur as a torch CPU conv, as torchvision
does), per core and on all granted cores, and (b) on the device with ssl4polyp_amd.data.DeviceAugmenter.
usage: python scratch/bench_input_aug.py [B]"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image, ImageEnhance
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (B, 576, 720, 3), dtype=np.uint8)   # Hyperkvasir-like decoded frames


def host_one(a, g):
    im = Image.fromarray(a).resize((224, 224), Image.BILINEAR)
    fns = [lambda i: ImageEnhance.Brightness(i).enhance(float(g.uniform(0.6, 1.4))), lambda i: ImageEnhance.Contrast(i).enhance(float(g.uniform(0.5, 1.5))),
           lambda i: ImageEnhance.Color(i).enhance(float(g.uniform(0.75, 1.25))), lambda i: Image.merge("HSV", i.convert("HSV").split()).convert("RGB")]
    for k in g.permutation(4):
        im = fns[k](im)
    t = torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float()[None]
    sg = float(g.uniform(0.001, 2.0))
    x = torch.linspace(-12, 12, 25)
    k1 = torch.exp(-0.5 * (x / sg) ** 2); k1 = k1 / k1.sum()
    k2 = (k1[:, None] * k1[None, :])[None, None].expand(3, 1, 25, 25)
    t = torch.nn.functional.conv2d(torch.nn.functional.pad(t, (12, 12, 12, 12), mode="reflect"), k2, groups=3).round().to(torch.uint8)
    im = Image.fromarray(t[0].permute(1, 2, 0).numpy())
    if g.random() < 0.5: im = im.transpose(Image.FLIP_LEFT_RIGHT)
    if g.random() < 0.5: im = im.transpose(Image.FLIP_TOP_BOTTOM)
    im = im.rotate(float(g.uniform(-180, 180)), Image.NEAREST)
    return (torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float().div(255) - 0.45) / 0.225


torch.set_num_threads(1)
g = np.random.default_rng(1)
t0 = time.perf_counter()
for i in range(B):
    host_one(frames[i], g)
dt = time.perf_counter() - t0
cores = len(os.sched_getaffinity(0))
print(f"host (Pillow + torch CPU conv), 1 core: {B / dt:.1f} img/s  -> {cores} cores, perfect scaling: {B / dt * cores:.0f} img/s")
if torch.cuda.is_available():
    from ssl4polyp_amd.data import DeviceAugmenter
    dev = torch.device("cuda", 0)
    aug = DeviceAugmenter(dev)
    xd = torch.from_numpy(frames).to(dev)
    gen = torch.Generator().manual_seed(0)
    for _ in range(3): aug(xd, generator=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 20
    for _ in range(n): aug(xd, generator=gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"device (DeviceAugmenter, incl. host parameter draws / uploads), B={B}: {B * n / dt:.0f} img/s, {dt / n * 1e3:.2f} ms per batch")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p = __import__("ssl4polyp_amd.data", fromlist=["x"]).draw_train_params(B, gen)
    e0.record()
    for _ in range(n): aug(xd, params=p)
    e1.record(); torch.cuda.synchronize()
    print(f"device kernels only: {e0.elapsed_time(e1) / n:.3f} ms per batch of {B} = {B * n / e0.elapsed_time(e1) * 1e3:.0f} img/s")
