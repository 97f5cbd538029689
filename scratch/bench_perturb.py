"""Eval-time perturbations (transforms.py:143-203) of a 224 x 224 batch: DevicePerturber against PIL on one host core.
usage: python scratch/bench_perturb.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image, ImageEnhance, ImageFilter, ImageDraw
from ssl4polyp_amd import data as D
dev = torch.device("cuda", 0)
B = 256
x = np.random.Generator(np.random.PCG64(1)).integers(0, 256, (B, 224, 224, 3), dtype=np.uint8)
for kind, mk in (("blur sigma 2", lambda b: {"variant": "blur_2"}), ("bc 1.2 / 0.8", lambda b: {"variant": "bc_b1p2_c0p8"}),
                 ("occ 0.15", lambda b: {"variant": "occ_a0p15", "frame_id": b}), ("jpeg 30", lambda b: {"variant": "jpeg_30"}),
                 ("mixed", lambda b: [{"variant": "blur_2"}, {"variant": "bc_b1p2_c0p8"}, {"variant": "occ_a0p15", "frame_id": b}, {"variant": "clean"}, {"variant": "jpeg_30"}][b % 5])):
    rows = [mk(b) for b in range(B)]
    pert = D.DevicePerturber(dev)
    xd = torch.from_numpy(x).to(dev)
    for _ in range(3): pert(xd, rows)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): pert(xd, rows)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    m = 64
    for b in range(m):
        im = Image.fromarray(x[b]); p = D.perturbation_plan(rows[b])
        if p[0] == "blur": im = im.filter(ImageFilter.GaussianBlur(radius=p[1]))
        elif p[0] == "bc":
            im = ImageEnhance.Brightness(im).enhance(p[1]); im = ImageEnhance.Contrast(im).enhance(p[2])
        elif p[0] == "jpeg":
            import io
            buf = io.BytesIO(); im.save(buf, format="JPEG", quality=p[1], optimize=False, subsampling=0); buf.seek(0); im = Image.open(buf).convert("RGB")
        elif p[0] == "occ":
            r = D.occlusion_rect(p[1], p[2], 224, 224); im = im.copy(); ImageDraw.Draw(im).rectangle(list(r), fill=(0, 0, 0))
        np.asarray(im)
    t_host = (time.perf_counter() - t0) / m
    print(f"{kind:14s}: device {B / t_dev:10.0f} img/s ({t_dev * 1e3:.2f} ms per {B}, host planning included), PIL one core {1 / t_host:8.0f} img/s")
