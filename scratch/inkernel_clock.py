#!/usr/bin/env python3
"""In-kernel clock of the GEMM k-loops INSIDE the training step (MI355X_MICROARCH.md "DVFS give-back" (6)):
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped at kernel entry and at the end of the k-loop of every wave of the
ring GEMMs and of the grouped weight-gradient kernel -- diagnostic build only:

    python -c "import __graft_entry__ as g; g.build_library(lib=g.LIB.replace('.so', '_stamp.so'), extra_flags=['-DPM_GEMM_STAMP'])"
    POLYPMAE_LIB=ssl4polyp_amd/lib/libpolypmae_stamp.so python scratch/inkernel_clock.py [cls|mae|qkv|zeros] ...

cls / mae: >= 2 s of the benched step, then ONE step with the stamp buffer armed (every GEMM launch of the step writes its rows;
a row holds the stamps of the last launch that used that block index -- each row is self-consistent, which is all the quotient
needs).  qkv: the stand-alone forward qkv GEMM, back to back, random data; zeros: the same on all-zero operands (the guide: ~2.39 GHz).
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ssl4polyp_amd import _lib  # noqa: E402
from ssl4polyp_amd.engine import Kernels  # noqa: E402

ROWS_GEMM, ROWS_GROUP = 4096, 512


def clocks(buf, lo, hi):
    s = buf.view(-1, 8, 16)[lo:hi].cpu().double()
    dt_c, dt_r = s[:, :, 9] - s[:, :, 8], s[:, :, 13] - s[:, :, 12]
    ok = (dt_r > 200) & (dt_c > 0)  # >= 2 us of k-loop
    if ok.sum() == 0:
        return None
    mhz = (dt_c[ok] / dt_r[ok] * 100.0).sort().values
    us = (dt_r[ok] / 100.0).sort().values
    q = lambda v, f: float(v[min(len(v) - 1, int(len(v) * f))])
    return {"waves": int(ok.sum()), "clock_mhz": {"p10": round(q(mhz, .1)), "median": round(q(mhz, .5)), "p90": round(q(mhz, .9))},
            "entry_to_loop_end_us_median": round(q(us, .5), 1)}


def main():
    what = sys.argv[1:] or ["cls", "mae", "qkv", "zeros"]
    lib = _lib.load()
    lib.pm_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import ssl4polyp_amd
    ssl4polyp_amd.reserve_streams(dev)
    buf = torch.zeros((ROWS_GEMM + ROWS_GROUP) * 8 * 16, dtype=torch.int64, device=dev)
    out = {"lib": _lib.LIB_PATH, "method": "d(s_memtime)/d(s_memrealtime) x 100 MHz, kernel entry -> end of k-loop, per wave"}
    for w in what:
        if w in ("cls", "mae"):
            batch = 64 if w == "cls" else 256
            model, ddp, opt = bench.build(w, "bf16", dev, 1, batch)
            imgs, labels = bench.make_batch(w, batch, dev, 0)
            step = bench.make_step(w, ddp, opt, imgs, labels)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 3.0:
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                n += 10
            dt = (time.perf_counter() - t0) / n
            buf.zero_()
            torch.cuda.synchronize()
            lib.pm_debug_gemm_stamps(buf.data_ptr())
            step()
            torch.cuda.synchronize()
            lib.pm_debug_gemm_stamps(None)
            out[w] = {"ms_per_step_stamp_build": round(dt * 1e3, 3), "ring_gemms": clocks(buf, 0, ROWS_GEMM),
                      "wgrad_group": clocks(buf, ROWS_GEMM, ROWS_GEMM + ROWS_GROUP)}
            del model, ddp, opt, step
            torch.cuda.empty_cache()
        else:
            k = Kernels("bf16")
            M, D = 12608, 768
            mk = (lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)) if w == "zeros" else \
                (lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16))
            x, W, o = mk(M, D), mk(3 * D, D), torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
            b = torch.zeros(3 * D, device=dev)
            fn = lambda: k.linear_fwd(x, W, b, o, M, 3 * D, D)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 2.5:
                for _ in range(200):
                    fn()
                torch.cuda.synchronize()
                n += 200
            dt = (time.perf_counter() - t0) / n
            buf.zero_()
            torch.cuda.synchronize()
            lib.pm_debug_gemm_stamps(buf.data_ptr())
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            lib.pm_debug_gemm_stamps(None)
            out["qkv_gemm_" + ("zeros" if w == "zeros" else "random")] = {"us_per_launch_back_to_back": round(dt * 1e6, 2),
                                                                          "ring_gemms": clocks(buf, 0, ROWS_GEMM)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
