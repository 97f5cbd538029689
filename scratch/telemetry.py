#!/usr/bin/env python3
"""Clock / power sampler for the question "what bounds the step" (VERDICT r3 item 4).

    python scratch/telemetry.py OUT.csv [--hz 50] [--seconds 600] &      # beside bench.py, killed by the caller

Reads, for EVERY device rocm_smi lists (the box shows the whole host; the busy one is picked afterwards), at --hz:
sclk / mclk (rsmi_dev_gpu_clk_freq_get: the DPM level the SMU reports), socket power (current, else average), power cap,
junction temperature, busy percent.  Never touches HIP.  The guide's caveat applies and is the reason the in-kernel clock is
measured separately (scratch/inkernel_clock.sh): `pp_dpm_sclk` reads up to ~10 % above the clock an MFMA loop really holds.
"""
import argparse
import ctypes
import signal
import sys
import time


class Freq(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * 33)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--hz", type=float, default=50.0)
    ap.add_argument("--seconds", type=float, default=900.0)
    a = ap.parse_args()
    lib = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
    if lib.rsmi_init(ctypes.c_uint64(0)) != 0:
        sys.exit("rsmi_init failed")
    n = ctypes.c_uint32(0)
    lib.rsmi_num_monitor_devices(ctypes.byref(n))
    ndev = n.value
    stop = [False]
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__(0, True))
    signal.signal(signal.SIGINT, lambda *_: stop.__setitem__(0, True))

    def clk(d, kind):
        f = Freq()
        if lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(d), ctypes.c_int(kind), ctypes.byref(f)) != 0 or f.current >= 33:
            return float("nan")
        return f.frequency[f.current] / 1e6

    def u64(fn, d, *extra):
        v = ctypes.c_uint64(0)
        if fn(ctypes.c_uint32(d), *extra, ctypes.byref(v)) != 0:
            return float("nan")
        return float(v.value)

    def i64(fn, d, *extra):
        v = ctypes.c_int64(0)
        if fn(ctypes.c_uint32(d), *extra, ctypes.byref(v)) != 0:
            return float("nan")
        return float(v.value)

    def busy(d):
        v = ctypes.c_uint32(0)
        if lib.rsmi_dev_busy_percent_get(ctypes.c_uint32(d), ctypes.byref(v)) != 0:
            return float("nan")
        return float(v.value)

    def membusy(d):
        v = ctypes.c_uint32(0)
        if lib.rsmi_dev_memory_busy_percent_get(ctypes.c_uint32(d), ctypes.byref(v)) != 0:
            return float("nan")
        return float(v.value)

    caps = [u64(lib.rsmi_dev_power_cap_get, d, ctypes.c_uint32(0)) / 1e6 for d in range(ndev)]
    with open(a.out, "w") as fh:
        fh.write("# power caps (W) per device: " + ",".join(f"{c:.0f}" for c in caps) + "\n")
        fh.write("t_unix,dev,sclk_mhz,mclk_mhz,power_w,temp_junction_c,busy_pct,mem_busy_pct\n")
        t_end = time.time() + a.seconds
        period = 1.0 / a.hz
        nxt = time.time()
        while not stop[0] and time.time() < t_end:
            for d in range(ndev):
                t = time.time()
                p = u64(lib.rsmi_dev_current_socket_power_get, d)
                if p != p:
                    p = u64(lib.rsmi_dev_power_ave_get, d, ctypes.c_uint32(0))
                tj = i64(lib.rsmi_dev_temp_metric_get, d, ctypes.c_uint32(1), ctypes.c_int(0)) / 1e3  # junction, current
                fh.write(f"{t:.4f},{d},{clk(d, 0):.0f},{clk(d, 4):.0f},{p / 1e6:.1f},{tj:.1f},{busy(d):.0f},{membusy(d):.0f}\n")
            fh.flush()
            nxt += period
            dt = nxt - time.time()
            if dt > 0:
                time.sleep(dt)
            else:
                nxt = time.time()
    lib.rsmi_shut_down()


if __name__ == "__main__":
    main()
