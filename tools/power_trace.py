"""Socket power and shader clock of the GPU while a command runs, sampled through librocm_smi64 as fast as the library answers
(show with a counter which kernels sit at the power cap).

    python tools/power_trace.py out.csv -- <command ...>

Parent process: rsmi only (no HIP context).  Prints mean / p95 / max power, the clock's distribution and the sampling rate; the CSV
holds every sample (t_ms, power_W, sclk_MHz).  The power cap is read once (rsmi_dev_power_cap_get)."""
import ctypes as ct
import subprocess
import sys
import time


class Freqs(ct.Structure):
    _fields_ = [("has_deep_sleep", ct.c_bool), ("num_supported", ct.c_uint32), ("current", ct.c_uint32), ("frequency", ct.c_uint64 * 33)]


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    smi = ct.CDLL("/opt/rocm/lib/librocm_smi64.so")
    rc = smi.rsmi_init(ct.c_uint64(0))
    dev = 0
    cap = ct.c_uint64(0)
    if rc != 0:      # no library access: the rocm-smi command line instead (a few samples per second)
        print("power_trace: rsmi_init -> %d, falling back to the rocm-smi command line" % rc)
        import json

        def cli():
            try:
                d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)
                card = d.get("card0", next(iter(d.values())))
                p = next((float(v) for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), float("nan"))
                c = next((float(str(v).strip("()Mhz")) for k, v in card.items() if "sclk" in k.lower()), float("nan"))
                return p, c
            except Exception:      # noqa: BLE001
                return float("nan"), float("nan")
        child = subprocess.Popen(cmd)
        t0 = time.perf_counter()
        rows = []
        while child.poll() is None:
            p, c = cli()
            rows.append(((time.perf_counter() - t0) * 1e3, p, c))
        with open(out, "w") as fh:
            fh.write("t_ms,power_W,sclk_MHz\n")
            for r in rows:
                fh.write("%.2f,%.1f,%.0f\n" % r)
        pw = [r[1] for r in rows if r[1] == r[1]]
        print("  %d samples; power max %.0f W, mean %.0f W" % (len(rows), max(pw) if pw else float("nan"), sum(pw) / max(len(pw), 1)))
        sys.exit(child.returncode)
    smi.rsmi_dev_power_cap_get(dev, 0, ct.byref(cap))

    def power():
        p, typ = ct.c_uint64(0), ct.c_int(0)
        if smi.rsmi_dev_power_get(dev, ct.byref(p), ct.byref(typ)) == 0:
            return p.value * 1e-6
        if smi.rsmi_dev_current_socket_power_get(dev, ct.byref(p)) == 0:
            return p.value * 1e-6
        if smi.rsmi_dev_power_ave_get(dev, 0, ct.byref(p)) == 0:
            return p.value * 1e-6
        return float("nan")

    f = Freqs()

    def sclk():
        if smi.rsmi_dev_gpu_clk_freq_get(dev, 0, ct.byref(f)) == 0 and f.current < 33:
            return f.frequency[f.current] * 1e-6
        return float("nan")

    idle = [(power(), sclk()) for _ in range(20)]
    child = subprocess.Popen(cmd)
    t0 = time.perf_counter()
    rows = []
    while child.poll() is None:
        rows.append(((time.perf_counter() - t0) * 1e3, power(), sclk()))
    dt = time.perf_counter() - t0
    with open(out, "w") as fh:
        fh.write("t_ms,power_W,sclk_MHz\n")
        for r in rows:
            fh.write("%.2f,%.1f,%.0f\n" % r)
    pw = sorted(r[1] for r in rows if r[1] == r[1])
    ck = sorted(r[2] for r in rows if r[2] == r[2])
    # the busy part: samples above the midpoint between idle and peak power
    idle_p = sum(p for p, _ in idle) / len(idle)
    busy = [r for r in rows if r[1] == r[1] and pw and r[1] > idle_p + 0.5 * (pw[-1] - idle_p)]
    q = lambda v, x: v[min(len(v) - 1, int(x * len(v)))] if v else float("nan")
    print("power_trace: %d samples in %.2f s (%.0f Hz); power cap %.0f W; idle before the run %.0f W / %.0f MHz" %
          (len(rows), dt, len(rows) / max(dt, 1e-9), cap.value * 1e-6, idle_p, idle[0][1]))
    print("  whole run : power mean %.0f W, p50 %.0f, p95 %.0f, max %.0f;  sclk p05 %.0f, p50 %.0f, p95 %.0f MHz" %
          (sum(pw) / max(len(pw), 1), q(pw, 0.5), q(pw, 0.95), pw[-1] if pw else float("nan"), q(ck, 0.05), q(ck, 0.5), q(ck, 0.95)))
    if busy:
        bp = sorted(r[1] for r in busy)
        bc = sorted(r[2] for r in busy if r[2] == r[2])
        print("  busy part : %d samples (%.2f s); power mean %.0f W, p05 %.0f, p50 %.0f, p95 %.0f;  sclk mean %.0f, p05 %.0f, p50 %.0f, p95 %.0f MHz" %
              (len(busy), (busy[-1][0] - busy[0][0]) * 1e-3, sum(bp) / len(bp), q(bp, 0.05), q(bp, 0.5), q(bp, 0.95), sum(bc) / max(len(bc), 1), q(bc, 0.05), q(bc, 0.5), q(bc, 0.95)))
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
