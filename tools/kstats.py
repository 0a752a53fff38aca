"""Average duration of chosen kernels from a rocprofv3 kernel_stats.csv:  python tools/kstats.py <csv> [label]
One line: the predictor GEMM, the four hop-8 LVC layers (by dilation), their sum, and the time of all kernels per sample call."""
import csv
import re
import sys


def main():
    path, label = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    rows = list(csv.DictReader(open(path)))
    pick = {}
    total = 0.0
    calls_gemm = 1
    for r in rows:
        n, avg, calls = r["Name"], float(r["AverageNs"]) * 1e-3, int(r["Calls"])
        if "fdk" not in n:
            continue
        total += float(r["TotalDurationNs"]) * 1e-3
        m = re.search(r"k_lvc_h8m<(\d+)", n)
        if m:
            pick["h8_d" + m.group(1)] = avg
        elif "k_kp_gemm_h2" in n or "k_kp_gemm_w" in n:
            pick["gemm"] = avg
            calls_gemm = calls
        elif "k_first_conv" in n:
            pick["first"] = avg
        elif "k_dblock_h2<4" in n:
            pick["dblock4"] = avg
        else:
            m = re.search(r"k_lvc_h2<(\d+), *(\d+), *(\w+), *(\d+)", n)
            if m:
                pick["h%s_d%s%s%s" % (m.group(1), m.group(2), "F" if m.group(3) in ("true", "1") else "", "U" if int(m.group(4)) else "")] = avg
    h8 = sum(v for k, v in pick.items() if k.startswith("h8_"))
    keys = ["gemm", "h8_d1", "h8_d3", "h8_d9", "h8_d27"]
    print("%-14s " % label + "  ".join("%s %.1f" % (k, pick.get(k, float("nan"))) for k in keys) +
          "  | gemm+h8 %.1f  | all kernels per step %.1f us" % (pick.get("gemm", 0.0) + h8, total / max(calls_gemm, 1)) +
          "  | " + "  ".join("%s %.1f" % (k, v) for k, v in sorted(pick.items()) if k.startswith("h64") or k.startswith("h256") or k in ("first", "dblock4")))


if __name__ == "__main__":
    main()
