"""Phase timeline of the fused hop-256 variants from tools/ubench/lvc_h2_timeline_v{0,1,2} (round 6): mean time of every stamp after the
workgroup's start, steady state; slots 1..4 are the variant's own phases (FD_STAMP_X), 5 the staging barrier, 6 LVC start, 7 the end."""
import sys

import numpy as np

LABELS = {
    0: {5: "staging barrier", 6: "LVC start (conv + halo + barrier + kernel split done)", 7: "end"},
    1: {1: "UP: xin image staged, barrier 1 passed", 2: "UP: matrix tiles + parking done", 3: "UP: barrier 2 passed", 4: "UP: x read back from the parking area",
        5: "staging barrier", 6: "LVC start", 7: "end"},
    2: {5: "staging barrier", 6: "LVC start", 1: "FINAL: LVC + gate done, fold starts", 2: "FINAL: 28 partial sums per lane written", 3: "FINAL: barrier passed",
        4: "FINAL: column sums stored", 7: "end"},
}


def report(path, variant):
    d = np.fromfile(path, dtype=np.int64).reshape(-1, 10)
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    st = (d[:, :8] - t0) / 100.0
    span = st[:, 7].max()
    steady = (st[:, 0] > 0.25 * span) & (st[:, 0] < 0.75 * span)
    rel = st - st[:, :1]
    print(f"{path}: {len(d)} workgroups, {span:.1f} us first stamp to last; lifetime of a workgroup (steady state) {rel[steady, 7].mean():.2f} us")
    prev = 0.0
    for slot, name in sorted(LABELS[variant].items(), key=lambda kv: rel[steady, kv[0]].mean()):
        m = rel[steady, slot].mean()
        print(f"    +{m:6.2f} us (phase {m - prev:5.2f})  {name}")
        prev = m


if __name__ == "__main__":
    report(sys.argv[1], int(sys.argv[2]))
