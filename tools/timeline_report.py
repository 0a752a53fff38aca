"""Phase timeline of k_lvc_h2<256> from tools/ubench/lvc_h2_timeline (one 10 x int64 record per workgroup: s_memrealtime at the
eight phase boundaries of wave 0, HW_ID, XCC_ID).  Prints the steady-state phase lengths, how many workgroups are alive / waiting
for their loads at a time, and how the two workgroups of a CU sit relative to each other."""
import sys

import numpy as np

NAMES = ["loads + staging", "residual to registers", "conv (MFMA) + y split", "halo columns", "barrier", "kernel split",
         "LVC (MFMA) + gate + stores"]


def report(path):
    d = np.fromfile(path, dtype=np.int64).reshape(-1, 10)
    st = (d[:, :8] - d[:, 0].min()) / 100.0                      # 100 MHz -> us
    hw, xcc = d[:, 8], d[:, 9] & 0xF
    cu = xcc * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xF)
    span = st[:, 7].max()
    steady = (st[:, 0] > 0.25 * span) & (st[:, 0] < 0.75 * span)
    dur = np.diff(st, axis=1)
    life = st[:, 7] - st[:, 0]
    print(f"{path}: {len(d)} workgroups on {len(np.unique(cu))} CUs, {span:.1f} us from the first stamp to the last")
    print(f"  lifetime of a workgroup (steady state): {life[steady].mean():.2f} us")
    for n, v in zip(NAMES, dur[steady].mean(0)):
        print(f"    {n:28s} {v:5.2f} us  {100 * v / life[steady].mean():4.1f} %")
    ts = np.arange(0.25 * span, 0.75 * span, 0.25)
    alive = np.array([((st[:, 0] <= t) & (st[:, 7] > t)).sum() for t in ts])
    loading = np.array([((st[:, 0] <= t) & (st[:, 1] > t)).sum() for t in ts])
    print(f"  alive at a time: {alive.mean():.0f}; in the load phase: mean {loading.mean():.0f}, min {loading.min()}, max {loading.max()}"
          f" (std {loading.std():.0f}) -- a convoy would swing between 0 and all of them")
    # the partner of a workgroup = the one on the same CU whose life overlaps its start
    frac = []
    order = np.argsort(st[:, 0])
    last = {}
    for i in order:
        if steady[i] and cu[i] in last:
            j = last[cu[i]]
            if st[j, 7] > st[i, 0]:
                frac.append((st[i, 0] - st[j, 0]) / life[j])
        last[cu[i]] = i
    frac = np.array(frac)
    hist, _ = np.histogram(frac, bins=10, range=(0, 1))
    print("  a workgroup starts when its CU partner is at this fraction of its own life (deciles):", (hist / hist.sum()).round(2).tolist())


if __name__ == "__main__":
    for p in sys.argv[1:]:
        report(p)
