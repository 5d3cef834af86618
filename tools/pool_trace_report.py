"""Reads the table a -DHIPDEC_POOL_TRACE build of libheifhip.so appends to $HIPDEC_POOL_TRACE (decoder.hip: launch_all) and prints, for the LAST run in
the file, how the CABAC work pool fills and drains: per 42 ms bucket the share of wave-time spent waiting for work, and how many waves had run their last
row x ms before the kernel's end.  usage: python tools/pool_trace_report.py <file>"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8192, 32)
print("runs in the file:", a.shape[0])
r = a[-1].astype(np.float64)
used = r[:, 4] > 0
r = r[used]
tick = 1e-5  # ms per 100 MHz tick
t0 = r[:, 0].min()
end = r[:, 2].max()
print("waves that ran rows: %d, kernel span %.1f ms" % (used.sum(), (end - t0) * tick))
print("rows per wave: mean %.0f  min %.0f  max %.0f" % (r[:, 4].mean(), r[:, 4].min(), r[:, 4].max()))
span = (r[:, 2] - r[:, 0]).sum()
print("wave-time: in rows %.1f %%, waiting for work %.1f %%" % (100 * r[:, 5].sum() / span, 100 * r[:, 3].sum() / span))
print("waiting for work by 42 ms bucket of the run (share of 8192 waves x bucket):")
for k in range(24):
    w = r[:, 8 + k].sum()
    if w == 0 and k * (1 << 22) * tick > (end - t0) * tick: break
    print("  %4.0f - %4.0f ms  %5.1f %%" % (k * (1 << 22) * tick, (k + 1) * (1 << 22) * tick, 100 * w / (len(r) * (1 << 22))))
print("waves whose last row ended more than x ms before the kernel's end:")
for x in (1, 2, 5, 10, 20, 40, 60, 80, 100, 150, 200):
    print("  %4d ms: %5.1f %%" % (x, 100 * ((end - r[:, 1]) * tick > x).mean()))
