#!/usr/bin/env python3
"""Occupancy timeline of the last diagonal fill recorded with SWG_TRACE=<file>.

usage: SWG_TRACE=gpurun_out/trace.txt python bench.py --steps 1 --warmup 0 ... ; python tools/trace_timeline.py gpurun_out/trace.txt
"""
import sys
import numpy as np

def main(path, buckets=28):
    fills, cur = [], None
    for line in open(path):
        if line.startswith("#"):
            cur = {"hdr": line[1:].strip(), "rows": []}
            fills.append(cur)
        elif cur is not None:
            cur["rows"].append([int(x) for x in line.split()])
    f = fills[-1]
    a = np.array(f["rows"], dtype=np.uint64).astype(np.int64)
    a = a[a[:, 4] > 0]
    packed = a[:, 5].copy()
    a[:, 5] = packed & 0xFFFFFFFF            # blocks
    events = (packed >> 32) & 0xFFF          # work-queue events of the wavefront
    ev_us = (packed >> 44) / 100.0           # time spent in them
    t0 = a[:, 3].min()
    start = (a[:, 3] - t0) / 100.0  # microseconds (100 MHz ticks)
    end = (a[:, 4] - t0) / 100.0
    total = end.max()
    print(f["hdr"], "| fills recorded:", len(fills))
    print("span %.1f us, waves %d" % (total, len(a)))
    for c in np.unique(a[:, 0]):
        m = a[:, 0] == c
        e, s, nb = end[m], start[m], a[m, 5]
        print("class %d: waves %d start[min %.1f med %.1f max %.1f] end[p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f] blocks[min %d med %d max %d]"
              % (c, m.sum(), s.min(), np.median(s), s.max(), *np.percentile(e, [10, 50, 90, 99]), e.max(), nb.min(), np.median(nb), nb.max()))
        dur = e - s
        rate = dur / np.maximum(nb, 1)
        print("         us per block of 4 rows: p10 %.3f p50 %.3f p90 %.3f" % tuple(np.percentile(rate, [10, 50, 90])))
        if events[m].max() > 0:
            print("         queue events per wave: med %d max %d; us in events per wave: p50 %.1f p90 %.1f max %.1f; us per event p50 %.2f"
                  % (np.median(events[m]), events[m].max(), *np.percentile(ev_us[m], [50, 90]), ev_us[m].max(),
                     np.median(ev_us[m] / np.maximum(events[m], 1))))
    edges = np.linspace(0, total, buckets + 1)
    print("active waves per time bucket (%.0f us each):" % (total / buckets))
    for c in np.unique(a[:, 0]):
        m = a[:, 0] == c
        act = [int(((start[m] < hi) & (end[m] > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        print("  class %d:" % c, " ".join("%5d" % x for x in act))

if __name__ == "__main__":
    main(sys.argv[1])
