"""Dev tool: per-block clocks of eight strips (SA_DEBUG_BLOCKS_FILE, -DSA_DEBUG_HANDOFF library): how evenly does a
strip advance?  Prints, per tracked strip, the distribution of its 64-step block times."""
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
stride, S = np.frombuffer(raw[:8], np.int32)
t = np.frombuffer(raw[8:], np.int64).reshape(8, stride).astype(np.float64) / 100.0  # us
wg = len(sys.argv) > 2 and sys.argv[2] == "wg"
mid = (S // 2) & ~3
picks = [0, 1, mid - 1, mid, mid + 1, mid + 2, mid + 3, mid + 4] if wg else [0, 1, 2, 3, S // 2, S // 2 + 1, S - 2, S - 1]
t0 = t[0, 0]
for q in range(8):
    d = np.diff(t[q][t[q] > 0])
    d = d[2:-2]
    pct = np.percentile(d, [1, 10, 50, 90, 99, 99.9])
    print(f"strip {picks[q]:4d}: first block at {t[q, 0] - t0:9.1f} us, {len(d)} blocks, mean {d.mean():.3f} us, min {d.min():.3f}, "
          f"p1/p10/p50/p90/p99/p99.9 {pct[0]:.3f}/{pct[1]:.3f}/{pct[2]:.3f}/{pct[3]:.3f}/{pct[4]:.3f}/{pct[5]:.3f}, max {d.max():.3f}; "
          f"time above 1.1 x median: {np.clip(d - 1.1 * pct[2], 0, None).sum():.1f} us")
# separation between consecutive tracked strips, per block index (how far behind, in us, is strip s+1 at the same block)
for a, b in (((0, 1), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)) if wg else ((0, 1), (1, 2), (2, 3), (4, 5), (6, 7))):
    m = (t[a] > 0) & (t[b] > 0)
    sep = (t[b] - t[a])[m]
    k = len(sep)
    print(f"strips {picks[a]}->{picks[b]}: separation at block 1 {sep[1]:.2f} us, 10% {sep[k // 10]:.2f}, 50% {sep[k // 2]:.2f}, 90% {sep[9 * k // 10]:.2f}, last {sep[-2]:.2f}")
