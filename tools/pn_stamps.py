"""Dev tool: reads the per-workgroup stamp dump of an -DT2L_EXP_PN_STAMPS build (T2L_PN_STAMPS=<file>) of pn_sa_kernel<128,256,256,64>
(round 5's measurement of the one-wave-per-SIMD form; the stamp hooks went with that kernel, DESIGN 3.6 keeps the result):
per workgroup [start, prologue end, self-round end, end, 4 round starts, hw id] in 10 ns ticks. Prints the phase durations and, per CU,
how much of the launch the CU had a workgroup resident (gaps between consecutive workgroups on the same CU)."""
import sys
from collections import defaultdict

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 16)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
dur = (a[:, 3] - a[:, 0]) * 0.01
print(f"{len(a)} workgroups; launch span {(a[:, 3].max() - t0) * 0.01:.1f} us; workgroup duration mean {dur.mean():.1f} us (min {dur.min():.1f}, max {dur.max():.1f})")
print(f"  prologue (loads + FPS) {((a[:, 1] - a[:, 0]) * 0.01).mean():.1f} us; self round {((a[:, 2] - a[:, 1]) * 0.01).mean():.1f} us; "
      f"centre rounds {((a[:, 3] - a[:, 2]) * 0.01).mean():.1f} us; round 1..3: "
      + ", ".join(f"{((a[:, 5 + i] - a[:, 4 + i]) * 0.01).mean():.2f}" for i in range(3)) + " us")
hw = a[:, 8]
xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
key = xcc * 10000 + se * 100 + sh * 50 + cu
per = defaultdict(list)
for k, s, e in zip(key, a[:, 0], a[:, 3]):
    per[int(k)].append((int(s), int(e)))
busy, gaps, conc = [], [], 0
for k, v in per.items():
    v.sort()
    b = sum(e - s for s, e in v)
    busy.append(b * 0.01)
    for (s0, e0), (s1, e1) in zip(v, v[1:]):
        gaps.append((s1 - e0) * 0.01)
        conc += s1 < e0
span = (a[:, 3].max() - t0) * 0.01
print(f"{len(per)} CUs seen; workgroups per CU {len(a) / len(per):.1f}; resident time per CU / span: mean {np.mean(busy) / span:.2f} "
      f"(min {np.min(busy) / span:.2f}, max {np.max(busy) / span:.2f}); overlapping pairs {conc}")
g = np.array(gaps)
print(f"gap between consecutive workgroups of a CU: mean {g.mean():.2f} us, median {np.median(g):.2f}, p90 {np.percentile(g, 90):.2f}, max {g.max():.1f}")
