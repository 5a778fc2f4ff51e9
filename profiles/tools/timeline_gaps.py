#!/usr/bin/env python
"""Where does the wall clock of the timed rollout go?  Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and, for the
LAST block of `steps` env steps forward + backward in it (bench.py --no-roofline: the timed region), prints: wall span, the sum of
kernel durations by kernel, and the idle time between consecutive kernels grouped by what ran before the gap."""
import csv
import collections
import re
import sys

path, steps, sub = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("plb::", "")))
rows.sort()
# the timed region: the last steps * (sub - 1) fused forward launches and everything up to the last p2g_grad
idx_fwd = [i for i, r in enumerate(rows) if r[2].startswith("k_g2p_p2g")]
idx_bwd = [i for i, r in enumerate(rows) if r[2].startswith("k_p2g_grad")]
first = idx_fwd[-steps * (sub - 1)]
# walk back to the k_p2g that opens that env step (and its kinematics kernels)
while first > 0 and not rows[first][2].startswith("k_fk_chain<"):
    first -= 1
last = idx_bwd[-1]
while last + 1 < len(rows) and rows[last + 1][2].startswith(("k_fk_chain_grad", "k_set_action")):
    last += 1
seg = rows[first:last + 1]
span = (seg[-1][1] - seg[0][0]) * 1e-3
busy = collections.Counter()
calls = collections.Counter()
gaps = collections.Counter()
ngaps = collections.Counter()
end = seg[0][0]
prev = None
for s, e, name in seg:
    busy[name] += (e - s) * 1e-3
    calls[name] += 1
    if prev is not None:
        g = (s - end) * 1e-3
        if g > 0:
            gaps[prev] += g
            ngaps[prev] += 1
    end = max(end, e)
    prev = name
n_sub = steps * sub
print(f"timed region: {len(seg)} launches, span {span / 1e3:.3f} ms = {span / n_sub:.2f} us per fwd+bwd substep; kernels {sum(busy.values()) / n_sub:.2f} us, idle {sum(gaps.values()) / n_sub:.2f} us per substep")
print("kernel                                   calls   us/substep   avg us | idle after it: us/substep   avg us")
for name, t in busy.most_common():
    print(f"{name[:40]:40s} {calls[name]:6d} {t / n_sub:10.2f} {t / calls[name]:9.2f} | {gaps[name] / n_sub:12.2f} {gaps[name] / max(ngaps[name], 1):10.2f}")
