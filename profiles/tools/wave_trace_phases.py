"""p50 / p90 cycles between consecutive phase marks of the traced kernels (gpurun_out/trace.npy from wave_trace_run.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'trace.npy')).astype(np.int64)
for ki, name in ((0, 'g2p_p2g'), (1, 'g2p_grad'), (2, 'p2g_grad')):
    a = t[ki]
    a = a[(a[:, 0] > 0) & (a[:, 0] < 10**17)]                 # rows a wave wrote (the buffer is not cleared: the rest is stale data)
    # (the counters of the eight XCDs are not aligned: only differences inside a row mean anything)
    # marks that were stamped (non-zero), in column order
    cols = [c for c in range(1, 11) if (a[:, c] > 0).mean() > 0.5]
    good = np.all(a[:, cols] >= a[:, [0]], axis=1) & np.all(a[:, cols] - a[:, [0]] < 10_000_000, axis=1)
    a = a[good]
    out, prev = [], a[:, 0]
    for c in cols:
        d = a[:, c] - prev
        prev = a[:, c]
        out.append(f"m{c - 1}: {int(np.median(d))}/{int(np.percentile(d, 90))}")
    life = a[:, cols[-1]] - a[:, 0] if cols else np.zeros(1)
    print(f"{name}: waves {len(a)} lifetime p50 {int(np.median(life))} p90 {int(np.percentile(life, 90))} | " + "  ".join(out))
