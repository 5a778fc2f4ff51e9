"""Per-workgroup lifetime along the launch index (gpurun_out/trace.npy from wave_trace_run.py): does the cost of a workgroup depend on where its
chunk lies on the storage order?  Prints p50 wave lifetime and p50 of each phase by index decile of the fused forward kernel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'trace.npy')).astype(np.int64)
for ki, name in ((0, 'g2p_p2g'), (1, 'g2p_grad'), (2, 'p2g_grad')):
    a = t[ki]
    idx = np.arange(len(a))
    ok = (a[:, 0] > 0) & (a[:, 0] < 10**17)
    cols = [c for c in range(1, 11) if (a[ok][:, c] > 0).mean() > 0.5]
    ok &= np.all(a[:, cols] >= a[:, [0]], axis=1) & np.all(a[:, cols] - a[:, [0]] < 10_000_000, axis=1)
    a, idx = a[ok], idx[ok]
    life = a[:, cols[-1]] - a[:, 0]
    n = idx.max() + 1
    print(f"{name}: {len(a)} waves, lifetime p50 {int(np.median(life))}")
    for d in range(10):
        m = (idx >= d * n // 10) & (idx < (d + 1) * n // 10)
        ph = [int(np.median(a[m][:, c] - (a[m][:, cols[i - 1]] if i else a[m][:, 0]))) for i, c in enumerate(cols)]
        print(f"  index decile {d}: life p50 {int(np.median(life[m])):6d} p90 {int(np.percentile(life[m], 90)):6d}  phases {ph}")
