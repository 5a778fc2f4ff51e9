"""How a launch ends (gpurun_out/trace.npy from wave_trace_run.py): rows are grouped by clock domain (the shader clocks of different CU groups
are not aligned: clusters of start stamps), and per group -- 64 waves = 16 workgroups of two CUs in the headline run -- the span of the kernel,
the start of the last first-round wave, how long the last round of waves takes to drain, and how many waves are alive at 1/8 .. 7/8 of the span."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'trace.npy')).astype(np.int64)
for ki, name, cap in ((0, 'g2p_p2g', 32), (1, 'g2p_grad', 32), (2, 'p2g_grad', 16)):
    a = t[ki]
    ok = (a[:, 0] > 0) & (a[:, 0] < 10**17)
    cols = [c for c in range(1, 11) if (a[ok][:, c] > 0).mean() > 0.5]
    s, e = a[ok, 0], a[ok, cols[-1]]
    order = np.argsort(s)
    gaps = np.nonzero(np.diff(s[order]) > 300000)[0]
    bounds = np.concatenate([[0], gaps + 1, [len(order)]])
    spans, ramps, tails, alive = [], [], [], []
    for k in range(len(bounds) - 1):
        idx = order[bounds[k]:bounds[k + 1]]
        if len(idx) < 56 or len(idx) > 64:
            continue
        s0 = s[idx].min()
        st, en = np.sort(s[idx] - s0), np.sort(e[idx] - s0)
        spans.append(en[-1]); ramps.append(st[cap - 1]); tails.append(en[-1] - en[-cap])
        alive.append([int(((s[idx] - s0 <= tt) & (e[idx] - s0 > tt)).sum()) for tt in np.linspace(0, en[-1], 9)[1:-1]])
    print(f"{name}: {len(spans)} groups of ~64 waves ({cap} resident at once); span p50 {int(np.median(spans))} cycles; the {cap} first-round waves have "
          f"started after {int(np.median(ramps))}; the last {cap} waves end over {int(np.median(tails))} cycles; waves alive at 1/8..7/8 of the span (median): "
          f"{np.median(np.array(alive), axis=0).astype(int).tolist()}")
