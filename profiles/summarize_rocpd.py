#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2) rocpd sqlite output: per-kernel launch count / average duration
(`--kernel-trace --stats` equivalent) and, when present, per-kernel PMC sums averaged per dispatch.

    python profiles/summarize_rocpd.py gpurun_out/prof/trace/r01_results.db [more.db ...] > profiles/rNN_xxx.txt
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:plb::)?(k_[a-z0-9_]+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else name.split("(")[0][:60]


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        tabs = {r[0].split("_0000")[0]: r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
        kd, ks, pe, ip = tabs["rocpd_kernel_dispatch"], tabs["rocpd_info_kernel_symbol"], tabs["rocpd_pmc_event"], tabs["rocpd_info_pmc"]
        names = {r[0]: short(r[1]) for r in cur.execute(f"select id, display_name from '{ks}'")}
        stat = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        ev2k = {}
        for kid, start, end, ev in cur.execute(f"select kernel_id, start, end, event_id from '{kd}'"):
            s = stat[names[kid]]
            d = (end - start) * 1e-3
            s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
            ev2k[ev] = names[kid]
        total = sum(s[1] for s in stat.values())
        print(f"# {path}")
        print(f"{'kernel':44s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for k, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:44s} {s[0]:7d} {s[1]:12.1f} {s[1] / s[0]:10.2f} {s[2]:10.2f} {s[3]:10.2f} {100 * s[1] / total:6.2f}")
        pmc_names = {r[0]: r[1] for r in cur.execute(f"select id, name from '{ip}'")}
        acc = defaultdict(lambda: defaultdict(float))
        for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from '{pe}'"):
            if ev in ev2k:
                acc[ev2k[ev]][pmc_names[pid]] += val
        if acc:
            ctrs = sorted({c for v in acc.values() for c in v})
            print("\n# PMC counters, average per dispatch")
            print(f"{'kernel':44s} " + " ".join(f"{c:>22s}" for c in ctrs))
            for k, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
                if k in acc:
                    print(f"{k:44s} " + " ".join(f"{acc[k].get(c, 0) / s[0]:22.1f}" for c in ctrs))
        print()


if __name__ == "__main__":
    main()
