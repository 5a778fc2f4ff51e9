#!/usr/bin/env python3
"""rocprofv3 PMC passes of `python bench.py` -> profiles/rNN_pmc.json, the file bench.py reads roofline.traffic from.

    # on the GPU box, separate passes (a --pmc run must not be combined with the trace domains):
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o r --  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o r --  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d gpurun_out/pmc_sq -o r -- ...
    # here:
    python profiles/tools/pmc_summary.py --workload config3_cube128 --dtype f32 --steps 20 --warmup 5 --out profiles/r03_pmc.json \\
        gpurun_out/pmc_fetch/*/r_results.db gpurun_out/pmc_write/*/r_results.db [gpurun_out/pmc_sq/*/r_results.db]

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB): on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so a wide coalesced read is under-counted by exactly 2 (MI355X_MICROARCH.md, HBM section).
"""
import argparse
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:plb::)?(k_[a-z0-9_]+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else name.split("(")[0][:60]


def read_db(path):
    """-> {kernel: {"calls": n, "avg_us": t, counter: average per dispatch}}"""
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = {r[0].split("_0000")[0]: r[0] for r in cur.execute("select name from sqlite_master where type='table'")}
    kd, ks, pe, ip = tabs["rocpd_kernel_dispatch"], tabs["rocpd_info_kernel_symbol"], tabs["rocpd_pmc_event"], tabs["rocpd_info_pmc"]
    names = {r[0]: short(r[1]) for r in cur.execute(f"select id, display_name from '{ks}'")}
    calls, dur, ev2k = defaultdict(int), defaultdict(float), {}
    for kid, start, end, ev in cur.execute(f"select kernel_id, start, end, event_id from '{kd}'"):
        calls[names[kid]] += 1
        dur[names[kid]] += (end - start) * 1e-3
        ev2k[ev] = names[kid]
    pmc = {r[0]: r[1] for r in cur.execute(f"select id, name from '{ip}'")}
    acc = defaultdict(lambda: defaultdict(float))
    for ev, pid, val in cur.execute(f"select event_id, pmc_id, value from '{pe}'"):
        if ev in ev2k:
            acc[ev2k[ev]][pmc[pid]] += val
    return {k: dict({"calls": calls[k], "avg_us": dur[k] / calls[k]}, **{c: v / calls[k] for c, v in acc[k].items()}) for k in calls}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--dtype", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, required=True, help="--steps of the profiled bench.py command")
    ap.add_argument("--warmup", type=int, required=True, help="--warmup of the profiled bench.py command")
    a = ap.parse_args()
    merged = defaultdict(dict)
    for p in a.dbs:
        for k, v in read_db(p).items():
            merged[k].update({c: x for c, x in v.items() if c not in ("calls", "avg_us")})
            merged[k].setdefault("calls", v["calls"])
    # bench.py names kernels without the k_ prefix and template arguments ("g2p_p2g"); keep both spellings
    kernels = {}
    for k, v in merged.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        m = re.match(r"k_([a-z0-9_]+)<(float|double)((?:, (?:true|false|-?\d+))*)>", k)
        key = k
        if m:
            key = m.group(1)
            flags = [t == "true" for t in re.findall(r"true|false", m.group(3))]
            # template flags: k_p2g<T, WRITE_F, DET>, k_grid_op<T, CLEAR>, k_g2p_p2g / k_g2p_grad / k_grid_mass<T, DET>
            if key == "p2g":
                if flags and not flags[0]:
                    key = "p2g_recompute"
                if len(flags) > 1 and flags[1]:
                    key += "_det"
            elif key == "grid_op":
                if flags and flags[0]:
                    key = "grid_op_clear"
            elif flags and flags[0]:
                key += "_det"
            if (m.group(2) == "float") != (a.dtype == "f32"):
                continue
        kernels[key] = v
    out = {"workload": a.workload, "dtype": a.dtype, "steps": a.steps, "warmup": a.warmup, "source": "profiles/" + os.path.basename(a.out) + " <- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --steps " + str(a.steps) + " --warmup " + str(a.warmup) + "` (profiles/tools/pmc_summary.py)",
           "formula": "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) KB, averaged per dispatch", "kernels": kernels}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0)):
        if "hbm_bytes_per_launch" in v:
            print(f"{k:24s} {v['calls']:6d} launches  {v['hbm_bytes_per_launch'] * 1e-6:10.2f} MB per launch")


if __name__ == "__main__":
    sys.exit(main())
