#!/usr/bin/env python3
"""The calibration file behind bench.py's `roofline.valu` (SURVEY 8(d)'s secondary, vector-ALU roof):

    python profiles/tools/valu_calibration.py gpurun_out/r6a/valu_calibration.txt gpurun_out/r6a/stall_pmc.json \\
        --workload config3_cube128 --dtype f32 --out profiles/r06_valu_calibration.json

* the first file is the table `profiles/microbench/valu_calibration.bin` prints: cycles per wave-instruction per SIMD of 36 instruction
  streams at W = 1, 2, 4, 8 waves per SIMD, each at the shader clock the stream really ran at (s_memtime against s_memrealtime);
* the second is the per-kernel average of the `rocprofv3 --pmc SQ_INSTS_VALU_*` passes of `python bench.py` (profiles/tools/r06_session.sh
  merges them with pmc_summary.read_db): the DYNAMIC instruction mix of every hot kernel by the classes the counters distinguish.

Each counter class is priced with the stream(s) that stand for it (CLASSES below), at the column --waves (default 4: what the scatter
kernels run at); a kernel's mix per wave = its counters / SQ_WAVES.  Instructions no class counter claims (moves, selects, compares,
DPP moves: SQ_INSTS_VALU minus the classes) are priced as "default".  bench.valu_roof turns this into issue microseconds per kernel.
What the number is NOT: a model of dependent-issue stalls, of the LDS or memory pipes, or of co-issue -- it is the time the SIMDs need to
ISSUE the vector instructions if nothing else ever stood in the way, i.e. a lower bound of the kernel time and, over the measured time,
the fraction that bound explains."""
import argparse
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# class -> (PMC counter, names of the microbenchmark streams whose mean prices it)
CLASSES = {
    "fma_f32": ("SQ_INSTS_VALU_FMA_F32", ["v_fma_f32 d,d,m,m2  (3 distinct VGPR sources)"]),
    "mul_f32": ("SQ_INSTS_VALU_MUL_F32", ["v_mul_f32 d,d,m"]),
    "add_f32": ("SQ_INSTS_VALU_ADD_F32", ["v_add_f32 d,d,m"]),
    "trans_f32": ("SQ_INSTS_VALU_TRANS_F32", ["v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"]),
    "int32": ("SQ_INSTS_VALU_INT32", ["v_add_u32 d,d,m", "v_lshl_add_u32 d,d,2,m", "v_and_b32 d,d,m", "v_bfe_u32 d,d,m,5"]),
    "int64": ("SQ_INSTS_VALU_INT64", ["v_lshl_add_u64 d,d,2,m64 (64-bit address add)"]),
    "cvt": ("SQ_INSTS_VALU_CVT", ["v_cvt_f64_f32", "v_cvt_f32_i32"]),
    "fma_f64": ("SQ_INSTS_VALU_FMA_F64", ["v_fma_f64"]),
    "mul_f64": ("SQ_INSTS_VALU_MUL_F64", ["v_fma_f64"]),
    "add_f64": ("SQ_INSTS_VALU_ADD_F64", ["v_add_f64"]),
    "trans_f64": ("SQ_INSTS_VALU_TRANS_F64", ["v_fma_f64"]),
}
DEFAULT_STREAMS = ["v_mov_b32 d,m", "v_cndmask_b32 d,d,m,vcc"]


def parse_table(text):
    """{stream name: {W: (cycles, GHz)}} from the microbenchmark's table."""
    out = {}
    for ln in text.splitlines():
        cols = re.findall(r"W=(\d+):\s*([\d.]+) cyc @\s*([\d.]+) GHz", ln)
        if not cols:
            continue
        name = ln[:ln.index("W=")].strip()
        out[name] = {int(w): (float(c), float(g)) for w, c, g in cols}
    return out


def bench_name(short, dtype):
    """'k_g2p_p2g<float, false>' -> 'g2p_p2g' (bench.py's kernel names); None for another scalar type or a kernel bench.py does not list."""
    m = re.match(r"k_([a-z0-9_]+)<(float|double)((?:, (?:true|false|-?\d+))*)>", short)
    if not m or (m.group(2) == "float") != (dtype == "f32"):
        return None
    key, flags = m.group(1), [t == "true" for t in re.findall(r"true|false", m.group(3))]
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
    return key


def build(table, pmc, workload, dtype, waves, source):
    def price(streams):
        vals = [table[s][waves][0] for s in streams if s in table and waves in table[s]]
        if not vals:
            raise SystemExit(f"the table has no W={waves} column for any of {streams}")
        return sum(vals) / len(vals)

    cyc = {c: price(streams) for c, (_cnt, streams) in CLASSES.items()}
    cyc["default"] = price(DEFAULT_STREAMS)
    ghz = [g for cols in table.values() for w, (_c, g) in cols.items() if w == waves]
    kernels = {}
    for short, v in pmc.items():
        name = bench_name(short, dtype)
        nw = v.get("SQ_WAVES")
        if name is None or not nw or "SQ_INSTS_VALU" not in v:
            continue
        mix = {c: v[cnt] / nw for c, (cnt, _s) in CLASSES.items() if v.get(cnt)}
        rest = v["SQ_INSTS_VALU"] / nw - sum(mix.values())
        if rest > 0:
            mix["other"] = rest
        kernels[name] = {"mix_per_wave": mix, "waves": nw, "valu_per_wave": v["SQ_INSTS_VALU"] / nw}
    return {"workload": workload, "dtype": dtype, "clock_ghz": sum(ghz) / len(ghz), "simds": 1024, "waves_per_simd_column": waves,
            "source": source, "cycles_per_wave_instruction": cyc, "kernels": kernels,
            "classes": {c: {"counter": cnt, "streams": s} for c, (cnt, s) in CLASSES.items()}, "default_streams": DEFAULT_STREAMS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("table")
    ap.add_argument("pmc_json")
    ap.add_argument("--workload", default="config3_cube128")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--waves", type=int, default=4, help="column of the table: waves per SIMD")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    table = parse_table(open(a.table).read())
    pmc = json.load(open(a.pmc_json))
    cal = build(table, pmc, a.workload, a.dtype, a.waves,
                f"{os.path.basename(a.table)} (microbench/valu_calibration.hip, W = {a.waves}) x {os.path.basename(a.pmc_json)} (rocprofv3 --pmc SQ_INSTS_VALU_* of bench.py)")
    with open(a.out, "w") as f:
        json.dump(cal, f, indent=1, sort_keys=True)
    print(f"clock {cal['clock_ghz']:.3f} GHz; cycles per wave-instruction: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(cal["cycles_per_wave_instruction"].items())))
    for k, v in sorted(cal["kernels"].items(), key=lambda kv: -kv[1]["valu_per_wave"]):
        cycles = sum(n * cal["cycles_per_wave_instruction"].get(c, cal["cycles_per_wave_instruction"]["default"]) for c, n in v["mix_per_wave"].items())
        print(f"{k:18s} {v['valu_per_wave']:8.0f} vector instructions per wave = {cycles:9.0f} issue cycles; waves {v['waves']:.0f}")


if __name__ == "__main__":
    main()
