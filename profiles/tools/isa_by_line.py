#!/usr/bin/env python3
"""Static instruction profile of one kernel by SOURCE LINE: compile the device code with line tables
(hipcc <Makefile flags> -gline-tables-only --cuda-device-only -S csrc/plmpm_capi.hip -o capi.s), then
    isa_by_line.py capi.s k_g2p_p2gIfLb0 [--top 40] [--by-function]
attributes every instruction of the kernel's listing to the .loc in force (file:line of the innermost inlined frame) and prints the
lines that own the most vector instructions, with a breakdown by class.  A rolled loop body counts once (static counts); what the
tool is for is finding WHERE the non-arithmetic instructions of a kernel come from (address arithmetic, selects, moves, waits) --
the questions a timing cannot answer and a listing of 3 000 unattributed instructions does not answer either."""
import argparse
import collections
import re


def classify(op):
    if op.startswith("v_pk_"): return "pk"
    if re.match(r"v_(fma|fmac|mad|mac)_f32", op): return "fma32"
    if re.match(r"v_(mul|add|sub|subrev|min|max)_f32", op): return "muladd32"
    if re.match(r"v_.*_f64", op) and not op.startswith("v_cvt"): return "f64"
    if op.startswith("v_cvt"): return "cvt"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op): return "trans"
    if re.match(r"v_(cndmask|mov|readlane|readfirstlane|writelane|swap|accvgpr)", op): return "sel/mov"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "cmp"
    if op.startswith("v_"): return "int"
    if op.startswith("ds_"): return "lds"
    if re.match(r"(global|flat|buffer|scratch)_", op): return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="fragment of the mangled kernel name")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--by-function", action="store_true", help="aggregate by the function named in the .loc's inlined-at chain is not available in "
                    "line tables; this aggregates by FILE instead")
    a = ap.parse_args()
    files, per, inside, cur = {}, collections.defaultdict(collections.Counter), False, ("?", 0)
    dpp = collections.Counter()
    for ln in open(a.asm):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        if not inside:
            if re.match(r"^[A-Za-z_.$][\w.$]*:", ln) and a.kernel in ln and not ln.startswith(".L"):
                inside = True
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".size"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        key = cur[0] if a.by_function else cur
        per[key][c] += 1
        if "dpp" in op or " row_" in s or "quad_perm" in s:
            dpp[key] += 1
    vec = ("fma32", "muladd32", "pk", "f64", "cvt", "trans", "sel/mov", "cmp", "int")
    tot = collections.Counter()
    for k, c in per.items():
        for cls, n in c.items():
            tot[cls] += n
    nv = sum(tot[c] for c in vec)
    print(f"kernel *{a.kernel}*: {sum(tot.values())} instructions, {nv} vector: " + ", ".join(f"{c} {tot[c]}" for c in vec + ('lds', 'vmem', 'salu', 'wait')))
    rows = sorted(per.items(), key=lambda kv: -sum(kv[1][c] for c in vec))[:a.top]
    for key, c in rows:
        v = sum(c[x] for x in vec)
        where = key if a.by_function else f"{key[0]}:{key[1]}"
        print(f"{where:28s} vec {v:5d} ({100.0 * v / max(nv, 1):4.1f} %)  " + " ".join(f"{x} {c[x]}" for x in vec + ('lds', 'vmem', 'salu', 'wait') if c[x]) + (f"  [dpp {dpp[key]}]" if dpp[key] else ""))


if __name__ == "__main__":
    main()
