#!/usr/bin/env python3
"""Static instruction statistics of the hot kernels from a device assembly listing (no GPU needed).

    hipcc <Makefile FLAGS> --cuda-device-only -S plasticinelab_amd/csrc/plmpm_capi.hip -o /tmp/capi.s
    python profiles/tools/isa_stats.py /tmp/capi.s [kernel-name-substring ...]

Per kernel: VGPRs, scratch bytes, LDS bytes, occupancy (waves per SIMD the registers allow), and the STATIC count of vector
instructions by class (a rolled loop body counts once: multiply by trip counts for the dynamic mix -- the PMC passes in
profiles/ give that).  Classes follow profiles/microbench/valu_calibration.hip, so that `--cycles calibration.json` can price a
listing in issue cycles."""
import collections
import json
import re
import subprocess
import sys

CLASSES = [
    ("fma_f32", r"^v_(fma|fmac|mad|mac)_f32"), ("pk_f32", r"^v_pk_(fma|mul|add)_f32"), ("mul_add_f32", r"^v_(mul|add|sub|subrev|max|min)_f32"),
    ("dpp", r"_dpp$|_dpp "), ("cvt", r"^v_cvt_"), ("trans", r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_"), ("f64", r"^v_\w+_f64"),
    ("int_mul", r"^v_(mul_lo|mul_hi|mad_u64|mad_i64|mad_u32|mad_i32)"), ("int", r"^v_(add|sub|subrev|lshl|lshr|ashr|and|or|xor|bfe|bfi|lshlrev|lshrrev|ashrrev|add3|lshl_add|add_lshl|lshl_or|and_or|or3|min|max|med3|not|ffbh|bcnt|mbcnt|perm|alignbit|addc|subb|subbrev)_"),
    ("cmp", r"^v_cmp"), ("select", r"^v_cndmask"), ("mov", r"^v_(mov|accvgpr|readlane|readfirstlane|writelane|swap|pk_mov)"), ("mfma", r"^v_mfma"),
]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def parse(path):
    kernels, cur, body = {}, None, None
    meta = {}
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), collections.Counter()
            kernels[cur] = body
            continue
        if cur and re.match(r"^\s*s_endpgm", line):
            body["__end__"] += 1
        if cur and body is not None:
            t = line.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                pass
            else:
                op = t.split()[0]
                full = t.split(";")[0]
                body["__all__"] += 1
                if op.startswith("v_"):
                    body["valu"] += 1
                    if "dpp" in full.split()[0] or " row_" in full or " quad_perm" in full:
                        body["dpp"] += 1
                    else:
                        for name, rx in CLASSES:
                            if name != "dpp" and re.search(rx, op):
                                body[name] += 1
                                break
                        else:
                            body["other_valu"] += 1
                elif op.startswith("ds_"):
                    body["lds"] += 1
                elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
                    body["vmem"] += 1
                    if op.startswith("scratch_"):
                        body["scratch_ops"] += 1
                elif op.startswith("s_"):
                    body["salu"] += 1
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            meta_k = m.group(1)
            meta[meta_k] = {}
            continue
        m = re.match(r"^\s*\.amdhsa_(next_free_vgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\d+)", line)
        if m and meta:
            meta[list(meta)[-1]][m.group(1)] = int(m.group(2))
    return kernels, meta


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path, pats = args[0], args[1:] or ["k_g2p_p2gIfLb0", "k_p2g_gradIf", "k_g2p_gradIfLb0", "k_grid_op_gradIf", "k_grid_opIfLb0"]
    cyc = None
    for a in sys.argv[1:]:
        if a.startswith("--cycles="):
            cyc = json.load(open(a.split("=", 1)[1]))["cycles"]
    kernels, meta = parse(path)
    names = demangle(list(kernels))
    for k, body in kernels.items():
        if not any(p in k for p in pats) or k not in meta:
            continue
        md = meta[k]
        vg = md.get("next_free_vgpr", 0)
        alloc = -(-vg // 8) * 8
        occ = min(8, 512 // alloc) if alloc else 8
        print(f"{names[k][:100]}")
        print(f"   VGPRs {vg} (alloc {alloc}, {occ} waves/SIMD), scratch {md.get('private_segment_fixed_size', 0)} B, LDS {md.get('group_segment_fixed_size', 0)} B; "
              f"static: {body['valu']} VALU, {body['salu']} SALU, {body['lds']} LDS, {body['vmem']} VMEM ({body['scratch_ops']} scratch)")
        cls = {n: body[n] for n, _ in CLASSES if body[n]}
        cls["other"] = body["other_valu"]
        print("   " + "  ".join(f"{n} {v}" for n, v in cls.items()))
        if cyc:
            print("   static issue cycles: %.0f" % sum(cyc.get(n, cyc.get("default", 4.0)) * v for n, v in cls.items()))


if __name__ == "__main__":
    main()
