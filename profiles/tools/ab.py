#!/usr/bin/env python
"""A/B runs of libplmpm.so variants in ONE process tree on one box:  ab.py OUTDIR [--reps R] [--args "..."] NAME[:ENV=V,...] ...
Each NAME is exp_libs/libplmpm_NAME.so ("default" = the in-tree library); runs alternate (a b c a b c) so that clock /
thermal drift hits all arms alike.  Prints one line per arm: substeps/s of every rep, kernel microseconds (HIP events)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    argv = sys.argv[1:]
    out = argv.pop(0)
    reps, extra = 2, "--steps 20 --warmup 5 --no-cpu-baseline"
    while argv and argv[0].startswith("--"):
        k = argv.pop(0)
        if k == "--reps":
            reps = int(argv.pop(0))
        elif k == "--args":
            extra = argv.pop(0)
    arms = argv
    os.makedirs(out, exist_ok=True)
    res = {a: [] for a in arms}
    for r in range(reps):
        for a in arms:
            name, _, envs = a.partition(":")
            env = dict(os.environ)
            if name != "default":
                env["PLMPM_LIB"] = os.path.join(ROOT, "exp_libs", f"libplmpm_{name}.so")
            for kv in filter(None, envs.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra.split(), env=env, capture_output=True, text=True, cwd=ROOT)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(f"[{a}] FAILED rc={p.returncode}: {p.stderr[-400:]}", flush=True)
                continue
            d = json.loads(line[-1])
            with open(os.path.join(out, f"{name}_{r}.json"), "w") as f:
                f.write(line[-1] + "\n")
            res[a].append(d)
    for a in arms:
        if not res[a]:
            continue
        vals = " / ".join(f"{d['value']:.0f}" for d in res[a])
        ks = res[a][-1].get("roofline", {}).get("kernels", {})
        kk = "  ".join(f"{k} {sum(d['roofline']['kernels'][k]['avg_us'] for d in res[a] if 'roofline' in d) / len(res[a]):.1f}" for k in ks)
        ssum = " / ".join(f"{d['roofline']['substep_kernel_sum_us']:.1f}" for d in res[a] if "roofline" in d)
        print(f"{a:28s} substeps/s {vals} | us/substep {ssum} | {kk} | loss {res[a][-1]['final_loss']:.6f}", flush=True)


if __name__ == "__main__":
    main()
