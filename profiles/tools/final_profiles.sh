#!/bin/bash
# final profiles of round 5 (run from the repo root on the GPU box): the driver's command, its rocprofv3 kernel trace, the PMC passes
# (separate --pmc runs, kernel trace / stats only -- never combined with other trace domains), the slab host cost, the GPU test tier
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5final
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> /tmp/kt.err
python $R/profiles/summarize_rocpd.py $(find /tmp/p_kt -name "*_results.db" | head -1) > $O/kernel_trace_stats.txt
P="--steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline --no-secondary"
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o r -- python $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o r -- python $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d /tmp/p_sq -o r -- python $R/bench.py $P > /dev/null 2>&1
cd $R
python profiles/tools/pmc_summary.py --workload config3_cube128 --dtype f32 --steps 20 --warmup 5 --out $O/pmc.json $(find /tmp/p_fetch -name "*_results.db" | head -1) $(find /tmp/p_write -name "*_results.db" | head -1) $(find /tmp/p_sq -name "*_results.db" | head -1) > $O/pmc_summary.txt 2>&1
tail -5 $O/pmc_summary.txt
PLMPM_PEER_FUSED=1 python profiles/tools/slab_host_cost.py --world 8 --peer --kernels > $O/slab_host_cost_fused.txt 2>&1
PLMPM_PEER_FUSED=0 python profiles/tools/slab_host_cost.py --world 8 --peer --kernels > $O/slab_host_cost_unfused.txt 2>&1
tail -3 $O/slab_host_cost_fused.txt $O/slab_host_cost_unfused.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.txt
cat $O/smoke.txt
# N = 2 over gloo on this one GPU, the driver's flags: the configs[3] point at full size inside the same world, loss and scaling checks
PLB_DIST_BACKEND=gloo PLB_PEER_HALOS=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_gpus2.err | grep '^{' > $O/bench_gpus2.json
# N = 8 over gloo on this one GPU: all three lines of the 8-GPU run at 10 % of the secondary sizes (the full sizes need 8 GPUs' HBM)
PLB_DIST_BACKEND=gloo PLB_PEER_HALOS=1 PLB_SECONDARY_SCALE=0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 8 --steps 4 --warmup 1 --no-cpu-baseline 2> $O/bench_gpus8.err | grep '^{' > $O/bench_gpus8.json
python - <<PY
import json
for f in ("bench_gpus2.json", "bench_gpus8.json"):
    try:
        d = json.load(open("$O/" + f))
        print(f, d["value"], d["loss_check"], d.get("strong_scaling_eff"), d["halo_transport"][:40])
        for p in d["secondary"]:
            print("   ", {k: p.get(k) for k in ("workload", "value", "job_frac", "loss_check", "strong_scaling_eff", "error", "build_and_warmup_s")})
    except Exception as e:
        print(f, "FAILED", e)
PY
