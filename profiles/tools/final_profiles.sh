#!/bin/bash
# final profiles of round 4 (run from the repo root on the GPU box)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> /tmp/kt.err
python $R/profiles/summarize_rocpd.py $(find /tmp/p_kt -name "*_results.db" | head -1) > $O/kernel_trace_stats.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_fetch -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_write -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d /tmp/p_sq -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
cd $R
python profiles/tools/pmc_summary.py --workload config3_cube128 --dtype f32 --steps 20 --warmup 5 --out $O/pmc.json $(find /tmp/p_fetch -name "*_results.db" | head -1) $(find /tmp/p_write -name "*_results.db" | head -1) $(find /tmp/p_sq -name "*_results.db" | head -1) > $O/pmc_summary.txt 2>&1
tail -5 $O/pmc_summary.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
