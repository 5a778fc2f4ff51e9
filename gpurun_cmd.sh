cd /root/repo
python bench.py > gpurun_out/r01_bench_default.json 2> gpurun_out/r01_bench_default.err
tail -c 600 gpurun_out/r01_bench_default.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_final/trace -o r01 -- python /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/r01_bench_under_rocprof.json 2> /root/repo/gpurun_out/prof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/prof_final/pmc_fetch -o r01 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> /root/repo/gpurun_out/prof_pmc1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/prof_final/pmc_write -o r01 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> /root/repo/gpurun_out/prof_pmc2.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS -d /root/repo/gpurun_out/prof_final/pmc_sq -o r01 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> /root/repo/gpurun_out/prof_pmc3.err
find /root/repo/gpurun_out/prof_final -name "*.db" | head
