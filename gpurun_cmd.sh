cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_r1b -o r01b -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof_r1b_bench.json 2> /root/repo/gpurun_out/prof_r1b.err
ls -R /root/repo/gpurun_out/prof_r1b | head
