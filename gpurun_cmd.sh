cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_shapes.py 2>&1 | tail -2
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_x.json'))
print(d['value'], d['ms_per_step'], d['roofline']['substep_kernel_sum_us'], d['roofline']['substep_frac'])
for k,v in d['roofline']['kernels'].items(): print(k, round(v['avg_us'],1))
PY
