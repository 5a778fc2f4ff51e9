cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_shapes.py 2>&1 | tail -2
python tests/dbg_drift.py 2>&1 | tail -8
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_x.json'))
print(d['value'], d['ms_per_step'], d['roofline']['substep_kernel_sum_us'], d['roofline']['substep_frac'])
for k,v in d['roofline']['kernels'].items(): print(k, round(v['avg_us'],1), v['launches'])
PY
