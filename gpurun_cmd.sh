cd /root/repo
for r in 0 1 2 4 8; do
PLMPM_RESORT_STEPS=$r python bench.py --steps 24 --warmup 1 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('R=$r steps', d['steps'], 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))"
done
