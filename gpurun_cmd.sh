cd /root/repo
for v in split lossfk; do
cp exp_libs/$v.so plasticinelab_amd/libplmpm.so
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', d['value'], d['ms_per_step'])"
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
