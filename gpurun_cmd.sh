cd /root/repo
python -m pytest tests/test_gpu_rollout.py -m gpu -x -q -k "resort" 2>&1 | tail -12
