cd /root/repo
python -m pytest tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -4
