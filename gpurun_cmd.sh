cd /root/repo
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8
