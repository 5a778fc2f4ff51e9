cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_shapes.py -m gpu -x -q -k rollingpin 2>&1 | tail -5
