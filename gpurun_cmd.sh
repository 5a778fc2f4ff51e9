cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_rollout.py -m gpu -x -q -k checkpointed 2>&1 | tail -8
