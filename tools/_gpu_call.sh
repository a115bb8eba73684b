cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ddpg.py tests/test_gpu_rddpg.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "not full_size" 2>&1 | tail -12
