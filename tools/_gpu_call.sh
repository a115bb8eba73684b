cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mqmix.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -15
