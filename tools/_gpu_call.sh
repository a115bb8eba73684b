cd $GRAFT_REPO_ROOT
python tools/ddpg_phases.py 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_ddpg.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -3
python bench.py --workload maddpg_spread --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | cut -c1-200
python bench.py --workload maddpg_spread --steps 300 --warmup 30 --no-cpu-baseline --no-graph 2>/dev/null | cut -c1-200
