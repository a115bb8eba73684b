cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ddpg.py -x -q -m gpu 2>&1 | tail -5
for w in maddpg_spread matd3_spread; do
timeout 300 python bench.py --workload $w --steps 500 --warmup 52 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
done
timeout 300 python bench.py --workload maddpg_spread --steps 504 --warmup 48 --no-cpu-baseline --steps-per-replay 8 2>&1 | tail -1 | cut -c1-260
