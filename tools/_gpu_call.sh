cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu -k "ddpg or graph or store" 2>&1 | tail -5
for w in maddpg_spread matd3_spread; do
timeout 300 python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
done
timeout 300 python bench.py --workload maddpg_spread --steps 500 --warmup 50 --no-cpu-baseline --host-indices 2>&1 | tail -1 | cut -c1-260
export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python bench.py --workload maddpg_spread --steps 300 --warmup 30 --no-cpu-baseline > /tmp/b.txt 2>&1
python tools/trace_gaps.py /tmp/prof --last 1500 > gpurun_out/gaps_maddpg.txt
cat gpurun_out/gaps_maddpg.txt
