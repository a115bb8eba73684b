mkdir -p gpurun_out/r02n
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r02n/bench.json 2> gpurun_out/r02n/bench.err; cat gpurun_out/r02n/bench.json; tail -3 gpurun_out/r02n/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r02n/prof -o q -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r02n/bench_prof.json 2> gpurun_out/r02n/bench_prof.err
db=$(find gpurun_out/r02n/prof -name "*.db" | head -1); python tools/rocprof_db_stats.py $db gpurun_out/r02n/kernel_stats.csv | grep -i "total kernel"; grep "episode_copy_kernel<true" gpurun_out/r02n/kernel_stats.csv | cut -c1-60,200-400
cat gpurun_out/r02n/bench_prof.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline'])"
rm -rf gpurun_out/r02n/prof
python bench.py --workload rmatd3_MMM2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
