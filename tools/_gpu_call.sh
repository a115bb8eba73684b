cd $GRAFT_REPO_ROOT
export OPE_BENCH_SELFTEST=1
for w in 3s5z maddpg_spread rmatd3_3m; do
  timeout 300 python bench.py --workload $w --gpus 2 --steps 10 --warmup 3 --episodes 64 2>gpurun_out/selftest_$w.err | tail -1 | cut -c1-420
  echo "rc=$? tracebacks=$(grep -c Traceback gpurun_out/selftest_$w.err)"; grep -B2 -A8 "Traceback" gpurun_out/selftest_$w.err | tail -14; grep "ope.dist" gpurun_out/selftest_$w.err | head -2
done
unset OPE_BENCH_SELFTEST
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3
