mkdir -p gpurun_out/r02i
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_qmix.py tests/test_gpu_mqmix.py tests/test_gpu_ckpt.py tests/test_gpu_store.py tests/test_gpu_ddpg.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r02i/pytest.txt 2>&1; tail -15 gpurun_out/r02i/pytest.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r02i/prof -o q -- python bench.py --steps 100 --warmup 20 --episodes 256 --no-cpu-baseline > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
db=$(find gpurun_out/r02i/prof -name "*.db" | head -1); python tools/rocprof_db_stats.py $db gpurun_out/r02i/kernel_stats.csv | cut -c1-150 | head -24
rm -rf gpurun_out/r02i/prof
python bench.py --steps 200 --warmup 20 --episodes 256 --no-cpu-baseline 2>/dev/null | cut -c1-260
