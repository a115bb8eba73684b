mkdir -p gpurun_out/r02m
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/r02m/pytest_gpu.txt 2>&1; tail -8 gpurun_out/r02m/pytest_gpu.txt
python bench.py --workload rmatd3_MMM2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-300
python bench.py --workload MMM2 --steps 60 --warmup 10 --no-cpu-baseline --episodes 512 2>/dev/null | cut -c1-300
python bench.py --workload 3m --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-300
