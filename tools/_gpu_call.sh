cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "qmix and not graph" 2>&1 | tail -2
for r in 0 160 224; do
OPE_WGRAD_ROWS=$r timeout 600 python bench.py --steps 300 --warmup 30 --episodes 256 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
done
