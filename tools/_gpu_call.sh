mkdir -p gpurun_out/r02g
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_store.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for v in "default" "OPE_GATHER_SMALL=0" "OPE_GATHER_TILE=1024" "OPE_GATHER_TILE=4096" "OPE_GATHER_UNROLL=4" "OPE_GATHER_FLOATS=4096"; do
  if [ "$v" = default ]; then envs=""; else envs="$v"; fi
  env $envs rocprofv3 --kernel-trace --stats -d gpurun_out/r02g/prof_$v -o q -- python bench.py --steps 60 --warmup 10 --episodes 5000 --no-cpu-baseline > gpurun_out/r02g/bench_$v.json 2> gpurun_out/r02g/bench_$v.err
  db=$(find gpurun_out/r02g/prof_$v -name "*.db" | head -1)
  echo "== $v"; python tools/rocprof_db_stats.py $db | grep "episode_copy_kernel<true\|^void  \|total kernel" | head -3
  python - "$db" <<'PY'
import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
for r in c.execute("select name,count(*),avg(end-start),min(end-start) from kernels where name like '%episode_copy_kernel<true%' group by name"):
    print("   gather:", r[1], "calls avg %.2f us min %.2f us" % (r[2]/1e3, r[3]/1e3))
PY
  rm -rf gpurun_out/r02g/prof_$v
done
