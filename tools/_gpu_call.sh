mkdir -p gpurun_out/r02h
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r02h/q_$c -o x -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/r02h/q_$c.json 2> gpurun_out/r02h/q_$c.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r02h/r_$c -o x -- python bench.py --workload rmatd3_MMM2 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02h/r_$c.json 2> gpurun_out/r02h/r_$c.err
done
find gpurun_out/r02h -name "*counter_collection.csv" | head
for d in q r; do for c in FETCH_SIZE WRITE_SIZE; do f=$(find gpurun_out/r02h/${d}_$c -name "*counter_collection.csv" | head -1); grep "episode_copy_kernel<true" $f > gpurun_out/r02h/${d}_$c.csv.tmp; head -1 $f > gpurun_out/r02h/${d}_$c.csv; cat gpurun_out/r02h/${d}_$c.csv.tmp >> gpurun_out/r02h/${d}_$c.csv; rm gpurun_out/r02h/${d}_$c.csv.tmp; rm -rf gpurun_out/r02h/${d}_$c; done; done
python tools/gather_traffic.py 3s5z:32:5000 95563264 gpurun_out/r02h/q_FETCH_SIZE.csv gpurun_out/r02h/q_WRITE_SIZE.csv
python tools/gather_traffic.py MMM2:128:512 $(python -c "print(2*128*3186968)") gpurun_out/r02h/r_FETCH_SIZE.csv gpurun_out/r02h/r_WRITE_SIZE.csv
cp profiles/gather_traffic.json gpurun_out/r02h/
cat gpurun_out/r02h/q_WRITE_SIZE.json; cat gpurun_out/r02h/r_WRITE_SIZE.json | cut -c1-600
