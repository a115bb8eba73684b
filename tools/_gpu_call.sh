cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 100 --warmup 20 --episodes 256 --no-cpu-baseline > /tmp/b1.txt 2>&1
find /tmp/prof -type f | head
cp $(find /tmp/prof -name "*kernel_stats*" | head -1) gpurun_out/r02z_qmix3s5z_kernel_stats.csv
rm -rf /tmp/prof2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -- python bench.py --workload maddpg_spread --steps 400 --warmup 40 --no-cpu-baseline > /tmp/b2.txt 2>&1
cp $(find /tmp/prof2 -name "*kernel_stats*" | head -1) gpurun_out/r02z_maddpg_spread_kernel_stats.csv
head -8 gpurun_out/r02z_maddpg_spread_kernel_stats.csv | cut -c1-150
