#!/bin/bash
# Collects the per-round evidence on the GPU box: kernel traces and two counter passes per QMIX workload.
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r03k "3s5z 3s5z_gall"'
# Writes gpurun_out/<tag>/{<w>_kernel_stats.csv, <w>_pmc_inst.txt, <w>_pmc_wait.txt, <w>_pmc_hbm.txt, bench_<w>.json}; copy what is to be
# judged into profiles/ (tools/kernel_roofline.py turns the first two into profiles/kernel_roofline.json).
# Counter passes never carry a trace domain besides --kernel-trace (see the brief: --pmc with sys/hip/hsa traces is refused).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r03}
WLS=${2:-3s5z}
O=gpurun_out/$TAG
mkdir -p $O
for w in $WLS; do
  timeout 300 python bench.py --workload $w --episodes 1000 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"; cut -c1-300 $O/bench_$w.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o p -- python bench.py --workload $w --episodes 1000 --steps 40 --warmup 10 --repeats 2 --no-cpu-baseline > $O/kt_$w.json 2> $O/kt_$w.log
  echo "kt $w rc=$?"; cp "$(find $O/kt_$w -name '*kernel_stats.csv' | head -1)" $O/${w}_kernel_stats.csv
  pass() { # name, counters...
    n=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${w}_$n -o p -- python bench.py --workload $w --episodes 1000 --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline > $O/pmc_${w}_$n.json 2> $O/pmc_${w}_$n.log
    echo "pmc $w $n rc=$?"; python tools/pmc_table.py "$(find $O/pmc_${w}_$n -name '*counter_collection.csv' | head -1)" > $O/${w}_pmc_$n.txt
  }
  pass inst SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32
  pass wait SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  rm -rf $O/kt_$w $O/pmc_${w}_*/
done
ls $O | head -40
