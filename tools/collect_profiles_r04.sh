#!/bin/bash
# Round-4 evidence, one gpurun call:  gpurun --timeout 1500 -- 'bash tools/collect_profiles_r04.sh r04e'
#   every BASELINE workload: bench line with the in-run per-kernel table + rocprofv3 --kernel-trace --stats of the same command
#   3s5z (headline) and maddpg_spread: SQ instruction / wait counter passes and FETCH_SIZE / WRITE_SIZE (one counter per pass)
#   3s5z --lazy-obs: FETCH_SIZE / WRITE_SIZE again (observations read in place from the store: what moves where)
# Counter passes never carry a trace domain besides --kernel-trace (--pmc with sys/hip/hsa traces is refused on this pool).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r04e}
O=gpurun_out/$TAG
mkdir -p $O
eps() { case $1 in 3m|3s5z|MMM2|3s5z_gall) echo "--episodes 1000";; *) echo "";; esac; }
steps() { case $1 in rmatd3_MMM2) echo "--steps 12 --warmup 4";; maddpg_spread|matd3_spread) echo "--steps 200 --warmup 40";; *) echo "--steps 40 --warmup 10";; esac; }
bench() { w=$1; shift; timeout 400 python bench.py --workload $w $(eps $w) $(steps $w) --no-cpu-baseline "$@" > $O/bench_$w$SUF.json 2> $O/bench_$w$SUF.err; echo "bench $w$SUF rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$w$SUF.json | head -1)"; }
kt() { w=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w$SUF -o p -- python bench.py --workload $w $(eps $w) $(steps $w) --repeats 2 --no-cpu-baseline --no-kernel-table "$@" > $O/kt_$w$SUF.json 2> $O/kt_$w$SUF.log
  echo "kt $w$SUF rc=$?"; cp "$(find $O/kt_$w$SUF -name '*kernel_stats.csv' | head -1)" $O/${w}${SUF}_kernel_stats.csv; rm -rf $O/kt_$w$SUF; }
pass() { w=$1; n=$2; shift; shift; extra=""; if [ "$SUF" = "_lazy" ]; then extra="--lazy-obs"; fi
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${w}${SUF}_$n -o p -- python bench.py --workload $w $(eps $w) --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-kernel-table $extra > $O/pmc_${w}${SUF}_$n.json 2> $O/pmc_${w}${SUF}_$n.log
  echo "pmc $w$SUF $n rc=$?"; python tools/pmc_table.py "$(find $O/pmc_${w}${SUF}_$n -name '*counter_collection.csv' | head -1)" > $O/${w}${SUF}_pmc_$n.txt; rm -rf $O/pmc_${w}${SUF}_$n; }
SUF=""
for w in 3s5z maddpg_spread 3m rmatd3_MMM2 MMM2 3s5z_gall matd3_spread; do bench $w; kt $w; done
for w in 3s5z maddpg_spread; do
  pass $w inst SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32
  pass $w wait SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
  pass $w fetch FETCH_SIZE
  pass $w write WRITE_SIZE
done
SUF="_lazy"
bench 3s5z --lazy-obs
pass 3s5z fetch FETCH_SIZE
pass 3s5z write WRITE_SIZE
ls $O | head -60
