#!/bin/bash
# Round-6 evidence, one gpurun call:  gpurun --timeout 1500 -- 'bash tools/collect_profiles_r06.sh r06f'
#   every BASELINE workload: bench line with the in-run per-kernel table + rocprofv3 --kernel-trace --stats of the same command
#   3s5z (headline, the DEFAULT bench command: 5 000 episodes resident) and maddpg_spread: SQ instruction / wait counter passes and
#   FETCH_SIZE / WRITE_SIZE (one counter group per pass)
# Counter passes never carry a trace domain besides --kernel-trace (--pmc with sys/hip/hsa traces is refused on this pool).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r06f}
O=gpurun_out/$TAG
mkdir -p $O
eps() { case $1 in 3m|MMM2|3s5z_gall) echo "--episodes 1000";; *) echo "";; esac; }
steps() { case $1 in rmatd3_MMM2) echo "--steps 12 --warmup 4";; maddpg_spread|matd3_spread) echo "--steps 200 --warmup 40";; *) echo "--steps 40 --warmup 10";; esac; }
bench() { w=$1; shift; timeout 400 python bench.py --workload $w $(eps $w) $(steps $w) --no-cpu-baseline "$@" > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$w.json | head -1)"; }
kt() { w=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o p -- python bench.py --workload $w $(eps $w) $(steps $w) --repeats 2 --no-cpu-baseline --no-kernel-table --no-full-length --no-gather-extras "$@" > $O/kt_$w.json 2> $O/kt_$w.log
  echo "kt $w rc=$?"; cp "$(find $O/kt_$w -name '*kernel_stats.csv' | head -1)" $O/${w}_kernel_stats.csv; rm -rf $O/kt_$w; }
pass() { w=$1; n=$2; shift; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_${w}_$n -o p -- python bench.py --workload $w $(eps $w) --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-kernel-table --no-full-length $(case $w in 3s5z) echo --no-gather-extras;; esac) > $O/pmc_${w}_$n.json 2> $O/pmc_${w}_$n.log
  echo "pmc $w $n rc=$?"; cp "$(find $O/pmc_${w}_$n -name '*counter_collection.csv' | head -1)" $O/${w}_pmc_${n}_raw.csv 2>/dev/null
  python tools/pmc_table.py "$(find $O/pmc_${w}_$n -name '*counter_collection.csv' | head -1)" > $O/${w}_pmc_$n.txt; rm -rf $O/pmc_${w}_$n; }
timeout 300 python bench.py > $O/bench_default_with_cpu.json 2> $O/bench_default_with_cpu.err; echo "default bench rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_default_with_cpu.json | head -1)"
for w in 3s5z maddpg_spread 3m rmatd3_MMM2 MMM2 3s5z_gall matd3_spread; do bench $w; kt $w; done
for w in 3s5z maddpg_spread; do
  pass $w inst SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32
  pass $w wait SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
  pass $w fetch FETCH_SIZE
  pass $w write WRITE_SIZE
done
# (every gather launch of those passes stops at the sampled episodes' terminations -- live_only, --no-gather-extras: no plain gathers in the process)
python tools/gather_traffic.py 3s5z:32:5000:live_only $(python -c "import json;print(json.load(open('$O/bench_3s5z.json'))['roofline']['algorithmic_bytes_per_launch'])") $O/3s5z_pmc_fetch_raw.csv $O/3s5z_pmc_write_raw.csv > $O/gather_traffic.txt 2>&1
cp profiles/gather_traffic.json $O/gather_traffic.json
rm -f $O/*_raw.csv
ls $O | head -80
# round 6: the headline command with every padded row computed (ope_qmix_cfg.live_rows = 1 by the process default), the driver's own command
# line, and the plan as a launch of its own in front of the step
OPE_LIVE_ROWS=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_3s5z_padded_rows.json 2> $O/bench_3s5z_padded_rows.err; echo "padded rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_3s5z_padded_rows.json | head -1)"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_command_steps20_warmup5.json 2> $O/bench_driver_command.err; echo "driver rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_driver_command_steps20_warmup5.json | head -1)"
timeout 300 python bench.py --no-cpu-baseline --no-early-plan > $O/bench_3s5z_plan_in_front_of_the_step.json 2> $O/bench_3s5z_plan_in_front.err; echo "no-early-plan rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_3s5z_plan_in_front_of_the_step.json | head -1)"
rm -f gpurun_out/parity_errors.jsonl
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
cp gpurun_out/parity_errors.jsonl $O/parity_errors.jsonl 2>/dev/null
ls $O | wc -l
