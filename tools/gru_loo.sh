#!/bin/bash
# Leave-one-out decomposition of gru_fwd4_kernel and gru_bwd4_kernel (profiles/r06_gru4_decomposition.txt): the headline step with every padded row computed
# (OPE_LIVE_ROWS=0: the plain instantiation carries the variants), libope_exp.so, OPE_GRU_EXP = bit mask (ope_gru4.hip). TIMING ONLY.
cd ${GRAFT_REPO_ROOT:-.}
export OPE_LIB_PATH=$PWD/off-policy_amd/libope_exp.so OPE_LIVE_ROWS=0
for e in 0 1 2 4 8 16 3 10 17 31 32 33 34 40 63; do
  OPE_GRU_EXP=$e python bench.py --no-cpu-baseline --no-full-length --steps 40 --warmup 10 --repeats 2 --episodes 512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['roofline_step']['per_kernel']['kernels'] if 'gru_fwd4' in x['kernel']][0]
print('OPE_GRU_EXP=%-3s gru_fwd4 %7.2f us   (step %.4f ms)  %s' % ('$e', k['avg_us'], d['ms_per_step'], k['kernel'][:60]))"
done
for e in 0 2 4 8 16 24 30; do
  OPE_GRUB_EXP=$e OPE_GRUB_W=4 python bench.py --no-cpu-baseline --no-full-length --steps 40 --warmup 10 --repeats 2 --episodes 512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[x for x in d['roofline_step']['per_kernel']['kernels'] if 'gru_bwd4' in x['kernel']][0]
print('OPE_GRUB_EXP=%-3s gru_bwd4 %7.2f us   (step %.4f ms)  %s' % ('$e', k['avg_us'], d['ms_per_step'], k['kernel'][:60]))"
done
