#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box: kernel-trace statistics of the default
# bench command, then separate PMC passes (one counter group per run) for HBM traffic.
# usage: tools/profile_round.sh <tag>      (run through gpurun; results land in gpurun_out/prof_<tag>/)
set -u
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/bench_under_trace.log 2>&1 || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --tiled-8k-only > $out/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; exit 1; }
done
echo done
