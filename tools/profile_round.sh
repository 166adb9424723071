#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box: kernel-trace statistics of the default
# bench command and of the headline part alone, then separate PMC passes (one counter group per run) for HBM traffic.
# usage: tools/profile_round.sh <tag>      (run through gpurun; results land in gpurun_out/prof_<tag>/)
set -u
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/bench_under_trace.log 2>&1 || exit 1
echo "trace of the default command done"
# the headline kernel alone: in the default command the camera path and the analytic whole-table line launch the same
# kernel on frames whose sky it does not shade, and the 1080p configuration launches it on a quarter of the pixels,
# which pulls that kernel's average below the headline launches'
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_headline -- python3 bench.py --no-callers --no-cpu-baseline --no-configs0 > $out/bench_headline_under_trace.log 2>&1 || exit 1
echo "trace of the headline part done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs0 > $out/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; exit 1; }
  echo "pmc pass $name done"
done
# the radius-cutoff workloads one at a time: their three kernels move different bytes per workload
for wl in tiled_resolve_4k_256 tiled_resolve_4k_4096 tiled_resolve_8k tiled_resolve_8k_discontinuous; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $out/wl_${wl}__$c -- python3 bench.py --only-tiled $wl --steps 6 --warmup 2 > $out/wl_${wl}__$c.log 2>&1 || { echo "pmc pass $wl $c failed"; exit 1; }
  done
  echo "workload passes $wl done"
done
echo done
