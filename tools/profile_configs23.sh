#!/bin/bash
# rocprofv3 evidence for the batch-1 / VideoResNet configurations (BASELINE configs 2 and 3), run on the MI355X box through gpurun:
#   gpurun --timeout 1100 -- 'bash tools/profile_configs23.sh <tag>'   then   python tools/collect_configs23.py <tag>
# Per configuration two passes: --kernel-trace --stats, and the SQ counters of tools/profile_round.sh (no --pmc beside a runtime trace;
# the program sits directly after `--`).
set -e
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof23_$TAG
rm -rf $O; mkdir -p $O
cd $R
COMMON="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-other-configs"
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
run() {   # name, bench arguments
  local n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${n}_stats -- python3 bench.py "$@" $COMMON > $O/${n}_stats.log 2>&1
  echo "$n stats done"
  timeout -k 10 200 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/${n}_sq -- python3 bench.py "$@" $COMMON > $O/${n}_sq.log 2>&1
  echo "$n sq done"
}
run vrn_r2plus1d_18_bs1 --model r2plus1d_18 --batch 1 --frames 16
run vrn_r2plus1d_18_bs8 --model r2plus1d_18 --batch 8 --frames 16
run i3d_bs1 --batch 1 --frames 64
find $O -name '*_agent_info.csv' -delete
du -sh $O
