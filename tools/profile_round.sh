#!/bin/bash
# Refresh the evidence under profiles/ at the current binary (run on the MI355X box through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh <tag>'   then   python tools/collect_profiles.py <tag>
# Separate rocprofv3 passes: kernel stats (single / multi stream), FETCH_SIZE, WRITE_SIZE, SQ counters; no --pmc beside a
# runtime trace; the program sits directly after `--`.
set -e
TAG=${1:-cur}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
echo bench done
timeout -k 10 200 python tools/profile_ops.py > $O/per_layer_serial.txt 2> $O/per_layer.err
echo per-layer done
FLK_SINGLE_STREAM=1 timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > $O/single.log 2>&1
echo single done
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > $O/multi.log 2>&1
echo multi done
FLK_SINGLE_STREAM=1 timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > $O/fetch.log 2>&1
echo fetch done
FLK_SINGLE_STREAM=1 timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > $O/write.log 2>&1
echo write done
FLK_SINGLE_STREAM=1 timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > $O/sq.log 2>&1
echo sq done
# keep only what collect_profiles.py reads (gpurun_out is capped at 64 MiB)
find $O -name '*_agent_info.csv' -delete
find $O/single -name '*_kernel_trace.csv' -delete      # the multi-stream trace feeds tools/timeline.py
du -sh $O
