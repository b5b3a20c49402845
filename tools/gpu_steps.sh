#!/bin/bash
# Run GPU steps one after another on the gpurun box: a step that FAILS (assertion, rc < 124) is logged and the next one runs; a step
# that times out or is killed (rc >= 124) ends the call -- no further GPU step is started behind a hung one.
#   tools/gpu_steps.sh <tag> <<'EOS'
#   600 python -m pytest tests/... > gpurun_out/x.log 2>&1
#   EOS
tag=$1
mkdir -p gpurun_out
log=gpurun_out/${tag}_steps.log
: > "$log"
while IFS= read -r line; do
  [ -z "$line" ] && continue
  secs=${line%% *}
  cmd=${line#* }
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd"
  rc=$?
  echo "rc=$rc $(( $(date +%s) - start ))s: $cmd" | tee -a "$log"
  if [ $rc -ge 124 ]; then echo "step timed out / was killed: stopping" | tee -a "$log"; exit $rc; fi
done
exit 0
