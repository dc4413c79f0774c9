#!/bin/bash
# Runs the per-kernel GPU parity tests one group per process; logs under gpurun_out/.  Stops at the first abnormal
# exit (signal / timeout) so no further GPU work is started after a fault.
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt
status=0
for k in "scale_cast or colsum" layernorm gemm attention "glu or dwconv or batchnorm" "normalize or logmel or im2col" "ctc or lid_score" novograd; do
  name=$(echo "$k" | tr ' ' '_')
  timeout -k 10 420 python -m pytest tests/test_gpu_ops.py -m gpu -q -rA -k "$k" -p no:cacheprovider > "gpurun_out/ops_$name.log" 2>&1
  rc=$?
  echo "group [$k] rc=$rc $(grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/ops_$name.log | tail -3 | tr '\n' ' ')"
  if [ $rc -gt 1 ]; then echo "abnormal exit, stopping sweep"; status=$rc; break; fi
  [ $rc -ne 0 ] && status=1
done
exit $status
