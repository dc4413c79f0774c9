#!/bin/bash
# Model-level GPU parity + the previously failing op groups.  One process per group; stop on abnormal exit.
mkdir -p gpurun_out
status=0
run() {
  name=$1; shift
  timeout -k 10 420 "$@" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "[$name] rc=$rc $(grep -E '^(FAILED|ERROR)|passed|failed' gpurun_out/$name.log | tail -4 | tr '\n' ' ')"
  if [ $rc -gt 1 ]; then echo "abnormal exit, stopping"; exit $rc; fi
  [ $rc -ne 0 ] && status=1
}
run model python -m pytest tests/test_gpu_model.py -m gpu -q -rA -p no:cacheprovider
run ops_front python -m pytest tests/test_gpu_ops.py -m gpu -q -rA -p no:cacheprovider -k "normalize or logmel or im2col or ctc or lid_score or novograd"
exit $status
