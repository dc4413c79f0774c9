#!/bin/bash
# Quick GPU pass: all -m gpu tests in one process per file + headline bench (no profile).
mkdir -p gpurun_out
fail=0
step() {
  name=$1; shift
  t0=$(date +%s)
  timeout -k 10 ${TMO:-600} "$@" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "[$name] rc=$rc in $(( $(date +%s) - t0 ))s :: $(grep -E '^(FAILED|ERROR)|passed|failed|"metric"' gpurun_out/$name.log | tail -4 | cut -c1-300 | tr '\n' ' ')"
  if [ $rc -gt 1 ]; then echo "abnormal exit, stopping"; exit $rc; fi
  [ $rc -ne 0 ] && fail=1
}
step t_ops   python -m pytest tests/test_gpu_ops.py -m gpu -q -rA -p no:cacheprovider -k "${OPS_K:-test}"
step t_model python -m pytest tests/test_gpu_model.py -m gpu -q -rA -p no:cacheprovider
step t_train python -m pytest tests/test_gpu_train.py -m gpu -q -rA -p no:cacheprovider
step t_dp    python -m pytest tests/test_gpu_dp.py -m gpu -q -rA -p no:cacheprovider
step bench   python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:---no-cpu-baseline}
exit $fail
