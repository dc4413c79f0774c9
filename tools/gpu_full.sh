#!/bin/bash
# Full GPU pass: every -m gpu test (one process per file), the headline bench, and a rocprofv3 kernel trace of the bench.
mkdir -p gpurun_out
fail=0
step() {
  name=$1; shift
  echo "=== $name"; t0=$(date +%s)
  timeout -k 10 ${TMO:-600} "$@" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "[$name] rc=$rc in $(( $(date +%s) - t0 ))s :: $(grep -E '^(FAILED|ERROR)|passed|failed|"metric"' gpurun_out/$name.log | tail -3 | cut -c1-400 | tr '\n' ' ')"
  if [ $rc -gt 1 ]; then echo "abnormal exit, stopping"; exit $rc; fi
  [ $rc -ne 0 ] && fail=1
}
step t_ops   python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider
step t_model python -m pytest tests/test_gpu_model.py -m gpu -q -rA -p no:cacheprovider
step t_train python -m pytest tests/test_gpu_train.py -m gpu -q -rA -p no:cacheprovider
step bench   python bench.py --steps 20 --warmup 5
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TMO=900 step prof rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
find gpurun_out/prof -name "*kernel_stats*" | head -3
exit $fail
