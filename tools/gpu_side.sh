#!/bin/bash
# deferred weight-gradient overlap (LIDK_SIDE_STREAM=1): model parity tests and bench, beside the single-stream default
mkdir -p gpurun_out
for m in 0 1; do
  export LIDK_SIDE_STREAM=$m
  timeout -k 10 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/t_model_side$m.log 2>&1; echo "side=$m t_model rc=$? $(tail -1 gpurun_out/t_model_side$m.log)"
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_side$m.log 2>&1; echo "side=$m bench rc=$? $(grep '^{' gpurun_out/bench_side$m.log | cut -c1-200)"
done
