#!/bin/bash
# Round-2 evidence in one gpurun call: every -m gpu test, the default bench line, a rocprofv3 kernel trace of the bench and
# the two PMC passes for HBM traffic.  Everything lands in gpurun_out/ (copy what is to be judged into profiles/r02/).
mkdir -p gpurun_out
echo "=== tests"; timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r02_tests.log 2>&1; echo "tests rc=$? :: $(tail -1 gpurun_out/r02_tests.log)"
echo "=== bench"; timeout -k 10 900 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"; grep "bench " gpurun_out/r02_bench.err | tail -12
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "=== rocprofv3 stats"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 --resident 1 > gpurun_out/r02_prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/r02_prof -name "*kernel_stats*" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r02_kernel_stats.csv && head -12 gpurun_out/r02_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/r02_prof
echo "=== PMC"; bash tools/gpu_pmc_bench.sh 2>&1 | tail -8
