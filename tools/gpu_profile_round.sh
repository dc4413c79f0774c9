#!/bin/bash
# Round-end evidence: the full bench line and the rocprofv3 kernel stats of the same command.  Outputs in gpurun_out/
# (copy bench_full.json / kernel_stats.csv into profiles/rNN/).  HBM traffic counters: tools/gpu_pmc_bench.sh.
# The profiled run times the same steps without the Cavg / fit legs (their ~3 000 extra steps make the raw trace exceed what
# gpurun copies back); the raw trace is deleted once the stats are extracted.
mkdir -p gpurun_out && rm -rf gpurun_out/prof
timeout -k 10 500 python3 bench.py $BENCH_ARGS > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_full.json; cut -c1-400 gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 $BENCH_ARGS > gpurun_out/prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv; head -12 gpurun_out/kernel_stats.csv | cut -c1-160
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ] && [ -f tools/timeline_busy.py ]; then python3 tools/timeline_busy.py "$t" gpurun_out/step_timeline.txt > gpurun_out/timeline_busy.txt 2>&1; tail -6 gpurun_out/timeline_busy.txt; fi
rm -rf gpurun_out/prof
