#!/bin/bash
# rocprofv3 kernel stats + queue-busy summary of the XLS-R ragged fine-tune workload -> gpurun_out/xlsr_finetune_*
mkdir -p gpurun_out; rm -rf gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C="--no-cpu-baseline --cavg-steps 0 --fit-epochs 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --model xlsr --ragged --wavlm-regime ${REGIME:-finetune} --steps 10 --warmup 30 $C > gpurun_out/prof_xlsr.log 2>&1; echo "prof rc=$?"
grep '^{' gpurun_out/prof_xlsr.log | cut -c1-200
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/xlsr_${REGIME:-finetune}_kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
TIMELINE_DELIM=wav_layernorm_kernel python3 tools/timeline_busy.py "$t" gpurun_out/xlsr_step_timeline.txt > gpurun_out/xlsr_${REGIME:-finetune}_timeline_busy.txt 2>&1; tail -3 gpurun_out/xlsr_${REGIME:-finetune}_timeline_busy.txt
rm -rf gpurun_out/prof gpurun_out/xlsr_step_timeline.txt
