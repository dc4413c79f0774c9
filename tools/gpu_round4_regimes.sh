#!/bin/bash
# Round-4 evidence for BASELINE configs 4 and 5 (gpurun_out/ -> profiles/r04/): WavLM regimes, wav2vec2 Base and XLS-R 300M (wav2vec2
# Large) with bucketed 1-10 s utterances, ragged Conformer - each as a bench line; then rocprofv3 kernel stats + the queue-busy
# summary (tools/timeline_busy.py: union of kernel intervals against the step time = "GPU-bound or host-bound", measured) for the
# WavLM fine-tune, wav2vec2 ragged, XLS-R ragged fine-tune and ragged Conformer workloads.  Ragged workloads warm up for 50-60 steps: every
# (shape, language) pair pays its graph captures once, and a short warm-up reports those instead of the steady state.
mkdir -p gpurun_out
C="--no-cpu-baseline --cavg-steps 0 --fit-epochs 0"
for r in frozen finetune; do
  timeout -k 10 300 python3 bench.py --model wavlm --wavlm-regime $r --steps 40 --warmup 8 $C 2>&1 | grep '^{' > gpurun_out/bench_wavlm_$r.json; echo "wavlm $r rc=$? $(cut -c1-260 gpurun_out/bench_wavlm_$r.json)"
done
timeout -k 10 300 python3 bench.py --model w2v2 --ragged --steps 40 --warmup 50 $C 2>&1 | grep '^{' > gpurun_out/bench_w2v2_ragged.json; echo "w2v2 ragged rc=$? $(cut -c1-260 gpurun_out/bench_w2v2_ragged.json)"
timeout -k 10 300 python3 bench.py --ragged --steps 60 --warmup 60 $C 2>&1 | grep '^{' > gpurun_out/bench_conformer_ragged.json; echo "conformer ragged rc=$? $(cut -c1-260 gpurun_out/bench_conformer_ragged.json)"
for r in frozen finetune; do
  timeout -k 10 400 python3 bench.py --model xlsr --ragged --wavlm-regime $r --steps 30 --warmup 50 $C > gpurun_out/bench_xlsr_$r.log 2>&1; grep '^{' gpurun_out/bench_xlsr_$r.log > gpurun_out/bench_xlsr_ragged_$r.json; echo "xlsr $r rc=$? $(cut -c1-260 gpurun_out/bench_xlsr_ragged_$r.json)"; tail -3 gpurun_out/bench_xlsr_$r.log | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
prof() {   # tag, bench args, delimiter kernel, warm-up steps
  rm -rf gpurun_out/prof
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py $2 --steps 12 --warmup ${4:-4} $C > gpurun_out/prof_$1.log 2>&1; echo "prof $1 rc=$?"
  grep '^{' gpurun_out/prof_$1.log | cut -c1-200
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${1}_kernel_stats.csv
  t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
  TIMELINE_DELIM=${3:-wavlm_conv0_stats_kernel} python3 tools/timeline_busy.py "$t" gpurun_out/${1}_step_timeline.txt > gpurun_out/${1}_timeline_busy.txt 2>&1; tail -3 gpurun_out/${1}_timeline_busy.txt
  rm -rf gpurun_out/prof gpurun_out/${1}_step_timeline.txt
}
prof wavlm_finetune "--model wavlm --wavlm-regime finetune"
prof w2v2_ragged "--model w2v2 --ragged" wavlm_conv0_stats_kernel 50
prof xlsr_finetune "--model xlsr --ragged --wavlm-regime finetune" wav_layernorm_kernel 30
prof conformer_ragged "--ragged" wav_stats_kernel 60
