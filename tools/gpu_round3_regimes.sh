#!/bin/bash
# Round-3 evidence for BASELINE configs 4 and 5: WavLM regimes, wav2vec2 backbone with bucketed 1-10 s utterances, ragged Conformer.
mkdir -p gpurun_out
for r in frozen finetune; do
  timeout -k 10 300 python3 bench.py --model wavlm --wavlm-regime $r --steps 40 --warmup 8 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 2>&1 | grep '^{' > gpurun_out/bench_wavlm_$r.json; echo "wavlm $r rc=$? $(cut -c1-260 gpurun_out/bench_wavlm_$r.json)"
done
timeout -k 10 300 python3 bench.py --model w2v2 --ragged --steps 40 --warmup 8 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 2>&1 | grep '^{' > gpurun_out/bench_w2v2_ragged.json; echo "w2v2 ragged rc=$? $(cut -c1-260 gpurun_out/bench_w2v2_ragged.json)"
timeout -k 10 300 python3 bench.py --ragged --steps 60 --warmup 10 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 2>&1 | grep '^{' > gpurun_out/bench_conformer_ragged.json; echo "conformer ragged rc=$? $(cut -c1-260 gpurun_out/bench_conformer_ragged.json)"
