#!/usr/bin/env python3
"""Feature-path microbenchmark: raw waveform -> log-mel (lidk_wav2mel) on BASELINE config 2's batch (64 x 3 s), timed by replaying
a captured hipGraph of N back-to-back calls between two HIP events.  Algorithmic bytes = 4 L + 4 * 80 * F per utterance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch  # noqa: E402
from lidk import ops  # noqa: E402

B, L, PAD, N = 64, 48000, 16, 20
dev = "cuda:0"
torch.manual_seed(0)
wav = torch.randn(B, L, device=dev)
spans = torch.tensor([[[10, 20, 5, 12], [100, 110, 30, 40]]] * B, dtype=torch.int32, device=dev)
out = torch.empty(B, 1 + (L + 2 * PAD) // 160, 80, device=dev)
for _ in range(3):
    ops.wav2mel(wav, pad=PAD, spans=spans, out=out, seed=7)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(N):
        ops.wav2mel(wav, pad=PAD, spans=spans, out=out, seed=7)
g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / N)
t = sorted(ts)[len(ts) // 2]
nbytes = B * (4 * L + 4 * 80 * out.shape[1])
print(f"wav2mel B={B} L={L}: {t * 1e3:.1f} us per batch, {nbytes / (t * 1e-3) / 1e9:.0f} GB/s = {nbytes / (t * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s "
      f"(LIDK_STFT_V1={os.environ.get('LIDK_STFT_V1', '0')})")
