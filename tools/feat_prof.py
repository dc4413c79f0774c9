#!/usr/bin/env python3
"""Eager loop of lidk_wav2mel for rocprofv3 --kernel-trace --stats (no graph capture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops
B, L, PAD = 64, 48000, 16
wav = torch.randn(B, L, device="cuda:0")
spans = torch.tensor([[[10, 20, 5, 12], [100, 110, 30, 40]]] * B, dtype=torch.int32, device="cuda:0")
out = torch.empty(B, 1 + (L + 2 * PAD) // 160, 80, device="cuda:0")
for _ in range(30):
    ops.wav2mel(wav, pad=PAD, spans=spans, out=out, seed=7)
torch.cuda.synchronize()
print("done")
