"""BASELINE config 1 trained to convergence on the learnable synthetic corpus, as a pure function of seeds.
``oracle/gen_golden_r2.py`` trains the REFERENCE model on ``train_set()`` in the build container, scores ``heldout()`` with it
and stores checkpoint + scores + Cavg in tests/golden/cfg1_trained.npz; the GPU test scores the same held-out utterances with
the HIP path from the same checkpoint."""
import torch

L2V = {"a": 30, "b": 40, "c": 50}
L2I = {"a": 0, "b": 1, "c": 2}
DIMS = dict(n_blocks=2, encoder_dim=64, dim_head=16, heads=4, last_dim_head=8)
SECONDS, TEXT_LEN = 1.0, 8
TRAIN_ITEMS, HELD_ITEMS = 64, 24            # per language -> 72 held-out utterances
STEPS, BATCH, LR = 2000, 8, 0.01


def _ds(train, items, seed):
    from lid.raw_datasets import SyntheticMergedDataset
    return SyntheticMergedDataset(train, L2I, L2V, items_per_lang=items, seconds=SECONDS, text_len=TEXT_LEN, seed=seed,
                                  transcript="tones", type="mel", pad=16)


def train_set():
    """-> (wav (192, 16000) f32, texts (192, 8) int64, lang index (192,))"""
    ds = _ds(True, TRAIN_ITEMS, 1234)
    n = len(ds)
    return (torch.stack([ds.waveform(i) for i in range(n)]), torch.stack([ds.text(i) for i in range(n)]),
            torch.tensor([i // TRAIN_ITEMS for i in range(n)]))


def heldout():
    """-> (wav (72, 16000) f32, texts (72, 8), target language index (72,))"""
    ds = _ds(False, HELD_ITEMS, 1235)
    n = len(ds)
    return (torch.stack([ds.waveform(i) for i in range(n)]), torch.stack([ds.text(i) for i in range(n)]),
            torch.tensor([i // HELD_ITEMS for i in range(n)]))
