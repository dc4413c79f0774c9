"""Pins oracle/optim.py and oracle/metrics.py against traces of the reference's own ccml.optim / lid.cavg / lid.tokenizer."""
import numpy as np
import torch

from conftest import load_npz
from oracle import metrics as om
from oracle import optim as oo


def test_novograd_and_tristage_trace():
    g = load_npz("optim_trace.npz")
    params = [torch.from_numpy(g[f"p{j}_init"]).clone() for j in range(3)]
    states = [oo.NovogradState() for _ in range(3)]
    sched = oo.TriStage(lr=0.01, max_update=50)
    lrs = g["lrs"]
    np.testing.assert_allclose([sched.lr_at(k) for k in range(len(lrs))], lrs, rtol=1e-12)
    for k in range(4):
        grads = [None if (k == 1 and j == 2) else torch.from_numpy(g[f"g{k}_{j}"]) for j in range(3)]
        oo.novograd_step(params, grads, states, lr=float(lrs[k]), weight_decay=1e-5)
        for j in range(3):
            np.testing.assert_allclose(params[j].numpy(), g[f"p{j}_after{k}"], rtol=1e-6, atol=1e-7)


def test_cavg_kats():
    g = load_npz("metrics_kat.npz")
    for case in range(3):
        scores, tgt = g[f"cavg{case}_scores"], g[f"cavg{case}_tgt"]
        pairs = om.make_pairs(scores.tolist(), tgt.tolist())
        assert om.cavg(pairs, scores.shape[1]) == float(g[f"cavg{case}_value"])


def test_ctc_greedy_collapse_kat():
    g = load_npz("metrics_kat.npz")
    for seq, n, ref in zip(g["ctc_seqs"], g["ctc_lens"], g["ctc_decoded"]):
        ids = om.ctc_greedy_collapse(seq[:n].tolist(), blank=6)
        assert "".join(chr(ord("a") + i) for i in ids) == str(ref)


def test_score_to_prob():
    p = om.score_to_prob([-0.5, -0.25, -1.0])
    assert abs(sum(p) - 1) < 1e-12 and p[1] > p[0] > p[2]
