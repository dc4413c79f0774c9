"""Holds the PRODUCT's host-side code (not the oracle) to the vectors captured from the reference (tests/golden/):
``ccml/optim/tri_state.py``, ``lid/cavg.py``, ``lid/eer.py``, ``lid/tokenizer.py``, ``LidSuperviseModule.score_to_prob``
and the mel filterbank the HIP log-mel kernel is fed with (``lidk/ops.py::melscale_fbanks``)."""
import numpy as np
import pytest
import torch

from conftest import load_npz


def test_product_tristage_matches_reference_lr_trace():
    """ccml/optim/tri_state.py against the LR sequence the reference's TriStageLRSchedule produced (optim_trace.npz)."""
    from ccml.optim.tri_state import TriStageLRSchedule
    g = load_npz("optim_trace.npz")
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=0.01)
    sched = TriStageLRSchedule(optimizer=opt, phase_ratio=[0.1, 0.4, 0.5], init_lr_scale=0.05, final_lr_scale=0.02, max_update=50,
                               lr=0.01)
    got = [opt.param_groups[0]["lr"]]
    for _ in range(len(g["lrs"]) - 1):
        opt.step()
        sched.step()
        got.append(opt.param_groups[0]["lr"])
    np.testing.assert_allclose(got, g["lrs"], rtol=1e-12)
    # save / resume carries no extra counter: a fresh schedule loaded from the state dict continues the same sequence
    opt2 = torch.optim.SGD([p], lr=0.01)
    s2 = TriStageLRSchedule(optimizer=opt2, phase_ratio=[0.1, 0.4, 0.5], init_lr_scale=0.05, final_lr_scale=0.02, max_update=50,
                            lr=0.01)
    for _ in range(7):
        opt2.step()
        s2.step()
    s3 = TriStageLRSchedule(optimizer=torch.optim.SGD([p], lr=0.01), phase_ratio=[0.1, 0.4, 0.5], init_lr_scale=0.05,
                            final_lr_scale=0.02, max_update=50, lr=0.01)
    s3.load_state_dict(s2.state_dict())
    assert s3.lr_at(s3.last_epoch) == pytest.approx(float(g["lrs"][7]), rel=1e-12)


def test_product_cavg_and_metric_class_match_reference_kats():
    """lid/cavg.py::get_cavg (vectorised rewrite) and lid/eer.py::CAvg against lid/cavg.py:82-117 of the reference."""
    from lid.cavg import get_cavg
    from lid.eer import CAvg
    g = load_npz("metrics_kat.npz")
    for case in range(3):
        scores, tgt = g[f"cavg{case}_scores"], g[f"cavg{case}_tgt"]
        n_lang = scores.shape[1]
        pairs = [(j, int(t), float(row[j])) for row, t in zip(scores, tgt) for j in range(n_lang)]
        lo, hi = min(p[2] for p in pairs), max(p[2] for p in pairs)
        cavgs, mn = get_cavg(pairs, n_lang, lo, hi, 20, 0.5)
        assert len(cavgs) == 21 and round(mn, 4) == float(g[f"cavg{case}_value"])
        m = CAvg(num_class=n_lang)
        for row, t in zip(scores.tolist(), tgt.tolist()):             # one utterance per update, as val_loop does
            m.update([row], [t])
        assert m.compute() == float(g[f"cavg{case}_value"])
        m.reset()
        assert m.compute() == 0.0


def test_product_tokenizer_greedy_collapse_matches_reference_kat():
    from lid.tokenizer import CTCTokenizer
    g = load_npz("metrics_kat.npz")
    tok = CTCTokenizer([chr(ord("a") + i) for i in range(6)])
    dec = tok.ctc_decode(torch.from_numpy(g["ctc_seqs"]), torch.from_numpy(g["ctc_lens"]))
    assert dec == [str(s) for s in g["ctc_decoded"]]
    assert tok.blank_id == 6 and tok.export_vocab() == list("abcdef")
    assert tok.encoder("fab?c").tolist() == [5, 0, 1, 2]                       # unknown symbols are dropped
    assert tok.decoder(torch.tensor([[0, 1, 2, 0]]), torch.tensor([3])) == ["abc"]


def test_product_score_to_prob_matches_reference_formula():
    """LidSuperviseModule.score_to_prob: p = -1/(s - 1e-9), normalised (reference lid/LidModule_ASR_Supervised.py:316-318)."""
    from lid.LidModule_ASR_Supervised import LidSuperviseModule
    from oracle import metrics as om
    rng = np.random.RandomState(0)
    for _ in range(20):
        s = (-rng.rand(14) * 2).tolist()
        got = LidSuperviseModule.score_to_prob(s)
        np.testing.assert_allclose(got, om.score_to_prob(s), rtol=1e-14)
        assert abs(sum(got) - 1) < 1e-12 and int(np.argmax(got)) == int(np.argmax(s))
    assert LidSuperviseModule.score_to_prob([0.0, -0.5])[0] > 0.999999           # all-blank head: s = 0 -> p ~ 1e9 (the quirk)


def test_product_eer_matches_roc_brentq():
    """lid/eer.py::EER2.compute against the reference's recipe (lid/eer.py:59-64): sklearn roc_curve + brentq on
    1 - x - interp1d(fpr, tpr)(x).  sklearn and scipy are in the image; the reference's formula is restated here in 3 lines."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve
    from lid.eer import EER2
    rng = np.random.RandomState(3)
    for n_utt, n_lang in ((50, 3), (200, 14), (9, 2)):
        scores = rng.rand(n_utt, n_lang)
        tgt = rng.randint(0, n_lang, n_utt)
        scores[np.arange(n_utt), tgt] += 0.4 * rng.rand(n_utt)
        m = EER2(num_class=n_lang)
        for row, t in zip(scores.tolist(), tgt.tolist()):
            m.update([row], [t])
        fpr, tpr, _ = roc_curve(m.pos_list, m.score_list)
        want = brentq(lambda x: 1.0 - x - interp1d(list(fpr), list(tpr))(x), 0, 1.0)
        assert m.compute() == pytest.approx(want, abs=1e-9)


def test_mel_filterbank_cross_check_against_an_independent_implementation():
    """The oracle's and the product's HTK filterbanks are the same formula typed twice; hold BOTH to an
    independent implementation that ships in the image: transformers.audio_utils.mel_filter_bank(htk, norm=None)."""
    from transformers.audio_utils import mel_filter_bank
    from lidk.ops import melscale_fbanks as product_fb
    from oracle.features import melscale_fbanks as oracle_fb
    ind = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                          sampling_rate=16000, norm=None, mel_scale="htk")
    assert ind.shape == (257, 80)
    for name, fb in (("oracle", oracle_fb()), ("product", product_fb())):
        err = float(np.abs(fb.numpy().astype(np.float64) - ind).max())
        print(f"[mel fb {name} vs transformers] max abs diff {err:.2e}")
        assert err <= 2e-5, name
    assert torch.equal(oracle_fb(), product_fb())


def test_result_file_formats(tmp_path):
    """lid/test_supervised.py writers: the reference's TSV layouts (lid/test_supervised.py:250-288)."""
    import csv
    from lid.test_supervised import write_to_csv, write_to_file
    res = str(tmp_path / "out" / "result.txt")
    write_to_file(res, [("a.wav", "hello"), ("b.wav", "")])
    rows = list(csv.DictReader(open(res), delimiter="\t"))
    assert rows == [{"wav_name": "a.wav", "text": "hello"}, {"wav_name": "b.wav", "text": ""}]
    path = write_to_csv(res, ["xy", "z"], ["xv", "z"], [[0.7, 0.2, 0.1], [0.1, 0.1, 0.8]], lang="Swahili")
    assert path.endswith("out/Swahili.csv")
    rows = list(csv.DictReader(open(path), delimiter="\t"))
    assert list(rows[0]) == ["true", "pred", "Persian", "Swahili", "Vietnamese"]
    assert rows[1]["true"] == "z" and float(rows[1]["Vietnamese"]) == 0.8
