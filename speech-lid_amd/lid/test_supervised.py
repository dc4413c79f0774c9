"""On-disk result formats of the reference's test harness (lid/test_supervised.py:250-288), so that downstream scoring
scripts read this build's outputs unchanged:

  * ``write_to_file``  - submission file, TSV with header ``wav_name<TAB>text``;
  * ``write_to_csv``   - per-language validation dump ``<dir>/<lang>.csv``, TSV with header ``true pred <lang columns>``: one row
    per utterance with the reference text, the greedy transcript and the LID probability of every language.  The reference
    hard-codes its three language column names; here the columns are the module's languages in index order.

``score_dataset`` fills both from a dataset with the module's own inference path (HIP features, all heads, device-side
greedy decode, ``score_to_prob``).  The reference's harness also drives a kenlm language model and an HTTP enhancement server
(lid/test_supervised.py:90-200): private services outside the hot path, not rebuilt."""
import csv
import os
from typing import Dict, List, Sequence

import torch


def write_to_file(result_file: str, datas: Sequence[Sequence[str]]):
    """datas: [(wav_name, text), ...]"""
    os.makedirs(os.path.dirname(os.path.abspath(result_file)), exist_ok=True)
    with open(result_file, "w", newline="") as f:
        writer = csv.DictWriter(f, fieldnames=["wav_name", "text"], delimiter="\t")
        writer.writeheader()
        for name, text in datas:
            writer.writerow({"wav_name": name, "text": text})


def write_to_csv(result_file: str, trues: List[str], preds: List[str], probs: List[Sequence[float]], lang: str = "none",
                 lang_names: Sequence[str] = ("Persian", "Swahili", "Vietnamese")):
    """Writes ``<dirname(result_file)>/<lang>.csv`` (the reference derives the path the same way)."""
    out = os.path.join(os.path.dirname(os.path.abspath(result_file)), lang + ".csv")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    fields = ["true", "pred"] + list(lang_names)
    with open(out, "w", newline="") as f:
        writer = csv.DictWriter(f, fieldnames=fields, delimiter="\t")
        writer.writeheader()
        for t, p, pr in zip(trues, preds, probs):
            row = {"true": t, "pred": p}
            row.update({name: pr[i] for i, name in enumerate(lang_names)})
            writer.writerow(row)
    return out


def score_dataset(module, dataset, result_file: str, device=None) -> Dict[str, float]:
    """Transcribe + language-score every utterance of ``dataset`` (items ``(wav, text ids, path, lang)``) with ``module``
    (a LidSuperviseModule on the GPU) and write the reference's result files.  Returns {"acc": ..., "cavg": ...}."""
    from lid.audio_processor import normalize_wav
    from lid.eer import CAvg
    device = device or next(module.model.parameters()).device
    index2lang = {v: k for k, v in module.lang2index_dict.items()}
    names = [index2lang[i] for i in range(len(index2lang))]
    per_lang: Dict[str, list] = {n: [] for n in names}
    submission, metric, correct = [], CAvg(num_class=len(names)), 0
    was = module.model.training
    module.model.eval()
    for i in range(len(dataset)):
        wav, text, path, lang = dataset[i]
        # the reference's harness normalises the waveform itself and then calls infer_tensor (pad = 0 features)
        x = normalize_wav(wav.reshape(1, -1).to(device))
        texts, lid_asr, _ = module.infer_tensor(x, module.sr, None)
        prob = module.score_to_prob(lid_asr[0].tolist())
        pred_lang = names[int(torch.tensor(prob).argmax())]
        correct += int(pred_lang == lang)
        metric.update([prob], [module.lang2index_dict[lang]])
        tok = module.tokenizer_dict[lang]
        truth = tok.decoder(text.reshape(1, -1), torch.tensor([text.numel()]))[0]
        per_lang[lang].append((truth, texts[lang][0], prob))
        submission.append((os.path.basename(path), texts[pred_lang][0]))
    module.model.train(was)
    write_to_file(result_file, submission)
    for lang, rows in per_lang.items():
        if rows:
            write_to_csv(result_file, [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], lang, names)
    return {"acc": correct / max(len(dataset), 1), "cavg": metric.compute()}
