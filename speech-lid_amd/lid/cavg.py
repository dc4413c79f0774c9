"""Average detection cost Cavg for language recognition (reference: lid/cavg.py:82-117, arXiv:1706.09742).

``get_cavg(pairs, lang_num, min_score, max_score, bins, p_target)`` keeps the reference signature and result; the
implementation tallies the (language, target, score) triples once per threshold with numpy instead of three nested loops."""
import numpy as np


def get_cavg(pairs, lang_num, min_score, max_score, bins=20, p_target=0.5):
    arr = np.asarray([(p[0], p[1], p[2]) for p in pairs], dtype=np.float64)
    lang, tgt, score = arr[:, 0].astype(int), arr[:, 1].astype(int), arr[:, 2]
    is_tgt = lang == tgt
    n_tgt = np.bincount(lang[is_tgt], minlength=lang_num).astype(np.float64)
    non_idx = lang[~is_tgt] * lang_num + tgt[~is_tgt]
    n_non = np.bincount(non_idx, minlength=lang_num * lang_num).astype(np.float64)
    step = (max_score - min_score) / bins
    p_non = (1 - p_target) / (lang_num - 1)
    cavgs = []
    for section in range(bins + 1):
        thr = min_score + section * step
        miss = np.bincount(lang[is_tgt & (score < thr)], minlength=lang_num)
        fa = np.bincount((lang * lang_num + tgt)[(~is_tgt) & (score >= thr)], minlength=lang_num * lang_num)
        p_miss = np.divide(miss, n_tgt, out=np.zeros(lang_num), where=n_tgt > 0)
        p_fa = np.divide(fa, n_non, out=np.zeros(lang_num * lang_num), where=n_non > 0).reshape(lang_num, lang_num).sum(1)
        cavgs.append(float(np.sum(p_target * p_miss + p_non * p_fa) / lang_num))
    return cavgs, min(cavgs)
