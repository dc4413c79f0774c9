"""Oracle: LID score transform, Cavg and greedy CTC collapse (test infrastructure)."""
from typing import List, Sequence, Tuple


def score_to_prob(scores: Sequence[float]) -> List[float]:
    """lid/LidModule_ASR_Supervised.py:316-318 — p_i = -1/(s_i - 1e-9), normalised to sum 1."""
    prob = [(-1 / (s - 1e-9)) for s in scores]
    tot = sum(prob)
    return [p / tot for p in prob]


def make_pairs(probs: Sequence[Sequence[float]], targets: Sequence[int]) -> List[Tuple[int, int, float]]:
    """CAvg.update — lid/eer.py:76-79: one (lang_j, target, p_j) triple per utterance and language."""
    return [(j, t, p[j]) for p, t in zip(probs, targets) for j in range(len(p))]


def cavg(pairs, lang_num: int, bins: int = 20, p_target: float = 0.5) -> float:
    """CAvg.compute/get_cavg — lid/eer.py:81-123 (= lid/cavg.py:82-117): 21 thresholds between the
    global min and max score, per-language 0.5*P_miss + 0.5/(C-1)*sum P_fa, min over thresholds, round 4."""
    lo = min(p[2] for p in pairs)
    hi = max(p[2] for p in pairs)
    step = (hi - lo) / bins
    best = None
    for section in range(bins + 1):
        thr = lo + section * step
        tot = 0.0
        for lang in range(lang_num):
            n_tgt = n_miss = 0.0
            n_non = [0.0] * lang_num
            n_fa = [0.0] * lang_num
            for (l, t, s) in pairs:
                if l != lang:
                    continue
                if t == lang:
                    n_tgt += 1
                    n_miss += s < thr
                else:
                    n_non[t] += 1
                    n_fa[t] += s >= thr
            p_miss = n_miss / n_tgt if n_tgt else 0.0
            p_fa = sum(f / n for f, n in zip(n_fa, n_non) if n)
            tot += p_target * p_miss + (1 - p_target) / (lang_num - 1) * p_fa
        c = tot / lang_num
        best = c if best is None else min(best, c)
    return round(best, 4)


def ctc_greedy_collapse(ids: Sequence[int], blank: int) -> List[int]:
    """CTCTokenizer.ctc_decode inner loop — lid/tokenizer.py:60-65."""
    out, prev = [], blank
    for p in ids:
        if (p != prev or prev == blank) and p != blank:
            out.append(p)
        prev = p
    return out
