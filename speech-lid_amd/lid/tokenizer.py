"""CTCTokenizer: vocabulary <-> ids, greedy CTC collapse (reference: lid/tokenizer.py:10-100; prefix beam search is out of
scope)."""
from typing import List, Union

import torch


class CTCTokenizer:
    def __init__(self, vocab: Union[str, list]) -> None:
        if isinstance(vocab, str):
            with open(vocab) as f:
                symbols = [line.rstrip("\n") for line in f]
        elif isinstance(vocab, (list, tuple)):
            symbols = list(vocab)
        else:
            raise Exception("vocab is neither a path nor a list")
        self.labels_map = dict(enumerate(symbols))
        self.s2labels_map = {s: i for i, s in self.labels_map.items()}
        self.blank_id = len(symbols)

    def export_vocab(self) -> List[str]:
        return [self.labels_map[i] for i in range(len(self.labels_map))]

    def encoder(self, text: str) -> torch.Tensor:
        return torch.LongTensor([self.s2labels_map[c] for c in text if c in self.s2labels_map])

    def ctc_decode(self, predictions: torch.Tensor, predictions_len: torch.Tensor = None) -> List[str]:
        """Greedy path -> text: drop repeats not separated by blank, then drop blanks.  One device->host copy per batch."""
        ids = predictions.long().cpu().tolist()
        lens = predictions_len.long().cpu().tolist() if predictions_len is not None else [len(r) for r in ids]
        out = []
        for row, n in zip(ids, lens):
            prev, chars = self.blank_id, []
            for p in row[:n]:
                if p != self.blank_id and (p != prev or prev == self.blank_id):
                    chars.append(self.labels_map[p])
                prev = p
            out.append("".join(chars))
        return out

    def ids_to_text(self, ids: torch.Tensor, lengths: torch.Tensor) -> List[str]:
        """Collapsed symbol ids (B, T) + counts (B,) from the device-side greedy decode (lidk.ops.ctc_greedy) -> text.
        One small device->host copy per batch; no per-frame work on the host."""
        rows, lens = ids.cpu().tolist(), lengths.cpu().tolist()
        return ["".join(self.labels_map[c] for c in row[:n]) for row, n in zip(rows, lens)]

    def decoder(self, targets: torch.Tensor, target_lengths: torch.Tensor) -> List[str]:
        ids = targets.long().cpu().tolist()
        lens = target_lengths.long().cpu().tolist()
        return ["".join(self.labels_map.get(c, "_") for c in row[:n]) for row, n in zip(ids, lens)]
