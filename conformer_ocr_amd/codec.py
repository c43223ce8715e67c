"""Label <-> text codec with the interface of `kraken.lib.codec.PytorchCodec` that the reference
uses (pred.py:41,144,163,189,205; dataset.py:25,128,169-174).  kraken is a third-party package that
is not available here; this is a restatement of its published behaviour, host
Python because it runs once per decoded line on a handful of labels."""
from __future__ import annotations

from typing import Dict, List, Sequence, Set, Tuple, Union


class PytorchCodec:
    """`c2l` maps a grapheme to its label sequence (labels >= 1; 0 is the CTC blank)."""

    def __init__(self, charset: Union[Dict[str, Sequence[int]], Sequence[str], str], strict: bool = False):
        if isinstance(charset, dict):
            self.c2l = {k: [int(x) for x in v] for k, v in charset.items()}
        else:
            self.c2l = {k: [v] for v, k in enumerate(sorted(set(charset)), start=1)}
        self.strict = strict
        self.l2c: Dict[Tuple[int, ...], str] = {}
        for k, v in self.c2l.items():
            if any(x < 1 for x in v):
                raise ValueError(f'label sequence of {k!r} contains the blank / a negative label')
            if tuple(v) in self.l2c:
                raise ValueError(f'duplicate label sequence for {k!r}')
            self.l2c[tuple(v)] = k
        self.c_sorted = sorted(self.c2l.keys(), key=len, reverse=True)
        self.l_max_len = max((len(k) for k in self.l2c), default=0)
        self._l2c1 = {k[0]: v for k, v in self.l2c.items()} if self.l_max_len == 1 else None      # 1:1 codecs: one dict lookup per label

    def __len__(self) -> int:
        return len(self.c2l)

    @property
    def is_valid(self) -> bool:
        return len(self.l2c) == len(self.c2l)

    @property
    def max_label(self) -> int:
        return max((l for labels in self.c2l.values() for l in labels), default=0)

    def encode(self, s: str) -> List[int]:
        """Greedy longest-grapheme-match encoding to a flat label list."""
        labels: List[int] = []
        idx = 0
        while idx < len(s):
            for code in self.c_sorted:
                if s.startswith(code, idx):
                    labels.extend(self.c2l[code])
                    idx += len(code)
                    break
            else:
                if self.strict:
                    raise KeyError(f'Non-encodable sequence {s[idx:idx + 5]}... encountered.')
                idx += 1
        return labels

    def decode(self, labels: Sequence[Tuple[int, int, int, float]]) -> List[Tuple[str, int, int, float]]:
        """(label, start, end, conf) records -> (char, start, end, conf) records by longest match of
        label subsequences against `l2c`; undecodable labels are skipped."""
        if self._l2c1 is not None and not self.strict:
            get = self._l2c1.get
            return [(c, st, en, cf) for lab, st, en, cf in labels for code in (get(int(lab)),) if code is not None for c in code]
        start = [x[1] for x in labels]
        end = [x[2] for x in labels]
        con = [x[3] for x in labels]
        labs = tuple(int(x[0]) for x in labels)
        decoded: List[Tuple[str, int, int, float]] = []
        idx = 0
        while idx < len(labs):
            for i in range(min(self.l_max_len, len(labs) - idx), 0, -1):
                code = self.l2c.get(labs[idx:idx + i])
                if code is not None:
                    decoded.extend((c, start[idx], end[idx + i - 1], max(con[idx:idx + i])) for c in code)
                    idx += i
                    break
            else:
                if self.strict:
                    raise KeyError(f'Non-decodable sequence {labs[idx:idx + 5]}... encountered.')
                idx += 1
        return decoded

    def merge(self, codec: 'PytorchCodec') -> Tuple['PytorchCodec', Set]:
        c2l = dict(self.c2l)
        nxt = self.max_label + 1
        for k in codec.c2l:
            if k not in c2l:
                c2l[k] = [nxt]
                nxt += 1
        return PytorchCodec(c2l), set()


def ascii_codec(num_classes: int) -> PytorchCodec:
    """1:1 stand-in codec for synthetic runs: label i -> one printable character."""
    chars = [chr(c) for c in range(0x21, 0x7f)] + [chr(c) for c in range(0xa1, 0x2000)]
    return PytorchCodec({chars[i - 1]: [i] for i in range(1, num_classes)})
