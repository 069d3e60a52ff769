"""CLIP byte-level BPE tokenizer (host side, integer work).

Same observable behaviour as the reference's `tokenize` / `SimpleTokenizer.encode`
(rdm/modules/custom_clip/clip.py:127-143, simple_tokenizer.py:62-132): lower-cased, whitespace-collapsed text is
split by the CLIP regex, every piece is mapped to printable byte symbols and merged greedily by merge rank;
ids = [SOT] + pieces + [EOT], zero padded / truncated to 77.  `ftfy.fix_text` is not available offline and is
skipped (identity on already-clean captions; KATs in tests/golden/tokenizer.npz).
"""
import gzip
import html
import os
from functools import lru_cache

import numpy as np
import regex

_VOCAB_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz")
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       regex.IGNORECASE)
_END = "</w>"


def _byte_symbols():
    """GPT-2 style table: every byte value gets a printable unicode stand-in."""
    keep = [b for b in range(256) if (33 <= b <= 126) or (161 <= b <= 172) or (174 <= b <= 255)]
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class BPETokenizer:
    def __init__(self, vocab_file: str = _VOCAB_FILE, n_merges: int = 49152 - 256 - 2):
        lines = gzip.open(vocab_file, "rt", encoding="utf-8").read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:n_merges + 1]]
        self.rank = {m: i for i, m in enumerate(merges)}
        sym = _byte_symbols()
        # id order: byte symbols in printable-first order, their word-final forms, merges, specials
        ordered = [chr(b) for b in range(256) if sym[b] == chr(b)] + [sym[b] for b in range(256) if sym[b] != chr(b)]
        vocab = ordered + [s + _END for s in ordered] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.ids = {tok: i for i, tok in enumerate(vocab)}
        self.sym = sym
        self.sot, self.eot = self.ids["<|startoftext|>"], self.ids["<|endoftext|>"]

    @lru_cache(maxsize=65536)
    def _merge(self, piece: str):
        parts = list(piece[:-1]) + [piece[-1] + _END]
        while len(parts) > 1:
            best, where = None, -1
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, where = r, i
            if best is None:
                break
            a, b = parts[where], parts[where + 1]
            out, i = [], 0
            while i < len(parts):                      # merge every occurrence of the best pair, left to right
                if i < len(parts) - 1 and parts[i] == a and parts[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(parts[i]); i += 1
            parts = out
        return tuple(parts)

    def encode(self, text: str):
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        out = []
        for piece in _SPLIT.findall(text):
            if piece in ("<|startoftext|>", "<|endoftext|>"):
                out.append(self.ids[piece]); continue
            mapped = "".join(self.sym[b] for b in piece.encode("utf-8"))
            out.extend(self.ids[p] for p in self._merge(mapped))
        return out


@lru_cache(maxsize=1)
def default_tokenizer():
    return BPETokenizer()


def tokenize(texts, context_length: int = 77):
    """-> int64 numpy [len(texts), context_length] (rdm/modules/custom_clip/clip.py:127-143)."""
    if isinstance(texts, str):
        texts = [texts]
    tk = default_tokenizer()
    out = np.zeros((len(texts), context_length), dtype=np.int64)
    for i, t in enumerate(texts):
        ids = [tk.sot] + tk.encode(t) + [tk.eot]
        if len(ids) > context_length:
            print(f"WARNING: Input of length {len(ids)} is too long for context length {context_length}. Cutting.")
            ids = ids[:context_length]
        out[i, :len(ids)] = ids
    return out
