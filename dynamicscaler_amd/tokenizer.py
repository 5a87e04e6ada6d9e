"""CLIP's byte-level BPE tokenizer (the `open_clip.tokenize` behind FrozenOpenCLIPEmbedder, lvdm/modules/encoders/condition.py:211).

open_clip_torch (requirements.txt:23) and its vocabulary file `bpe_simple_vocab_16e6.txt.gz` are not in this image, so this
module implements the published algorithm and takes the vocabulary file as an argument:

    tok = ClipBpeTokenizer("/path/to/bpe_simple_vocab_16e6.txt.gz")
    ids = tok(["a prompt", "another"])            # LongTensor [2, 77]: <start_of_text> ... <end_of_text>, zero padded
    embedder = FrozenOpenCLIPEmbedder(tokenizer=tok)

Algorithm (CLIP / open_clip `SimpleTokenizer`): text -> html-unescape, collapse whitespace, lower-case -> split by the CLIP
pattern (special tokens, the English contractions, letter runs, single digits, other non-space runs) -> every piece to its UTF-8
bytes, each byte mapped to a printable unicode character -> byte-pair merges in rank order with `</w>` marking the end of a piece
-> ids.  The id space is [256 byte symbols, the same 256 with `</w>`, one id per merge in file order, <start_of_text>,
<end_of_text>]; from the shipped file only the first 48 894 merges are used (vocabulary 49 408).  `ftfy.fix_text` (mojibake
repair, a no-op on clean text) runs when the package is importable.

Pinned like the CLIP towers (open_clip absent): against an independent implementation of the same algorithm -- transformers'
`CLIPTokenizer` on a synthetic vocabulary (tests/test_host_cpu.py::test_clip_bpe_tokenizer_vs_independent_implementation).
Host-side only; no GPU work.
"""
import gzip
import html
import re as _re
from functools import lru_cache

import torch

try:                                            # third-party `regex` for the unicode classes \\p{L} / \\p{N}
    import regex
except ImportError:                             # pragma: no cover
    regex = None

N_MERGES_CLIP = 49152 - 256 - 2                 # merges CLIP keeps from its vocabulary file


@lru_cache()
def byte_symbols():
    """byte value -> printable unicode character: the printable latin-1 bytes map to themselves, the other 68 to U+0100..."""
    keep = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _symbol_order():
    """The order in which CLIP numbers the 256 byte symbols: the self-mapped bytes first (ascending), then the remapped ones."""
    keep = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    rest = [b for b in range(256) if b not in keep]
    sym = byte_symbols()
    return [sym[b] for b in keep + rest]


class ClipBpeTokenizer:
    SOT, EOT = "<start_of_text>", "<end_of_text>"

    def __init__(self, bpe_path=None, merges=None, context_length=77):
        """bpe_path: CLIP's vocabulary file (gzip or plain text: a header line, then one merge "left right" per line);
        merges: the same as a list of (left, right) pairs (tests)."""
        if regex is None:
            raise ImportError("ClipBpeTokenizer needs the `regex` package (unicode letter / number classes)")
        if merges is None:
            if bpe_path is None:
                raise ValueError("ClipBpeTokenizer: pass bpe_path (open_clip's bpe_simple_vocab_16e6.txt.gz is not shipped) or merges")
            opener = gzip.open if str(bpe_path).endswith(".gz") else open
            with opener(bpe_path, "rb") as f:
                lines = f.read().decode("utf-8").split("\n")
            merges = [tuple(ln.split()) for ln in lines[1:N_MERGES_CLIP + 1] if ln.strip()]
        merges = [tuple(m) for m in merges][:N_MERGES_CLIP]
        if any(len(m) != 2 for m in merges):
            raise ValueError("ClipBpeTokenizer: every merge must be a (left, right) pair")
        symbols = _symbol_order()
        vocab = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + [self.SOT, self.EOT]
        self.ids = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.ids[self.SOT], self.ids[self.EOT]
        self.context_length = context_length
        self._pieces = {self.SOT: (self.SOT,), self.EOT: (self.EOT,)}
        self._pattern = regex.compile(
            regex.escape(self.SOT) + "|" + regex.escape(self.EOT) + r"|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
            regex.IGNORECASE)

    @property
    def vocab_size(self):
        return len(self.ids)

    @staticmethod
    def clean(text):
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return _re.sub(r"\s+", " ", text).strip().lower()

    def _merge(self, piece):
        """Byte-pair merges of one piece (already in byte symbols): repeatedly join the adjacent pair of lowest rank."""
        done = self._pieces.get(piece)
        if done is not None:
            return done
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word, word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    out.append(word[i] + word[i + 1])
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        done = self._pieces[piece] = tuple(word)
        return done

    def encode(self, text):
        sym = byte_symbols()
        out = []
        for piece in self._pattern.findall(self.clean(text)):
            mapped = "".join(sym[b] for b in piece.encode("utf-8"))
            out.extend(self.ids[t] for t in self._merge(mapped))
        return out

    def __call__(self, texts, context_length=None):
        """list[str] (or one str) -> LongTensor [b, context_length]: <start_of_text> ids <end_of_text>, zero padded; a text that is
        too long is cut and its last id set to <end_of_text> (open_clip.tokenize)."""
        if isinstance(texts, str):
            texts = [texts]
        n = self.context_length if context_length is None else context_length
        result = torch.zeros((len(texts), n), dtype=torch.long)
        for i, text in enumerate(texts):
            toks = [self.sot] + self.encode(text) + [self.eot]
            if len(toks) > n:
                toks = toks[:n]
                toks[-1] = self.eot
            result[i, :len(toks)] = torch.tensor(toks, dtype=torch.long)
        return result
