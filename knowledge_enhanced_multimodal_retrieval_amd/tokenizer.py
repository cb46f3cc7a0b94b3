"""`clip.tokenize` for the build: byte-level BPE of openai/CLIP, restated from its published algorithm.

The reference calls ``clip.tokenize(texts, truncate=True)`` (/root/reference/src/clip/eval/evaluator_baseline.py:112,118;
src/clip/eval/evaluator.py:126,132) and feeds the ``[B, 77]`` ids to ``encode_text``.  The tokenizer needs the merge
table ``bpe_simple_vocab_16e6.txt.gz`` that ships with the ``clip`` package; it is not part of this repository (no
network in the build container).  It is looked up in this order:

1. ``$KEMR_BPE_VOCAB``, 2. ``<this package>/bpe_simple_vocab_16e6.txt.gz``, 3. ``<repo>/clip/bpe_simple_vocab_16e6.txt.gz``,
4. ``~/.cache/clip/bpe_simple_vocab_16e6.txt.gz``.

Without the table a deterministic **hash tokenizer** is used (word -> stable id in [1, SOT)), with a warning: ids are
then not OpenAI's, which is fine for synthetic benchmarking / plumbing tests and wrong for pretrained weights.
``ftfy`` is not installed here; text cleaning is ``html.unescape`` + whitespace collapse + lower-casing.
"""
from __future__ import annotations

import gzip
import hashlib
import html
import os
import warnings
from functools import lru_cache
from typing import List, Sequence, Union

import regex as re
import torch

SOT, EOT, CONTEXT = 49406, 49407, 77
_PAT = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                  re.IGNORECASE)


def _vocab_candidates():
    here = os.path.dirname(os.path.abspath(__file__))
    name = "bpe_simple_vocab_16e6.txt.gz"
    env = os.environ.get("KEMR_BPE_VOCAB")
    return [p for p in (env, os.path.join(here, name), os.path.join(os.path.dirname(here), "clip", name),
                        os.path.expanduser(os.path.join("~", ".cache", "clip", name))) if p]


def find_vocab():
    for p in _vocab_candidates():
        if os.path.exists(p):
            return p
    return None


@lru_cache()
def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return set(zip(word[:-1], word[1:]))


def clean(text: str) -> str:
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip().lower()


class BPETokenizer:
    """Byte-level BPE with the 49 408-entry CLIP vocabulary (256 bytes, 256 end-of-word bytes, merges, 2 specials)."""

    def __init__(self, bpe_path: str):
        self.path = bpe_path
        self.byte_encoder = bytes_to_unicode()
        merges = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in merges[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab.extend("".join(m) for m in merges)
        vocab.extend(["<|startoftext|>", "<|endoftext|>"])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in re.findall(_PAT, clean(text)):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids


class HashTokenizer:
    """Offline stand-in: one stable id per word piece (NOT OpenAI's ids; see module docstring)."""

    def encode(self, text: str) -> List[int]:
        out = []
        for tok in re.findall(_PAT, clean(text)):
            h = int.from_bytes(hashlib.blake2s(tok.encode("utf-8"), digest_size=4).digest(), "little")
            out.append(1 + h % (SOT - 1))
        return out


_tokenizer = None
_allow_hash = False


def allow_hash_tokenizer(flag: bool = True) -> None:
    """Opt in to the hash tokenizer when the BPE vocabulary is absent (synthetic-data runs, tests)."""
    global _allow_hash
    _allow_hash = bool(flag)


def hash_tokenizer_allowed() -> bool:
    return _allow_hash or os.environ.get("KEMR_ALLOW_HASH_TOKENIZER", "") == "1"


def get_tokenizer():
    """The BPE tokenizer when its vocabulary file is found; without it: an error, unless the hash tokenizer was allowed
    explicitly (its ids are not OpenAI's: text embeddings of a pretrained model would be meaningless)."""
    global _tokenizer
    if _tokenizer is None or (isinstance(_tokenizer, HashTokenizer) and not hash_tokenizer_allowed()):
        path = find_vocab()
        if path:
            _tokenizer = BPETokenizer(path)
        elif hash_tokenizer_allowed():
            warnings.warn("CLIP BPE vocabulary not found (%s): using the deterministic hash tokenizer (explicitly allowed); "
                          "token ids are not OpenAI's" % ", ".join(_vocab_candidates()), RuntimeWarning, stacklevel=2)
            _tokenizer = HashTokenizer()
        else:
            raise FileNotFoundError(
                "CLIP BPE vocabulary not found (%s).  Put bpe_simple_vocab_16e6.txt.gz there, or opt in to the hash tokenizer "
                "(synthetic-data runs: --synthetic, tokenizer.allow_hash_tokenizer(), KEMR_ALLOW_HASH_TOKENIZER=1)"
                % ", ".join(_vocab_candidates()))
    return _tokenizer


def tokenizer_name() -> str:
    t = get_tokenizer()
    return "bpe(%s)" % t.path if isinstance(t, BPETokenizer) else "hash (NOT OpenAI ids)"


def tokenize(texts: Union[str, Sequence[str]], context_length: int = CONTEXT, truncate: bool = False) -> torch.Tensor:
    """-> int32 [len(texts), context_length]: SOT, ids..., EOT, zero padding; over-long inputs are cut and end in EOT
    when ``truncate`` else raise (upstream ``clip.tokenize`` contract)."""
    if isinstance(texts, str):
        texts = [texts]
    tok = get_tokenizer()
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, text in enumerate(texts):
        ids = [SOT] + tok.encode(text) + [EOT]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = EOT
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out
