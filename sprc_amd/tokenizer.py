"""WordPiece tokeniser for the text side of `inference` (R9; host-side).

The reference uses the third-party `transformers.BertTokenizer("bert-base-uncased")` plus one added
special token "[DEC]" (lavis/models/blip2_models/blip2.py:30-34) and calls it as
``tokenizer(text, padding="max_length", truncation=True, max_length=32, return_tensors="pt")``
(align_prompt.py:323-329).  The vocabulary file is a network fetch and is not shipped: pass its
path (``vocab.txt`` of bert-base-uncased) via ``SPRC_BERT_VOCAB`` or the constructor.  The
algorithm below is the published BERT one (basic tokenisation: clean, lower-case, NFD accent
strip, punctuation split, CJK isolation; then greedy longest-match-first WordPiece with "##"
continuation, 100-char word cap) and is checked against the installed `transformers`
implementation on synthetic vocabularies in tests/test_host.py and fuzzed against it in
tests/test_tokenizer_fuzz.py (10 k random strings over three synthetic vocabularies; the real
vocabulary too when SPRC_BERT_VOCAB is set).
"""
from __future__ import annotations

import os
import re
import unicodedata
from typing import Dict, List, Sequence

import torch


def _is_whitespace(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class TokenBatch:
    def __init__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return TokenBatch(self.input_ids.to(device), self.attention_mask.to(device))


class BertWordPieceTokenizer:
    def __init__(self, vocab_file: str | None = None, do_lower_case: bool = True, extra_special: Sequence[str] = ("[DEC]",)):
        vocab_file = vocab_file or os.environ.get("SPRC_BERT_VOCAB")
        if not vocab_file or not os.path.isfile(vocab_file):
            raise FileNotFoundError(
                "bert-base-uncased vocab.txt not found: set SPRC_BERT_VOCAB (the reference downloads it, "
                "blip2.py:32; it is not redistributable here) or call the *_ids entry points with token ids")
        with open(vocab_file, encoding="utf-8") as f:
            toks = [line.rstrip("\n") for line in f]
        self.vocab: Dict[str, int] = {t: i for i, t in enumerate(toks)}
        for t in extra_special:                       # tokenizer.add_special_tokens({"bos_token": "[DEC]"})
            if t not in self.vocab:
                self.vocab[t] = len(self.vocab)
        self.do_lower_case = do_lower_case
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in ("[UNK]", "[CLS]", "[SEP]", "[PAD]"))
        self.never_split = {"[UNK]", "[CLS]", "[SEP]", "[PAD]", "[MASK]", *extra_special}
        # special-token literals are cut out of the RAW text before any normalisation (transformers: the added-tokens trie of
        # PreTrainedTokenizer.tokenize / the AddedVocabulary of the fast tokenizers): "x[MASK]y" -> "x", [MASK], "y"; case-sensitive
        self._special_re = re.compile("(" + "|".join(re.escape(t) for t in sorted(self.never_split, key=len, reverse=True)) + ")")

    def __len__(self) -> int:
        return len(self.vocab)

    # ---- basic tokenisation ----
    def _basic(self, text: str) -> List[str]:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_cjk(cp):
                out.append(f" {ch} ")
            else:
                out.append(" " if _is_whitespace(ch) else ch)
        text = unicodedata.normalize("NFC", "".join(out))
        words: List[str] = []
        for tok in text.strip().split():
            if self.do_lower_case:
                tok = tok.lower()
                tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
            cur = ""
            for ch in tok:                               # split on punctuation, keeping it
                if _is_punct(ch):
                    if cur:
                        words.append(cur)
                        cur = ""
                    words.append(ch)
                else:
                    cur += ch
            if cur:
                words.append(cur)
        return words

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > 100:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str, max_length: int) -> List[int]:
        ids: List[int] = []
        for i, seg in enumerate(self._special_re.split(text)):
            if i % 2 == 1:                                   # a special-token literal
                ids.append(self.vocab[seg])
                continue
            for w in self._basic(seg):
                ids.extend(self._wordpiece(w))
        ids = ids[: max_length - 2]                      # truncation=True, truncation_side="right"
        return [self.cls] + ids + [self.sep]

    def __call__(self, text, padding="max_length", truncation=True, max_length=32, return_tensors="pt") -> TokenBatch:
        if isinstance(text, str):
            text = [text]
        if padding != "max_length" or not truncation or return_tensors != "pt":
            raise ValueError("only padding='max_length', truncation=True, return_tensors='pt' are supported")
        ids = torch.full((len(text), max_length), self.pad, dtype=torch.int64)
        mask = torch.zeros((len(text), max_length), dtype=torch.int64)
        for i, t in enumerate(text):
            e = self.encode(t, max_length)
            ids[i, : len(e)] = torch.tensor(e, dtype=torch.int64)
            mask[i, : len(e)] = 1
        return TokenBatch(ids, mask)
