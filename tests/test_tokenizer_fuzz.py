"""R9 (WordPiece) fuzzed against the installed third-party implementation (`transformers.BertTokenizer`; the reference pins transformers
4.36.2 and calls it at blip2.py:30-34 / align_prompt.py:323-329).  The real bert-base-uncased vocabulary is a network fetch, so the
comparison runs on SYNTHETIC vocabularies (three seeds: whole words, "##" continuation pieces, single characters incl. accented / CJK /
punctuation entries, the special tokens and the added "[DEC]") over >= 10 000 random strings mixing unicode categories, long words,
control characters, exotic whitespace and special-token literals; and -- when SPRC_BERT_VOCAB names the real vocab.txt -- on that too."""
import os
import random
import tempfile
from pathlib import Path

import pytest
import torch

SPECIALS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
LATIN = "abcdefghijklmnopqrstuvwxyz"
ACCENTED = "àáâãäåçèéêëìíîïñòóôõöùúûüýÿāăąćčďēęěğīłńňōőřśšşţťūůűźżž"
COMBINING = "̧̀́̂̃̈̊"
CJK = "中文字漢字日本語한국어㐀𠀀"
PUNCT = "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~«»‘’“”…—–·¿¡、。「」"
SPACES = [" ", "  ", "\t", "\n", "\r", " ", " ", "　", "​", " \n "]
CONTROL = ["\x00", "\x01", "\x1f", "\x7f", "�", "‎", "﻿", "­"]
OTHER = "ßøæœđŋþƒαβγδεζηθабвгдежзابتثאבגד१२३€£¥°±²³µ¹º¼½¾×÷ℓ™Ω"


def _make_vocab(seed: int) -> list:
    rng = random.Random(seed)
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(20)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = list(LATIN + "0123456789") + rng.sample(list(ACCENTED), 10) + rng.sample(list(CJK), 5) + rng.sample(list(PUNCT), 30) + \
        rng.sample(list(OTHER), 15)
    toks += chars + ["##" + c for c in LATIN + "0123456789"] + ["##" + c for c in rng.sample(list(OTHER), 6)]
    seen = set(toks)
    for _ in range(400):                                    # whole words and continuation pieces of 2 .. 7 letters
        w = "".join(rng.choice(LATIN) for _ in range(rng.randint(2, 7)))
        for t in (w, "##" + w)[: rng.randint(1, 2)] if rng.random() < 0.5 else ("##" + w,):
            if t not in seen:
                seen.add(t)
                toks.append(t)
    for w in ["the", "dog", "is", "now", "stand", "##ing", "and", "by", "him", "##self", "make", "it", "more", "color", "##ful", "remove",
              "second", "person", "two", "dogs", "instead", "of", "one", "caf", "##e", "sep", "dec", "mask", "##dec", "shirt", "##s"]:
        if w not in seen:
            seen.add(w)
            toks.append(w)
    return toks


def _random_text(rng: random.Random, words: list) -> str:
    parts = []
    for _ in range(rng.randint(0, 14)):
        r = rng.random()
        if r < 0.35:
            w = rng.choice(words)
            if rng.random() < 0.3:
                w = w.upper() if rng.random() < 0.5 else w.capitalize()
            if rng.random() < 0.3:
                w += rng.choice(words)
            parts.append(w)
        elif r < 0.50:
            parts.append("".join(rng.choice(LATIN + ACCENTED + OTHER) for _ in range(rng.randint(1, 12))))
        elif r < 0.58:
            base = "".join(rng.choice(LATIN) for _ in range(rng.randint(1, 6)))
            parts.append("".join(c + (rng.choice(COMBINING) if rng.random() < 0.4 else "") for c in base))
        elif r < 0.66:
            parts.append("".join(rng.choice(CJK + LATIN) for _ in range(rng.randint(1, 5))))
        elif r < 0.78:
            parts.append(rng.choice(words) + "".join(rng.choice(PUNCT) for _ in range(rng.randint(1, 3))) + rng.choice(words))
        elif r < 0.84:
            parts.append(rng.choice(SPECIALS + ["[DEC]", "[dec]", "[SEP]x", "x[MASK]", "[ SEP ]", "[UNUSED3]", "[unused3]"]))
        elif r < 0.90:
            parts.append(rng.choice(CONTROL).join(rng.choice(words) for _ in range(2)))
        elif r < 0.94:
            parts.append(rng.choice(LATIN) * rng.choice([99, 100, 101, 150]))          # max_input_chars_per_word = 100
        else:
            parts.append(str(rng.randint(0, 10 ** rng.randint(1, 9))))
    return "".join(p + rng.choice(SPACES) for p in parts)


def _compare(vocab_file: Path, texts: list, max_length: int = 32):
    from transformers import BertTokenizer
    from sprc_amd.tokenizer import BertWordPieceTokenizer
    hf = BertTokenizer.from_pretrained(str(vocab_file.parent), do_lower_case=True)
    hf.add_special_tokens({"bos_token": "[DEC]"})                              # blip2.py:33
    mine = BertWordPieceTokenizer(str(vocab_file))
    assert len(mine) == len(hf)
    bad = []
    for s in range(0, len(texts), 500):
        chunk = texts[s:s + 500]
        a = hf(chunk, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt")
        b = mine(chunk, padding="max_length", truncation=True, max_length=max_length, return_tensors="pt")
        eq = (a.input_ids == b.input_ids).all(1) & (a.attention_mask == b.attention_mask).all(1)
        bad += [(chunk[i], a.input_ids[i].tolist(), b.input_ids[i].tolist()) for i in torch.nonzero(~eq).flatten().tolist()]
    return bad


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_wordpiece_fuzz_against_transformers(seed):
    toks = _make_vocab(seed)
    words = [t for t in toks if not t.startswith("[") and not t.startswith("##") and len(t) > 1]
    rng = random.Random(1000 + seed)
    texts = [_random_text(rng, words) for _ in range(3400)]
    with tempfile.TemporaryDirectory() as d:
        vp = Path(d) / "vocab.txt"
        vp.write_text("\n".join(toks) + "\n", encoding="utf-8")
        bad = _compare(vp, texts)
        long_bad = _compare(vp, texts[:300], max_length=128)                   # the padded width is the caller's, not a constant
    assert not bad and not long_bad, f"{len(bad)} of {len(texts)} strings differ; first: {bad[:3] or long_bad[:3]!r}"


@pytest.mark.skipif(not os.environ.get("SPRC_BERT_VOCAB") or not os.path.isfile(os.environ.get("SPRC_BERT_VOCAB", "")),
                    reason="SPRC_BERT_VOCAB does not name bert-base-uncased's vocab.txt (a network fetch in the reference, blip2.py:32)")
def test_wordpiece_on_the_real_vocabulary():
    """With the real vocabulary at hand R9 stops being "parity unpinned": the same fuzz + CIRR-style captions on bert-base-uncased."""
    import shutil
    src = Path(os.environ["SPRC_BERT_VOCAB"])
    toks = src.read_text(encoding="utf-8").split("\n")
    assert len([t for t in toks if t]) == 30522 and toks[101] == "[CLS]" and toks[102] == "[SEP]"
    words = [t for t in toks[2000:12000] if t.isalpha()]
    rng = random.Random(7)
    texts = [_random_text(rng, words) for _ in range(10000)] + [
        "is a darker shade of brown and has a longer tail", "Remove the \"second\" person (left)!", "has a v-neck; and it's (more) colourful"]
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(src, Path(d) / "vocab.txt")
        bad = _compare(Path(d) / "vocab.txt", texts)
    assert not bad, f"{len(bad)} of {len(texts)} strings differ; first: {bad[:3]!r}"
