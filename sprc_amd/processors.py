"""Text processors of the model protocol (host-side, pure string work).

`BlipCaptionProcessor` mirrors lavis/processors/blip_processors.py:28-68 (R8): it is what
`load_model_and_preprocess` hands back as ``txt_processors["eval"]`` and what the evaluation
loops apply to every caption (validate_blip.py:389, :185; cirr_test_submission.py:167).
"""
from __future__ import annotations

import re


class BlipCaptionProcessor:
    def __init__(self, prompt: str = "", max_words: int = 50):
        self.prompt = prompt
        self.max_words = max_words

    def __call__(self, caption: str) -> str:
        return self.prompt + self.pre_caption(caption)

    @classmethod
    def from_config(cls, cfg=None):
        cfg = cfg or {}
        return cls(prompt=cfg.get("prompt", ""), max_words=cfg.get("max_words", 50))

    def pre_caption(self, caption: str) -> str:
        caption = re.sub(r"([.!\"()*#:;~])", " ", caption.lower())   # punctuation -> space
        caption = re.sub(r"\s{2,}", " ", caption)                     # collapse whitespace runs
        caption = caption.rstrip("\n").strip(" ")
        words = caption.split(" ")
        if len(words) > self.max_words:                               # truncate to max_words
            caption = " ".join(words[: self.max_words])
        return caption


def fiq_compose_caption(c1: str, c2: str) -> str:
    """FashionIQ's two relative captions -> one (validate_blip.py:180-184, R10)."""
    return f"{c1.strip('.?, ').capitalize()} and {c2.strip('.?, ')}"
