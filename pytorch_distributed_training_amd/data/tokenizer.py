"""Tokenizer front-end: the ``AutoTokenizer`` calls of the reference
(``tokenizer(s1, s2, truncation=True, max_length=None)`` and ``tokenizer.pad(...)``,
/root/reference/test_data_parallelism.py:69,75,98-99) on the C++ WordPiece implementation
(csrc/runtime/wordpiece.cpp).  A ``vocab.txt`` must be available locally - there is no hub access."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

from ..ops import native
from .collate import PadCollator


class WordPieceTokenizer:
    def __init__(self, vocab_file: str, do_lower_case: bool = False, model_max_length: int = 512):
        self._tok = native().WordPiece(vocab_file, do_lower_case, 100)
        self.model_max_length = model_max_length
        self.pad_token_id = self._tok.pad_token_id
        self.vocab_size = self._tok.vocab_size

    @classmethod
    def from_pretrained(cls, directory_or_vocab: str, **kw) -> "WordPieceTokenizer":
        import os

        path = directory_or_vocab
        if os.path.isdir(path):
            path = os.path.join(path, "vocab.txt")
        kw.setdefault("do_lower_case", "uncased" in os.path.basename(os.path.dirname(path) or path).lower())
        return cls(path, **kw)

    def tokenize(self, text: str) -> List[str]:
        return self._tok.tokenize(text)

    def __call__(self, text: Union[str, Sequence[str]], text_pair: Union[None, str, Sequence[str]] = None,
                 truncation: bool = True, max_length: Optional[int] = None) -> Dict[str, list]:
        limit = (max_length or self.model_max_length) if truncation else 0
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        pairs = [] if text_pair is None else ([text_pair] if isinstance(text_pair, str) else list(text_pair))
        enc = self._tok.encode_pairs(texts, pairs, limit)
        out = {"input_ids": [e[0] for e in enc], "token_type_ids": [e[1] for e in enc],
               "attention_mask": [[1] * len(e[0]) for e in enc]}
        return {k: v[0] for k, v in out.items()} if single else out

    def pad(self, examples, padding: str = "longest", max_length: Optional[int] = None, return_tensors: str = "pt"):
        return PadCollator(self.pad_token_id, padding, max_length)(examples)
