"""Synthetic GLUE/MRPC-shaped sentence-pair data.

The reference tokenises GLUE/MRPC from the HF hub (/root/reference/test_data_parallelism.py:69-87):
3 668 training / 408 validation sentence pairs, WordPiece ids, ``[CLS] s1 [SEP] s2 [SEP]`` with
token types 0/1, ragged lengths padded per batch.  No hub, vocabulary or dataset is reachable here, so
this generator produces batches of exactly that *shape* with a learnable rule: a pair is a "paraphrase"
(label 1) when sentence 2 is a noisy copy of sentence 1, otherwise both are independent draws.  A model
that can attend across the separator learns it, which is what the end-to-end convergence tests use.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

MRPC_TRAIN, MRPC_VALIDATION = 3668, 408
CLS, SEP, PAD = 101, 102, 0


class SyntheticPairDataset(Dataset):
    def __init__(self, num_examples: int = MRPC_TRAIN, vocab_size: int = 28996, seq_len: Optional[int] = None,
                 min_len: int = 8, max_len: int = 64, seed: int = 0, copy_noise: float = 0.15,
                 special_floor: int = 1000, roberta: bool = False, data_vocab: Optional[int] = None,
                 rule: str = "copy", topics: int = 8):
        """``seq_len``: if given every example has exactly this many tokens (fixed-shape benchmark data),
        otherwise sentence lengths are ragged in ``[min_len, max_len]`` per sentence.  ``data_vocab``: draw the
        word tokens from only this many ids.  ``rule``: what makes a pair a "paraphrase" (label 1) -
        ``"copy"``: sentence 2 is a noisy copy of sentence 1 (needs token-by-token comparison across the separator);
        ``"topic"``: both sentences draw their words from the same one of ``topics`` disjoint slices of the vocabulary,
        label 0 pairs from two different slices (a sentence-pair relation a randomly initialised encoder picks up
        within a few hundred steps - the convergence tests and logs use it)."""
        if rule not in ("copy", "topic", "marker"):
            raise ValueError("rule must be 'copy', 'topic' or 'marker'")
        rng = np.random.default_rng(seed)
        self.examples: List[Dict[str, List[int]]] = []
        lo = min(special_floor, vocab_size // 4)
        if data_vocab:
            vocab_size = min(vocab_size, lo + int(data_vocab))
        cls, sep = (0, 2) if roberta else (CLS, SEP)
        cls, sep = min(cls, vocab_size - 1), min(sep, vocab_size - 1)
        self.roberta = roberta
        for _ in range(num_examples):
            if seq_len is not None:
                body = seq_len - 3
                l1 = body // 2
                l2 = body - l1
            else:
                l1 = int(rng.integers(min_len, max_len + 1))
                l2 = int(rng.integers(min_len, max_len + 1))
            label = int(rng.integers(0, 2))
            if rule == "marker":  # (a bag-of-words sanity rule: sentence 2 from the upper / lower half of the vocabulary)
                mid = (lo + vocab_size) // 2
                s1 = rng.integers(lo, vocab_size, size=l1)
                s2 = rng.integers(mid, vocab_size, size=l2) if label == 1 else rng.integers(lo, mid, size=l2)
            elif rule == "topic":
                width = max(1, (vocab_size - lo) // topics)
                t1 = int(rng.integers(0, topics))
                t2 = t1 if label == 1 else int((t1 + rng.integers(1, topics)) % topics)
                s1 = rng.integers(lo + t1 * width, lo + (t1 + 1) * width, size=l1)
                s2 = rng.integers(lo + t2 * width, lo + (t2 + 1) * width, size=l2)
            else:
                s1 = rng.integers(lo, vocab_size, size=l1)
            if rule in ("topic", "marker"):
                pass
            elif label == 1:
                base = np.resize(s1, l2)
                noise = rng.random(l2) < copy_noise
                s2 = np.where(noise, rng.integers(lo, vocab_size, size=l2), base)
            else:
                s2 = rng.integers(lo, vocab_size, size=l2)
            ids = [cls] + s1.tolist() + [sep] + s2.tolist() + [sep]
            types = [0] * (l1 + 2) + [1] * (l2 + 1)
            ex = {"input_ids": ids, "attention_mask": [1] * len(ids), "labels": label}
            ex["token_type_ids"] = [0] * len(ids) if roberta else types
            self.examples.append(ex)

    def __len__(self) -> int:
        return len(self.examples)

    def __getitem__(self, i: int) -> Dict[str, List[int]]:
        return self.examples[i]


def fixed_batch(batch: int, seq_len: int, vocab_size: int, device, seed: int = 0, first_id: int = 0) -> Dict[str, torch.Tensor]:
    """One device-resident full-length batch (benchmark input): no padding, both token types present.  The
    samples carry the global ids ``first_id ... first_id + batch - 1``."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000 if vocab_size > 2000 else 4, vocab_size, (batch, seq_len), generator=g)
    ids[:, 0] = min(CLS, vocab_size - 1)
    half = seq_len // 2
    ids[:, half] = min(SEP, vocab_size - 1)
    ids[:, -1] = min(SEP, vocab_size - 1)
    types = torch.zeros(batch, seq_len, dtype=torch.long)
    types[:, half + 1:] = 1
    return {
        "input_ids": ids.to(device),
        "token_type_ids": types.to(device),
        "attention_mask": torch.ones(batch, seq_len, dtype=torch.long, device=device),
        "labels": torch.randint(0, 2, (batch,), generator=g).to(device),
        "sample_ids": torch.arange(first_id, first_id + batch, dtype=torch.int32).to(device),
    }
