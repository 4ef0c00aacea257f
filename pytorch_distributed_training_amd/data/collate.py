"""Batch collation: what ``tokenizer.pad(examples, padding="longest", return_tensors="pt")`` does at
/root/reference/test_data_parallelism.py:95-99 (and the fixed ``max_length=128`` padding of its TPU branch)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


class BatchDict(dict):
    """dict of tensors with the ``.to(device)`` the reference calls on its batches."""

    def to(self, device, non_blocking: bool = False) -> "BatchDict":
        for k, v in list(self.items()):
            if isinstance(v, torch.Tensor):
                self[k] = v.to(device, non_blocking=non_blocking)
        return self

    def pin_memory(self) -> "BatchDict":
        for k, v in list(self.items()):
            if isinstance(v, torch.Tensor):
                self[k] = v.pin_memory()
        return self


class PadCollator:
    def __init__(self, pad_token_id: int = 0, padding: str = "longest", max_length: Optional[int] = None,
                 pad_to_multiple_of: Optional[int] = None, varlen: bool = False):
        """``varlen``: also emit ``cu_seqlens`` (int32 [B+1], first packed row of every sample), ``token_index``
        (int64 [T], the flat positions ``b * S + s`` of the real tokens) and ``lengths`` (the per-sample token counts
        as a host list): the native path then runs the encoder on the T real tokens only - no GEMM row, LayerNorm
        row or attention key is spent on padding."""
        self.varlen = varlen
        if padding not in ("longest", "max_length"):
            raise ValueError("padding must be 'longest' or 'max_length'")
        self.pad_token_id, self.padding, self.max_length, self.multiple = pad_token_id, padding, max_length, pad_to_multiple_of

    def __call__(self, examples: List[Dict]) -> BatchDict:
        n = max(len(e["input_ids"]) for e in examples)
        if self.padding == "max_length":
            if self.max_length is None:
                raise ValueError("padding='max_length' needs max_length")
            n = self.max_length
        if self.multiple:
            n = (n + self.multiple - 1) // self.multiple * self.multiple
        pad_values = {"input_ids": self.pad_token_id, "attention_mask": 0, "token_type_ids": 0}
        out = BatchDict()
        for key, pad in pad_values.items():
            if key not in examples[0]:
                continue
            t = torch.full((len(examples), n), pad, dtype=torch.long)
            for i, e in enumerate(examples):
                v = e[key][:n]
                t[i, :len(v)] = torch.as_tensor(v, dtype=torch.long)
            out[key] = t
        lab = "labels" if "labels" in examples[0] else ("label" if "label" in examples[0] else None)
        if lab is not None:
            out["labels"] = torch.as_tensor([e[lab] for e in examples], dtype=torch.long)
        if self.varlen:
            lens = [min(len(e["input_ids"]), n) for e in examples]
            cu = [0]
            for v in lens:
                cu.append(cu[-1] + v)
            out["cu_seqlens"] = torch.as_tensor(cu, dtype=torch.int32)
            out["token_index"] = torch.cat([torch.arange(v, dtype=torch.long) + i * n for i, v in enumerate(lens)])
            out["lengths"] = lens  # (a host list: stays on the host when the batch moves to the device)
        if "sample_ids" in examples[0]:  # global ids of the samples (data/loader.py adds them): sample-keyed dropout
            out["sample_ids"] = torch.as_tensor([e["sample_ids"] for e in examples], dtype=torch.int32)
        return out
