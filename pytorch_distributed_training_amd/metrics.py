"""GLUE/MRPC metric (accuracy + binary F1), the role of ``load_metric("glue", "mrpc")`` in the reference
(/root/reference/test_data_parallelism.py:71,159-164).  Counts are accumulated on the device as a 2x2
confusion matrix; ``compute()`` is the only host read."""
from __future__ import annotations

from typing import Dict, Optional

import torch


class MRPCMetric:
    def __init__(self):
        self.conf: Optional[torch.Tensor] = None  # [[tn, fp], [fn, tp]]

    def add_batch(self, predictions: torch.Tensor, references: torch.Tensor, valid: Optional[int] = None) -> None:
        """`valid`: use only the first `valid` gathered samples (drops the wrap-around duplicates that a
        sharded eval loader appends to even out the last batch)."""
        p, r = predictions.reshape(-1).long(), references.reshape(-1).long()
        if valid is not None:
            p, r = p[:valid], r[:valid]
        idx = (r * 2 + p).clamp_(0, 3)
        c = torch.bincount(idx, minlength=4).reshape(2, 2)
        self.conf = c if self.conf is None else self.conf + c.to(self.conf.device)

    def compute(self) -> Dict[str, float]:
        if self.conf is None:
            return {"accuracy": 0.0, "f1": 0.0}
        (tn, fp), (fn, tp) = self.conf.tolist()
        total = tn + fp + fn + tp
        acc = (tn + tp) / total if total else 0.0
        f1 = 2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) else 0.0
        self.conf = None
        return {"accuracy": acc, "f1": f1}


def load_metric(*_names) -> MRPCMetric:
    """``load_metric("glue", "mrpc")`` spelled like the reference."""
    return MRPCMetric()
