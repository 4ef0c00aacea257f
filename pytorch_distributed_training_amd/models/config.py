"""Encoder architecture descriptions.

The reference hard-codes its checkpoints as string literals
(``bert-large-cased`` at /root/reference/test_data_parallelism.py:69,112 and
``bert-base-cased`` at /root/reference/test_model_parallelism.py:193,230-238).
Here the same architectures are named presets of one dataclass, so every other
hard-coded constant of the reference becomes a field that can be overridden.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, replace
from typing import Dict, Optional


@dataclass
class EncoderConfig:
    """Shape + regularisation hyper-parameters of a BERT-family encoder."""

    model_type: str = "bert"  # "bert" | "roberta"
    vocab_size: int = 28996
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    classifier_dropout: Optional[float] = None
    layer_norm_eps: float = 1e-12
    initializer_range: float = 0.02
    pad_token_id: int = 0
    num_labels: int = 2
    hidden_act: str = "gelu"

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def head_dropout(self) -> float:
        p = self.classifier_dropout
        return self.hidden_dropout_prob if p is None else p

    def validate(self) -> None:
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")
        if self.hidden_act != "gelu":
            raise ValueError("only the exact-erf GELU of BERT/RoBERTa is implemented")
        if self.model_type not in ("bert", "roberta"):
            raise ValueError(f"unknown model_type {self.model_type!r}")

    # -- HF config.json round trip -------------------------------------------------
    def to_hf_dict(self) -> Dict:
        d = asdict(self)
        d["architectures"] = [
            "BertForSequenceClassification" if self.model_type == "bert" else "RobertaForSequenceClassification"
        ]
        d["position_embedding_type"] = "absolute"
        return d

    def save(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, "config.json")
        with open(path + ".tmp", "w") as f:
            json.dump(self.to_hf_dict(), f, indent=2, sort_keys=True)
        os.replace(path + ".tmp", path)

    @classmethod
    def from_hf_dict(cls, d: Dict) -> "EncoderConfig":
        names = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in names}
        if "num_labels" not in kw and "id2label" in d:
            kw["num_labels"] = len(d["id2label"])
        cfg = cls(**kw)
        cfg.validate()
        return cfg

    @classmethod
    def load(cls, directory: str) -> "EncoderConfig":
        with open(os.path.join(directory, "config.json")) as f:
            return cls.from_hf_dict(json.load(f))


_PRESETS: Dict[str, EncoderConfig] = {
    # bert-base-cased: L=12 H=768 A=12 F=3072 V=28996 (108.3 M parameters)
    "bert-base-cased": EncoderConfig(),
    # bert-large-cased: L=24 H=1024 A=16 F=4096 V=28996 (333.6 M parameters)
    "bert-large-cased": EncoderConfig(
        hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096
    ),
    # roberta-large: byte-level BPE vocab, 514 positions (2 reserved by padding_idx offset)
    "roberta-large": EncoderConfig(
        model_type="roberta",
        vocab_size=50265,
        hidden_size=1024,
        num_hidden_layers=24,
        num_attention_heads=16,
        intermediate_size=4096,
        max_position_embeddings=514,
        type_vocab_size=1,
        layer_norm_eps=1e-5,
        pad_token_id=1,
    ),
    "roberta-base": EncoderConfig(
        model_type="roberta",
        vocab_size=50265,
        max_position_embeddings=514,
        type_vocab_size=1,
        layer_norm_eps=1e-5,
        pad_token_id=1,
    ),
    # tiny shapes for unit tests (every dim still a multiple of the 64-wide kernel tiles)
    "bert-tiny": EncoderConfig(
        vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
        max_position_embeddings=128,
    ),
    # the same tiny encoder without dropout: deterministic loss curves for parity tests
    "bert-tiny-nodrop": EncoderConfig(
        vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
        max_position_embeddings=128, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
    ),
    "roberta-tiny": EncoderConfig(
        model_type="roberta", vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
        intermediate_size=512, max_position_embeddings=130, type_vocab_size=1, layer_norm_eps=1e-5,
        pad_token_id=1,
    ),
}
_ALIASES = {"bert-base": "bert-base-cased", "bert-large": "bert-large-cased"}


def preset(name: str, **overrides) -> EncoderConfig:
    """Return a copy of a named architecture, optionally with fields overridden."""
    key = _ALIASES.get(name, name)
    if key not in _PRESETS:
        raise KeyError(f"unknown model preset {name!r}; known: {sorted(_PRESETS) + sorted(_ALIASES)}")
    cfg = replace(_PRESETS[key], **overrides)
    cfg.validate()
    return cfg


def preset_names():
    return sorted(_PRESETS) + sorted(_ALIASES)
