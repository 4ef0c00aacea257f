"""HF-compatible checkpoints + resumable training state.

The reference only ever *loads* weights (``from_pretrained``, /root/reference/test_data_parallelism.py:112) and
never saves anything, so an elastic restart recomputes from scratch.  Here:

* :func:`save_pretrained` / :func:`load_pretrained` write / read the HF directory layout
  (``config.json`` + ``model.safetensors``; ``pytorch_model.bin`` is accepted on load with
  ``weights_only=True``).  Parameter names are HF's, so a directory written here loads into
  ``transformers.BertForSequenceClassification`` and vice versa.  Legacy ``bert.embeddings.position_ids``
  buffers and ``LayerNorm.gamma/beta`` spellings are tolerated on load.
* :func:`save_training_state` / :func:`load_training_state` add the sidecar needed to resume exactly:
  optimizer moments, step count, loss scale, LR-schedule position, epoch / in-epoch offset, RNG counters.
  A pipeline stage saves only its own layer range (``stage{K}`` suffix); rank 0 of each data-parallel group
  writes.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from .models.bert import build_model
from .models.config import EncoderConfig

WEIGHTS = "model.safetensors"
WEIGHTS_BIN = "pytorch_model.bin"
STATE = "training_state.pt"


def _unwrap(model: nn.Module) -> nn.Module:
    return getattr(model, "module", model)


def _hf_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    sd = _unwrap(model).state_dict()
    return {k: v.detach().to("cpu", copy=True).contiguous() for k, v in sd.items() if not k.startswith("_")}


def save_pretrained(model: nn.Module, directory: str, suffix: str = "") -> str:
    os.makedirs(directory, exist_ok=True)
    m = _unwrap(model)
    if hasattr(m, "config"):
        m.config.save(directory)
    from safetensors.torch import save_file

    name = WEIGHTS if not suffix else f"model.{suffix}.safetensors"
    path = os.path.join(directory, name)
    tmp = path + ".tmp"
    save_file(_hf_state_dict(m), tmp, metadata={"format": "pt"})
    os.replace(tmp, path)
    return path


_LEGACY = ((".LayerNorm.gamma", ".LayerNorm.weight"), (".LayerNorm.beta", ".LayerNorm.bias"))


def _read_weights(directory: str) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    files = sorted(f for f in os.listdir(directory) if f.endswith(".safetensors"))
    if files:
        from safetensors.torch import load_file

        for f in files:
            sd.update(load_file(os.path.join(directory, f)))
    elif os.path.exists(os.path.join(directory, WEIGHTS_BIN)):
        sd = torch.load(os.path.join(directory, WEIGHTS_BIN), map_location="cpu", weights_only=True)
    else:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {directory}")
    out = {}
    for k, v in sd.items():
        for old, new in _LEGACY:
            if k.endswith(old):
                k = k[: -len(old)] + new
        if k.endswith("position_ids") or k.endswith("token_type_ids"):
            continue  # non-parameter buffers of older HF checkpoints
        out[k] = v
    return out


def load_state_dict_into(model: nn.Module, sd: Dict[str, torch.Tensor], strict: bool = True):
    """Copy `sd` into the model's (possibly arena-backed) parameters, then refresh the 16-bit copy."""
    m = _unwrap(model)
    own = dict(m.named_parameters())
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    head_only = all(k.startswith("classifier") or ".pooler." in k for k in missing)
    if strict and (unexpected or (missing and not head_only)):
        raise KeyError(f"state_dict mismatch: missing {missing[:4]}, unexpected {unexpected[:4]}")
    with torch.no_grad():
        for k, p in own.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(p.shape):
                    raise ValueError(f"shape mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(p.shape)}")
                p.data.copy_(sd[k].to(p.device, p.dtype))
    arena = getattr(m, "arena", None)
    if arena is not None:
        arena.sync_compute_copy()
    return missing, unexpected


def load_pretrained(directory: str, num_labels: Optional[int] = None, **overrides) -> nn.Module:
    """``AutoModelForSequenceClassification.from_pretrained(directory)``: config + weights from disk.  A
    checkpoint without a classification head (pre-training weights) gets a freshly initialised one."""
    cfg = EncoderConfig.load(directory)
    if num_labels is not None:
        cfg.num_labels = num_labels
    for k, v in overrides.items():
        setattr(cfg, k, v)
    model = build_model(cfg)
    load_state_dict_into(model, _read_weights(directory), strict=True)
    return model


# ---------------------------------------------------------------------------------------------------------
def save_training_state(directory: str, model: nn.Module, optimizer=None, scheduler=None, extra: Optional[dict] = None,
                        suffix: str = "") -> str:
    """Weights (HF layout) + everything needed to continue the run bit-for-bit on the same world size."""
    save_pretrained(model, directory, suffix)
    m = _unwrap(model)
    state = {"extra": extra or {}}
    opt = getattr(optimizer, "optimizer", optimizer)
    if opt is not None:
        state["optimizer"] = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in opt.state_dict().items()} \
            if not isinstance(opt, torch.optim.Optimizer) else opt.state_dict()
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    rt = getattr(m, "runtime", None)
    if rt is not None:
        state["rng"] = {"state": rt.rng_state.cpu(), "micro": rt.micro}
    state["torch_rng"] = torch.get_rng_state()
    name = STATE if not suffix else f"training_state.{suffix}.pt"
    path = os.path.join(directory, name)
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)
    with open(os.path.join(directory, "latest.json" if not suffix else f"latest.{suffix}.json"), "w") as f:
        json.dump({"weights": WEIGHTS if not suffix else f"model.{suffix}.safetensors", "state": name,
                   "extra": {k: v for k, v in (extra or {}).items() if isinstance(v, (int, float, str, bool))}}, f)
    return path


def load_training_state(directory: str, model: nn.Module, optimizer=None, scheduler=None, suffix: str = "") -> dict:
    from safetensors.torch import load_file

    wname = WEIGHTS if not suffix else f"model.{suffix}.safetensors"
    load_state_dict_into(model, load_file(os.path.join(directory, wname)), strict=True)
    name = STATE if not suffix else f"training_state.{suffix}.pt"
    state = torch.load(os.path.join(directory, name), map_location="cpu", weights_only=False)
    opt = getattr(optimizer, "optimizer", optimizer)
    if opt is not None and "optimizer" in state:
        opt.load_state_dict(state["optimizer"])
    if scheduler is not None and "scheduler" in state:
        scheduler.load_state_dict(state["scheduler"])
    m = _unwrap(model)
    rt = getattr(m, "runtime", None)
    if rt is not None and "rng" in state:
        rt.rng_state.copy_(state["rng"]["state"].to(rt.rng_state.device))
        rt.micro = int(state["rng"]["micro"])
    if "torch_rng" in state:
        torch.set_rng_state(state["torch_rng"])
    return state.get("extra", {})


def has_checkpoint(directory: Optional[str], suffix: str = "") -> bool:
    if not directory:
        return False
    return os.path.exists(os.path.join(directory, "latest.json" if not suffix else f"latest.{suffix}.json"))
