"""HF-compatible checkpoints + resumable training state.

The reference only ever *loads* weights (``from_pretrained``, /root/reference/test_data_parallelism.py:112) and
never saves anything, so an elastic restart recomputes from scratch.  Here:

* :func:`save_pretrained` / :func:`load_pretrained` write / read the HF directory layout
  (``config.json`` + ``model.safetensors``; ``pytorch_model.bin`` is accepted on load with
  ``weights_only=True``).  Parameter names are HF's, so a directory written here loads into
  ``transformers.BertForSequenceClassification`` and vice versa.  Legacy ``bert.embeddings.position_ids``
  buffers and ``LayerNorm.gamma/beta`` spellings are tolerated on load.
* :func:`save_training_state` / :func:`load_training_state` add the sidecar needed to resume exactly:
  optimizer moments, step count, loss scale, LR-schedule position, epoch / in-epoch batch offset, RNG counters.
  A checkpoint is a step-numbered subdirectory published atomically through ``latest.json``.  A model-parallel
  rank saves only its own part (``stage{K}`` suffix); rank 0 of each data-parallel group writes.
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


def save_pretrained(model: nn.Module, directory: str, suffix: str = "", write_config: bool = True) -> str:
    os.makedirs(directory, exist_ok=True)
    m = _unwrap(model)
    if write_config and hasattr(m, "config"):
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


def from_transformers(hf_model, **overrides) -> nn.Module:
    """This framework's model from an in-memory ``transformers`` Bert / Roberta ``ForSequenceClassification`` (what the
    reference builds with ``AutoModelForSequenceClassification.from_pretrained``,
    /root/reference/test_data_parallelism.py:112): same configuration, same weights - parameter names are the HF names,
    so the state dict is taken as it is."""
    cfg = EncoderConfig.from_hf_dict(hf_model.config.to_dict())
    for k, v in overrides.items():
        setattr(cfg, k, v)
    model = build_model(cfg)
    load_state_dict_into(model, {k: v.detach() for k, v in hf_model.state_dict().items()}, strict=True)
    return model


def to_transformers(model: nn.Module):
    """The inverse of :func:`from_transformers`: a ``transformers`` model of the matching architecture holding this
    model's (master, fp32) weights."""
    import transformers

    inner = _unwrap(model)
    hf_cfg = transformers.AutoConfig.for_model(**inner.config.to_hf_dict())
    hf = transformers.AutoModelForSequenceClassification.from_config(hf_cfg)
    missing, unexpected = hf.load_state_dict(_hf_state_dict(model), strict=False)
    missing = [k for k in missing if not k.endswith("position_ids")]  # (a buffer of some transformers versions)
    if missing or unexpected:
        raise KeyError(f"state dicts do not line up: missing {missing[:4]}, unexpected {list(unexpected)[:4]}")
    return hf


# ---------------------------------------------------------------------------------------------------------
# Training-state checkpoints.  Layout of `directory`:
#
#   checkpoint-<step>-<epoch>-<offset>/config.json, model[.<suffix>].safetensors, training_state[.<suffix>].pt
#   latest[.<suffix>].json            -> names the newest complete subdirectory of this writer
#
# A subdirectory is filled first (every file through a temporary name + os.replace) and only then published by
# replacing latest.json atomically, so a crash at any point leaves the previous checkpoint intact and never pairs
# new weights with old optimizer moments.  Each subdirectory is itself an HF model directory.  The writers of a
# model-parallel replica (suffix "stage0" / "stage1") share the subdirectories and publish separately;
# `common_tag` lets them agree on the newest checkpoint all of them completed.
def _tag_of(extra: dict) -> str:
    return f"{int(extra.get('global_step', 0)):08d}-{int(extra.get('epoch', 0))}-{int(extra.get('micro_offset', 0))}"


def _tag_key(tag: str):
    """Order of a tag: (optimizer step, epoch, batches consumed) as integers - only the step field is zero-padded, so as
    strings "…-10-0" would sort before "…-9-57"."""
    try:
        return tuple(int(x) for x in tag.split("-"))
    except ValueError:
        return (-1,)


def _names(suffix: str):
    if not suffix:
        return WEIGHTS, STATE, "latest.json"
    return f"model.{suffix}.safetensors", f"training_state.{suffix}.pt", f"latest.{suffix}.json"


def _atomic_json(path: str, obj) -> None:
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(obj, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def save_training_state(directory: str, model: nn.Module, optimizer=None, scheduler=None, extra: Optional[dict] = None,
                        suffix: str = "", write_config: bool = True, keep: int = 2) -> str:
    """Weights (HF layout) + everything needed to continue the run bit-for-bit on the same world size.  `extra`
    carries the position of the run (``epoch``, ``global_step``, ``micro_offset`` = batches of the epoch already
    consumed).  Only one writer of a shared directory should pass ``write_config=True``."""
    extra = dict(extra or {})
    wname, sname, lname = _names(suffix)
    tag = _tag_of(extra)
    sub = os.path.join(directory, f"checkpoint-{tag}")
    os.makedirs(sub, exist_ok=True)
    save_pretrained(model, sub, suffix, write_config=write_config)
    m = _unwrap(model)
    state = {"extra": extra}
    opt = getattr(optimizer, "optimizer", optimizer)
    if opt is not None:
        state["optimizer"] = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in opt.state_dict().items()} \
            if not isinstance(opt, torch.optim.Optimizer) else opt.state_dict()
    if scheduler is not None:
        state["scheduler"] = scheduler.state_dict()
    rt = getattr(m, "runtime", None)
    if rt is not None:
        # (rng_state_global: the stream of batches that carry sample ids; a graphed step advances its offset on the
        # device at every replay, so it is part of the position of the run)
        state["rng"] = {"state": rt.rng_state.cpu(), "global": rt.rng_state_global.cpu(), "micro": rt.micro,
                        "window": rt.arena.window}
    state["torch_rng"] = torch.get_rng_state()
    path = os.path.join(sub, sname)
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)
    _atomic_json(os.path.join(directory, lname), {
        "dir": f"checkpoint-{tag}", "weights": wname, "state": sname,
        "extra": {k: v for k, v in extra.items() if isinstance(v, (int, float, str, bool))}})
    # older checkpoints: this writer removes its own files, the directory goes with its last file
    mine = sorted((t for t in available_tags(directory, suffix) if t != tag), key=_tag_key)
    for old in mine[:max(0, len(mine) - (keep - 1))]:
        odir = os.path.join(directory, f"checkpoint-{old}")
        for name in (wname, sname) + (("config.json",) if write_config else ()):
            try:
                os.remove(os.path.join(odir, name))
            except OSError:
                pass
        try:
            os.rmdir(odir)
        except OSError:
            pass
    return path


def available_tags(directory: str, suffix: str = ""):
    """Tags of the checkpoints under `directory` for which this writer's two files are complete."""
    wname, sname, _ = _names(suffix)
    out = []
    if not directory or not os.path.isdir(directory):
        return out
    for d in os.listdir(directory):
        if d.startswith("checkpoint-") and all(os.path.exists(os.path.join(directory, d, n)) for n in (wname, sname)):
            out.append(d[len("checkpoint-"):])
    return sorted(out, key=_tag_key)


def latest_tag(directory: Optional[str], suffix: str = "") -> Optional[str]:
    """Tag of the checkpoint this writer published last, or None."""
    if not directory:
        return None
    path = os.path.join(directory, _names(suffix)[2])
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f).get("dir", "")
    return d[len("checkpoint-"):] if d.startswith("checkpoint-") else None


def common_tag(directory: Optional[str], suffix: str, group=None) -> Optional[str]:
    """Newest checkpoint every rank of `group` can load (collective).  The writers of a model-parallel replica
    publish one after the other, so after a crash one of them may be a checkpoint ahead of the other."""
    import torch.distributed as dist

    mine = set(available_tags(directory, suffix)) if directory else set()
    newest = latest_tag(directory, suffix)
    mine = {t for t in mine if newest is not None and _tag_key(t) <= _tag_key(newest)}  # nothing newer than what was published
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return max(mine, key=_tag_key) if mine else None
    every = [None] * dist.get_world_size(group)
    dist.all_gather_object(every, sorted(mine, key=_tag_key), group=group)
    common = set(every[0]).intersection(*map(set, every[1:]))
    return max(common, key=_tag_key) if common else None


def load_training_state(directory: str, model: nn.Module, optimizer=None, scheduler=None, suffix: str = "",
                        tag: Optional[str] = None) -> dict:
    """Restore what :func:`save_training_state` wrote (the newest published checkpoint, or `tag`); returns `extra`.
    Nothing from the files is executed: safetensors for the weights, ``weights_only=True`` for the state."""
    from safetensors.torch import load_file

    wname, sname, _ = _names(suffix)
    tag = tag or latest_tag(directory, suffix)
    if tag is None:
        raise FileNotFoundError(f"no published checkpoint in {directory}")
    sub = os.path.join(directory, f"checkpoint-{tag}")
    load_state_dict_into(model, load_file(os.path.join(sub, wname)), strict=True)
    state = torch.load(os.path.join(sub, sname), map_location="cpu", weights_only=True)
    opt = getattr(optimizer, "optimizer", optimizer)
    if opt is not None and "optimizer" in state:
        opt.load_state_dict(state["optimizer"])
    if scheduler is not None and "scheduler" in state:
        scheduler.load_state_dict(state["scheduler"])
    m = _unwrap(model)
    rt = getattr(m, "runtime", None)
    if rt is not None and "rng" in state:
        rt.rng_state.copy_(state["rng"]["state"].to(rt.rng_state.device))
        if "global" in state["rng"]:
            rt.rng_state_global.copy_(state["rng"]["global"].to(rt.rng_state_global.device))
        rt.micro = int(state["rng"]["micro"])
        rt.arena.window = int(state["rng"].get("window", rt.arena.window))
    if "torch_rng" in state:
        torch.set_rng_state(state["torch_rng"])
    return state.get("extra", {})


def has_checkpoint(directory: Optional[str], suffix: str = "") -> bool:
    return latest_tag(directory, suffix) is not None
