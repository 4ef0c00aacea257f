"""Flat parameter / gradient arenas.

All parameters of a model are re-homed into ONE contiguous fp32 "master" tensor, their gradients into
a second one of the same layout, and (for 16-bit training) a third tensor holds the fp16/bf16 compute
copy that the kernels read.  Consequences, all of them the point of the design:

* the fused AdamW is a single kernel over the whole model (or one per bucket);
* a data-parallel bucket is a contiguous *slice* of the gradient arena - the weight-gradient kernels
  write straight into it, and RCCL reduces it in place (no bucket copy-in / copy-out as in torch DDP,
  whose reducer the reference relies on at /root/reference/test_data_parallelism.py:125-127);
* Q, K and V weights are adjacent, so one ``[3H, H]`` view is the fused QKV GEMM weight;
* ``state_dict()`` still shows the HF names, because ``nn.Parameter.data`` are views into the arena.

The arena is laid out in REVERSE forward order (head, last layer, ..., first layer, embeddings):
address order == the order in which backward finishes gradients == bucket launch order.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

ALIGN = 64  # elements; 256 B in fp32, 128 B in 16-bit


@dataclass
class Segment:
    """A run of parameters that becomes gradient-complete at one point of backward."""

    name: str
    params: List[Tuple[str, nn.Parameter]]
    start: int = 0
    end: int = 0
    bucket: int = -1  # first bucket the segment's gradients lie in
    buckets: Tuple[int, ...] = ()  # every bucket whose completion this segment's completion counts towards
    # gradients of this segment are produced by native kernels that can overwrite (beta=0) on the first
    # backward after zero_grad, so the slice needs no memset
    native_overwrite: bool = False


def _align(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


def default_segments(model: nn.Module, cap_elems: int = 6 << 20) -> List[Segment]:
    """Generic fallback: registration order cut into ~25 MB (fp32) runs, like DDP's bucket cap."""
    segs: List[Segment] = []
    cur: List[Tuple[str, nn.Parameter]] = []
    size = 0
    for name, p in model.named_parameters():
        cur.append((name, p))
        size += p.numel()
        if size >= cap_elems:
            segs.append(Segment(f"seg{len(segs)}", cur))
            cur, size = [], 0
    if cur:
        segs.append(Segment(f"seg{len(segs)}", cur))
    return segs


def bert_segments(model, prefix: str = "") -> List[Segment]:
    """Forward-ordered segments of a (possibly stage-sliced) BERT/RoBERTa classifier."""
    base = model.base_model
    bp = f"{prefix}{model.base_model_prefix}."
    segs: List[Segment] = []
    if base.embeddings is not None:
        e = base.embeddings
        segs.append(Segment(f"{prefix}embeddings", [
            (f"{bp}embeddings.word_embeddings.weight", e.word_embeddings.weight),
            (f"{bp}embeddings.position_embeddings.weight", e.position_embeddings.weight),
            (f"{bp}embeddings.token_type_embeddings.weight", e.token_type_embeddings.weight),
            (f"{bp}embeddings.LayerNorm.weight", e.LayerNorm.weight),
            (f"{bp}embeddings.LayerNorm.bias", e.LayerNorm.bias),
        ], native_overwrite=True))
    for lyr in base.encoder.layers():
        lp = f"{bp}encoder.layer.{lyr.index}."
        segs.append(Segment(f"{prefix}layer{lyr.index}", [(lp + n, p) for n, p in lyr.arena_order()],
                            native_overwrite=True))
    head: List[Tuple[str, nn.Parameter]] = []
    if base.pooler is not None:
        head += [(f"{bp}pooler.dense.weight", base.pooler.dense.weight), (f"{bp}pooler.dense.bias", base.pooler.dense.bias)]
    if getattr(model, "with_head", False):
        head += [(f"{prefix}classifier.{n}", p) for n, p in model.classifier.named_parameters()]
    if head:
        segs.append(Segment(f"{prefix}head", head))
    return segs


class ParamArena:
    def __init__(self, model: nn.Module, device: torch.device, compute_dtype: Optional[torch.dtype] = None,
                 min_bucket_elems: int = 1 << 20, segments: Optional[List[Segment]] = None,
                 guard_elems: Optional[int] = None, max_bucket_elems: Optional[int] = None,
                 tail_bucket_elems: Optional[int] = None):
        self.model = model
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        if segments is None:
            segments = model.arena_segments() if hasattr(model, "arena_segments") else default_segments(model)
        named = dict(model.named_parameters())
        listed = {id(p) for s in segments for _, p in s.params}
        missing = [n for n, p in named.items() if id(p) not in listed]
        if missing:
            raise ValueError(f"arena_segments() does not cover parameters: {missing[:5]} ...")
        # reverse forward order => address order is gradient-completion order
        self.segments: List[Segment] = list(reversed(segments))
        off = 0
        self.offsets: Dict[str, Tuple[int, int]] = {}
        seen = set()
        for seg in self.segments:
            seg.start = off
            for name, p in seg.params:
                if id(p) in seen:
                    raise ValueError(f"parameter {name} appears twice (tied weights are not supported)")
                seen.add(id(p))
                self.offsets[name] = (off, p.numel())
                off = _align(off + p.numel()) if p.numel() % ALIGN else off + p.numel()
            off = _align(off)
            seg.end = off
        self.numel = off
        if guard_elems is None:
            from ..utils.debug import guard_elems as _default_guard

            guard_elems = _default_guard()
        self.guard_elems = int(guard_elems)
        self._backing: Dict[str, torch.Tensor] = {}
        self.master = self._alloc("master", torch.float32)
        self.grad = self._alloc("grad", torch.float32)
        self.p16 = self._alloc("p16", compute_dtype) if compute_dtype else None
        with torch.no_grad():
            for seg in self.segments:
                for name, p in seg.params:
                    o, n = self.offsets[name]
                    view = self.master[o:o + n].view(p.shape)
                    view.copy_(p.data)
                    p.data = view
                    p.grad = self.grad[o:o + n].view(p.shape)
        # buckets: merge small neighbouring segments (e.g. the ~1 M-parameter head joins the last layer)
        self.bucket_bounds: List[int] = [0]
        self.bucket_segments: List[int] = []
        count = 0
        for i, seg in enumerate(self.segments):
            seg.bucket = len(self.bucket_bounds) - 1
            count += 1
            last = i == len(self.segments) - 1
            if seg.end - self.bucket_bounds[-1] >= min_bucket_elems or last:
                self.bucket_bounds.append(seg.end)
                self.bucket_segments.append(count)
                count = 0
        if len(self.bucket_bounds) > 2 and self.bucket_bounds[-1] - self.bucket_bounds[-2] < min_bucket_elems:
            # a small tail joins the previous bucket
            self.bucket_bounds.pop(-2)
            self.bucket_segments[-2] += self.bucket_segments[-1]
            self.bucket_segments.pop()
            for seg in self.segments:
                seg.bucket = min(seg.bucket, len(self.bucket_bounds) - 2)
        self._split_large_buckets(max_bucket_elems, tail_bucket_elems)
        self.written = [False] * len(self.segments)  # gradient slice holds this step's values already
        self.grad_alt: Optional[torch.Tensor] = None  # second gradient lane (enable_alt)
        self.written_alt: List[bool] = []
        self.window = 0  # accumulation windows started so far (= optimizer steps taken, on every rank alike)
        self.seg_index = {s.name: i for i, s in enumerate(self.segments)}
        self.param_segment: Dict[int, int] = {id(p): i for i, s in enumerate(self.segments) for _, p in s.params}
        if self.p16 is not None:
            self.sync_compute_copy()

    def _split_large_buckets(self, max_elems: Optional[int], tail_elems: Optional[int]) -> None:
        """Cut buckets that are too large for one collective into equal pieces.  A bucket is one segment or a run of
        small ones, and a segment is as large as its parameters: the embedding tables of BERT-large are 119 MB of fp32
        gradients, complete only at the very end of backward - as one all-reduce, all of it is exposed behind the
        backward pass, and nothing (the optimizer on the pieces that have landed, the next piece on another channel)
        can overlap with it.  `max_elems` (default 64 MB of fp32, ``PDT_MAX_BUCKET_MB``) bounds every bucket,
        `tail_elems` (default 32 MB, ``PDT_TAIL_BUCKET_MB``) the last one of backward.  Every piece waits for all
        segments of the bucket it was cut from (a piece does not follow parameter boundaries)."""
        import os

        if max_elems is None:
            max_elems = int(float(os.environ.get("PDT_MAX_BUCKET_MB", "64")) * (1 << 20) / 4)
        if tail_elems is None:
            tail_elems = int(float(os.environ.get("PDT_TAIL_BUCKET_MB", "32")) * (1 << 20) / 4)
        unit = 8 * ALIGN  # piece sizes: a multiple of the parameter alignment and of any rank count up to 8 (512 elements)
        old_bounds, old_expected = self.bucket_bounds, self.bucket_segments
        nb = len(old_bounds) - 1
        bounds, expected, parts = [0], [], []
        for b in range(nb):
            lo, hi = old_bounds[b], old_bounds[b + 1]
            cap = min(max_elems, tail_elems) if b == nb - 1 else max_elems
            k = max(1, -(-(hi - lo) // cap)) if cap > 0 else 1
            step = -(-((hi - lo) // k) // unit) * unit if k > 1 else hi - lo
            mine = []
            for j in range(k):
                end = hi if j == k - 1 else min(hi, lo + (j + 1) * step)
                if end <= bounds[-1]:
                    continue
                mine.append(len(bounds) - 1)
                bounds.append(end)
                expected.append(old_expected[b])
            parts.append(tuple(mine))
        self.bucket_bounds, self.bucket_segments = bounds, expected
        for seg in self.segments:
            seg.buckets = parts[seg.bucket]
            seg.bucket = seg.buckets[0]

    # ------------------------------------------------------------------------------------------
    _GUARD32, _GUARD16 = 0x7FC0DEAD, 0x7E5A  # NaN patterns: a guard word that leaks into arithmetic is visible

    def _alloc(self, name: str, dtype: torch.dtype) -> torch.Tensor:
        """Zeroed arena tensor; with guards it is the middle of a larger allocation fenced by sentinel words."""
        g = self.guard_elems
        if g == 0:
            return torch.zeros(self.numel, device=self.device, dtype=dtype)
        back = torch.zeros(self.numel + 2 * g, device=self.device, dtype=dtype)
        bits = back.view(torch.int32 if back.element_size() == 4 else torch.int16)
        pattern = self._GUARD32 if back.element_size() == 4 else self._GUARD16
        bits[:g].fill_(pattern)
        bits[g + self.numel:].fill_(pattern)
        self._backing[name] = back
        return back[g:g + self.numel]

    def _padding_index(self) -> torch.Tensor:
        """Element indices of the alignment padding between parameters (never written by anyone)."""
        if getattr(self, "_pad_idx", None) is None:
            used = torch.zeros(self.numel, dtype=torch.bool)
            for o, n in self.offsets.values():
                used[o:o + n] = True
            self._pad_idx = (~used).nonzero().flatten().to(self.device)
        return self._pad_idx

    def check_guards(self) -> None:
        """Raise :class:`ArenaGuardError` if a guard word changed or alignment padding is non-zero (host sync)."""
        from ..utils.debug import ArenaGuardError

        g = self.guard_elems
        for name, back in self._backing.items():
            bits = back.view(torch.int32 if back.element_size() == 4 else torch.int16)
            pattern = self._GUARD32 if back.element_size() == 4 else self._GUARD16
            for side, region in (("below", bits[:g]), ("above", bits[g + self.numel:])):
                bad = (region != pattern).nonzero().flatten()
                if bad.numel():
                    raise ArenaGuardError(f"{name} arena: {bad.numel()} guard words {side} the arena were overwritten "
                                          f"(first at guard offset {int(bad[0])})")
        pad = self._padding_index()
        if pad.numel():
            for name, t in (("master", self.master), ("grad", self.grad), ("p16", self.p16)):
                if t is None:
                    continue
                bad = (t[pad] != 0).nonzero().flatten()
                if bad.numel():
                    e = int(pad[bad[0]])
                    owner = max((o, n) for n, (o, _) in self.offsets.items() if o <= e)[1]
                    raise ArenaGuardError(f"{name} arena: alignment padding after '{owner}' is non-zero at element {e}")

    def view16(self, name: str, shape: Sequence[int], count: int = 1) -> torch.Tensor:
        """16-bit view starting at parameter `name`, spanning `count` adjacent parameters."""
        o, _ = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        return self.p16[o:o + n].view(*shape)

    def gview(self, name: str, shape: Sequence[int], lane: int = 0) -> torch.Tensor:
        o, _ = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        return (self.grad if lane == 0 else self.grad_alt)[o:o + n].view(*shape)

    # -- second gradient lane ---------------------------------------------------------------------------
    # Two backward passes that accumulate into ONE buffer must run one after the other.  With a second full-size
    # gradient buffer (1.3 GB for BERT-large: nothing next to 288 GB) the micro-batches of an accumulation window
    # alternate between two lanes whose backward passes run concurrently; merge_alt() adds the second lane into
    # the first before anything (bucket all-reduce, overflow check, optimizer) reads the gradients.
    def enable_alt(self) -> None:
        if self.grad_alt is None:
            self.grad_alt = torch.zeros_like(self.grad)
            self.written_alt = [False] * len(self.segments)

    def merge_alt(self, seg_idx: Optional[int] = None) -> None:
        """grad (+)= grad_alt for one segment (or all) that the second lane wrote in this window."""
        if self.grad_alt is None:
            return
        for i in (range(len(self.segments)) if seg_idx is None else (seg_idx,)):
            if not self.written_alt[i]:
                continue
            seg = self.segments[i]
            dst, src = self.grad[seg.start:seg.end], self.grad_alt[seg.start:seg.end]
            overwrite = seg.native_overwrite and self.p16 is not None and self.master.is_cuda and not self.written[i]
            if overwrite:
                dst.copy_(src)  # (the first lane never touched the slice: it may hold the previous window's values)
            else:
                dst.add_(src)
            self.written[i] = True
            self.written_alt[i] = False
            if not (seg.native_overwrite and self.p16 is not None and self.master.is_cuda):
                src.zero_()  # accumulate-only producers (the head) expect zeros at their first touch

    def mview(self, name: str, shape: Sequence[int]) -> torch.Tensor:
        o, _ = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        return self.master[o:o + n].view(*shape)

    def adjacent(self, names: Sequence[str]) -> bool:
        pos = None
        for n in names:
            o, k = self.offsets[n]
            if pos is not None and o != pos:
                return False
            pos = o + k
        return True

    def sync_compute_copy(self) -> None:
        """Refresh the 16-bit compute copy from the fp32 master (after init / checkpoint load)."""
        if self.p16 is None:
            return
        if self.master.is_cuda:
            from . import native

            native().cast_to_16(self.master, self.p16)
        else:
            self.p16.copy_(self.master)

    def zero_grad(self) -> None:
        """Start a new accumulation window.  Slices whose producers overwrite on first touch are not memset."""
        self.window += 1
        for i, seg in enumerate(self.segments):
            self.written[i] = False
            if not (seg.native_overwrite and self.p16 is not None and self.master.is_cuda):
                self.grad[seg.start:seg.end].zero_()
            if self.grad_alt is not None and self.written_alt[i]:  # (a window that ended without its merge)
                self.written_alt[i] = False
                if not (seg.native_overwrite and self.p16 is not None and self.master.is_cuda):
                    self.grad_alt[seg.start:seg.end].zero_()

    def zero_unwritten(self) -> None:
        """Before the optimizer reads the gradients: a slice that zero_grad() left alone because its native producer
        overwrites it, but whose producer did NOT run in this window (a layer outside the graph of this step), still
        holds the previous window's values - clear it, so that "no gradient" means zero for every kind of segment."""
        if self.p16 is None or not self.master.is_cuda:
            return
        for i, seg in enumerate(self.segments):
            if seg.native_overwrite and not self.written[i]:
                self.grad[seg.start:seg.end].zero_()

    def take_accumulate_flag(self, seg_idx: int, lane: int = 0) -> bool:
        """True if the native producer must add to the slice (of gradient lane `lane`), False if it may overwrite it."""
        flags = self.written if lane == 0 else self.written_alt
        acc = flags[seg_idx]
        flags[seg_idx] = True
        return acc

    def state_bytes(self) -> int:
        per = 8 + (self.p16.element_size() if self.p16 is not None else 0)
        return self.numel * per
