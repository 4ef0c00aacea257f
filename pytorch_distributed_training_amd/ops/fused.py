"""Coarse autograd nodes over the native kernels.

PyTorch autograd is kept only as the OUTER tape: embeddings, each encoder layer, the pooler and the
classification head are one node each.  A node's backward writes parameter gradients straight into the
gradient arena (overwriting on the first micro-batch of an accumulation window, adding afterwards) and
tells the runtime that its arena segment is complete, which is what lets the bucket reducer start the
all-reduce of a layer while backward is still running on the layers below it.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch

from . import native
from .arena import ParamArena

SITE_EMBED = 0x7F000000
SITE_HEAD = 0x7F000010
SITE_HEAD_PRE = 0x7F000011


class RngKey:
    """What the dropout sites of one forward pass key their counters with (kept by every autograd node, so that
    backward replays exactly its own forward's masks whatever ran in between)."""

    __slots__ = ("state", "counter", "sample_ids")

    def __init__(self, state: torch.Tensor, counter: int, sample_ids: Optional[torch.Tensor]):
        self.state, self.counter, self.sample_ids = state, counter, sample_ids

    def rows(self, lo: int, hi: int) -> "RngKey":
        return RngKey(self.state, self.counter, None if self.sample_ids is None else self.sample_ids[lo:hi].contiguous())


class NativeRuntime:
    """Per-model state shared by the native nodes: dropout streams, completion hook.

    Two ways to key a dropout mask:

    * a batch that carries ``sample_ids`` (the loaders of this framework add them): counters derive from (seed,
      optimizer-step window, site, GLOBAL sample id, position, column).  A mask belongs to the sample, whichever
      rank, pipeline micro-batch or padded length it is processed with, so data-parallel x model-parallel runs
      reproduce the loss curve of a single GPU on the same global batch with dropout on.
    * no ids: (rank-specific seed, per-rank micro-batch counter, site, local row, column) - distinct but
      reproducible streams per rank, as a seeded torch generator per process would give.
    """

    def __init__(self, arena: ParamArena, seed: int = 0, global_seed: Optional[int] = None):
        self.arena = arena
        self.rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=arena.device)
        gs = seed if global_seed is None else global_seed
        self.rng_state_global = torch.tensor([gs ^ 0x5BD1E995, 0], dtype=torch.int64, device=arena.device)
        self.micro = 0  # per-rank micro-batch counter (the no-ids keying)
        self.key = RngKey(self.rng_state, 0, None)  # of the forward pass in progress
        self.segment_done: Optional[Callable[[int], None]] = None  # set by the DP reducer wrapper
        # segments whose completion is reported later, by release(): a segment that gets part of its gradient outside
        # its native node (the word-embedding table of a tower-parallel rank, filled by a second backward call)
        self.hold: set = set()
        self._held: List[int] = []
        # gradient lane of the forward pass in progress (ParamArena.enable_alt): captured by every native node, its
        # backward writes the parameter gradients of that lane.  merge_on_done: the backward in progress is the last
        # of its accumulation window - each segment's second lane is added to the first as soon as the segment is done,
        # i.e. before the data-parallel reducer sees it.
        self.lane = 0
        self.merge_on_done = False

    def reseed(self, seed: int) -> None:
        self.rng_state[0] = seed

    def next_micro(self) -> int:
        self.micro = (self.micro + 1) & 0x7FFFFFFF
        self.key = RngKey(self.rng_state, self.micro, None)
        return self.micro

    def begin_forward(self, sample_ids: Optional[torch.Tensor] = None, train: bool = True) -> RngKey:
        """Start the forward pass of one (micro-)batch; returns the key its dropout sites use.  An evaluation pass
        (``train=False``: dropout off) consumes nothing and carries no ids."""
        if not train:
            self.key = RngKey(self.rng_state, self.micro, None)
        elif sample_ids is None:
            self.next_micro()
        else:
            ids = sample_ids.reshape(-1).to(device=self.arena.device, dtype=torch.int32).contiguous()
            # (the window counter advances with every zero_grad(), i.e. once per optimizer step on every rank)
            self.key = RngKey(self.rng_state_global, self.arena.window & 0x7FFFFFFF, ids)
        return self.key

    def done(self, seg: int) -> None:
        if self.merge_on_done:
            self.arena.merge_alt(seg)
        if seg in self.hold:
            self._held.append(seg)
        elif self.segment_done is not None:
            self.segment_done(seg)

    def release(self) -> None:
        """Report the held segments that completed since the last call."""
        held, self._held = self._held, []
        if self.segment_done is not None:
            for seg in held:
                self.segment_done(seg)


class LayerNative:
    """16-bit weight views and fp32 gradient views of one encoder layer, in the order csrc expects."""

    def __init__(self, arena: ParamArena, prefix: str, cfg, segment: int, runtime: NativeRuntime):
        H, Fd = cfg.hidden_size, cfg.intermediate_size
        a = prefix + "attention."
        qkv_w = [a + f"self.{n}.weight" for n in ("query", "key", "value")]
        qkv_b = [a + f"self.{n}.bias" for n in ("query", "key", "value")]
        if not (arena.adjacent(qkv_w) and arena.adjacent(qkv_b)):
            raise RuntimeError("Q/K/V parameters are not adjacent in the arena")
        spec = [
            (qkv_w[0], (3 * H, H)), (qkv_b[0], (3 * H,)),
            (a + "output.dense.weight", (H, H)), (a + "output.dense.bias", (H,)),
            (a + "output.LayerNorm.weight", (H,)), (a + "output.LayerNorm.bias", (H,)),
            (prefix + "intermediate.dense.weight", (Fd, H)), (prefix + "intermediate.dense.bias", (Fd,)),
            (prefix + "output.dense.weight", (H, Fd)), (prefix + "output.dense.bias", (H,)),
            (prefix + "output.LayerNorm.weight", (H,)), (prefix + "output.LayerNorm.bias", (H,)),
        ]
        self.w: List[torch.Tensor] = [arena.view16(n, s) for n, s in spec]
        self.g: List[torch.Tensor] = [arena.gview(n, s) for n, s in spec]
        self._spec, self._g_alt = spec, None
        self.segment = segment
        self.rt = runtime
        self.arena = arena
        self.site_offset = 0  # distinguishes the dropout streams of equal layer indices in different towers

    def grads(self, lane: int) -> List[torch.Tensor]:
        if lane == 0:
            return self.g
        if self._g_alt is None:
            self._g_alt = [self.arena.gview(n, s, 1) for n, s in self._spec]
        return self._g_alt


class _LayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layer, mask):
        nat: LayerNative = layer._native
        cfg = layer.cfg
        packed = mask.cu is not None  # unpadded batch: x is [T, H], sample b = rows cu[b] .. cu[b+1]
        B, S, H = (mask.batch, mask.seq, x.shape[-1]) if packed else x.shape
        m, kvl = mask.native()
        train = layer.training
        key = nat.rt.key
        x2 = x if packed else x.reshape(B * S, H)
        outs = native().layer_forward(
            x2, m, kvl, nat.w, key.state, B, S, cfg.num_attention_heads,
            cfg.attention_probs_dropout_prob if train else 0.0, cfg.hidden_dropout_prob if train else 0.0,
            cfg.layer_norm_eps, key.counter, layer.index + nat.site_offset, key.sample_ids, mask.cu, mask.rowmap)
        ctx.layer, ctx.mask, ctx.x2, ctx.saved = layer, (m, kvl, mask.cu, mask.rowmap), x2, outs[1:]
        ctx.meta = (B, S, H, train, key, packed)
        ctx.lane = nat.rt.lane
        return outs[0] if packed else outs[0].view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        nat: LayerNative = layer._native
        cfg = layer.cfg
        B, S, H, train, key, packed = ctx.meta
        acc = nat.arena.take_accumulate_flag(nat.segment, ctx.lane)
        dx = native().layer_backward(
            dy.reshape(-1, H), ctx.x2, list(ctx.saved), ctx.mask[0], ctx.mask[1], nat.w, nat.grads(ctx.lane), key.state,
            B, S, cfg.num_attention_heads,
            cfg.attention_probs_dropout_prob if train else 0.0, cfg.hidden_dropout_prob if train else 0.0,
            key.counter, layer.index + nat.site_offset, acc, key.sample_ids, ctx.mask[2], ctx.mask[3])
        ctx.saved = ctx.x2 = None
        nat.rt.done(nat.segment)
        return (dx if packed else dx.view(B, S, H)), None, None


def encoder_layer_forward(layer, hidden, mask, rng=None):
    nat = layer._native
    if hidden.dtype != nat.w[0].dtype:
        hidden = hidden.to(nat.w[0].dtype)
    if hidden.dim() == 2 and mask.cu is None:
        raise ValueError("2-D hidden states are the packed rows of an unpadded batch: the mask must carry cu_seqlens")
    if not hidden.requires_grad and torch.is_grad_enabled():
        # the stack must stay on the tape even when its input is a constant (e.g. a pipeline stage whose
        # input arrived from the previous rank): parameter gradients are produced inside backward()
        hidden = hidden.detach().requires_grad_()
    return _LayerFn.apply(hidden.contiguous(), layer, mask)


class _StackFn(torch.autograd.Function):
    """A run of encoder layers as one autograd node (csrc: stack_forward / stack_backward).  Same kernels, same
    dropout keys, same gradient-segment bookkeeping as a chain of _LayerFn nodes - the per-layer work that needs
    Python (accumulate flags, the segment-done callback of the data-parallel reducer) is done before resp. from
    inside the one C++ call."""

    @staticmethod
    def forward(ctx, x, layers, mask):
        first = layers[0]
        cfg = first.cfg
        packed = mask.cu is not None
        B, S, H = (mask.batch, mask.seq, x.shape[-1]) if packed else x.shape
        m, kvl = mask.native()
        train = first.training
        key = first._native.rt.key
        x2 = x if packed else x.reshape(B * S, H)
        ids = [l.index + l._native.site_offset for l in layers]
        outs = native().stack_forward(
            x2, m, kvl, [l._native.w for l in layers], key.state, B, S, cfg.num_attention_heads,
            cfg.attention_probs_dropout_prob if train else 0.0, cfg.hidden_dropout_prob if train else 0.0,
            cfg.layer_norm_eps, key.counter, ids, key.sample_ids, mask.cu, mask.rowmap)
        ctx.layers, ctx.mask, ctx.x2, ctx.ids = layers, (m, kvl, mask.cu, mask.rowmap), x2, ids
        ctx.saved = [outs[i * 11:(i + 1) * 11] for i in range(len(layers))]
        ctx.meta = (B, S, H, train, key, packed)
        ctx.lane = first._native.rt.lane
        y = outs[-11]
        return y if packed else y.view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        layers = ctx.layers
        cfg = layers[0].cfg
        rt = layers[0]._native.rt
        B, S, H, train, key, packed = ctx.meta
        lane = ctx.lane
        acc = [int(l._native.arena.take_accumulate_flag(l._native.segment, lane)) for l in layers]
        saved, ctx.saved = ctx.saved, None

        def layer_done(i: int) -> None:
            saved[i] = None
            rt.done(layers[i]._native.segment)

        dx = native().stack_backward(
            dy.reshape(-1, H), ctx.x2, saved, ctx.mask[0], ctx.mask[1], [l._native.w for l in layers],
            [l._native.grads(lane) for l in layers], key.state, B, S, cfg.num_attention_heads,
            cfg.attention_probs_dropout_prob if train else 0.0, cfg.hidden_dropout_prob if train else 0.0,
            key.counter, ctx.ids, acc, key.sample_ids, ctx.mask[2], ctx.mask[3], layer_done)
        ctx.x2 = None
        return (dx if packed else dx.view(B, S, H)), None, None


def encoder_stack_forward(layers, hidden, mask, rng=None):
    """`layers` (consecutive BertLayer modules adopted by one arena) applied to `hidden` as one autograd node;
    PDT_LAYER_NODES=1 keeps one node per layer."""
    if not layers:
        return hidden
    # (without a tape - evaluation - the layers run one by one: every layer's saved tensors die with its call, whereas the
    # one-call form returns those of the whole stack at once)
    if len(layers) == 1 or not torch.is_grad_enabled() or os.environ.get("PDT_LAYER_NODES") == "1":
        for lyr in layers:
            hidden = encoder_layer_forward(lyr, hidden, mask, rng)
        return hidden
    nat = layers[0]._native
    if hidden.dtype != nat.w[0].dtype:
        hidden = hidden.to(nat.w[0].dtype)
    if hidden.dim() == 2 and mask.cu is None:
        raise ValueError("2-D hidden states are the packed rows of an unpadded batch: the mask must carry cu_seqlens")
    if not hidden.requires_grad and torch.is_grad_enabled():
        hidden = hidden.detach().requires_grad_()  # (see encoder_layer_forward)
    return _StackFn.apply(hidden.contiguous(), layers, mask)


# --------------------------------------------------------------------------------------------------
class EmbedNative:
    def __init__(self, arena: ParamArena, prefix: str, cfg, segment: int, runtime: NativeRuntime):
        V, P, R, H = cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size, cfg.hidden_size
        spec = [
            (prefix + "word_embeddings.weight", (V, H)), (prefix + "position_embeddings.weight", (P, H)),
            (prefix + "token_type_embeddings.weight", (R, H)), (prefix + "LayerNorm.weight", (H,)),
            (prefix + "LayerNorm.bias", (H,)),
        ]
        self.w = [arena.view16(n, s) for n, s in spec]
        self.g = [arena.gview(n, s) for n, s in spec]
        self._spec, self._g_alt = spec, None
        self.segment, self.rt, self.arena = segment, runtime, arena
        self.site_offset = 0

    grads = LayerNative.grads


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, emb, ids, type_ids, pos_ids, inputs_embeds, rowmap):
        nat: EmbedNative = emb._native
        cfg = emb.cfg
        # (packed rows of an unpadded batch: pos_ids is [T]; the row table alone defines the dropout rows)
        B, S = pos_ids.shape if rowmap is None else (1, pos_ids.numel())
        train = emb.training
        key = nat.rt.key
        idf = ids.reshape(-1).contiguous() if ids is not None else None
        ie = inputs_embeds.reshape(B * S, -1).contiguous() if inputs_embeds is not None else None
        pf, tf = pos_ids.reshape(-1).contiguous(), type_ids.reshape(-1).contiguous()
        y, pre, stats = native().embed_forward(idf, pf, tf, ie, *nat.w, key.state, cfg.layer_norm_eps,
                                               cfg.hidden_dropout_prob if train else 0.0, key.counter,
                                               SITE_EMBED + nat.site_offset, key.sample_ids, S, rowmap)
        ctx.emb, ctx.idx, ctx.saved = emb, (idf, pf, tf), (pre, stats)
        ctx.meta = (B, S, train, key, ie is not None, rowmap)
        ctx.lane = nat.rt.lane
        return y if rowmap is not None else y.view(B, S, cfg.hidden_size)

    @staticmethod
    def backward(ctx, dy):
        emb = ctx.emb
        nat: EmbedNative = emb._native
        cfg = emb.cfg
        B, S, train, key, has_ie, rowmap = ctx.meta
        idf, pf, tf = ctx.idx
        pre, stats = ctx.saved
        acc = nat.arena.take_accumulate_flag(nat.segment, ctx.lane)
        dsum = native().embed_backward(dy.reshape(B * S, -1), pre, stats, idf, pf, tf, nat.w[3], nat.grads(ctx.lane),
                                       key.state, cfg.hidden_dropout_prob if train else 0.0, key.counter,
                                       SITE_EMBED + nat.site_offset, cfg.pad_token_id, acc, key.sample_ids, S, rowmap)
        ctx.saved = None
        nat.rt.done(nat.segment)
        d_ie = None
        if has_ie:
            d_ie = dsum.view(B * S, -1) if rowmap is not None else dsum.view(B, S, -1)
        return None, None, None, None, None, d_ie, None


def embeddings_forward(emb, input_ids, token_type_ids, pos_ids, inputs_embeds, rng=None):
    nat: EmbedNative = emb._native
    if inputs_embeds is not None and inputs_embeds.dtype != nat.w[0].dtype:
        inputs_embeds = inputs_embeds.to(nat.w[0].dtype)
    # `anchor` ties the node to the tape: its gradient is never materialised (grads go to the arena)
    anchor = emb.LayerNorm.weight
    return _EmbedFn.apply(anchor, emb, None if inputs_embeds is not None else input_ids, token_type_ids, pos_ids,
                          inputs_embeds, None)


def embeddings_forward_packed(emb, ids, type_ids, pos_ids, rowmap, inputs_embeds=None):
    """Embeddings of the T real tokens of an unpadded batch (index arguments [T]; `inputs_embeds`, when given instead
    of `ids`, the packed [T, H] word embeddings); returns the packed [T, H] rows."""
    nat: EmbedNative = emb._native
    if inputs_embeds is not None and inputs_embeds.dtype != nat.w[0].dtype:
        inputs_embeds = inputs_embeds.to(nat.w[0].dtype)
    return _EmbedFn.apply(emb.LayerNorm.weight, emb, None if inputs_embeds is not None else ids, type_ids, pos_ids,
                          inputs_embeds, rowmap)


# --------------------------------------------------------------------------------------------------
class DenseNative:
    """A Linear(H,H)+tanh evaluated on token 0 (BERT pooler / RoBERTa head dense)."""

    def __init__(self, arena: ParamArena, wname: str, bname: str, H: int, segment: int, runtime: NativeRuntime):
        self.w, self.b = arena.view16(wname, (H, H)), arena.view16(bname, (H,))
        self.gw, self.gb = arena.gview(wname, (H, H)), arena.gview(bname, (H,))
        self._names, self._alt = (wname, (H, H), bname, (H,)), None
        self.segment, self.rt, self.arena = segment, runtime, arena

    def grads(self, lane: int):
        """(weight gradient, bias gradient) views of gradient lane `lane`; these producers always ADD, so a write into
        the second lane is flagged here for the merge."""
        if lane == 0:
            return self.gw, self.gb
        if self._alt is None:
            wn, ws, bn, bs = self._names
            self._alt = (self.arena.gview(wn, ws, 1), self.arena.gview(bn, bs, 1))
        self.arena.written_alt[self.segment] = True
        return self._alt


class _DenseTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, nat):
        out = native().dense_tanh_forward(first, nat.w, nat.b)
        ctx.nat, ctx.lane = nat, nat.rt.lane
        # an OUTPUT kept as a plain ctx attribute would close a reference cycle (ctx -> out -> grad_fn -> ctx)
        # that keeps the whole graph and its activations alive until the cycle collector runs
        ctx.save_for_backward(first, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        nat: DenseNative = ctx.nat
        first, out = ctx.saved_tensors
        dpre = (dout.float() * (1.0 - out.float() ** 2)).to(out.dtype)
        gw, gb = nat.grads(ctx.lane)
        dfirst = native().dense_backward(dpre.contiguous(), first, nat.w, gw, gb, True)
        nat.rt.done(nat.segment)  # the dense layer is the last producer of the head segment in backward order
        return dfirst, None


def dense_tanh(first_tok: torch.Tensor, nat: DenseNative) -> torch.Tensor:
    """tanh(first_tok @ W^T + b) for a [B, H] (possibly strided) 16-bit view of token 0."""
    if not first_tok.requires_grad and torch.is_grad_enabled():
        first_tok = first_tok.detach().requires_grad_()
    return _DenseTanhFn.apply(first_tok, nat)


class HeadNative:
    def __init__(self, arena: ParamArena, wname: str, bname: str, C: int, H: int, segment: int, runtime: NativeRuntime):
        self.w, self.b = arena.mview(wname, (C, H)), arena.mview(bname, (C,))
        self.gw, self.gb = arena.gview(wname, (C, H)), arena.gview(bname, (C,))
        self._names, self._alt = (wname, (C, H), bname, (C,)), None
        self.segment, self.rt, self.arena = segment, runtime, arena

    grads = DenseNative.grads


class _HeadFn(torch.autograd.Function):
    """dropout -> classifier -> mean cross-entropy; returns (logits, loss)."""

    @staticmethod
    def forward(ctx, pooled, nat, labels, p, site):
        key = nat.rt.key
        logits, loss = native().head_forward(pooled, nat.w, nat.b, labels, key.state, p, key.counter, site, key.sample_ids)
        ctx.nat, ctx.has_labels, ctx.lane = nat, labels is not None, nat.rt.lane
        ctx.save_for_backward(pooled, logits, *([labels] if labels is not None else []))  # logits is an output
        ctx.meta = (p, key, site)
        ctx.set_materialize_grads(False)  # an unused output arrives as None instead of a zero tensor
        return logits, loss

    @staticmethod
    def backward(ctx, dlogits, dloss):
        nat: HeadNative = ctx.nat
        p, key, site = ctx.meta
        pooled, logits = ctx.saved_tensors[:2]
        labels = ctx.saved_tensors[2] if ctx.has_labels else None
        gw, gb = nat.grads(ctx.lane)
        dp = native().head_backward(dloss if labels is not None else None, dlogits, pooled, nat.w, logits,
                                    labels, gw, gb, key.state, p, key.counter, site, True, False, key.sample_ids)
        return dp, None, None, None, None


def classifier_head(pooled: torch.Tensor, nat: HeadNative, labels: Optional[torch.Tensor], p: float,
                    site: int = SITE_HEAD):
    """(logits fp32 [B,C], loss or None) from a 16-bit pooled [B,H]."""
    if labels is not None:
        labels = labels.reshape(-1).to(torch.int64).contiguous()  # the kernel indexes labels densely
    logits, loss = _HeadFn.apply(pooled.contiguous(), nat, labels, p, site)
    return logits, (loss if labels is not None else None)


def dropout16(x: torch.Tensor, rt: NativeRuntime, p: float, site: int) -> torch.Tensor:
    """Counter-based dropout of a 16-bit [B, H] tensor (same mask on forward and backward by construction)."""
    if p <= 0.0:
        return x
    return _DropFn.apply(x, rt, p, site)


class _DropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rt, p, site):
        key = rt.key
        ctx.meta = (p, key, site)
        return native().dropout16(x, key.state, p, key.counter, site, key.sample_ids)

    @staticmethod
    def backward(ctx, dy):
        p, key, site = ctx.meta
        return native().dropout16(dy, key.state, p, key.counter, site, key.sample_ids), None, None, None
