"""BERT / RoBERTa sequence classifiers with HF-identical parameter names.

What the reference obtains from ``AutoModelForSequenceClassification.from_pretrained``
(/root/reference/test_data_parallelism.py:112, /root/reference/test_model_parallelism.py:230-238)
is implemented here from scratch.  The module tree mirrors the HF attribute names, so
``state_dict()`` keys are interchangeable with HF checkpoints
(``bert.encoder.layer.{i}.attention.self.query.weight`` ...).

Two execution paths share the same parameters:

* **torch path** – plain fp32 PyTorch math.  It is the numerical oracle of the test-suite
  and the CPU/gloo "plumbing" configuration.
* **native path** – once :func:`~pytorch_distributed_training_amd.ops.arena.ParamArena` has
  adopted the parameters, every encoder layer runs as ONE coarse autograd node whose
  forward/backward are hand-written gfx950 kernels (``ops/fused.py``).

The pieces the reference's model-parallel script reaches into are kept reachable:
``get_input_embeddings()``, ``.bert.pooler``, ``.bert.encoder``, ``.classifier``,
``.dropout``, ``.num_labels``, ``inputs_embeds=``, ``output_hidden_states=True`` and
``get_extended_attention_mask`` (/root/reference/test_model_parallelism.py:58,65,73-74,114-125).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import EncoderConfig, preset


@dataclass
class SequenceClassifierOutput:
    """Attribute- and key-addressable result, like the HF class of the same name."""

    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None
    last_hidden_state: Optional[torch.Tensor] = None
    pooler_output: Optional[torch.Tensor] = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return [v for v in (self.loss, self.logits, self.hidden_states, self.attentions) if v is not None][k]

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]


class MaskInfo:
    """Padding information of one batch in the forms the two paths need.

    ``mask`` is the ``[B, S]`` 0/1 attention mask; the torch path adds
    ``(1-mask) * finfo.min`` to the scores (HF semantics), the native attention kernel
    takes the mask as int32 plus a per-sample ``kv_len`` (one past the last attended key)
    so it can skip key blocks that are entirely padding.
    """

    def __init__(self, mask: Optional[torch.Tensor], batch: int, seq: int, device, cu_seqlens=None, rowmap=None):
        """`cu_seqlens` / `rowmap`: an UNPADDED batch (native path only) - the hidden states are the packed [T, H] rows
        of the real tokens, sample b owns rows cu_seqlens[b] .. cu_seqlens[b+1], `seq` is the longest sample and
        `rowmap[t]` the dropout row (sample id * 512 + position) of packed row t."""
        self.batch, self.seq = batch, seq
        self.cu, self.rowmap = cu_seqlens, rowmap
        if mask is None and cu_seqlens is None:
            mask = torch.ones(batch, seq, dtype=torch.int32, device=device)
        self.mask = mask
        self._ext = {}
        self._native = None

    def extended(self, dtype: torch.dtype) -> torch.Tensor:
        if dtype not in self._ext:
            self._ext[dtype] = get_extended_attention_mask(self.mask, dtype)
        return self._ext[dtype]

    def native(self):
        if self._native is None and self.cu is not None:
            dummy = torch.zeros(1, dtype=torch.int32, device=self.cu.device)  # (never read: every key of a sample is real)
            self._native = (dummy, dummy)
        if self._native is None:
            m = self.mask.to(torch.int32).contiguous()
            idx = torch.arange(1, self.seq + 1, device=m.device, dtype=torch.int32)
            kv_len = (m.ne(0).to(torch.int32) * idx).amax(dim=1).to(torch.int32).contiguous()
            self._native = (m, kv_len)
        return self._native


def get_extended_attention_mask(attention_mask: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """``[B,S]`` 0/1 mask -> additive ``[B,1,1,S]`` mask with ``finfo(dtype).min`` on padding."""
    ext = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - ext) * torch.finfo(dtype).min


# ----------------------------------------------------------------------------------------
# module tree (names == HF names)
# ----------------------------------------------------------------------------------------
class BertEmbeddings(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.cfg = cfg
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.dropout_p = cfg.hidden_dropout_prob
        self._native = None  # set by ParamArena.adopt

    def position_ids(self, input_ids: Optional[torch.Tensor], batch: int, seq: int, device) -> torch.Tensor:
        if self.cfg.model_type == "roberta":
            pad = self.cfg.pad_token_id
            if input_ids is None:
                return torch.arange(pad + 1, seq + pad + 1, device=device).expand(batch, seq)
            m = input_ids.ne(pad).to(torch.int64)
            return torch.cumsum(m, dim=1) * m + pad
        return torch.arange(seq, device=device).expand(batch, seq)

    def forward(self, input_ids=None, token_type_ids=None, inputs_embeds=None, rng=None):
        if inputs_embeds is not None:
            batch, seq = inputs_embeds.shape[:2]
            device = inputs_embeds.device
        else:
            batch, seq = input_ids.shape
            device = input_ids.device
        if seq > self.cfg.max_position_embeddings:
            raise ValueError(f"sequence length {seq} exceeds the position table ({self.cfg.max_position_embeddings})")
        if token_type_ids is None:
            token_type_ids = torch.zeros(batch, seq, dtype=torch.long, device=device)
        pos = self.position_ids(input_ids, batch, seq, device)
        if self._native is not None and device.type == "cuda":
            from ..ops import fused

            return fused.embeddings_forward(self, input_ids, token_type_ids, pos, inputs_embeds, rng)
        if inputs_embeds is None:
            inputs_embeds = self.word_embeddings(input_ids)
        x = inputs_embeds + self.position_embeddings(pos) + self.token_type_embeddings(token_type_ids)
        x = self.LayerNorm(x)
        return F.dropout(x, self.dropout_p, self.training)


class BertSelfAttention(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        h = cfg.hidden_size
        self.query = nn.Linear(h, h)
        self.key = nn.Linear(h, h)
        self.value = nn.Linear(h, h)


class BertSelfOutput(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class BertAttention(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.self = BertSelfAttention(cfg)
        self.output = BertSelfOutput(cfg)


class BertIntermediate(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.dense = nn.Linear(cfg.intermediate_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class BertLayer(nn.Module):
    """One post-LN transformer block (attention + FFN)."""

    def __init__(self, cfg: EncoderConfig, index: int):
        super().__init__()
        self.cfg = cfg
        self.index = index
        self.attention = BertAttention(cfg)
        self.intermediate = BertIntermediate(cfg)
        self.output = BertOutput(cfg)
        self._native = None  # set by ParamArena.adopt

    def arena_order(self) -> List[Tuple[str, nn.Parameter]]:
        """Parameter order inside the flat arena.

        Q/K/V weights (and biases) are adjacent so that one ``[3H, H]`` view over them is
        the weight of the fused QKV GEMM, and its gradient is written by a single wgrad.
        """
        a, s, o = self.attention, self.attention.self, self.attention.output
        return [
            ("attention.self.query.weight", s.query.weight),
            ("attention.self.key.weight", s.key.weight),
            ("attention.self.value.weight", s.value.weight),
            ("attention.self.query.bias", s.query.bias),
            ("attention.self.key.bias", s.key.bias),
            ("attention.self.value.bias", s.value.bias),
            ("attention.output.dense.weight", o.dense.weight),
            ("attention.output.dense.bias", o.dense.bias),
            ("attention.output.LayerNorm.weight", o.LayerNorm.weight),
            ("attention.output.LayerNorm.bias", o.LayerNorm.bias),
            ("intermediate.dense.weight", self.intermediate.dense.weight),
            ("intermediate.dense.bias", self.intermediate.dense.bias),
            ("output.dense.weight", self.output.dense.weight),
            ("output.dense.bias", self.output.dense.bias),
            ("output.LayerNorm.weight", self.output.LayerNorm.weight),
            ("output.LayerNorm.bias", self.output.LayerNorm.bias),
        ]

    def forward(self, hidden: torch.Tensor, mask: MaskInfo, rng=None) -> torch.Tensor:
        if self._native is not None and hidden.is_cuda:
            from ..ops import fused

            return fused.encoder_layer_forward(self, hidden, mask, rng)
        return self._forward_torch(hidden, mask)

    def _forward_torch(self, x: torch.Tensor, mask: MaskInfo) -> torch.Tensor:
        cfg = self.cfg
        b, s, h = x.shape
        a, d = cfg.num_attention_heads, cfg.head_dim
        sa, so = self.attention.self, self.attention.output

        def heads(t):
            return t.view(b, s, a, d).transpose(1, 2)

        q, k, v = heads(sa.query(x)), heads(sa.key(x)), heads(sa.value(x))
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
        scores = scores + mask.extended(scores.dtype)
        probs = F.dropout(torch.softmax(scores, dim=-1), cfg.attention_probs_dropout_prob, self.training)
        ctx = torch.matmul(probs, v).transpose(1, 2).reshape(b, s, h)
        x = so.LayerNorm(F.dropout(so.dense(ctx), cfg.hidden_dropout_prob, self.training) + x)
        inter = F.gelu(self.intermediate.dense(x))
        y = F.dropout(self.output.dense(inter), cfg.hidden_dropout_prob, self.training)
        return self.output.LayerNorm(y + x)


class BertEncoder(nn.Module):
    def __init__(self, cfg: EncoderConfig, layer_range: Optional[Tuple[int, int]] = None):
        super().__init__()
        self.cfg = cfg
        lo, hi = layer_range if layer_range is not None else (0, cfg.num_hidden_layers)
        self.layer_range = (lo, hi)
        # ModuleDict keyed by the GLOBAL layer index keeps HF names on a pipeline stage
        self.layer = nn.ModuleDict({str(i): BertLayer(cfg, i) for i in range(lo, hi)})

    def layers(self) -> List[BertLayer]:
        return [self.layer[str(i)] for i in range(*self.layer_range)]

    def forward(self, hidden_states, attention_mask=None, mask_info: Optional[MaskInfo] = None,
                output_hidden_states=False, rng=None):
        """Run the stack.  ``attention_mask`` may be a ``[B,S]`` 0/1 mask or the additive
        ``[B,1,1,S]`` form that ``get_extended_attention_mask`` returns."""
        b, s = hidden_states.shape[:2]
        if mask_info is None:
            m = attention_mask
            if m is not None and m.dim() == 4:
                m = (m[:, 0, 0, :] == 0).to(torch.int32)
            mask_info = MaskInfo(m, b, s, hidden_states.device)
        layers = self.layers()
        if (not output_hidden_states and hidden_states.is_cuda and layers
                and all(l._native is not None for l in layers)):
            from ..ops import fused

            return (fused.encoder_stack_forward(layers, hidden_states, mask_info, rng), None)
        all_h = [hidden_states] if output_hidden_states else None
        for lyr in layers:
            hidden_states = lyr(hidden_states, mask_info, rng)
            if output_hidden_states:
                all_h.append(hidden_states)
        return (hidden_states, tuple(all_h) if output_hidden_states else None)


class BertPooler(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self._native = None

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        first = hidden_states[:, 0]
        if self._native is not None and first.is_cuda:
            from ..ops import fused

            return fused.dense_tanh(first.to(self._native.w.dtype), self._native)
        return torch.tanh(self.dense(first.to(self.dense.weight.dtype)))


class BertModel(nn.Module):
    def __init__(self, cfg: EncoderConfig, add_pooling_layer=True, layer_range=None,
                 with_embeddings=True):
        super().__init__()
        self.cfg = cfg
        self.embeddings = BertEmbeddings(cfg) if with_embeddings else None
        self.encoder = BertEncoder(cfg, layer_range)
        self.pooler = BertPooler(cfg) if add_pooling_layer else None

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, inputs_embeds=None,
                output_hidden_states=False, rng=None, hidden_states=None):
        """`hidden_states`: input of a pipeline stage that has no embeddings (output of the previous stage)."""
        if self.embeddings is None:
            if hidden_states is None:
                raise ValueError("this stage has no embeddings: pass hidden_states= from the previous stage")
            x = hidden_states
        else:
            x = self.embeddings(input_ids, token_type_ids, inputs_embeds, rng)
        b, s = x.shape[:2]
        mask = MaskInfo(attention_mask, b, s, x.device)
        last, all_h = self.encoder(x, mask_info=mask, output_hidden_states=output_hidden_states, rng=rng)
        pooled = self.pooler(last) if self.pooler is not None else None
        return SequenceClassifierOutput(last_hidden_state=last, pooler_output=pooled, hidden_states=all_h)


class _ClassifierBase(nn.Module):
    """Shared plumbing of the two sequence-classification heads."""

    base_model_prefix = "bert"

    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        cfg.validate()
        self.config = cfg
        self.num_labels = cfg.num_labels
        self.arena = None  # ParamArena once adopted
        self.runtime = None  # ops.fused.NativeRuntime once the native path is enabled
        self._head_native = None

    def arena_segments(self, prefix: str = ""):
        from ..ops.arena import bert_segments

        return bert_segments(self, prefix)

    def enable_native(self, arena, seed: int = 0, prefix: str = "", runtime=None, site_offset: int = 0, global_seed=None):
        """Route embeddings / layers / pooler / head through the gfx950 kernels.  `arena` must hold this
        model's parameters with a 16-bit compute copy."""
        from ..ops import fused

        if arena.p16 is None:
            raise ValueError("the native path needs a 16-bit compute copy (fp16 or bf16 mode)")
        cfg = self.config
        if cfg.head_dim != 64:
            raise ValueError("native attention kernels are specialised for head_dim == 64")
        rt = runtime or fused.NativeRuntime(arena, seed, global_seed)
        self.arena, self.runtime = arena, rt
        base = self.base_model
        bp = f"{prefix}{self.base_model_prefix}."
        if base.embeddings is not None:
            base.embeddings._native = fused.EmbedNative(arena, bp + "embeddings.", cfg,
                                                        arena.seg_index[f"{prefix}embeddings"], rt)
            base.embeddings._native.site_offset = site_offset
        for lyr in base.encoder.layers():
            lyr._native = fused.LayerNative(arena, f"{bp}encoder.layer.{lyr.index}.", cfg,
                                            arena.seg_index[f"{prefix}layer{lyr.index}"], rt)
            lyr._native.site_offset = site_offset
        if getattr(self, "with_head", False):
            self._enable_native_head(arena, rt, prefix, arena.seg_index[f"{prefix}head"])
        return rt

    def _begin_forward(self, sample_ids=None):
        """`sample_ids` ([B] global ids of the batch's samples, added by this framework's loaders): key the dropout
        masks by sample instead of by rank-local row (ops.fused.NativeRuntime)."""
        if self.runtime is not None:
            self.runtime.begin_forward(sample_ids, train=self.training and torch.is_grad_enabled())

    # -- unpadded ("varlen") execution of the native path ----------------------------------------------------------
    def _can_run_unpadded(self, input_ids, cu_seqlens, token_index, inputs_embeds, hidden_states, output_hidden_states):
        if (cu_seqlens is None or token_index is None or self.runtime is None or inputs_embeds is not None
                or output_hidden_states):
            return False
        if self.base_model.embeddings is not None:
            return input_ids is not None and input_ids.is_cuda and hidden_states is None
        # a later pipeline stage: its input is already the packed [T, H] block of the previous stage
        return hidden_states is not None and hidden_states.is_cuda and hidden_states.dim() == 2

    def _forward_unpadded(self, input_ids, token_type_ids, labels, sample_ids, cu_seqlens, token_index,
                          hidden_states=None, padded_shape=None, inputs_embeds=None, run_head=True):
        """The model (or one pipeline stage of it) on the T real tokens of a dynamically padded batch
        (T = cu_seqlens[-1]): embeddings, every GEMM, LayerNorm and attention work on packed [T, H] rows; the pooler
        reads the first row of every sample.  Numerically the padded computation restricted to the real tokens
        (padding never influences them); dropout masks are keyed by (sample, position), so they are those of the
        padded run too.  ``last_hidden_state`` of the result is the packed [T, H] tensor - the block a stage
        without a head hands to the next one, which passes it back in as `hidden_states` together with the
        `padded_shape` (B, S) of the batch it belongs to.  `inputs_embeds`: packed [T, H] word embeddings computed by
        the caller instead of a lookup of `input_ids` (tower ensembles share one lookup); `run_head=False` stops
        after the encoder stack."""
        from ..ops import fused

        base = self.base_model
        emb = base.embeddings
        lookup = emb is not None and inputs_embeds is None
        if not lookup and padded_shape is None:
            raise ValueError("packed rows need the attention mask of their batch (for the padded shape)")
        B, S = input_ids.shape if lookup else padded_shape
        tok = token_index
        b_of = torch.div(tok, S, rounding_mode="floor")
        sid = sample_ids.reshape(-1).to(torch.int64)[b_of] if sample_ids is not None else b_of
        rowmap = (sid * 512 + (tok - b_of * S)).to(torch.int32)
        cu = cu_seqlens.to(torch.int32)
        mask = MaskInfo(None, B, S, tok.device, cu_seqlens=cu, rowmap=rowmap)
        if emb is not None:
            ids = input_ids.reshape(-1)[tok] if lookup else None
            types = (token_type_ids.reshape(-1)[tok] if token_type_ids is not None else torch.zeros_like(tok))
            pos = emb.position_ids(input_ids if lookup else None, B, S, tok.device).reshape(-1)[tok]
            x = fused.embeddings_forward_packed(emb, ids, types, pos, rowmap, inputs_embeds)
        else:
            x = hidden_states
        x = fused.encoder_stack_forward(base.encoder.layers(), x, mask)
        if not (run_head and getattr(self, "with_head", False)):
            return SequenceClassifierOutput(last_hidden_state=x)
        first = x.index_select(0, cu[:-1].to(torch.int64)).unsqueeze(1)  # token 0 of every sample: [B, 1, H]
        res = self.head(first, labels)
        res.last_hidden_state = x
        return res

    @property
    def base_model(self) -> BertModel:
        return getattr(self, self.base_model_prefix)

    def get_input_embeddings(self):
        return self.base_model.get_input_embeddings()

    def get_extended_attention_mask(self, attention_mask, input_shape=None, device=None, dtype=torch.float32):
        m = get_extended_attention_mask(attention_mask, dtype)
        return m.to(device) if device is not None else m

    def init_weights(self, seed: Optional[int] = None) -> None:
        """HF initialisation: N(0, initializer_range) for Linear/Embedding weights, zero biases
        and padding row, unit LayerNorm."""
        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(seed)
        std = self.config.initializer_range
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    m.weight.copy_(torch.empty(m.weight.shape).normal_(0.0, std, generator=gen))
                    if m.bias is not None:
                        m.bias.zero_()
                elif isinstance(m, nn.Embedding):
                    m.weight.copy_(torch.empty(m.weight.shape).normal_(0.0, std, generator=gen))
                    if m.padding_idx is not None:
                        m.weight[m.padding_idx].zero_()
                elif isinstance(m, nn.LayerNorm):
                    m.weight.fill_(1.0)
                    m.bias.zero_()

    def num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters())


def _ce_loss(logits: torch.Tensor, labels: torch.Tensor, num_labels: int) -> torch.Tensor:
    return F.cross_entropy(logits.float().view(-1, num_labels), labels.view(-1))


class BertForSequenceClassification(_ClassifierBase):
    base_model_prefix = "bert"

    def __init__(self, cfg: EncoderConfig, layer_range=None, with_embeddings=True, with_head=True):
        super().__init__(cfg)
        self.bert = BertModel(cfg, add_pooling_layer=with_head, layer_range=layer_range,
                              with_embeddings=with_embeddings)
        self.with_head = with_head
        if with_head:
            self.dropout = nn.Dropout(cfg.head_dropout)
            self.classifier = nn.Linear(cfg.hidden_size, cfg.num_labels)
        self.init_weights()

    def _enable_native_head(self, arena, rt, prefix, seg):
        from ..ops import fused

        H, bp = self.config.hidden_size, f"{prefix}bert."
        self.bert.pooler._native = fused.DenseNative(arena, bp + "pooler.dense.weight", bp + "pooler.dense.bias", H, seg, rt)
        self._head_native = fused.HeadNative(arena, f"{prefix}classifier.weight", f"{prefix}classifier.bias",
                                             self.num_labels, H, seg, rt)

    def head(self, last_hidden: torch.Tensor, labels=None) -> SequenceClassifierOutput:
        pooled = self.bert.pooler(last_hidden)
        if self._head_native is not None and pooled.is_cuda:
            from ..ops import fused

            logits, loss = fused.classifier_head(pooled, self._head_native, labels,
                                                 self.config.head_dropout if self.training else 0.0)
            return SequenceClassifierOutput(loss=loss, logits=logits)
        logits = self.classifier(self.dropout(pooled))
        loss = _ce_loss(logits, labels, self.num_labels) if labels is not None else None
        return SequenceClassifierOutput(loss=loss, logits=logits)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, labels=None,
                inputs_embeds=None, output_hidden_states=False, return_dict=True, rng=None, hidden_states=None,
                sample_ids=None, cu_seqlens=None, token_index=None, **unused):
        self._begin_forward(sample_ids)
        if self._can_run_unpadded(input_ids, cu_seqlens, token_index, inputs_embeds, hidden_states, output_hidden_states):
            return self._forward_unpadded(input_ids, token_type_ids, labels, sample_ids, cu_seqlens, token_index, hidden_states,
                                          attention_mask.shape if attention_mask is not None else None)
        out = self.bert(input_ids, attention_mask, token_type_ids, inputs_embeds, output_hidden_states, rng,
                        hidden_states)
        if not self.with_head:  # a pipeline stage that ends inside the encoder stack
            return out
        res = self.head(out.last_hidden_state, labels)
        res.hidden_states = out.hidden_states
        res.last_hidden_state = out.last_hidden_state
        return res


class RobertaClassificationHead(nn.Module):
    def __init__(self, cfg: EncoderConfig):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.dropout = nn.Dropout(cfg.head_dropout)
        self.out_proj = nn.Linear(cfg.hidden_size, cfg.num_labels)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        x = self.dropout(features[:, 0].to(self.dense.weight.dtype))
        x = self.dropout(torch.tanh(self.dense(x)))
        return self.out_proj(x)


class RobertaForSequenceClassification(_ClassifierBase):
    base_model_prefix = "roberta"

    def __init__(self, cfg: EncoderConfig, layer_range=None, with_embeddings=True, with_head=True):
        super().__init__(cfg)
        self.roberta = BertModel(cfg, add_pooling_layer=False, layer_range=layer_range,
                                 with_embeddings=with_embeddings)
        self.with_head = with_head
        if with_head:
            self.classifier = RobertaClassificationHead(cfg)
        self.init_weights()

    def _enable_native_head(self, arena, rt, prefix, seg):
        from ..ops import fused

        H, cp = self.config.hidden_size, f"{prefix}classifier."
        self._dense_native = fused.DenseNative(arena, cp + "dense.weight", cp + "dense.bias", H, seg, rt)
        self._head_native = fused.HeadNative(arena, cp + "out_proj.weight", cp + "out_proj.bias", self.num_labels, H,
                                             seg, rt)

    def head(self, last_hidden: torch.Tensor, labels=None) -> SequenceClassifierOutput:
        if self._head_native is not None and last_hidden.is_cuda:
            from ..ops import fused

            p = self.config.head_dropout if self.training else 0.0
            first = fused.dropout16(last_hidden[:, 0].contiguous(), self.runtime, p, fused.SITE_HEAD_PRE)
            pooled = fused.dense_tanh(first, self._dense_native)
            logits, loss = fused.classifier_head(pooled, self._head_native, labels, p)
            return SequenceClassifierOutput(loss=loss, logits=logits)
        logits = self.classifier(last_hidden)
        loss = _ce_loss(logits, labels, self.num_labels) if labels is not None else None
        return SequenceClassifierOutput(loss=loss, logits=logits)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, labels=None,
                inputs_embeds=None, output_hidden_states=False, return_dict=True, rng=None, hidden_states=None,
                sample_ids=None, cu_seqlens=None, token_index=None, **unused):
        self._begin_forward(sample_ids)
        if self._can_run_unpadded(input_ids, cu_seqlens, token_index, inputs_embeds, hidden_states, output_hidden_states):
            return self._forward_unpadded(input_ids, token_type_ids, labels, sample_ids, cu_seqlens, token_index, hidden_states,
                                          attention_mask.shape if attention_mask is not None else None)
        out = self.roberta(input_ids, attention_mask, token_type_ids, inputs_embeds, output_hidden_states, rng,
                        hidden_states)
        if not self.with_head:  # a pipeline stage that ends inside the encoder stack
            return out
        res = self.head(out.last_hidden_state, labels)
        res.hidden_states = out.hidden_states
        res.last_hidden_state = out.last_hidden_state
        return res


def build_model(name_or_cfg, layer_range=None, with_embeddings=True, with_head=True, **overrides) -> _ClassifierBase:
    """``AutoModelForSequenceClassification``-style factory (random init; see
    ``checkpoint.load_pretrained`` for weights on disk).  `layer_range` / `with_embeddings` / `with_head`
    build one stage of a layer-split model; parameter names keep the GLOBAL layer indices."""
    cfg = name_or_cfg if isinstance(name_or_cfg, EncoderConfig) else preset(name_or_cfg, **overrides)
    cls = RobertaForSequenceClassification if cfg.model_type == "roberta" else BertForSequenceClassification
    return cls(cfg, layer_range=layer_range, with_embeddings=with_embeddings, with_head=with_head)
