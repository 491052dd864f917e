"""CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) — Qwen3 graph + engine loop.

Restates the reference's model wiring (src/models/qwen3.rs:208-240,305-314,
372-392,487-505), step-input construction (src/engine/model_runner.rs:172-300)
and the engine step (src/engine/llm_engine.rs:155-197) on top of the C float ops
in nvr_oracle.c and the integer state machines in engine_oracle.py.  Decisions
for everything the reference leaves broken: SURVEY.md Appendix A.

Two numeric modes:
  * fp16=False — the reference's CPU path (device="cpu", f32 everywhere).
  * fp16=True  — "fp16-faithful": fp16 weights and fp16 storage between ops, f32
    inside an op — the rounding points of the fp16 GPU path (and of candle's
    fp16 tensors), so kernel parity can be tested at tight tolerance.
Logits are always f32 (decision A-21, DESIGN.md: the LM head keeps its f32
accumulators; the reference casts logits to f32 before sampling anyway,
src/layers/sampler.rs:37).
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional

import copy

import numpy as np

from . import (add, attn_paged, attn_prefill_varlen, embedding, fill_tokens, fill_weight, kv_store, linear,
               rmsnorm, rope_apply, rope_table, round_bf16, round_f16, sample, sample_key, silu_and_mul, weight_key,
               weight_scale)
from . import engine_oracle as eo

# tensor ids of the synthetic weight generator (shared with the product:
# nano-vllm-rs_amd/csrc/weights.h)
TID_QKV, TID_O, TID_GATE_UP, TID_DOWN = 0, 1, 2, 3
TID_BIAS = 4                                  # + tensor id: the bias of that projection (use_bias)
TID_EMBED, TID_LM_HEAD = 1 << 20, (1 << 20) + 1


@dataclass
class ModelConfig:
    """Qwen3Config, src/models/qwen3.rs:26-125 (+ head_dim override, A-17)."""
    vocab_size: int = 151936
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    head_dim: Optional[int] = None
    max_position_embeddings: int = 32768
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    tie_word_embeddings: bool = False
    init_std: float = 0.02
    seed: int = 0
    # Extension (DESIGN A-27; False = the reference graph, which has no such layers): the real Qwen3
    # checkpoints carry self_attn.q_norm / k_norm, an RMSNorm over head_dim on every q and k head between the projection and
    # RoPE.  They are restated with the reference's own RMSNorm::forward_simple (layernorm.rs:58-75) and rms_norm_eps.
    qk_norm: bool = False
    # Qwen3Config::use_bias (qwen3.rs:54-55,82: default false): a bias on the four projections of a layer — qkv_proj :167, o_proj :178,
    # gate_up_proj :276, down_proj :287; the row-parallel ones hold it on rank 0 only (linear.rs:206).  A-30: candle_nn::Linear::forward is a
    # matmul and a broadcast_add, two tensor ops, so the 16-bit modes round the product and then the sum (x·Wᵀ -> 16 bit, + b -> 16 bit).
    use_bias: bool = False

    def hd(self) -> int:                       # qwen3.rs:101-103
        return self.head_dim if self.head_dim else self.hidden_size // self.num_attention_heads

    def validate(self, tp: int = 1) -> None:   # qwen3.rs:106-124
        if self.head_dim is None and self.hidden_size % self.num_attention_heads != 0:
            raise ValueError("Hidden size must be divisible by number of attention heads")
        if self.num_attention_heads % tp != 0:
            raise ValueError("Number of attention heads must be divisible by tensor parallel size")
        if self.num_key_value_heads % tp != 0:
            raise ValueError("Number of key-value heads must be divisible by tensor parallel size")
        if self.intermediate_size % tp != 0:
            raise ValueError("Intermediate size must be divisible by tensor parallel size")


def qwen3_0_6b(**kw) -> ModelConfig:
    return ModelConfig(vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                       num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                       rope_theta=1e6, tie_word_embeddings=True, max_position_embeddings=32768, **kw)


def qwen3_8b(**kw) -> ModelConfig:
    return ModelConfig(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=36,
                       num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                       rope_theta=1e6, tie_word_embeddings=False, max_position_embeddings=32768, **kw)


def tiny(**kw) -> ModelConfig:
    d = dict(vocab_size=256, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
             num_attention_heads=4, num_key_value_heads=2, head_dim=None, rope_theta=10000.0,
             tie_word_embeddings=False, max_position_embeddings=512, init_std=0.1)
    d.update(kw)
    return ModelConfig(**d)


def small(**kw) -> ModelConfig:
    """Smallest shape the gfx950 kernels accept (head_dim 64, 16-multiples): GPU end-to-end parity."""
    d = dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
             num_attention_heads=4, num_key_value_heads=2, head_dim=64, rope_theta=10000.0,
             tie_word_embeddings=False, max_position_embeddings=2048, init_std=0.05)
    d.update(kw)
    return ModelConfig(**d)


class _CompactLayer(dict):
    """Layer weights kept as IEEE fp16 in host memory and widened to f32 when an op asks for them (exact: in fp16 mode every
    weight is an fp16 value already).  Halves the oracle's footprint so that Qwen3-8B (BASELINE configs[3]) fits a test box."""

    def __setitem__(self, k, v):
        super().__setitem__(k, v.astype(np.float16) if isinstance(v, np.ndarray) and v.ndim == 2 else v)

    def __getitem__(self, k):
        v = super().__getitem__(k)
        return v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float16 else v


class OracleModel:
    """One tensor-parallel rank of the Qwen3 graph with synthetic weights."""

    def __init__(self, cfg: ModelConfig, num_blocks: int, block_size: int, fp16: bool = True,
                 tp_rank: int = 0, tp_size: int = 1, max_pos: Optional[int] = None, compact: bool = False, bf16: bool = False):
        cfg.validate(tp_size)
        if bf16:
            fp16 = False                       # bf16 = True: the 16-bit type of every rounding point is bfloat16 (Config.dtype "bfloat16")
        self.bf16 = bf16
        if compact and not fp16:
            raise ValueError("compact weight storage is exact only in fp16 mode")
        self.compact = compact
        self.cfg, self.fp16, self.tp_rank, self.tp_size = cfg, fp16, tp_rank, tp_size
        self.D = cfg.hd()
        self.H = cfg.num_attention_heads // tp_size          # qwen3.rs:158
        self.KVH = cfg.num_key_value_heads // tp_size        # qwen3.rs:159
        self.I = cfg.intermediate_size // tp_size
        self.Hd = cfg.hidden_size
        self.V = cfg.vocab_size
        self.block_size, self.num_blocks = block_size, num_blocks
        self.scale = float(np.float32(1.0) / np.sqrt(np.float32(self.D)))  # attention.rs:45
        mp = max_pos or cfg.max_position_embeddings
        self.cos, self.sin = rope_table(self.D, mp, cfg.rope_theta)
        self._gen_weights()
        # create_kv_cache, model_runner.rs:364-396: [NB, bs, KVH/tp, D] per layer per K/V
        shape = (num_blocks, block_size, self.KVH, self.D)
        self.k_cache = [np.zeros(shape, np.float32) for _ in range(cfg.num_hidden_layers)]
        self.v_cache = [np.zeros(shape, np.float32) for _ in range(cfg.num_hidden_layers)]

    # -- synthetic weights, sharded by the reference's TP shape rules ---------
    def _gen_weights(self) -> None:
        c, r, tp = self.cfg, self.tp_rank, self.tp_size
        sc = weight_scale(c.init_std)
        Hg, KVHg, D, Hd, Ig = c.num_attention_heads, c.num_key_value_heads, self.D, self.Hd, c.intermediate_size
        f16 = self.fp16
        if self.bf16:                          # generated unrounded, then rounded to bfloat16 like the product's fill_weight
            _fw = fill_weight
            fill_w = lambda *a: round_bf16(_fw(*a[:-1], False))
        else:
            fill_w = fill_weight
        self.layers: List[Dict[str, np.ndarray]] = []
        for l in range(c.num_hidden_layers):
            key = lambda tid: weight_key(c.seed, l * 8 + tid)
            # QKVParallelLinear (linear.rs:300-340): global rows [q heads | k heads | v heads]
            q = fill_w(self.H * D, Hd, Hd, r * self.H * D, 0, key(TID_QKV), sc, f16)
            k = fill_w(self.KVH * D, Hd, Hd, Hg * D + r * self.KVH * D, 0, key(TID_QKV), sc, f16)
            v = fill_w(self.KVH * D, Hd, Hd, (Hg + KVHg) * D + r * self.KVH * D, 0, key(TID_QKV), sc, f16)
            # RowParallelLinear o_proj (linear.rs:180-268): global [Hd, H*D], columns sharded
            o = fill_w(Hd, self.H * D, Hg * D, 0, r * self.H * D, key(TID_O), sc, f16)
            # MergedColumnParallelLinear (linear.rs:378-454): global rows [gate | up], each sharded
            g = fill_w(self.I, Hd, Hd, r * self.I, 0, key(TID_GATE_UP), sc, f16)
            u = fill_w(self.I, Hd, Hd, Ig + r * self.I, 0, key(TID_GATE_UP), sc, f16)
            # down_proj row-parallel: global [Hd, I], columns sharded
            d = fill_w(Hd, self.I, Ig, 0, r * self.I, key(TID_DOWN), sc, f16)
            W = _CompactLayer() if self.compact else {}
            if c.use_bias:
                # one value per output feature, generated like a one-column weight (row = the GLOBAL output row: shards take their slices)
                bias = lambda rows, row0, tid: fill_w(rows, 1, 1, row0, 0, key(TID_BIAS + tid), sc, f16).reshape(rows)
                W["qkv_b"] = np.concatenate([bias(self.H * D, r * self.H * D, TID_QKV), bias(self.KVH * D, Hg * D + r * self.KVH * D, TID_QKV),
                                             bias(self.KVH * D, (Hg + KVHg) * D + r * self.KVH * D, TID_QKV)])
                W["gate_up_b"] = np.concatenate([bias(self.I, r * self.I, TID_GATE_UP), bias(self.I, Ig + r * self.I, TID_GATE_UP)])
                W["o_b"] = bias(Hd, 0, TID_O) if r == 0 else None             # linear.rs:206: only rank 0 has the bias
                W["down_b"] = bias(Hd, 0, TID_DOWN) if r == 0 else None
            for name, val in dict(qkv=np.concatenate([q, k, v], 0), o=o, gate_up=np.concatenate([g, u], 0), down=d,
                                  ln1=np.ones(Hd, np.float32), ln2=np.ones(Hd, np.float32),
                                  q_norm=np.ones(D, np.float32), k_norm=np.ones(D, np.float32)).items():
                W[name] = val
            self.layers.append(W)
        # Embedding replicated on every rank (skip C2); LM head vocab-sharded
        # (embed_head.rs:57-59), tied to the embedding when tie_word_embeddings (qwen3.rs:461-473).
        self.embed = fill_w(self.V, Hd, Hd, 0, 0, weight_key(c.seed, TID_EMBED), sc, f16)
        vs = self.V // tp
        self.vocab_start = r * vs
        self.vocab_end = self.V if r == tp - 1 else (r + 1) * vs
        if c.tie_word_embeddings:
            self.lm_head = self.embed[self.vocab_start:self.vocab_end]
        else:
            self.lm_head = fill_w(self.vocab_end - self.vocab_start, Hd, Hd, self.vocab_start, 0,
                                       weight_key(c.seed, TID_LM_HEAD), sc, f16)
        self.norm = np.ones(Hd, np.float32)

    # -- checkpoint tensors ---------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, np.ndarray]) -> List[str]:
        """Full (un-sharded) tensors by their HF / reference names (Qwen3Model::load_weights qwen3.rs:518-570) -> this rank's
        slices by the reference's shard rules: ColumnParallelLinear::load_weight narrows dim 0 by rank*out_per_partition
        (linear.rs:154-171), RowParallelLinear::load_weight dim 1 (:249-267), the LM head takes its vocabulary rows
        (embed_head.rs:57-59,142-161); qkv_proj is [q | k | v] rows, gate_up_proj [gate | up] rows (linear.rs:300-340,
        378-454).  Values are rounded to fp16 in fp16 mode.  Returns the names outside the reference graph."""
        if self.compact:
            raise ValueError("load_state_dict: not available with compact weight storage (slices are assigned in place)")
        c, r = self.cfg, self.tp_rank
        D, Hd, H, KVH, I = self.D, self.Hd, self.H, self.KVH, self.I
        Hg, KVHg, Ig = c.num_attention_heads, c.num_key_value_heads, c.intermediate_size
        skipped = []

        def want(a, shape, name):
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"Partition weight shape mismatch: {name} expected {list(shape)}, got {list(a.shape)}")
            return self._r(np.asarray(a, np.float32))
        for full, a in sd.items():
            name = full[6:] if full.startswith("model.") else full
            parts = name.split(".")
            if name == "embed_tokens.weight":
                self.embed = want(a, (self.V, Hd), full)
                if c.tie_word_embeddings:
                    self.lm_head = self.embed[self.vocab_start:self.vocab_end]
            elif name == "norm.weight":
                self.norm = want(a, (Hd,), full)
            elif name == "lm_head.weight":
                if c.tie_word_embeddings:
                    raise ValueError("lm_head.weight: tie_word_embeddings is set")
                self.lm_head = want(a, (self.V, Hd), full)[self.vocab_start:self.vocab_end]
            elif parts[0] == "layers" and parts[1].isdigit() and 0 <= int(parts[1]) < c.num_hidden_layers:
                W, rest = self.layers[int(parts[1])], ".".join(parts[2:])
                if rest == "input_layernorm.weight":
                    W["ln1"] = want(a, (Hd,), full)
                elif rest == "post_attention_layernorm.weight":
                    W["ln2"] = want(a, (Hd,), full)
                elif rest in ("self_attn.q_norm.weight", "self_attn.k_norm.weight") and c.qk_norm:
                    W["q_norm" if ".q_norm." in rest else "k_norm"] = want(a, (D,), full)      # per head_dim: replicated on every rank
                elif rest == "self_attn.q_proj.weight":
                    W["qkv"][:H * D] = want(a, (Hg * D, Hd), full)[r * H * D:(r + 1) * H * D]
                elif rest == "self_attn.k_proj.weight":
                    W["qkv"][H * D:(H + KVH) * D] = want(a, (KVHg * D, Hd), full)[r * KVH * D:(r + 1) * KVH * D]
                elif rest == "self_attn.v_proj.weight":
                    W["qkv"][(H + KVH) * D:] = want(a, (KVHg * D, Hd), full)[r * KVH * D:(r + 1) * KVH * D]
                elif rest == "self_attn.qkv_proj.weight":
                    w = want(a, ((Hg + 2 * KVHg) * D, Hd), full)
                    W["qkv"][:H * D] = w[r * H * D:(r + 1) * H * D]
                    W["qkv"][H * D:(H + KVH) * D] = w[Hg * D + r * KVH * D:Hg * D + (r + 1) * KVH * D]
                    W["qkv"][(H + KVH) * D:] = w[(Hg + KVHg) * D + r * KVH * D:(Hg + KVHg) * D + (r + 1) * KVH * D]
                elif rest == "self_attn.o_proj.weight":
                    W["o"] = np.ascontiguousarray(want(a, (Hd, Hg * D), full)[:, r * H * D:(r + 1) * H * D])
                elif rest == "mlp.gate_proj.weight":
                    W["gate_up"][:I] = want(a, (Ig, Hd), full)[r * I:(r + 1) * I]
                elif rest == "mlp.up_proj.weight":
                    W["gate_up"][I:] = want(a, (Ig, Hd), full)[r * I:(r + 1) * I]
                elif rest == "mlp.gate_up_proj.weight":
                    w = want(a, (2 * Ig, Hd), full)
                    W["gate_up"][:I] = w[r * I:(r + 1) * I]
                    W["gate_up"][I:] = w[Ig + r * I:Ig + (r + 1) * I]
                elif rest == "mlp.down_proj.weight":
                    W["down"] = np.ascontiguousarray(want(a, (Hd, Ig), full)[:, r * I:(r + 1) * I])
                elif c.use_bias and rest.endswith(".bias"):
                    if rest == "self_attn.q_proj.bias":
                        W["qkv_b"][:H * D] = want(a, (Hg * D,), full)[r * H * D:(r + 1) * H * D]
                    elif rest == "self_attn.k_proj.bias":
                        W["qkv_b"][H * D:(H + KVH) * D] = want(a, (KVHg * D,), full)[r * KVH * D:(r + 1) * KVH * D]
                    elif rest == "self_attn.v_proj.bias":
                        W["qkv_b"][(H + KVH) * D:] = want(a, (KVHg * D,), full)[r * KVH * D:(r + 1) * KVH * D]
                    elif rest == "self_attn.o_proj.bias":
                        if r == 0: W["o_b"] = want(a, (Hd,), full)
                    elif rest == "mlp.gate_proj.bias":
                        W["gate_up_b"][:I] = want(a, (Ig,), full)[r * I:(r + 1) * I]
                    elif rest == "mlp.up_proj.bias":
                        W["gate_up_b"][I:] = want(a, (Ig,), full)[r * I:(r + 1) * I]
                    elif rest == "mlp.down_proj.bias":
                        if r == 0: W["down_b"] = want(a, (Hd,), full)
                    else:
                        skipped.append(full)
                else:
                    skipped.append(full)
            else:
                skipped.append(full)
        return skipped

    def _lin(self, x: np.ndarray, W: dict, name: str) -> np.ndarray:
        """candle_nn::Linear::forward (linear.rs:12-24 wraps it): x·Wᵀ, then + bias when the layer has one (A-30: two tensor ops, two roundings)."""
        y = self._r(linear(x, W[name]))
        b = W.get(name + "_b") if self.cfg.use_bias else None
        return y if b is None else self._add(y, np.broadcast_to(b, y.shape).copy())

    def _r(self, x: np.ndarray) -> np.ndarray:
        return round_bf16(x) if self.bf16 else (round_f16(x) if self.fp16 else x)

    def _add(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:   # residual add, qwen3.rs:382,389, rounded to the storage type
        return round_bf16(add(a, b, round16=False)) if self.bf16 else add(a, b, round16=self.fp16)

    # -- one layer, split at the two all-reduce points so TP can be simulated --
    def attn_part(self, l: int, h: np.ndarray, positions, meta: dict) -> np.ndarray:
        """input_layernorm -> qkv -> rope -> store -> attention -> o_proj partial (qwen3.rs:208-240,378-381)."""
        W = self.layers[l]
        n = self._r(rmsnorm(h, W["ln1"], self.cfg.rms_norm_eps))
        qkv = self._lin(n, W, "qkv")
        T = h.shape[0]
        qd, kd = self.H * self.D, self.KVH * self.D
        q = qkv[:, :qd].reshape(T, self.H, self.D)                       # split_qkv linear.rs:331-340
        k = qkv[:, qd:qd + kd].reshape(T, self.KVH, self.D)
        v = np.ascontiguousarray(qkv[:, qd + kd:].reshape(T, self.KVH, self.D))
        if self.cfg.qk_norm:                                             # A-27: RMSNorm over head_dim, fp16 out, before RoPE
            eps = self.cfg.rms_norm_eps
            q = self._r(rmsnorm(np.ascontiguousarray(q).reshape(T * self.H, self.D), W["q_norm"], eps)).reshape(T, self.H, self.D)
            k = self._r(rmsnorm(np.ascontiguousarray(k).reshape(T * self.KVH, self.D), W["k_norm"], eps)).reshape(T, self.KVH, self.D)
        q = self._r(rope_apply(q, positions, self.cos, self.sin))        # rotary_embedding.rs:145-158
        k = self._r(rope_apply(k, positions, self.cos, self.sin))
        kv_store(k, v, meta["slot_mapping"], self.k_cache[l], self.v_cache[l])   # attention.rs:150-174
        if meta["is_prefill"] and meta.get("block_tables") is None:
            a = attn_prefill_varlen(q, k, v, meta["cu_seqlens_q"], self.scale)   # attention.rs:177-208
        else:
            cu = meta["cu_seqlens_q"] if meta["is_prefill"] else np.arange(T + 1, dtype=np.int32)
            a = attn_paged(q, cu, self.k_cache[l], self.v_cache[l], meta["block_tables"],
                           meta["context_lens"], self.scale)             # attention.rs:211-235
        a = self._r(a).reshape(T, qd)
        return self._lin(a, W, "o")

    def mlp_part(self, l: int, h: np.ndarray) -> np.ndarray:
        """post_attention_layernorm -> gate_up -> SiluAndMul -> down partial (qwen3.rs:305-314,385-388)."""
        W = self.layers[l]
        n = self._r(rmsnorm(h, W["ln2"], self.cfg.rms_norm_eps))
        gu = self._lin(n, W, "gate_up")
        act = self._r(silu_and_mul(gu))
        return self._lin(act, W, "down")

    def head_part(self, h: np.ndarray, meta: dict) -> np.ndarray:
        """final norm + last-token select + LM head shard (qwen3.rs:501-504, embed_head.rs:250-306)."""
        if meta["is_prefill"]:
            last = np.asarray(meta["cu_seqlens_q"][1:], dtype=np.int64) - 1     # embed_head.rs:272-289
            h = np.ascontiguousarray(h[last])
        n = self._r(rmsnorm(h, self.norm, self.cfg.rms_norm_eps))
        return linear(n, self.lm_head)                                   # f32 logits, A-21

    def embed_tokens(self, ids) -> np.ndarray:
        return embedding(ids, self.embed)


def forward_tp(ranks: List[OracleModel], ids, positions, meta: dict) -> np.ndarray:
    """Qwen3Model::forward (qwen3.rs:487-505) over tp ranks held in one process: the three
    exchange points (linear.rs:236-238 all-reduce, embed_head.rs:321-336 vocab gather) are
    plain sums / concatenations here."""
    m0 = ranks[0]
    h = m0.embed_tokens(ids)
    for l in range(m0.cfg.num_hidden_layers):
        parts = [m.attn_part(l, h, positions, meta) for m in ranks]
        o = parts[0] if len(parts) == 1 else m0._r(np.sum(parts, axis=0, dtype=np.float32))
        h = m0._add(h, o)                                                # qwen3.rs:382
        parts = [m.mlp_part(l, h) for m in ranks]
        d = parts[0] if len(parts) == 1 else m0._r(np.sum(parts, axis=0, dtype=np.float32))
        h = m0._add(h, d)                                                # qwen3.rs:389
    return np.concatenate([m.head_part(h, meta) for m in ranks], axis=1)


def build_meta(seqs: List[eo.Sequence], is_prefill: bool, block_size: int) -> tuple:
    """ModelRunner::prepare_inputs + create_context, model_runner.rs:159-300."""
    if is_prefill:
        p = eo.prepare_prefill(seqs, block_size)
        meta = dict(is_prefill=True, cu_seqlens_q=np.asarray(p["cu_seqlens_q"], np.int32),
                    slot_mapping=np.asarray(p["slot_mapping"], np.int32), block_tables=None)
        if "block_tables" in p:                              # A-23 chunks: earlier tokens through the block table (K8)
            meta.update(block_tables=np.asarray(p["block_tables"], np.int32), context_lens=np.asarray(p["context_lens"], np.int32))
    else:
        p = eo.prepare_decode(seqs, block_size)
        meta = dict(is_prefill=False, slot_mapping=np.asarray(p["slot_mapping"], np.int32),
                    context_lens=np.asarray(p["context_lens"], np.int32),
                    block_tables=np.asarray(p["block_tables"], np.int32))
    return np.asarray(p["input_ids"], np.int64), np.asarray(p["positions"], np.int64), meta


class OracleEngine:
    """LLMEngine::step loop (llm_engine.rs:155-197): schedule -> execute -> sample -> postprocess."""

    def __init__(self, cfg: ModelConfig, config: eo.Config, fp16: bool = True, tp_size: int = 1,
                 sample_seed: int = 0, max_pos: Optional[int] = None, compact: bool = False, bf16: bool = False):
        self.config = config
        self.scheduler = eo.Scheduler(config)
        nb = config.num_kvcache_blocks if config.num_kvcache_blocks is not None else 1000
        self.ranks = [OracleModel(cfg, nb, config.kvcache_block_size, fp16, r, tp_size, max_pos, compact, bf16)
                      for r in range(tp_size)]
        self.sample_seed = sample_seed
        self.step_count = 0
        self.trace: List[dict] = []
        self.finished: dict = {}               # seq_id -> finished Sequence (SequenceOutput emission, llm_engine.rs:188-196)
        self.last_batch: list = []

    def add_request(self, prompt: List[int], sp: eo.SamplingParams, seq_id: Optional[int] = None) -> eo.Sequence:
        seq = eo.Sequence(prompt, sp, self.config.kvcache_block_size, seq_id)
        self.scheduler.add_sequence(seq)
        return seq

    def execute_model(self, seqs, is_prefill) -> np.ndarray:
        ids, pos, meta = build_meta(seqs, is_prefill, self.config.kvcache_block_size)
        return forward_tp(self.ranks, ids, pos, meta)

    def sample_tokens(self, logits: np.ndarray, seqs) -> List[int]:
        """ModelRunner::sample_tokens, model_runner.rs:131-156 -> Sampler::batch_sample."""
        # Sampler::batch_sample, sampler.rs:221-254: when ANY row of the batch sets top_p, the rows that do not are given
        # top_p.unwrap_or(1.0) and still go through apply_top_p (:233-240); likewise top_k.unwrap_or(0) (:243-247), which A-18
        # reads as "disabled".  Restated literally here; the product treats p >= 1.0 as "no filter" (decision A-26: in exact
        # arithmetic p = 1.0 keeps every token; the f32 cumulative sum of :168-177 can only cut a tail of ~1e-7 total mass —
        # tests/test_oracle_kat.py::test_top_p_one_is_no_filter, tests/test_kernels_gpu.py::test_sampler_mixed_batch_defaults).
        any_top_p = any(s.sampling_params.top_p is not None for s in seqs)
        out = []
        for i, s in enumerate(seqs):
            sp = s.sampling_params
            key = sample_key(self.sample_seed, s.seq_id, s.num_completion_tokens())
            top_p = sp.top_p if sp.top_p is not None else (1.0 if any_top_p else None)
            out.append(sample(logits[i], sp.temperature, sp.top_k or 0, top_p, key))
        return out

    def step(self, forced_tokens: Optional[List[int]] = None) -> dict:
        seqs, is_prefill = self.scheduler.schedule()
        logits = self.execute_model(seqs, is_prefill)
        toks = self.sample_tokens(logits, seqs)
        rec = dict(is_prefill=is_prefill, seq_ids=[s.seq_id for s in seqs],
                   block_tables=[list(s.block_table) for s in seqs], tokens=list(toks), logits=logits)
        partial = [s.chunk_start + s.chunk_len < len(s) for s in seqs]     # A-23: prompts not finished by this step
        if any(partial):
            rec["tokens"] = [-1 if pt else t for t, pt in zip(rec["tokens"], partial)]
            toks = list(rec["tokens"])
        if forced_tokens is not None:          # teacher forcing for near-tie analysis
            toks = list(forced_tokens)
        self.scheduler.postprocess(seqs, toks)
        self.step_count += 1
        self.last_batch = seqs                 # the step's (updated) sequences: what generate_stream reports
        for s in seqs:
            if s.is_finished():
                self.finished[s.seq_id] = s
        return rec

    def generate(self, prompts, sp: eo.SamplingParams, on_output=None) -> List[eo.SequenceOutput]:
        """LLMEngine::generate / generate_stream (llm_engine.rs:70-128): string prompts go through the placeholder
        tokenizer (:220-230), one SamplingParams for all, loop until the scheduler is finished (:131-152); outputs in
        prompt order.  on_output(SequenceOutput) sees every sequence of every step's batch; truthy return = dropped
        receiver (:250-253), the loop stops."""
        if not prompts:
            return []
        seqs = [self.add_request(eo.tokenize(p) if isinstance(p, str) else list(p), copy.deepcopy(sp)) for p in prompts]
        while not self.scheduler.is_finished():
            self.step()
            if on_output is not None:
                for s in self.last_batch:
                    if on_output(eo.sequence_output(s)):
                        return []
        return [eo.sequence_output(self.finished.pop(s.seq_id)) for s in seqs]

    def run(self, max_steps: int = 1 << 30) -> List[dict]:
        out = []
        while not self.scheduler.is_finished() and len(out) < max_steps:
            out.append(self.step())
        return out
