"""CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) — integer state machines.

A literal Python restatement of the reference's Sequence / BlockManager /
Scheduler (ssvgopal/nano-vllm-rs @ 2025-07-18).  Deliberately naive: the same
containers (deque free list with O(n) remove, dict hash index, set of used ids,
sequences copied in and out of the queues) so that every observable — block
ids, block tables, cached-token counts, batch composition, preemption order,
statistics — follows from the cited Rust lines and nothing else.  The product
(nano-vllm-rs_amd/csrc) is an independent O(1) C++ implementation; tests compare
the two on scripted and randomised traces.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Pinning: the known answers of the reference's in-file unit
tests + public XXH64 vectors (B.3); see
tests/test_oracle_kat.py.  Ambiguities resolved per SURVEY.md Appendix A.
"""
from __future__ import annotations

import copy
import itertools
import struct
from collections import deque
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

# ---------------------------------------------------------------------------
# XXH64 — public xxHash spec (what xxhash_rust::xxh64::xxh64 implements,
# src/engine/block_manager.rs:7,122).  Pure Python; the C twin is
# oracle/nvr_oracle.c:nvo_xxh64 and both are checked against python-xxhash.
# ---------------------------------------------------------------------------
_M = (1 << 64) - 1
_P1, _P2, _P3, _P4, _P5 = (0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9,
                           0x85EBCA77C2B2AE63, 0x27D4EB2F165667C5)


def _rotl(x: int, r: int) -> int:
    return ((x << r) | (x >> (64 - r))) & _M


def _round(acc: int, inp: int) -> int:
    acc = (acc + inp * _P2) & _M
    return (_rotl(acc, 31) * _P1) & _M


def _merge(acc: int, v: int) -> int:
    acc ^= _round(0, v)
    return (acc * _P1 + _P4) & _M


def xxh64(data: bytes, seed: int = 0) -> int:
    n, p = len(data), 0
    if n >= 32:
        v1, v2, v3, v4 = (seed + _P1 + _P2) & _M, (seed + _P2) & _M, seed & _M, (seed - _P1) & _M
        while p + 32 <= n:
            a, b, c, d = struct.unpack_from("<4Q", data, p)
            v1, v2, v3, v4 = _round(v1, a), _round(v2, b), _round(v3, c), _round(v4, d)
            p += 32
        h = (_rotl(v1, 1) + _rotl(v2, 7) + _rotl(v3, 12) + _rotl(v4, 18)) & _M
        for v in (v1, v2, v3, v4):
            h = _merge(h, v)
    else:
        h = (seed + _P5) & _M
    h = (h + n) & _M
    while p + 8 <= n:
        (k,) = struct.unpack_from("<Q", data, p)
        h ^= _round(0, k)
        h = (_rotl(h, 27) * _P1 + _P4) & _M
        p += 8
    if p + 4 <= n:
        (k,) = struct.unpack_from("<I", data, p)
        h ^= (k * _P1) & _M
        h = (_rotl(h, 23) * _P2 + _P3) & _M
        p += 4
    while p < n:
        h ^= (data[p] * _P5) & _M
        h = (_rotl(h, 11) * _P1) & _M
        p += 1
    h ^= h >> 33
    h = (h * _P2) & _M
    h ^= h >> 29
    h = (h * _P3) & _M
    h ^= h >> 32
    return h


# ---------------------------------------------------------------------------
# SamplingParams — src/engine/sampling_params.rs:10-119
# ---------------------------------------------------------------------------
@dataclass
class SamplingParams:
    temperature: float = 1.0          # :34
    max_tokens: int = 64              # :35
    ignore_eos: bool = False          # :36
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    repetition_penalty: Optional[float] = None

    def is_greedy(self) -> bool:      # :86-88
        return self.temperature == 0.0

    def validate(self) -> None:       # :91-119
        if self.temperature < 0.0:
            raise ValueError(f"Temperature must be non-negative, got {self.temperature}")
        if self.max_tokens == 0:
            raise ValueError(f"Max tokens must be positive, got {self.max_tokens}")
        if self.top_p is not None and not (0.0 <= self.top_p <= 1.0):
            raise ValueError(f"Top-p must be between 0.0 and 1.0, got {self.top_p}")
        if self.top_k is not None and self.top_k == 0:
            raise ValueError(f"Top-k must be positive, got {self.top_k}")
        if self.repetition_penalty is not None and self.repetition_penalty <= 0.0:
            raise ValueError(f"Repetition penalty must be positive, got {self.repetition_penalty}")


# ---------------------------------------------------------------------------
# Sequence — src/engine/sequence.rs:50-237
# ---------------------------------------------------------------------------
WAITING, RUNNING, FINISHED, PREEMPTED, ERROR = range(5)   # sequence.rs:15-27
_SEQUENCE_COUNTER = itertools.count()                     # sequence.rs:12


def reset_sequence_counter() -> None:
    global _SEQUENCE_COUNTER
    _SEQUENCE_COUNTER = itertools.count()


class Sequence:
    def __init__(self, prompt_token_ids: List[int], sampling_params: SamplingParams,
                 block_size: int = 256, seq_id: Optional[int] = None):
        # sequence.rs:84-101; A-1: block_size comes from the engine config
        # (the reference hard-codes 256 at :99).
        self.seq_id = next(_SEQUENCE_COUNTER) if seq_id is None else seq_id
        self.status = WAITING
        self.token_ids = list(prompt_token_ids)
        self.last_token = prompt_token_ids[-1] if prompt_token_ids else 0
        self.num_tokens = len(prompt_token_ids)
        self.num_prompt_tokens = len(prompt_token_ids)
        self.num_cached_tokens = 0
        self.block_table: List[int] = []
        self.sampling_params = sampling_params
        self.block_size = block_size
        # chunked prefill (extension A-23; the reference schedules whole sequences, scheduler.rs:135-138): tokens whose K/V
        # are in the cache, and the token range [chunk_start, chunk_start + chunk_len) of the step the sequence is in
        self.num_computed_tokens = 0
        self.chunk_start = 0
        self.chunk_len = 0

    def __len__(self) -> int:                       # :104-106
        return self.num_tokens

    def num_completion_tokens(self) -> int:         # :135-137
        return self.num_tokens - self.num_prompt_tokens

    def completion_token_ids(self) -> List[int]:    # :130-132
        return self.token_ids[self.num_prompt_tokens:]

    def append_token(self, token_id: int) -> None:  # :150-154
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    def num_blocks(self) -> int:                    # :157-159
        return (self.num_tokens + self.block_size - 1) // self.block_size

    def num_cached_blocks(self) -> int:             # :162-164
        return self.num_cached_tokens // self.block_size

    def last_block_num_tokens(self) -> int:         # :167-174
        r = self.num_tokens % self.block_size
        return self.block_size if (r == 0 and self.num_tokens > 0) else r

    def get_block_tokens(self, block_idx: int) -> List[int]:   # :177-186
        start = block_idx * self.block_size
        end = min((block_idx + 1) * self.block_size, self.num_tokens)
        return [] if start >= self.num_tokens else self.token_ids[start:end]

    def should_stop(self, eos_token_id: Optional[int]) -> bool:  # :189-205
        if self.num_completion_tokens() >= self.sampling_params.max_tokens:
            return True
        if not self.sampling_params.ignore_eos and eos_token_id is not None:
            if self.last_token == eos_token_id:
                return True
        return False

    def is_finished(self) -> bool:                  # :140-142
        return self.status in (FINISHED, ERROR)

    def can_schedule(self) -> bool:                 # :145-147
        return self.status in (WAITING, PREEMPTED)

    def finish(self) -> None:                       # :208-210
        self.status = FINISHED

    def preempt(self) -> None:                      # :213-218
        self.status = PREEMPTED
        self.block_table.clear()
        self.num_cached_tokens = 0
        self.num_computed_tokens = 0

    def clone(self) -> "Sequence":                  # #[derive(Clone)] :49
        c = copy.copy(self)
        c.token_ids = list(self.token_ids)
        c.block_table = list(self.block_table)
        return c


# ---------------------------------------------------------------------------
# BlockManager — src/engine/block_manager.rs:12-361
# ---------------------------------------------------------------------------
class Block:
    def __init__(self, block_id: int):              # :27-34
        self.block_id = block_id
        self.ref_count = 0
        self.hash: Optional[int] = None
        self.token_ids: List[int] = []

    def update(self, h: int, token_ids: List[int]) -> None:   # :37-40
        self.hash = h
        self.token_ids = token_ids

    def reset(self) -> None:                        # :43-47 (ref_count = 1!)
        self.ref_count = 1
        self.hash = None
        self.token_ids = []

    def is_free(self) -> bool:                      # :50-52
        return self.ref_count == 0


class NoFreeBlocks(RuntimeError):
    pass


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):     # :91-106
        assert num_blocks > 0, "Number of blocks must be positive"
        assert block_size > 0, "Block size must be positive"
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: dict = {}
        self.free_block_ids = deque(range(num_blocks))        # A-2
        self.used_block_ids: set = set()

    @staticmethod
    def compute_hash(token_ids: List[int], prefix_hash: Optional[int] = None) -> int:  # :109-123, A-3
        data = b""
        if prefix_hash is not None:
            data += struct.pack("<Q", prefix_hash)
        data += struct.pack(f"<{len(token_ids)}q", *token_ids)
        return xxh64(data, 0)

    def _allocate_block(self, block_id: int) -> Block:        # :126-134
        assert self.blocks[block_id].is_free(), f"Block {block_id} is not free"
        self.blocks[block_id].reset()
        self.free_block_ids.remove(block_id)                  # VecDeque::retain
        self.used_block_ids.add(block_id)
        return self.blocks[block_id]

    def _deallocate_block(self, block_id: int) -> None:       # :137-149
        assert self.blocks[block_id].is_free(), f"Block {block_id} still has references"
        self.used_block_ids.discard(block_id)
        self.free_block_ids.append(block_id)
        h = self.blocks[block_id].hash
        if h is not None and self.hash_to_block_id.get(h) == block_id:
            del self.hash_to_block_id[h]

    def can_allocate(self, seq: Sequence) -> bool:            # :152-154, A-5
        return len(self.free_block_ids) >= seq.num_blocks()

    def allocate(self, seq: Sequence) -> None:                # :157-219, A-4
        if seq.block_table:
            raise RuntimeError("Sequence already has allocated blocks")
        if not self.can_allocate(seq):
            raise NoFreeBlocks("Not enough free blocks to allocate sequence")
        prefix_hash: Optional[int] = None
        cache_miss = False
        for block_idx in range(seq.num_blocks()):
            block_tokens = seq.get_block_tokens(block_idx)
            current_hash = (self.compute_hash(block_tokens, prefix_hash)
                            if len(block_tokens) == self.block_size else None)
            if current_hash is not None:
                existing = self.hash_to_block_id.get(current_hash)
                if existing is not None:
                    if not cache_miss and self.blocks[existing].token_ids == block_tokens:
                        seq.num_cached_tokens += self.block_size
                        if existing in self.used_block_ids:
                            self.blocks[existing].ref_count += 1
                        else:
                            self._allocate_block(existing)     # dead code per A-4
                        block_id = existing
                    else:
                        cache_miss = True
                        block_id = self._allocate_new_block(current_hash, block_tokens)
                else:
                    cache_miss = True
                    block_id = self._allocate_new_block(current_hash, block_tokens)
            else:
                cache_miss = True
                block_id = self._allocate_new_block(None, block_tokens)
            seq.block_table.append(block_id)
            prefix_hash = current_hash

    def _allocate_new_block(self, h: Optional[int], token_ids: List[int]) -> int:   # :222-237
        if not self.free_block_ids:
            raise NoFreeBlocks("No free blocks available")
        block_id = self.free_block_ids[0]
        block = self._allocate_block(block_id)
        if h is not None:
            block.update(h, token_ids)
            self.hash_to_block_id[h] = block_id
        else:
            block.token_ids = token_ids
        return block_id

    def deallocate(self, seq: Sequence) -> None:              # :240-252
        for block_id in reversed(seq.block_table):
            block = self.blocks[block_id]
            assert block.ref_count > 0, "Cannot remove reference from block with zero refs"
            block.ref_count -= 1
            if block.is_free():
                self._deallocate_block(block_id)
        seq.num_cached_tokens = 0
        seq.block_table.clear()

    def can_append(self, seq: Sequence) -> bool:              # :255-262
        if len(seq) % self.block_size == 1:
            return len(self.free_block_ids) > 0
        return True

    def may_append(self, seq: Sequence) -> None:              # :265-304
        if not seq.block_table:
            raise RuntimeError("Sequence has no allocated blocks")
        last_idx = len(seq.block_table) - 1
        last_id = seq.block_table[last_idx]
        last_block = self.blocks[last_id]
        if len(seq) % self.block_size == 1:
            if last_block.hash is not None:
                if not self.free_block_ids:
                    raise NoFreeBlocks("No free blocks for append")
                new_id = self.free_block_ids[0]
                self._allocate_block(new_id)
                seq.block_table.append(new_id)
        elif len(seq) % self.block_size == 0:
            if last_block.hash is None:
                block_tokens = seq.get_block_tokens(seq.num_blocks() - 1)
                prefix_hash = (self.blocks[seq.block_table[last_idx - 1]].hash
                               if len(seq.block_table) > 1 else None)
                h = self.compute_hash(block_tokens, prefix_hash)
                last_block.update(h, block_tokens)
                self.hash_to_block_id[h] = last_id

    def get_stats(self) -> dict:                              # :307-315
        return dict(total_blocks=self.num_blocks, free_blocks=len(self.free_block_ids),
                    used_blocks=len(self.used_block_ids), cached_blocks=len(self.hash_to_block_id),
                    block_size=self.block_size)


# ---------------------------------------------------------------------------
# Config (fields used by the scheduler) — src/config.rs:16-119
# ---------------------------------------------------------------------------
@dataclass
class Config:
    max_num_batched_tokens: int = 32768
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    eos_token_id: Optional[int] = None
    kvcache_block_size: int = 256
    num_kvcache_blocks: Optional[int] = None
    device: str = "cuda"
    dtype: str = "float16"
    enable_chunked_prefill: bool = False      # extension A-23 (the reference has no intra-sequence chunking)

    def validate(self) -> None:                               # config.rs:83-119 (path checks omitted)
        if self.kvcache_block_size % 256 != 0:
            raise ValueError(f"KV cache block size must be a multiple of 256, got {self.kvcache_block_size}")
        if not (1 <= self.tensor_parallel_size <= 8):
            raise ValueError(f"Tensor parallel size must be between 1 and 8, got {self.tensor_parallel_size}")
        if not (0.0 <= self.gpu_memory_utilization <= 1.0):
            raise ValueError("GPU memory utilization must be between 0.0 and 1.0")
        if self.device not in ("cuda", "cpu", "metal"):
            raise ValueError(f"Unsupported device: {self.device}")
        if self.dtype not in ("float16", "bfloat16", "float32"):
            raise ValueError(f"Unsupported dtype: {self.dtype}")


# ---------------------------------------------------------------------------
# Scheduler — src/engine/scheduler.rs:70-365
# ---------------------------------------------------------------------------
@dataclass
class SchedulerStats:                                         # scheduler.rs:38-66
    total_sequences: int = 0
    waiting_sequences: int = 0
    running_sequences: int = 0
    finished_sequences: int = 0
    preemptions: int = 0
    prefill_batches: int = 0
    decode_batches: int = 0
    avg_prefill_batch_size: float = 0.0
    avg_decode_batch_size: float = 0.0


class Scheduler:
    def __init__(self, config: Config):                       # :70-85
        self.block_manager = BlockManager(
            config.num_kvcache_blocks if config.num_kvcache_blocks is not None else 1000,
            config.kvcache_block_size)
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos_token_id = config.eos_token_id
        self.chunked = bool(getattr(config, "enable_chunked_prefill", False))
        self.waiting: deque = deque()
        self.running: deque = deque()
        self.stats = SchedulerStats()

    def is_finished(self) -> bool:                            # :88-90
        return not self.waiting and not self.running

    def add_sequence(self, seq: Sequence) -> None:            # :93-98
        seq.status = WAITING
        self.waiting.append(seq)
        self.stats.total_sequences += 1
        self._update_stats()

    def schedule(self) -> Tuple[List[Sequence], bool]:        # :103-116
        seqs = self._try_schedule_prefill()
        if seqs is not None:
            self.stats.prefill_batches += 1
            n = float(self.stats.prefill_batches)             # :283-288
            self.stats.avg_prefill_batch_size = (
                self.stats.avg_prefill_batch_size * (n - 1.0) + float(len(seqs))) / n
            return seqs, True
        seqs = self._try_schedule_decode()
        self.stats.decode_batches += 1
        n = float(self.stats.decode_batches)                  # :291-296
        self.stats.avg_decode_batch_size = (
            self.stats.avg_decode_batch_size * (n - 1.0) + float(len(seqs))) / n
        return seqs, False

    def _try_schedule_prefill_chunked(self) -> Optional[List[Sequence]]:
        """Extension A-23 — intra-sequence chunked prefill on top of :119-168: the head of the waiting queue is scheduled for
        min(remaining, budget left) tokens instead of being held back until its whole remainder fits the token budget.  Its
        blocks are allocated for the whole prompt when its first chunk is scheduled (can_allocate / allocate exactly as
        :141-149); a sequence whose prompt is not finished stays at the FRONT of the waiting queue with its blocks and closes
        the batch; its last chunk moves it to running like :152-165.  A step samples a token only for sequences whose
        prefill completed in it."""
        if not self.waiting:
            return None
        scheduled: List[Sequence] = []
        finished_prefill: List[Sequence] = []
        num_seqs = 0
        num_batched_tokens = 0
        while self.waiting:
            seq = self.waiting[0]
            if num_seqs >= self.max_num_seqs:
                break
            budget_left = self.max_num_batched_tokens - num_batched_tokens
            if budget_left <= 0:
                break
            if not seq.block_table:                       # first chunk: blocks for the whole prompt
                if not self.block_manager.can_allocate(seq):
                    break
                self.block_manager.allocate(seq)
                seq.num_computed_tokens = 0
            remaining = len(seq) - seq.num_computed_tokens
            chunk = min(remaining, budget_left)
            seq.chunk_start, seq.chunk_len = seq.num_computed_tokens, chunk
            num_seqs += 1
            num_batched_tokens += chunk
            scheduled.append(seq)
            if chunk < remaining:                         # budget exhausted inside this prompt
                break
            self.waiting.popleft()
            seq.status = RUNNING
            finished_prefill.append(seq)
        if not scheduled:
            return None
        for seq in finished_prefill:
            self.running.append(seq.clone())
        return scheduled

    def _try_schedule_prefill(self) -> Optional[List[Sequence]]:   # :119-168
        if self.chunked:
            return self._try_schedule_prefill_chunked()
        if not self.waiting:
            return None
        scheduled: List[Sequence] = []
        num_seqs = 0
        num_batched_tokens = 0
        while self.waiting:
            seq = self.waiting[0]
            if num_seqs >= self.max_num_seqs:
                break
            seq_tokens = len(seq) - seq.num_cached_tokens
            if num_batched_tokens + seq_tokens > self.max_num_batched_tokens:
                break
            if not self.block_manager.can_allocate(seq):
                break
            seq = self.waiting.popleft()
            self.block_manager.allocate(seq)
            num_seqs += 1
            num_batched_tokens += seq_tokens
            seq.status = RUNNING
            seq.chunk_start, seq.chunk_len = 0, len(seq)
            scheduled.append(seq)
        if not scheduled:
            return None
        for seq in scheduled:
            self.running.append(seq.clone())
        return scheduled

    def _try_schedule_decode(self) -> List[Sequence]:         # :171-223, A-16
        scheduled: List[Sequence] = []
        num_seqs = 0
        to_reschedule: List[Sequence] = []
        while self.running:
            seq = self.running.popleft()
            if num_seqs >= self.max_num_seqs:
                to_reschedule.append(seq)
                continue
            self_preempted = False
            while not self.block_manager.can_append(seq):
                if self.running:
                    self._preempt_sequence(self.running.pop())
                elif scheduled:
                    self._preempt_sequence(scheduled.pop())
                else:
                    self._preempt_sequence(seq)
                    self_preempted = True
                    break
            if not self_preempted and self.block_manager.can_append(seq):
                num_seqs += 1
                self.block_manager.may_append(seq)
                seq.chunk_start, seq.chunk_len = len(seq) - 1, 1
                scheduled.append(seq)
        for seq in reversed(to_reschedule):
            self.running.appendleft(seq)
        for seq in reversed(scheduled):
            self.running.appendleft(seq.clone())
        if not scheduled:
            raise RuntimeError("No sequences could be scheduled for decode")
        return scheduled

    def _preempt_sequence(self, seq: Sequence) -> None:       # :226-231
        seq.status = PREEMPTED
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)
        self.stats.preemptions += 1

    def postprocess(self, sequences: List[Sequence], token_ids: List[int]) -> None:   # :234-257
        if len(sequences) != len(token_ids):
            raise RuntimeError("Mismatch between sequences and token_ids length")
        for seq, token_id in zip(sequences, token_ids):
            seq.num_computed_tokens = seq.chunk_start + seq.chunk_len
            if seq.num_computed_tokens < len(seq):        # A-23: a prefill chunk that does not finish the prompt — no token yet
                continue
            seq.append_token(token_id)
            if seq.should_stop(self.eos_token_id):
                seq.finish()
                self.block_manager.deallocate(seq)
                self.running = deque(s for s in self.running if s.seq_id != seq.seq_id)   # :260-262
                self.stats.finished_sequences += 1
            else:
                for i, s in enumerate(self.running):          # :265-274
                    if s.seq_id == seq.seq_id:
                        self.running[i] = seq
                        break
                else:
                    self.running.append(seq)
        self._update_stats()

    def _update_stats(self) -> None:                          # :277-280
        self.stats.waiting_sequences = len(self.waiting)
        self.stats.running_sequences = len(self.running)

    def preempt_all(self) -> None:                            # :314-319
        seqs = list(self.running)
        self.running.clear()
        for seq in seqs:
            self._preempt_sequence(seq)
        for seq in self.waiting:                          # A-23: a partially prefilled prompt gives its blocks back too
            if seq.block_table:
                self.block_manager.deallocate(seq)
                seq.num_computed_tokens = 0

    def memory_pressure(self) -> float:                       # :322-329
        st = self.block_manager.get_stats()
        return 0.0 if st["total_blocks"] == 0 else 1.0 - st["free_blocks"] / st["total_blocks"]

    def get_queue_lengths(self) -> Tuple[int, int]:           # :309-311
        return len(self.waiting), len(self.running)


# ---------------------------------------------------------------------------
# Step-input builders — src/engine/model_runner.rs:172-300 (A-6, A-7)
# ---------------------------------------------------------------------------
def slot_of(seq: Sequence, pos: int, block_size: int) -> int:
    """A-6: slot(pos) = block_table[pos // bs] * bs + pos % bs."""
    return seq.block_table[pos // block_size] * block_size + pos % block_size


def prepare_prefill(seqs: List[Sequence], block_size: int) -> dict:
    """prepare_prefill_inputs :172-193 + create_prefill_context :222-263 (A-7: all
    tokens from position 0; slot mapping per A-6)."""
    if any(s.chunk_len and (s.chunk_start != 0 or s.chunk_len != len(s)) for s in seqs):
        # A-23: token ranges [chunk_start, chunk_start + chunk_len); earlier tokens are reached through the block table
        # (flash_attention_varlen_with_cache, attention.rs:211-222): context_lens = end of the chunk
        ids, pos, cu, slots, ctx = [], [], [0], [], []
        for s in seqs:
            a, b = s.chunk_start, s.chunk_start + s.chunk_len
            ids.extend(s.token_ids[a:b]); pos.extend(range(a, b)); cu.append(cu[-1] + (b - a))
            slots.extend(slot_of(s, p, block_size) for p in range(a, b))
            ctx.append(b)
        max_blocks = max([len(s.block_table) for s in seqs] or [1])
        bt = [list(s.block_table) + [-1] * (max_blocks - len(s.block_table)) for s in seqs]
        return dict(input_ids=ids, positions=pos, cu_seqlens_q=cu, slot_mapping=slots, context_lens=ctx, block_tables=bt,
                    max_blocks=max_blocks)
    ids, pos, cu, slots = [], [], [0], []
    max_len = 0
    for s in seqs:
        n = len(s)
        ids.extend(s.token_ids)
        pos.extend(range(n))
        cu.append(cu[-1] + n)
        max_len = max(max_len, n)
        slots.extend(slot_of(s, p, block_size) for p in range(n))
    return dict(input_ids=ids, positions=pos, cu_seqlens_q=cu, cu_seqlens_k=list(cu),
                max_seqlen_q=max_len, max_seqlen_k=max_len, slot_mapping=slots)


def prepare_decode(seqs: List[Sequence], block_size: int) -> dict:
    """prepare_decode_inputs :196-210 + create_decode_context :266-300."""
    ids = [s.last_token for s in seqs]
    pos = [len(s) - 1 for s in seqs]
    slots = [slot_of(s, len(s) - 1, block_size) for s in seqs]
    ctx = [len(s) for s in seqs]
    max_blocks = max([s.num_blocks() for s in seqs] or [1])
    bt = [list(s.block_table) + [-1] * (max_blocks - len(s.block_table)) for s in seqs]
    return dict(input_ids=ids, positions=pos, slot_mapping=slots, context_lens=ctx,
                block_tables=bt, max_blocks=max_blocks)


# ---------------------------------------------------------------------------
# Text in, SequenceOutput out — src/engine/llm_engine.rs:70-128,200-230 and
# src/engine/sequence.rs:30-47
# ---------------------------------------------------------------------------
TOKENIZE_MAX_CHARS = 100


def tokenize(text: str) -> List[int]:
    """LLMEngine::tokenize, llm_engine.rs:220-230 — the reference's placeholder:
    text.chars().map(|c| c as u32 as i64).take(100)."""
    return [ord(c) for c in text][:TOKENIZE_MAX_CHARS]


def detokenize(ids: List[int]) -> str:
    """Inverse of the placeholder tokenizer; ids that are not Unicode scalar values become U+FFFD."""
    return "".join(chr(i) if 0 <= i <= 0x10FFFF and not 0xD800 <= i <= 0xDFFF else "\ufffd" for i in ids)


@dataclass
class SequenceOutput:                                   # sequence.rs:30-47
    seq_id: int
    text: str
    token_ids: List[int]
    completion_token_ids: List[int]
    num_prompt_tokens: int
    num_completion_tokens: int
    status: int


def sequence_output(seq: "Sequence") -> SequenceOutput:
    comp = list(seq.completion_token_ids())
    return SequenceOutput(seq.seq_id, detokenize(comp), list(seq.token_ids), comp, seq.num_prompt_tokens, len(comp), seq.status)
