"""nano-vllm-rs_amd — Python host-side mirror of the reference's engine API over the C ABI.

The product is `libnvr.so` (C++ host + gfx950 HIP kernels, C ABI in include/nvr.h).  This package
only binds it with ctypes and mirrors the reference's public names (Config, SamplingParams,
Sequence, BlockManager, Scheduler, ModelRunner, LLMEngine — src/lib.rs:91-94) so that tests and
bench.py read like the reference's own tests.  There is no compute in Python and no CPU fallback:
if the library is missing, importing the bindings raises; if no MI355X is visible, device calls
return NVR_ERR_HIP and raise NvrError.

The directory name has hyphens; import it through `import nvr_import` (repo root), which registers
it as module `nano_vllm_rs_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence as Seq, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVR_LIBNVR") or os.path.join(_HERE, "libnvr.so")   # NVR_LIBNVR: another build of the library (A/B measurements)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nvr.h")


class NvrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[nvr {code}] {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile every HIP/C++ source for gfx950 into libnvr.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


# ---------------------------------------------------------------------------------- structs
class SamplingParamsC(C.Structure):
    _fields_ = [("temperature", C.c_float), ("max_tokens", C.c_uint64), ("ignore_eos", C.c_int32),
                ("has_top_p", C.c_int32), ("top_p", C.c_float), ("has_top_k", C.c_int32), ("top_k", C.c_uint64),
                ("has_repetition_penalty", C.c_int32), ("repetition_penalty", C.c_float)]


class ConfigC(C.Structure):
    _fields_ = [("max_num_batched_tokens", C.c_uint64), ("max_num_seqs", C.c_uint64), ("max_model_len", C.c_uint64),
                ("gpu_memory_utilization", C.c_float), ("tensor_parallel_size", C.c_uint64),
                ("enforce_eager", C.c_int32), ("has_eos", C.c_int32), ("eos_token_id", C.c_int64),
                ("kvcache_block_size", C.c_uint64), ("num_kvcache_blocks", C.c_int64),
                ("tensor_parallel_rank", C.c_uint64), ("device_ordinal", C.c_int32), ("sample_seed", C.c_uint64),
                ("skip_block_size_check", C.c_int32), ("decode_chain", C.c_uint32),
                ("recompute_cached_prefix", C.c_int32), ("enable_chunked_prefill", C.c_int32), ("async_decode", C.c_int32), ("shared_prefix_min_seqs", C.c_int32),
                ("device", C.c_char * 16), ("dtype", C.c_char * 16)]


class ModelConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_uint64), ("hidden_size", C.c_uint64), ("intermediate_size", C.c_uint64),
                ("num_hidden_layers", C.c_uint64), ("num_attention_heads", C.c_uint64),
                ("num_key_value_heads", C.c_uint64), ("head_dim", C.c_uint64), ("max_position_embeddings", C.c_uint64),
                ("rms_norm_eps", C.c_float), ("rope_theta", C.c_double), ("tie_word_embeddings", C.c_int32),
                ("init_std", C.c_float), ("seed", C.c_uint64), ("qk_norm", C.c_int32), ("use_bias", C.c_int32)]


class EngineStatsC(C.Structure):
    pass                                     # fields set below (needs SchedStatsC)


class HealthStatusC(C.Structure):
    _fields_ = [("is_healthy", C.c_int32), ("memory_pressure", C.c_double), ("active_sequences", C.c_uint64),
                ("waiting_sequences", C.c_uint64)]


class SequenceOutputC(C.Structure):                          # nvr_sequence_output (SequenceOutput, sequence.rs:30-47)
    _fields_ = [("seq_id", C.c_uint64), ("text", C.c_void_p), ("text_len", C.c_size_t), ("token_ids", C.c_void_p),
                ("num_tokens", C.c_size_t), ("completion_token_ids", C.c_void_p), ("num_prompt_tokens", C.c_size_t),
                ("num_completion_tokens", C.c_size_t), ("status", C.c_int32)]


STREAM_FN = C.CFUNCTYPE(C.c_int, C.POINTER(SequenceOutputC), C.c_void_p)


class BmStatsC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_blocks", "free_blocks", "used_blocks", "cached_blocks", "block_size")]


class BlockInfoC(C.Structure):
    _fields_ = [("block_id", C.c_uint64), ("ref_count", C.c_uint64), ("has_hash", C.c_int32), ("hash", C.c_uint64),
                ("num_tokens", C.c_uint64)]


class SchedStatsC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_sequences", "waiting_sequences", "running_sequences",
                                          "finished_sequences", "preemptions", "prefill_batches", "decode_batches")] + \
               [("avg_prefill_batch_size", C.c_double), ("avg_decode_batch_size", C.c_double)]


EngineStatsC._fields_ = [("scheduler", SchedStatsC), ("total_blocks", C.c_uint64), ("free_blocks", C.c_uint64),
                         ("used_blocks", C.c_uint64), ("utilization", C.c_double), ("is_running", C.c_int32)]


class StepInfoC(C.Structure):
    _fields_ = [("is_prefill", C.c_int32), ("num_seqs", C.c_uint64), ("num_tokens", C.c_uint64),
                ("num_finished", C.c_uint64)]


class AttnMetaC(C.Structure):
    _fields_ = [("is_prefill", C.c_int32), ("cu_seqlens_q", C.c_void_p), ("cu_seqlens_k", C.c_void_p),
                ("max_seqlen_q", C.c_int32), ("max_seqlen_k", C.c_int32), ("slot_mapping", C.c_void_p),
                ("context_lens", C.c_void_p), ("block_tables", C.c_void_p), ("max_blocks", C.c_int32),
                ("batch", C.c_int32), ("max_context_len", C.c_int32)]


_lib: Optional[C.CDLL] = None
_P = C.c_void_p
_SIGS = {
    # name: (restype, argtypes)
    "nvr_last_error": (C.c_char_p, []), "nvr_last_status": (C.c_int, []), "nvr_version": (C.c_char_p, []),
    "nvr_sampling_params_default": (None, [C.POINTER(SamplingParamsC)]),
    "nvr_sampling_params_validate": (C.c_int, [C.POINTER(SamplingParamsC)]),
    "nvr_config_default": (None, [C.POINTER(ConfigC)]), "nvr_config_validate": (C.c_int, [C.POINTER(ConfigC)]),
    "nvr_model_config_default": (None, [C.POINTER(ModelConfigC)]),
    "nvr_model_config_qwen3_0_6b": (None, [C.POINTER(ModelConfigC)]),
    "nvr_model_config_qwen3_8b": (None, [C.POINTER(ModelConfigC)]),
    "nvr_model_config_validate": (C.c_int, [C.POINTER(ModelConfigC), C.c_uint64]),
    "nvr_seq_create": (_P, [_P, C.c_size_t, C.POINTER(SamplingParamsC), C.c_size_t]),
    "nvr_seq_destroy": (None, [_P]), "nvr_seq_reset_id_counter": (None, []),
    "nvr_seq_id": (C.c_uint64, [_P]), "nvr_seq_status": (C.c_int32, [_P]), "nvr_seq_len": (C.c_size_t, [_P]),
    "nvr_seq_num_prompt_tokens": (C.c_size_t, [_P]), "nvr_seq_num_completion_tokens": (C.c_size_t, [_P]),
    "nvr_seq_num_cached_tokens": (C.c_size_t, [_P]), "nvr_seq_last_token": (C.c_int64, [_P]),
    "nvr_seq_num_computed_tokens": (C.c_size_t, [_P]), "nvr_seq_chunk": (None, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "nvr_seq_num_blocks": (C.c_size_t, [_P]), "nvr_seq_last_block_num_tokens": (C.c_size_t, [_P]),
    "nvr_seq_token_ids": (None, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "nvr_seq_block_table": (None, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "nvr_seq_append_token": (None, [_P, C.c_int64]), "nvr_seq_should_stop": (C.c_int, [_P, C.c_int, C.c_int64]),
    "nvr_seq_preempt": (None, [_P]), "nvr_seq_finish": (None, [_P]),
    "nvr_bm_create": (_P, [C.c_size_t, C.c_size_t]), "nvr_bm_destroy": (None, [_P]),
    "nvr_bm_compute_hash": (C.c_uint64, [_P, C.c_size_t, C.c_int, C.c_uint64]),
    "nvr_bm_can_allocate": (C.c_int, [_P, _P]), "nvr_bm_allocate": (C.c_int, [_P, _P]),
    "nvr_bm_deallocate": (C.c_int, [_P, _P]), "nvr_bm_can_append": (C.c_int, [_P, _P]),
    "nvr_bm_may_append": (C.c_int, [_P, _P]), "nvr_bm_get_stats": (C.c_int, [_P, C.POINTER(BmStatsC)]),
    "nvr_bm_get_block": (C.c_int, [_P, C.c_size_t, C.POINTER(BlockInfoC)]),
    "nvr_bm_free_list": (C.c_size_t, [_P, _P, C.c_size_t]),
    "nvr_sched_create": (_P, [C.POINTER(ConfigC)]), "nvr_sched_destroy": (None, [_P]),
    "nvr_sched_add_sequence": (C.c_int, [_P, _P]),
    "nvr_sched_schedule": (C.c_int, [_P, C.POINTER(_P), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "nvr_sched_postprocess": (C.c_int, [_P, C.POINTER(_P), _P, C.c_size_t]),
    "nvr_sched_is_finished": (C.c_int, [_P]), "nvr_sched_preempt_all": (None, [_P]),
    "nvr_sched_get_stats": (C.c_int, [_P, C.POINTER(SchedStatsC)]),
    "nvr_sched_get_block_stats": (C.c_int, [_P, C.POINTER(BmStatsC)]),
    "nvr_sched_queue_lengths": (None, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "nvr_sched_memory_pressure": (C.c_double, [_P]), "nvr_sched_block_manager": (_P, [_P]),
    "nvr_sched_take_finished": (C.c_size_t, [_P, C.POINTER(_P), C.c_size_t]),
    "nvr_runner_create": (_P, [C.POINTER(ConfigC), C.POINTER(ModelConfigC)]), "nvr_runner_destroy": (None, [_P]),
    "nvr_runner_execute_model": (C.c_int, [_P, C.POINTER(_P), C.c_size_t, C.c_int, C.POINTER(_P)]),
    "nvr_runner_sample_tokens": (C.c_int, [_P, C.POINTER(_P), C.c_size_t, _P]),
    "nvr_runner_load_tensor": (C.c_int, [_P, C.c_char_p, C.c_int, _P, C.c_int, _P]),
    "nvr_runner_copy_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t, _P, _P]),
    "nvr_runner_copy_logits": (C.c_int, [_P, _P, C.c_size_t]),
    "nvr_runner_num_kvcache_blocks": (C.c_uint64, [_P]),
    "nvr_runner_kv_cache": (C.c_int, [_P, C.c_size_t, C.POINTER(_P), C.POINTER(_P)]),
    "nvr_runner_stream": (_P, [_P]), "nvr_comm_unique_id": (C.c_int, [_P]),
    "nvr_runner_init_comm": (C.c_int, [_P, _P]), "nvr_runner_comm_selftest": (C.c_int, [_P]),
    "nvr_engine_create": (_P, [C.POINTER(ConfigC), C.POINTER(ModelConfigC)]), "nvr_engine_destroy": (None, [_P]),
    "nvr_engine_add_request": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(SamplingParamsC), C.POINTER(C.c_uint64)]),
    "nvr_engine_step": (C.c_int, [_P, C.POINTER(StepInfoC)]), "nvr_engine_is_finished": (C.c_int, [_P]),
    "nvr_engine_scheduler": (_P, [_P]), "nvr_engine_runner": (_P, [_P]),
    "nvr_engine_get_stats": (C.c_int, [_P, _P]), "nvr_engine_health_check": (C.c_int, [_P, _P]), "nvr_engine_shutdown": (C.c_int, [_P]),
    "nvr_tokenize": (C.c_int, [C.c_char_p, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "nvr_detokenize": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "nvr_engine_add_prompt": (C.c_int, [_P, C.c_char_p, C.c_size_t, _P, C.POINTER(C.c_uint64)]),
    "nvr_engine_generate": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "nvr_engine_generate_ids": (C.c_int, [_P, _P, _P, C.c_size_t, _P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "nvr_engine_generate_stream": (C.c_int, [_P, _P, _P, C.c_size_t, _P, STREAM_FN, _P]),
    "nvr_runner_replay_last_decode_graph": (C.c_int, [_P, C.c_int]),
    "nvr_local_group_create": (_P, [C.c_int]), "nvr_local_group_destroy": (None, [_P]),
    "nvr_runner_init_comm_local": (C.c_int, [_P, _P]), "nvr_local_group_set_p2p": (C.c_int, [_P, C.c_int]),
    "nvr_runner_p2p_export": (C.c_int, [_P, _P]), "nvr_runner_p2p_attach": (C.c_int, [_P, _P, _P]),
    "nvr_runner_p2p_disable": (C.c_int, [_P]), "nvr_runner_p2p_active": (C.c_int, [_P]),
    "nvr_runner_p2p_set_fenced": (C.c_int, [_P, C.c_int32]), "nvr_runner_p2p_fenced": (C.c_int, [_P]),
    "nvr_runner_p2p_reset": (C.c_int, [_P]), "nvr_runner_comm_drop_rccl": (C.c_int, [_P]),
    "nvr_engine_abort_last_batch": (C.c_int, [_P]), "nvr_engine_ahead_declined": (C.c_uint64, [_P]), "nvr_engine_ahead_launched": (C.c_uint64, [_P]),
    "nvr_engine_host_times": (None, [_P, _P]),
    "nvr_runner_last_prefill_kv_source": (C.c_int, [_P]),
    "nvr_runner_set_tp_prefill_overlap": (C.c_int, [_P, C.c_int32]), "nvr_runner_last_overlap_chunks": (C.c_int64, [_P]),
    "nvr_runner_last_decode_ragged": (C.c_int32, [_P]),
    "nvr_runner_last_shared_prefix_len": (C.c_int64, [_P]), "nvr_runner_last_shared_prefix_rows": (C.c_int64, [_P]),
    "nvr_engine_last_step": (None, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "nvr_engine_take_finished": (C.c_size_t, [_P, C.POINTER(_P), C.c_size_t]),
    "nvr_engine_last_batch": (C.c_size_t, [_P, C.POINTER(_P), C.c_size_t]),
    "nvr_device_count": (C.c_int, [C.POINTER(C.c_int)]), "nvr_device_set": (C.c_int, [C.c_int]),
    "nvr_device_name": (C.c_int, [C.c_char_p, C.c_size_t]),
    "nvr_device_mem_info": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nvr_device_malloc": (C.c_int, [C.POINTER(_P), C.c_size_t]), "nvr_device_free": (C.c_int, [_P]),
    "nvr_device_memset": (C.c_int, [_P, C.c_int, C.c_size_t]),
    "nvr_memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t]), "nvr_memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t]),
    "nvr_device_synchronize": (C.c_int, []),
    "nvr_stream_create": (C.c_int, [C.POINTER(_P)]), "nvr_stream_destroy": (C.c_int, [_P]),
    "nvr_stream_synchronize": (C.c_int, [_P]),
    "nvr_graph_capture_begin": (C.c_int, [_P]), "nvr_graph_capture_end": (C.c_int, [_P, C.POINTER(_P)]),
    "nvr_graph_launch": (C.c_int, [_P, _P]), "nvr_graph_destroy": (C.c_int, [_P]),
    "nvr_event_create": (C.c_int, [C.POINTER(_P)]), "nvr_event_destroy": (C.c_int, [_P]),
    "nvr_event_record": (C.c_int, [_P, _P]), "nvr_stream_wait_event": (C.c_int, [_P, _P]), "nvr_event_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "nvr_embedding": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P]),
    "nvr_rmsnorm": (C.c_int, [_P, _P, C.c_float, C.c_int64, C.c_int64, _P, _P]),
    "nvr_add_rmsnorm": (C.c_int, [_P, _P, _P, C.c_float, C.c_int64, C.c_int64, _P, _P]),
    "nvr_linear": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int, _P]),
    "nvr_lm_head": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    "nvr_argmax_partials": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, _P, C.c_int64, _P]),
    "nvr_linear_splitk": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "nvr_linear_tiled": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int64, C.c_int64, _P, C.c_int, _P]),
    "nvr_linear_splitk_tiled": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "nvr_linear_silu_mul_tiled": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "nvr_linear_qkv_rope_store_tiled": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P,
                                                  _P, _P, _P, _P]),
    "nvr_lm_head_tiled": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    "nvr_retile_weight": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, _P]),
    "nvr_decode_splitk_slices": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "nvr_linear_add_residual": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "nvr_add_rmsnorm_slabs": (C.c_int, [_P, _P, C.c_int64, _P, C.c_float, C.c_int64, C.c_int64, _P, _P]),
    "nvr_linear_silu_mul": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "nvr_linear_qkv_rope_store": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P,
                                            _P, _P, _P, _P]),
    "nvr_rope_store_kv": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    "nvr_qk_norm_rope_store_kv": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P, _P, _P, C.c_float, _P, _P, _P]),
    "nvr_rope_table": (C.c_int, [C.c_int64, C.c_int64, C.c_double, _P, _P]),
    "nvr_paged_attn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "nvr_paged_attn_decode": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(AttnMetaC), C.c_int64, C.c_int64, C.c_int64,
                                        C.c_int64, C.c_float, _P, _P, _P]),
    "nvr_paged_attn_decode_fused": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(AttnMetaC), C.c_int64, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_float, _P, _P, _P, _P]),
    "nvr_paged_attn_decode_shared": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(AttnMetaC), C.c_int64, C.c_int64, C.c_int64,
                                               C.c_int64, C.c_float, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "nvr_attn_prefill_varlen": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(AttnMetaC), C.c_int64, C.c_int64, C.c_int64,
                                          C.c_int64, C.c_float, _P, _P]),
    "nvr_attn_prefill_paged": (C.c_int, [_P, C.c_int64, _P, _P, C.POINTER(AttnMetaC), C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int64, C.c_float, _P, _P]),
    "nvr_silu_and_mul": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "nvr_activation": (C.c_int, [C.c_int32, _P, C.c_int64, C.c_int64, _P, _P]), "nvr_activation_type_from_str": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "nvr_add_bias": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "nvr_select_last_tokens": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P]),
    "nvr_argmax": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "nvr_sample_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "nvr_sample": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "nvr_sample_key": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "nvr_weight_key": (C.c_uint64, [C.c_uint64, C.c_uint64]), "nvr_weight_scale": (C.c_float, [C.c_double]),
    "nvr_fill_weight": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                  C.c_float, _P]),
    "nvr_fill_const": (C.c_int, [_P, C.c_int64, C.c_float, _P]),
    "nvr_ops_set_dtype": (C.c_int, [C.c_char_p]), "nvr_ops_dtype": (C.c_char_p, []),
}


def lib() -> C.CDLL:
    """Load libnvr.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(there is no CPU fallback for the hot path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def last_error() -> str:
    return lib().nvr_last_error().decode()


def check(rc: int) -> None:
    if rc != 0:
        raise NvrError(rc, last_error())


# ---------------------------------------------------------------------------------- host mirror
class SamplingParams:
    """SamplingParams, reference src/engine/sampling_params.rs:10-119."""

    def __init__(self, temperature: float = 1.0, max_tokens: int = 64, ignore_eos: bool = False,
                 top_p: Optional[float] = None, top_k: Optional[int] = None,
                 repetition_penalty: Optional[float] = None):
        self.temperature, self.max_tokens, self.ignore_eos = temperature, max_tokens, ignore_eos
        self.top_p, self.top_k, self.repetition_penalty = top_p, top_k, repetition_penalty

    def is_greedy(self) -> bool:
        return self.temperature == 0.0

    def to_c(self) -> SamplingParamsC:
        c = SamplingParamsC()
        c.temperature, c.max_tokens, c.ignore_eos = self.temperature, self.max_tokens, int(self.ignore_eos)
        c.has_top_p, c.top_p = int(self.top_p is not None), self.top_p or 0.0
        c.has_top_k, c.top_k = int(self.top_k is not None), self.top_k or 0
        c.has_repetition_penalty = int(self.repetition_penalty is not None)
        c.repetition_penalty = self.repetition_penalty or 0.0
        return c

    def validate(self) -> None:
        check(lib().nvr_sampling_params_validate(C.byref(self.to_c())))


class Config:
    """Config, reference src/config.rs:16-186 (fields of the hot path; model_path omitted)."""

    def __init__(self, **kw):
        c = ConfigC()
        lib().nvr_config_default(C.byref(c))
        self.c = c
        eos = kw.pop("eos_token_id", None)
        if eos is not None:
            c.has_eos, c.eos_token_id = 1, eos
        nb = kw.pop("num_kvcache_blocks", None)
        c.num_kvcache_blocks = -1 if nb is None else (-2 if nb == "auto" else nb)
        for k, v in kw.items():
            if not hasattr(c, k):
                raise AttributeError(f"Config has no field {k}")
            setattr(c, k, int(v) if isinstance(v, bool) else (v.encode() if isinstance(v, str) else v))

    def validate(self) -> None:
        check(lib().nvr_config_validate(C.byref(self.c)))


class ModelConfig:
    """Qwen3Config, reference src/models/qwen3.rs:26-125 (+ head_dim override)."""

    def __init__(self, preset: Optional[str] = None, **kw):
        m = ModelConfigC()
        {None: lib().nvr_model_config_default, "qwen3-0.6b": lib().nvr_model_config_qwen3_0_6b,
         "qwen3-8b": lib().nvr_model_config_qwen3_8b}[preset](C.byref(m))
        for k, v in kw.items():
            if not hasattr(m, k):
                raise AttributeError(f"ModelConfig has no field {k}")
            setattr(m, k, (v or 0) if k == "head_dim" else (int(v) if isinstance(v, bool) else v))
        self.c = m

    def validate(self, tp: int = 1) -> None:
        check(lib().nvr_model_config_validate(C.byref(self.c), tp))

    def head_dim(self) -> int:
        return self.c.head_dim or self.c.hidden_size // self.c.num_attention_heads


def _read_array(fn, handle, dtype):
    p, n = _P(), C.c_size_t()
    fn(handle, C.byref(p), C.byref(n))
    if n.value == 0:
        return []
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), (n.value,)).tolist()


class Sequence:
    """Sequence, reference src/engine/sequence.rs:50-237 (handle wrapper)."""

    def __init__(self, prompt_token_ids: Seq[int] = (), sampling_params: Optional[SamplingParams] = None,
                 block_size: int = 256, _handle=None):
        if _handle is not None:
            self.h, self.owned = _handle, False
            return
        arr = np.ascontiguousarray(prompt_token_ids, dtype=np.int64)
        sp = (sampling_params or SamplingParams()).to_c()
        self.h = lib().nvr_seq_create(arr.ctypes.data, arr.size, C.byref(sp), block_size)
        if not self.h:
            raise NvrError(-6, last_error())
        self.owned = True

    def __del__(self):
        if getattr(self, "owned", False) and self.h and _lib is not None:
            _lib.nvr_seq_destroy(self.h)
            self.h = None

    seq_id = property(lambda s: lib().nvr_seq_id(s.h))
    status = property(lambda s: lib().nvr_seq_status(s.h))
    num_prompt_tokens = property(lambda s: lib().nvr_seq_num_prompt_tokens(s.h))
    num_cached_tokens = property(lambda s: lib().nvr_seq_num_cached_tokens(s.h))
    num_computed_tokens = property(lambda s: lib().nvr_seq_num_computed_tokens(s.h))

    @property
    def chunk(self) -> Tuple[int, int]:
        """(start, length) of the token range of the step this sequence was last scheduled into (chunked prefill, A-23)"""
        a, n = C.c_size_t(), C.c_size_t()
        lib().nvr_seq_chunk(self.h, C.byref(a), C.byref(n))
        return a.value, n.value

    last_token = property(lambda s: lib().nvr_seq_last_token(s.h))
    token_ids = property(lambda s: _read_array(lib().nvr_seq_token_ids, s.h, np.int64))
    block_table = property(lambda s: _read_array(lib().nvr_seq_block_table, s.h, np.int32))

    def __len__(self) -> int:
        return lib().nvr_seq_len(self.h)

    def num_completion_tokens(self) -> int:
        return lib().nvr_seq_num_completion_tokens(self.h)

    def completion_token_ids(self) -> List[int]:
        return self.token_ids[self.num_prompt_tokens:]

    def num_blocks(self) -> int:
        return lib().nvr_seq_num_blocks(self.h)

    def last_block_num_tokens(self) -> int:
        return lib().nvr_seq_last_block_num_tokens(self.h)

    def append_token(self, t: int) -> None:
        lib().nvr_seq_append_token(self.h, t)

    def should_stop(self, eos: Optional[int]) -> bool:
        return bool(lib().nvr_seq_should_stop(self.h, int(eos is not None), eos or 0))


WAITING, RUNNING, FINISHED, PREEMPTED, ERROR = range(5)


class BlockManager:
    """BlockManager, reference src/engine/block_manager.rs:69-361."""

    def __init__(self, num_blocks: int, block_size: int, _handle=None):
        if _handle is not None:
            self.h, self.owned = _handle, False
            return
        self.h = lib().nvr_bm_create(num_blocks, block_size)
        if not self.h:
            raise NvrError(-6, last_error())
        self.owned = True

    def __del__(self):
        if getattr(self, "owned", False) and self.h and _lib is not None:
            _lib.nvr_bm_destroy(self.h)
            self.h = None

    @staticmethod
    def compute_hash(token_ids: Seq[int], prefix_hash: Optional[int] = None) -> int:
        a = np.ascontiguousarray(token_ids, dtype=np.int64)
        return lib().nvr_bm_compute_hash(a.ctypes.data, a.size, int(prefix_hash is not None), prefix_hash or 0)

    def can_allocate(self, s: Sequence) -> bool:
        return bool(lib().nvr_bm_can_allocate(self.h, s.h))

    def allocate(self, s: Sequence) -> None:
        check(lib().nvr_bm_allocate(self.h, s.h))

    def deallocate(self, s: Sequence) -> None:
        check(lib().nvr_bm_deallocate(self.h, s.h))

    def can_append(self, s: Sequence) -> bool:
        return bool(lib().nvr_bm_can_append(self.h, s.h))

    def may_append(self, s: Sequence) -> None:
        check(lib().nvr_bm_may_append(self.h, s.h))

    def get_stats(self) -> dict:
        st = BmStatsC()
        check(lib().nvr_bm_get_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in BmStatsC._fields_}

    def get_block(self, block_id: int) -> dict:
        b = BlockInfoC()
        check(lib().nvr_bm_get_block(self.h, block_id, C.byref(b)))
        return dict(block_id=b.block_id, ref_count=b.ref_count, hash=b.hash if b.has_hash else None,
                    num_tokens=b.num_tokens)

    def free_list(self) -> List[int]:
        n = self.get_stats()["total_blocks"]
        out = np.empty(n, dtype=np.int32)
        cnt = lib().nvr_bm_free_list(self.h, out.ctypes.data, n)
        return out[:cnt].tolist()


class Scheduler:
    """Scheduler, reference src/engine/scheduler.rs:13-365."""

    def __init__(self, config: Config, _handle=None):
        if _handle is not None:
            self.h, self.owned = _handle, False
        else:
            self.h = lib().nvr_sched_create(C.byref(config.c))
            if not self.h:
                raise NvrError(-6, last_error())
            self.owned = True
        self._cap = 4096

    def __del__(self):
        if getattr(self, "owned", False) and self.h and _lib is not None:
            _lib.nvr_sched_destroy(self.h)
            self.h = None

    def add_sequence(self, s: Sequence) -> None:
        check(lib().nvr_sched_add_sequence(self.h, s.h))
        s.owned = False                       # ownership moved (scheduler.rs:93 takes the Sequence by value)

    def schedule(self) -> Tuple[List[Sequence], bool]:
        out = (_P * self._cap)()
        n, pf = C.c_size_t(), C.c_int()
        check(lib().nvr_sched_schedule(self.h, out, self._cap, C.byref(n), C.byref(pf)))
        return [Sequence(_handle=out[i]) for i in range(n.value)], bool(pf.value)

    def postprocess(self, seqs: List[Sequence], token_ids: Seq[int]) -> None:
        if len(seqs) != len(token_ids):       # scheduler.rs:235-237
            raise NvrError(-4, "Mismatch between sequences and token_ids length")
        hs = (_P * len(seqs))(*[s.h for s in seqs])
        t = np.ascontiguousarray(token_ids, dtype=np.int64)
        check(lib().nvr_sched_postprocess(self.h, hs, t.ctypes.data, len(seqs)))

    def is_finished(self) -> bool:
        return bool(lib().nvr_sched_is_finished(self.h))

    def preempt_all(self) -> None:
        lib().nvr_sched_preempt_all(self.h)

    def get_stats(self) -> dict:
        st = SchedStatsC()
        check(lib().nvr_sched_get_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in SchedStatsC._fields_}

    def get_block_stats(self) -> dict:
        st = BmStatsC()
        check(lib().nvr_sched_get_block_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in BmStatsC._fields_}

    def get_queue_lengths(self) -> Tuple[int, int]:
        w, r = C.c_size_t(), C.c_size_t()
        lib().nvr_sched_queue_lengths(self.h, C.byref(w), C.byref(r))
        return w.value, r.value

    def memory_pressure(self) -> float:
        return lib().nvr_sched_memory_pressure(self.h)

    @property
    def block_manager(self) -> BlockManager:
        return BlockManager(0, 0, _handle=lib().nvr_sched_block_manager(self.h))

    def take_finished(self) -> List[Sequence]:
        out = (_P * self._cap)()
        n = lib().nvr_sched_take_finished(self.h, out, self._cap)
        res = []
        for i in range(n):
            s = Sequence(_handle=out[i])
            s.owned = True                    # caller destroys
            res.append(s)
        return res


def iter_safetensors(fn: str):
    """(name, array) for every tensor of one .safetensors file, read straight from the file: 8-byte little-endian header
    length, JSON header {name: {dtype, shape, data_offsets}}, raw little-endian payload.  F16 -> float16, F32 -> float32,
    BF16 -> uint16 bit patterns (numpy has no bfloat16; load_tensor passes them down as dtype 1 and the device side
    converts).  The safetensors package's numpy front end cannot return bf16 at all."""
    import json
    kinds = {"F16": (np.float16, 2), "BF16": (np.uint16, 2), "F32": (np.float32, 4)}
    with open(fn, "rb") as f:
        hlen = int.from_bytes(f.read(8), "little")
        header = json.loads(f.read(hlen).decode("utf-8"))
    data = np.memmap(fn, dtype=np.uint8, mode="r", offset=8 + hlen)
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        if meta["dtype"] not in kinds:
            raise NvrError(-10, f"safetensors: {name} has dtype {meta['dtype']} (F16, BF16 and F32 are supported)")
        dt, isz = kinds[meta["dtype"]]
        b, e = meta["data_offsets"]
        n = int(np.prod(meta["shape"], dtype=np.int64)) if meta["shape"] else 1
        if e - b != n * isz or e > data.size:
            raise NvrError(-4, f"safetensors: {name}: {e - b} bytes for shape {meta['shape']} of {meta['dtype']}")
        yield name, np.frombuffer(data, dtype=dt, count=n, offset=b).reshape(meta["shape"])


class ModelRunner:
    """ModelRunner, reference src/engine/model_runner.rs:19-464."""

    def __init__(self, config: Config, model_config: ModelConfig, _handle=None):
        self.vocab_local = model_config.c.vocab_size // max(1, config.c.tensor_parallel_size) if _handle is None else 0
        self.bf16 = bytes(config.c.dtype).split(b"\0")[0] == b"bfloat16"    # the runner's 16-bit type (Config.dtype, config.rs:51)
        self.f32 = bytes(config.c.dtype).split(b"\0")[0] == b"float32"      # the reference-precision path: 4-byte storage
        if _handle is not None:
            self.h, self.owned = _handle, False
            return
        self.h = lib().nvr_runner_create(C.byref(config.c), C.byref(model_config.c))
        if not self.h:
            raise NvrError(lib().nvr_last_status() or -8, last_error())
        self.owned = True

    def __del__(self):
        if getattr(self, "owned", False) and self.h and _lib is not None:
            _lib.nvr_runner_destroy(self.h)
            self.h = None

    def execute_model(self, seqs: List[Sequence], is_prefill: bool) -> int:
        hs = (_P * len(seqs))(*[s.h for s in seqs])
        p = _P()
        check(lib().nvr_runner_execute_model(self.h, hs, len(seqs), int(is_prefill), C.byref(p)))
        return p.value

    def sample_tokens(self, seqs: List[Sequence]) -> List[int]:
        hs = (_P * len(seqs))(*[s.h for s in seqs])
        out = np.empty(len(seqs), dtype=np.int64)
        check(lib().nvr_runner_sample_tokens(self.h, hs, len(seqs), out.ctypes.data))
        return out.tolist()

    def logits(self, rows: int, vocab: Optional[int] = None) -> np.ndarray:
        v = vocab or self.vocab_local
        out = np.empty((rows, v), dtype=np.float32)
        check(lib().nvr_runner_copy_logits(self.h, out.ctypes.data, rows))
        return out

    # -- weight loading (Qwen3Model::load_weights qwen3.rs:518-570, utils/loader.rs) --------------------
    def load_tensor(self, name: str, array: np.ndarray) -> None:
        """One full (un-sharded) checkpoint tensor by name; this rank's slice lands in the packed device parameter."""
        a = np.ascontiguousarray(array)
        if a.dtype == np.float16:
            dt = 0
        elif a.dtype == np.float32:
            dt = 2
        elif a.dtype == np.uint16:          # bfloat16 bits (numpy has no bf16)
            dt = 1
        else:
            a, dt = a.astype(np.float32), 2
        shape = (C.c_int64 * a.ndim)(*a.shape)
        check(lib().nvr_runner_load_tensor(self.h, name.encode(), dt, shape, a.ndim, a.ctypes.data))

    def load_safetensors(self, path: str, strict: bool = False) -> List[str]:
        """Every tensor of a .safetensors file (or of all such files in a directory); bf16 — the dtype of the published
        Qwen3 checkpoints — goes down as its 16-bit patterns and is converted on load (iter_safetensors).  Returns the names
        that are not part of the graph (biases unless ModelConfig(use_bias=1), A-30; q_norm / k_norm unless ModelConfig(qk_norm=1), A-27) and warns about them (a
        real Qwen3 checkpoint loaded without qk_norm runs WITHOUT its attention norms, SURVEY A-17); strict=True raises."""
        import warnings
        files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith(".safetensors")) if os.path.isdir(path) else [path]
        skipped = []
        for fn in files:
            for name, arr in iter_safetensors(fn):
                try:
                    self.load_tensor(name, arr)
                except NvrError as ex:
                    if ex.code != -10 or strict:
                        raise
                    skipped.append(name)
        if skipped:
            warnings.warn(f"load_safetensors: {len(skipped)} tensors are outside the reference's Qwen3 graph and were NOT loaded "
                          f"(e.g. {skipped[0]}): biases only with ModelConfig(use_bias=1), q/k-norm only with ModelConfig(qk_norm=1) (SURVEY A-17)", RuntimeWarning, stacklevel=2)
        return skipped

    def weight(self, local_name: str) -> np.ndarray:
        """A local packed tensor: "embed", "lm_head", "norm", "layers.N.{qkv,o,gate_up,down,ln1,ln2}" (+ q_norm / k_norm with qk_norm, qkv_b / gate_up_b and, on rank 0, o_b / down_b with use_bias) — as fp16, or, from a bfloat16
        runner (numpy has no bf16), as the f32 values of its bf16 elements (exact)."""
        r, c = C.c_int64(), C.c_int64()
        check(lib().nvr_runner_copy_weight(self.h, local_name.encode(), None, 0, C.byref(r), C.byref(c)))
        if self.f32:
            out = np.empty((r.value, c.value), np.float32)
            check(lib().nvr_runner_copy_weight(self.h, local_name.encode(), out.ctypes.data, out.size * 2, C.byref(r), C.byref(c)))
            return out[:, 0] if c.value == 1 else out
        out = np.empty((r.value, c.value), np.uint16 if self.bf16 else np.float16)
        check(lib().nvr_runner_copy_weight(self.h, local_name.encode(), out.ctypes.data, out.size, C.byref(r), C.byref(c)))
        if self.bf16:
            out = (out.astype(np.uint32) << 16).view(np.float32)
        return out[:, 0] if c.value == 1 else out

    def num_kvcache_blocks(self) -> int:
        return lib().nvr_runner_num_kvcache_blocks(self.h)

    def kv_cache(self, layer: int) -> Tuple[int, int]:
        k, v = _P(), _P()
        check(lib().nvr_runner_kv_cache(self.h, layer, C.byref(k), C.byref(v)))
        return k.value, v.value

    def comm_selftest(self) -> None:
        check(lib().nvr_runner_comm_selftest(self.h))

    def init_comm(self, unique_id: bytes) -> None:
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().nvr_runner_init_comm(self.h, buf))

    # one-shot peer-to-peer collectives (kernels/comm_p2p.hip): export my arena's hipIpc handle, attach everybody's
    def p2p_export(self) -> bytes:
        buf = (C.c_uint8 * 64)()
        check(lib().nvr_runner_p2p_export(self.h, buf))
        return bytes(buf)

    def p2p_attach(self, handles: Seq[bytes], devices: Optional[Seq[int]] = None) -> None:
        blob = (C.c_uint8 * (64 * len(handles))).from_buffer_copy(b"".join(handles))
        dv = (C.c_int32 * len(handles))(*devices) if devices is not None else None
        check(lib().nvr_runner_p2p_attach(self.h, blob, dv))

    def p2p_disable(self) -> None:
        check(lib().nvr_runner_p2p_disable(self.h))

    def p2p_active(self) -> bool:
        return bool(lib().nvr_runner_p2p_active(self.h))

    def p2p_set_fenced(self, on: bool) -> None:
        """protocol of the one-shot collectives: False = fence-free (default), True = r04's release / acquire fences; every rank alike"""
        check(lib().nvr_runner_p2p_set_fenced(self.h, 1 if on else 0))

    def p2p_fenced(self) -> bool:
        return bool(lib().nvr_runner_p2p_fenced(self.h))

    def p2p_reset(self) -> None:
        """After a collective timed out (NVR_ERR_RCCL): epochs and arrival flags back to their initial state; every rank calls it,
        then the control plane barriers."""
        check(lib().nvr_runner_p2p_reset(self.h))

    def comm_drop_rccl(self) -> None:
        check(lib().nvr_runner_comm_drop_rccl(self.h))

    def last_prefill_kv_source(self) -> int:
        """0: the last prefill's attention read K/V from the qkv buffer, 1: from contiguous cache rows, 2: through the block tables; -1: decode."""
        return int(lib().nvr_runner_last_prefill_kv_source(self.h))

    def set_tp_prefill_overlap(self, on: bool) -> None:
        """Tensor-parallel prefill exchange: 1 / True (default) = all-reduce of token chunk i on a second stream under the GEMM of chunk i + 1;
        2 = two micro-batches of whole sequences, each exchange under the other one's compute; 0 / False = serial on one stream."""
        check(lib().nvr_runner_set_tp_prefill_overlap(self.h, 2 if on == 2 else 1 if on else 0))

    def last_overlap_chunks(self) -> int:
        return int(lib().nvr_runner_last_overlap_chunks(self.h))

    def last_decode_ragged(self) -> bool:
        """The last decode step took the work-balanced attention launch for its ragged contexts (nvr_runner_last_decode_ragged)."""
        return bool(lib().nvr_runner_last_decode_ragged(self.h))

    def last_shared_prefix_len(self) -> int:
        """Tokens of the last decode step that went through the shared-prefix attention pass (0: plain paged attention)."""
        return int(lib().nvr_runner_last_shared_prefix_len(self.h))

    def last_shared_prefix_rows(self) -> int:
        """How many sequences of the last decode step were inside the sharing group (0: no shared pass)."""
        return int(lib().nvr_runner_last_shared_prefix_rows(self.h))


class LocalGroup:
    """In-process communicator (nvr_local_group_*): tensor-parallel ranks as runners of this process on one GPU,
    one host thread per rank.  Tests and bring-up only."""

    def __init__(self, nranks: int, p2p: bool = True):
        self.h = lib().nvr_local_group_create(nranks)
        if not self.h:
            raise NvrError(-7, last_error())
        self.nranks = nranks
        check(lib().nvr_local_group_set_p2p(self.h, 1 if p2p else 0))

    def attach(self, runner: "ModelRunner") -> None:
        check(lib().nvr_runner_init_comm_local(runner.h, self.h))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.nvr_local_group_destroy(self.h)
            self.h = None


def preload_rccl() -> None:
    """Load the ROCm RCCL of this image before anything else can bring its own copy under the same soname
    (torch bundles one).  Call before `import torch` in multi-rank programs."""
    lib()
    for name in ("/opt/rocm/lib/librccl.so.1", "librccl.so.1"):
        try:
            C.CDLL(name, mode=C.RTLD_GLOBAL)
            return
        except OSError:
            continue


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    check(lib().nvr_comm_unique_id(buf))
    return bytes(buf)


@dataclass
class SequenceOutput:
    """SequenceOutput, reference src/engine/sequence.rs:30-47."""
    seq_id: int
    text: str
    token_ids: List[int]
    completion_token_ids: List[int]
    num_prompt_tokens: int
    num_completion_tokens: int
    status: int

    @classmethod
    def from_c(cls, c: SequenceOutputC) -> "SequenceOutput":
        toks = np.ctypeslib.as_array(C.cast(c.token_ids, C.POINTER(C.c_int64)), (c.num_tokens,)).tolist() if c.num_tokens else []
        text = C.string_at(c.text, c.text_len).decode("utf-8") if c.text_len else ""
        return cls(c.seq_id, text, toks, toks[c.num_prompt_tokens:], c.num_prompt_tokens, c.num_completion_tokens, c.status)


def tokenize(text: str) -> List[int]:
    """LLMEngine::tokenize (llm_engine.rs:220-230): the reference's placeholder char tokenizer, through the C ABI."""
    b = text if isinstance(text, bytes) else text.encode("utf-8")
    out = (C.c_int64 * 100)()
    n = C.c_size_t()
    check(lib().nvr_tokenize(b, len(b), out, 100, C.byref(n)))
    return list(out[:n.value])


def detokenize(ids: Seq[int]) -> str:
    a = np.ascontiguousarray(ids, dtype=np.int64)
    n = C.c_size_t()
    check(lib().nvr_detokenize(a.ctypes.data, a.size, None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value + 1)
    check(lib().nvr_detokenize(a.ctypes.data, a.size, buf, n.value + 1, C.byref(n)))
    return buf.raw[:n.value].decode("utf-8")


def _text_args(prompts):
    enc = [p if isinstance(p, bytes) else p.encode("utf-8") for p in prompts]
    bufs = [C.create_string_buffer(b, len(b) + 1) for b in enc]
    ptrs = (_P * max(1, len(bufs)))(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * max(1, len(bufs)))(*[len(b) for b in enc])
    return ptrs, lens, bufs


class LLMEngine:
    """The hot loop of LLMEngine (reference src/engine/llm_engine.rs:155-197) over one ModelRunner."""

    def __init__(self, config: Config, model_config: ModelConfig):
        config.validate()
        self.h = lib().nvr_engine_create(C.byref(config.c), C.byref(model_config.c))
        if not self.h:
            raise NvrError(lib().nvr_last_status() or -8, last_error())
        self.config, self.model_config = config, model_config
        self.scheduler = Scheduler(config, _handle=lib().nvr_engine_scheduler(self.h))
        self.model_runner = ModelRunner(config, model_config, _handle=lib().nvr_engine_runner(self.h))
        self.model_runner.vocab_local = model_config.c.vocab_size // max(1, config.c.tensor_parallel_size)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.nvr_engine_destroy(self.h)
            self.h = None

    def add_request(self, prompt: Seq[int], sp: Optional[SamplingParams] = None) -> int:
        a = np.ascontiguousarray(prompt, dtype=np.int64)
        sid = C.c_uint64()
        spc = (sp or SamplingParams()).to_c()
        check(lib().nvr_engine_add_request(self.h, a.ctypes.data, a.size, C.byref(spc), C.byref(sid)))
        return sid.value

    def step(self) -> dict:
        info = StepInfoC()
        check(lib().nvr_engine_step(self.h, C.byref(info)))
        ids, toks, n = _P(), _P(), C.c_size_t()
        lib().nvr_engine_last_step(self.h, C.byref(ids), C.byref(toks), C.byref(n))
        k = n.value
        sid = np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_uint64)), (k,)).tolist() if k else []
        tk = np.ctypeslib.as_array(C.cast(toks, C.POINTER(C.c_int64)), (k,)).tolist() if k else []
        return dict(is_prefill=bool(info.is_prefill), num_seqs=info.num_seqs, num_tokens=info.num_tokens,
                    num_finished=info.num_finished, seq_ids=sid, tokens=tk)

    def is_finished(self) -> bool:
        return bool(lib().nvr_engine_is_finished(self.h))

    def get_stats(self) -> dict:                              # llm_engine.rs:312-327
        st = EngineStatsC()
        check(lib().nvr_engine_get_stats(self.h, C.byref(st)))
        sch = {n: getattr(st.scheduler, n) for n, _ in SchedStatsC._fields_}
        return dict(scheduler=sch, memory=dict(total_blocks=st.total_blocks, free_blocks=st.free_blocks, used_blocks=st.used_blocks,
                                               utilization=st.utilization), is_running=bool(st.is_running))

    def health_check(self) -> dict:                           # llm_engine.rs:330-342
        h = HealthStatusC()
        check(lib().nvr_engine_health_check(self.h, C.byref(h)))
        return dict(is_healthy=bool(h.is_healthy), memory_pressure=h.memory_pressure, active_sequences=h.active_sequences,
                    waiting_sequences=h.waiting_sequences)

    def shutdown(self) -> None:                               # llm_engine.rs:345-357
        check(lib().nvr_engine_shutdown(self.h))

    # ---- text in, SequenceOutput out (llm_engine.rs:70-128,200-230)
    def add_prompt(self, text: str, sp: Optional[SamplingParams] = None) -> int:
        b = text.encode("utf-8")
        sid = C.c_uint64()
        spc = (sp or SamplingParams()).to_c()
        check(lib().nvr_engine_add_prompt(self.h, b, len(b), C.byref(spc), C.byref(sid)))
        return sid.value

    def generate(self, prompts, sp: Optional[SamplingParams] = None) -> List["SequenceOutput"]:
        """LLMEngine::generate: prompts are strings (placeholder char tokenizer) or token-id lists; one
        SamplingParams for all; outputs in prompt order."""
        prompts = list(prompts)
        spc = (sp or SamplingParams()).to_c()
        outs, n = _P(), C.c_size_t()
        if prompts and not isinstance(prompts[0], (str, bytes)):
            arrs = [np.ascontiguousarray(p, dtype=np.int64) for p in prompts]
            ptrs = (_P * len(arrs))(*[a.ctypes.data for a in arrs])
            lens = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
            check(lib().nvr_engine_generate_ids(self.h, ptrs, lens, len(arrs), C.byref(spc), C.byref(outs), C.byref(n)))
        else:
            ptrs, lens, keep = _text_args(prompts)
            check(lib().nvr_engine_generate(self.h, ptrs, lens, len(prompts), C.byref(spc), C.byref(outs), C.byref(n)))
        if not n.value:
            return []
        arr = C.cast(outs, C.POINTER(SequenceOutputC * n.value)).contents
        return [SequenceOutput.from_c(arr[i]) for i in range(n.value)]

    def generate_stream(self, prompts: Seq[str], sp: Optional[SamplingParams] = None, on_output=None) -> List["SequenceOutput"]:
        """LLMEngine::generate_stream: on_output(SequenceOutput) is called for every sequence of every step's batch
        (cumulative text / tokens); returning a truthy value stops the stream (the dropped receiver).  Returns the
        list of everything delivered."""
        prompts = list(prompts)
        spc = (sp or SamplingParams()).to_c()
        got: List[SequenceOutput] = []
        err: List[BaseException] = []

        def _cb(po, _user):
            try:
                o = SequenceOutput.from_c(po.contents)
                got.append(o)
                return 1 if (on_output is not None and on_output(o)) else 0
            except BaseException as ex:                         # never unwind through the C frames
                err.append(ex)
                return 1
        ptrs, lens, keep = _text_args(prompts)
        check(lib().nvr_engine_generate_stream(self.h, ptrs, lens, len(prompts), C.byref(spc), STREAM_FN(_cb), None))
        if err:
            raise err[0]
        return got

    def last_batch(self) -> List[Sequence]:
        out = (_P * 4096)()
        n = lib().nvr_engine_last_batch(self.h, out, 4096)
        return [Sequence(_handle=out[i]) for i in range(n)]

    def take_finished(self) -> List[Sequence]:
        return self.scheduler.take_finished()

    def abort_last_batch(self) -> None:
        """Control-plane abort of the batch scheduled last (tensor-parallel ranks after a peer's NVR_ERR_RCCL)."""
        check(lib().nvr_engine_abort_last_batch(self.h))

    def ahead_declined(self) -> int:
        return int(lib().nvr_engine_ahead_declined(self.h))

    def ahead_launched(self) -> int:
        return int(lib().nvr_engine_ahead_launched(self.h))

    def host_times(self) -> dict:
        """microseconds spent inside Scheduler::schedule / ::postprocess so far and the steps counted (nvr_engine_host_times)"""
        out = (C.c_double * 3)()
        lib().nvr_engine_host_times(self.h, out)
        return dict(schedule_us=out[0], postprocess_us=out[1], steps=int(out[2]))


# ---------------------------------------------------------------------------------- device helpers
class DeviceBuffer:
    """A hipMalloc'ed buffer with numpy upload/download (tests and bench only)."""

    def __init__(self, nbytes: int):
        p = _P()
        check(lib().nvr_device_malloc(C.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 16))
        if a.nbytes:
            check(lib().nvr_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes))
        return b

    def to_numpy(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            check(lib().nvr_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self) -> None:
        check(lib().nvr_device_memset(self.ptr, 0, self.nbytes))

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.nvr_device_free(self.ptr)
            self.ptr = None


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synthetic_tokens(n: int, seed: int, stream: int, vocab: int) -> np.ndarray:
    """Deterministic prompt token ids uniform in [0, vocab): splitmix64(key ^ i) % vocab
    with key = nvr_weight_key(seed, stream) — the same counter-based generator as the synthetic weights."""
    key = np.uint64(lib().nvr_weight_key(seed, stream))
    with np.errstate(over="ignore"):
        r = _splitmix64(np.arange(n, dtype=np.uint64) ^ key)
    return (r % np.uint64(vocab)).astype(np.int64)


def device_count() -> int:
    n = C.c_int()
    rc = lib().nvr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().nvr_device_name(buf, 256))
    return buf.value.decode()


def synchronize() -> None:
    check(lib().nvr_device_synchronize())
