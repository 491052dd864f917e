// config_api.cpp — defaults and validation of SamplingParams (reference src/engine/sampling_params.rs:30-41,91-119) and Config
// (src/config.rs:54-71,83-119): host-only part of the C ABI (also linked into the sanitizer self-test).
#include <cmath>
#include <cstring>
#include <initializer_list>
#include "common.h"

extern "C" {
void nvr_sampling_params_default(nvr_sampling_params *sp) {          // sampling_params.rs:30-41
    std::memset(sp, 0, sizeof *sp);
    sp->temperature = 1.0f; sp->max_tokens = 64;
}
int nvr_sampling_params_validate(const nvr_sampling_params *sp) {    // sampling_params.rs:91-119
    if (sp->temperature < 0.0f) return nvr::fail(NVR_ERR_INVALID_ARG, "Temperature must be non-negative, got %g", sp->temperature);
    if (sp->max_tokens == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "Max tokens must be positive, got 0");
    if (sp->has_top_p && !(sp->top_p >= 0.0f && sp->top_p <= 1.0f)) return nvr::fail(NVR_ERR_INVALID_ARG, "Top-p must be between 0.0 and 1.0, got %g", sp->top_p);
    if (sp->has_top_k && sp->top_k == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "Top-k must be positive, got 0");
    if (sp->has_repetition_penalty && !(sp->repetition_penalty > 0.0f)) return nvr::fail(NVR_ERR_INVALID_ARG, "Repetition penalty must be positive, got %g", sp->repetition_penalty);
    return NVR_OK;
}
void nvr_config_default(nvr_config *c) {                             // config.rs:54-71
    std::memset(c, 0, sizeof *c);
    c->max_num_batched_tokens = 32768; c->max_num_seqs = 512; c->max_model_len = 4096;
    c->gpu_memory_utilization = 0.9f; c->tensor_parallel_size = 1; c->enforce_eager = 0;
    c->has_eos = 0; c->kvcache_block_size = 256; c->num_kvcache_blocks = -1;
    c->async_decode = 1;                                                  // launch-ahead of greedy decode steps: transparent (same batches, tokens, statistics)
    std::strcpy(c->device, "hip"); std::strcpy(c->dtype, "float16");     // config.rs:67-68 ("cuda" there)
}
static bool cfg_str_in(const char *v, size_t cap, std::initializer_list<const char *> set) {
    if (!std::memchr(v, 0, cap)) return false;
    for (const char *s : set) if (!std::strcmp(v, s)) return true;
    return false;
}
// a runner / engine can be built for this config (the HIP path, fp16 or bf16): validate() only checks the names, like the reference
int nvr_config_runnable(const nvr_config *c) {
    if (!cfg_str_in(c->device, sizeof c->device, {"hip", "cuda"}))
        return nvr::fail(NVR_ERR_UNSUPPORTED, "device '%s': this library is the MI355X (HIP) path; there is no CPU or Metal path", c->device);
    if (std::strcmp(c->dtype, "float16") != 0 && std::strcmp(c->dtype, "bfloat16") != 0 && std::strcmp(c->dtype, "float32") != 0)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "dtype '%s': float16, bfloat16 or float32", c->dtype);
    return NVR_OK;
}
int nvr_config_validate(const nvr_config *c) {                       // config.rs:83-119 (model_path checks n/a)
    if (!c->skip_block_size_check && c->kvcache_block_size % 256 != 0)
        return nvr::fail(NVR_ERR_INVALID_ARG, "KV cache block size must be a multiple of 256, got %lu", (unsigned long)c->kvcache_block_size);
    if (c->kvcache_block_size == 0) return nvr::fail(NVR_ERR_INVALID_ARG, "Block size must be positive");
    if (c->tensor_parallel_size < 1 || c->tensor_parallel_size > 8)
        return nvr::fail(NVR_ERR_INVALID_ARG, "Tensor parallel size must be between 1 and 8, got %lu", (unsigned long)c->tensor_parallel_size);
    if (!(c->gpu_memory_utilization >= 0.0f && c->gpu_memory_utilization <= 1.0f))
        return nvr::fail(NVR_ERR_INVALID_ARG, "GPU memory utilization must be between 0.0 and 1.0, got %g", c->gpu_memory_utilization);
    if (!cfg_str_in(c->device, sizeof c->device, {"hip", "cuda", "cpu", "metal"}))                       // config.rs:108-111 (+ "hip")
        return nvr::fail(NVR_ERR_INVALID_ARG, "Unsupported device: %.15s", c->device);
    if (!cfg_str_in(c->dtype, sizeof c->dtype, {"float16", "bfloat16", "float32"}))                      // config.rs:113-116
        return nvr::fail(NVR_ERR_INVALID_ARG, "Unsupported dtype: %.15s", c->dtype);
    if (c->decode_chain != 0 && c->decode_chain != 6)
        return nvr::fail(NVR_ERR_INVALID_ARG, "decode_chain must be 0 (reserved; 6 = the same six-launch chain), got %u", c->decode_chain);
    return NVR_OK;
}
}  // extern "C"
