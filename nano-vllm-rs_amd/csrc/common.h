// common.h — status codes, thread-local error text, HIP error plumbing.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>
#include "../../include/nvr.h"

namespace nvr {

std::string &last_error_slot();
int &last_status_slot();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Every NVR_* environment switch of the product, read ONCE when a runner / engine is created (never on a launch path).
// Diagnostics and test hooks only: the defaults are the product.
struct Env {
    bool trace_host = false;      // NVR_TRACE_HOST=1     host time per step phase, printed when the engine is destroyed
    bool tiled_weights = true;    // NVR_TILED_WEIGHTS=0  decode kernels read the row-major parameters (bit-identical; tests)
    bool lazy_logits = true;      // NVR_LAZY_LOGITS=0    greedy steps store their f32 logits too (tests)
    bool tp_no_comm = false;      // NVR_TP_NO_COMM=1     one tensor-parallel rank's compute without its collectives (profiling)
    bool tp_force_comm = false;   // NVR_TP_FORCE_COMM=1  enqueue the RCCL collectives even with one rank (tests on a 1-GPU box)
    bool tp_graph = true;         // NVR_TP_GRAPH=0       tensor-parallel decode steps run eagerly
    bool f32_fused_norm = true;   // NVR_F32_FUSED_NORM=0 float32 decode-sized steps keep the add + RMSNorm launch in front of their GEMVs (bit-identical; A/B, tests)
    bool attn_fused_merge = true; // NVR_ATTN_FUSED_MERGE=0 split-KV decode attention keeps its merge launch (bit-identical; A/B, tests)
    int max_graphs = 256;         // NVR_MAX_GRAPHS=n     captured decode graphs kept before the cache is flushed (test hook)
    int p2p_timeout_ms = 20000;   // NVR_P2P_TIMEOUT_MS=n how long a peer-to-peer collective waits for a peer before it gives up
    bool p2p_fenced = false;      // NVR_P2P_FENCED=1     the one-shot collectives keep r04's release / acquire fences (kernels/comm_p2p.hip; nvr_runner_p2p_set_fenced)
    int selftest_inject = 0;      // NVR_SELFTEST_INJECT=n test hook: nvr_runner_comm_selftest reports a failure on its first n calls of this runner (fallback dry runs)
    static Env read();
};

}  // namespace nvr

#define NVR_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return nvr::fail(NVR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                               \
    } while (0)
