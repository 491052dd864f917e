#include "common.h"
#include <atomic>
#include <cstdlib>
#include "sequence.h"
namespace nvr {
std::string &last_error_slot() { static thread_local std::string s; return s; }
int &last_status_slot() { static thread_local int c = 0; return c; }
int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    last_error_slot() = buf;
    last_status_slot() = code;
    return code;
}
std::atomic<uint64_t> g_sequence_counter{0};
Env Env::read() {
    Env e;
    auto flag = [](const char *name, bool dflt) { const char *v = std::getenv(name); return v && v[0] ? v[0] != '0' : dflt; };
    auto num = [](const char *name, int dflt, int lo, int hi) {
        const char *v = std::getenv(name);
        if (!v || !v[0]) return dflt;
        long x = std::strtol(v, nullptr, 10);
        return (int)(x < lo ? lo : x > hi ? hi : x);
    };
    e.trace_host = flag("NVR_TRACE_HOST", false);
    e.tiled_weights = flag("NVR_TILED_WEIGHTS", true);
    e.lazy_logits = flag("NVR_LAZY_LOGITS", true);
    e.tp_no_comm = flag("NVR_TP_NO_COMM", false);
    e.tp_force_comm = flag("NVR_TP_FORCE_COMM", false);
    e.tp_graph = flag("NVR_TP_GRAPH", true);
    e.attn_fused_merge = flag("NVR_ATTN_FUSED_MERGE", true);
    e.f32_fused_norm = flag("NVR_F32_FUSED_NORM", true);
    e.max_graphs = num("NVR_MAX_GRAPHS", 256, 1, 1 << 20);
    e.p2p_timeout_ms = num("NVR_P2P_TIMEOUT_MS", 20000, 1, 3600000);
    e.p2p_fenced = flag("NVR_P2P_FENCED", false);
    e.selftest_inject = num("NVR_SELFTEST_INJECT", 0, 0, 16);
    return e;
}
}
