#include "common.h"
#include <atomic>
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
}
