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

}  // namespace nvr

#define NVR_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return nvr::fail(NVR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                               \
    } while (0)
