// hip_check.h — the reference's error policy for GPU calls: print and exit
// (CUDA_CHECK, src/cuda/cuda_kernel.cuh:11-18).  The C-ABI itself never
// aborts; this macro is the caller-side policy of the gcn-hip program.  Library
// entry points (capi.cpp) use GCNHIP_RET instead and hand the code back.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include "gcnhip_driver.h"

struct GcnHipFailure : std::runtime_error {
    int code;
    GcnHipFailure(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

// throws; main.cpp turns it into "HIP_ASSERT: <msg> <file> <line>" + exit(code),
// capi.cpp into an int return value
#define GCNHIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        int _rc = (expr);                                                                      \
        if (_rc != 0) {                                                                        \
            char _buf[512];                                                                    \
            snprintf(_buf, sizeof _buf, "HIP_ASSERT: %s %s %d", gcnhip_error_string(_rc), __FILE__, __LINE__); \
            throw GcnHipFailure(_rc, _buf);                                                    \
        }                                                                                      \
    } while (0)
