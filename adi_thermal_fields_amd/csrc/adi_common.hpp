// adi_common.hpp -- host-side helpers shared by the translation units of libadi_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/adi_hip.h"

namespace adi {

// thread-local last-error string (adi_last_error)
char *err_buf();
int set_err(int code, const char *fmt, ...);

#define ADI_HIP_TRY(expr)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return adi::set_err(ADI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                            \
    } while (0)

#define ADI_CHECK_LAUNCH()                                                                      \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess)                                                                   \
            return adi::set_err(ADI_ERR_HIP, "kernel launch failed: %s (%s:%d)",               \
                                hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

#define ADI_REQUIRE(cond, ...)                                          \
    do {                                                                \
        if (!(cond)) return adi::set_err(ADI_ERR_ARG, __VA_ARGS__);     \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// rows-per-thread limit of the in-register partition kernels: 64 lanes x 16 rows
constexpr int kMaxFastLine = 1024;

}  // namespace adi
