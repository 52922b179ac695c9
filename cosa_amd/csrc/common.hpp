// common.hpp -- shared host-side helpers for libcosa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/cosa_hip.h"

namespace cosa {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define COSA_REQUIRE(cond, ...)                     \
    do {                                            \
        if (!(cond)) {                              \
            ::cosa::set_error(__VA_ARGS__);         \
            return COSA_EINVAL;                     \
        }                                           \
    } while (0)

#define COSA_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::cosa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                              __FILE__, __LINE__);                                        \
            return COSA_EHIP;                                                             \
        }                                                                                 \
    } while (0)

#define COSA_LAUNCH_CHECK() COSA_HIP_CHECK(hipGetLastError())

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// carve sub-buffers out of a caller-provided workspace
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T>
    T *take(size_t n) {
        T *r = reinterpret_cast<T *>(base + off);
        off = align_up(off + n * sizeof(T), 256);
        return r;
    }
};

constexpr int kMaxDil = 8;  // PAR dilations supported per call

}  // namespace cosa
