// Shared host-side plumbing for liblitcoder_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/litcoder_hip.h"

namespace lc {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline hipStream_t as_stream(lc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// After a kernel launch: surface launch-configuration errors without synchronising.
inline int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LC_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return LC_OK;
}

template <typename T>
inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

inline long long imin(long long a, long long b) { return a < b ? a : b; }

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that needs more than the default 64 KB of LDS.  The
// attribute is per DEVICE: set once per (kernel, current device), under a mutex, with the return code checked.
int ensure_dynamic_lds(const void* kernel, int bytes);

// Optional per-kernel-class timing with HIP events on the launch stream (lc_timing_enable).
enum TimingSlot {
    T_SWEEP_GEMM = 0, T_SWEEP_FINALIZE, T_GROUPED_GEMM, T_CHOL_SOLVE, T_LAMBDA_MAX, T_GRAM, T_ASSEMBLE,
    T_VAL_STATS, T_PEARSON, T_GATHER, T_SCATTER, T_SELECT, T_FIR, T_LANCZOS, T_CAST, T_COLSTATS, T_SPLIT16, T_SERIES, T_SERIES_SWEEP,
    T_SLOTS
};
bool timing_on(int slot);
void timing_begin(int slot, hipStream_t s);
void timing_end(int slot, hipStream_t s);

struct ScopedTimer {
    int slot; hipStream_t s; bool on;
    ScopedTimer(int slot_, hipStream_t s_) : slot(slot_), s(s_), on(timing_on(slot_)) { if (on) timing_begin(slot, s); }
    ~ScopedTimer() { if (on) timing_end(slot, s); }
};

}  // namespace lc

#define LC_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return lc::fail((code), __VA_ARGS__); } while (0)
#define LC_HIP(call) \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) return lc::fail(LC_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
