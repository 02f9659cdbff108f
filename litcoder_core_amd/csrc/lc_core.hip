// Library-level entry points: version, per-thread error string, device check.
#include "lc_common.h"
#include <cstring>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace lc {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace lc

namespace lc {
namespace {
struct EventPair { hipEvent_t a, b; };
std::mutex g_mu;
unsigned long long g_mask = 0;           // timed slots (bit per slot)
std::vector<EventPair> g_events[T_SLOTS];
std::vector<hipEvent_t> g_open[T_SLOTS];
const char* const g_names[T_SLOTS] = {
    "alpha_sweep_gemm", "alpha_sweep_finalize", "grouped_gemm", "batch_chol_solve", "lambda_max", "gram",
    "batch_assemble", "val_stats", "pearson_cols", "gather", "scatter_axpy", "select_group", "fir_delay",
    "lanczos_interp", "cast", "col_stats", "split_f16", "series_hat", "series_sweep_gemm"};
}  // namespace

bool timing_on(int slot) { return (g_mask >> slot) & 1ull; }

void timing_begin(int slot, hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_open[slot].push_back(e);
}

void timing_end(int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open[slot].empty()) return;
    EventPair p;
    p.a = g_open[slot].back();
    g_open[slot].pop_back();
    if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return; }
    (void)hipEventRecord(p.b, s);
    g_events[slot].push_back(p);
}
}  // namespace lc

namespace lc {
int ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    LC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({kernel, dev})) return LC_OK;
    LC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kernel, dev});
    return LC_OK;
}
}  // namespace lc

extern "C" int lc_version(void) { return 101; }

extern "C" int lc_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(lc::g_mu);
    lc::g_mask = on ? ~0ull : 0ull;
    return LC_OK;
}

extern "C" int lc_timing_enable_slots(uint64_t mask) {
    std::lock_guard<std::mutex> lk(lc::g_mu);
    lc::g_mask = mask;
    return LC_OK;
}

extern "C" int lc_timing_slots(void) { return lc::T_SLOTS; }

extern "C" const char* lc_timing_name(int slot) {
    return (slot >= 0 && slot < lc::T_SLOTS) ? lc::g_names[slot] : "";
}

extern "C" int lc_timing_read(int slot, double* total_ms, int* calls) {
    LC_REQUIRE(slot >= 0 && slot < lc::T_SLOTS && total_ms && calls, LC_E_BADARG, "lc_timing_read: bad argument");
    std::vector<lc::EventPair> ev;
    {
        std::lock_guard<std::mutex> lk(lc::g_mu);
        ev.swap(lc::g_events[slot]);
    }
    double sum = 0.0;
    for (auto& p : ev) {
        float ms = 0.f;
        LC_HIP(hipEventSynchronize(p.b));
        LC_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        sum += ms;
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    *total_ms = sum;
    *calls = (int)ev.size();
    return LC_OK;
}

extern "C" const char* lc_last_error(void) { return lc::g_err; }

extern "C" int lc_check_device(int dev) {
    hipDeviceProp_t prop;
    LC_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return lc::fail(LC_E_ARCH, "device %d is %s; this library carries gfx950 code only", dev, prop.gcnArchName);
    return LC_OK;
}

// A stream whose kernels may only run on the CUs whose bits are set in `mask` (words x 32 bits, bit i = CU i).
// The fit's main stream can be created this way to keep a few CUs free for the auxiliary stream's short fp64
// kernels, which otherwise wait for a whole MFMA-sweep workgroup to retire before they find a free slot.
extern "C" int lc_stream_create_cu_mask(const uint32_t* mask, int words, lc_stream_t* out) {
    LC_REQUIRE(mask && out && words > 0, LC_E_BADARG, "lc_stream_create_cu_mask: bad argument");
    hipStream_t s = nullptr;
    LC_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    *out = reinterpret_cast<lc_stream_t>(s);
    return LC_OK;
}

extern "C" int lc_stream_destroy(lc_stream_t stream) {
    LC_REQUIRE(stream, LC_E_BADARG, "lc_stream_destroy: null stream");
    LC_HIP(hipStreamDestroy(lc::as_stream(stream)));
    return LC_OK;
}

// ------------------------------------------------------------------ host <-> device movement of voxel panels
// Strided 2-D copy on a stream: `rows` rows of `width_bytes`, row pitches in bytes.  kind: 0 = host -> device,
// 1 = device -> host, 2 = device -> device.  The host side should be page-locked (the copy is then one DMA at link
// rate and really asynchronous).  Used for the column panels of the targets / weights (nested_cv.py:99-100, 293-296
// are the reference's host <-> device boundary).
extern "C" int lc_memcpy2d_async(void* dst, int64_t dst_pitch, const void* src, int64_t src_pitch, int64_t width_bytes,
                                 int64_t rows, int kind, lc_stream_t stream) {
    LC_REQUIRE(dst && src, LC_E_BADARG, "lc_memcpy2d_async: null pointer");
    LC_REQUIRE(width_bytes >= 0 && rows >= 0 && dst_pitch >= width_bytes && src_pitch >= width_bytes && kind >= 0 && kind <= 2,
               LC_E_SHAPE, "lc_memcpy2d_async: pitches must cover the row width, kind in 0..2");
    if (width_bytes == 0 || rows == 0) return LC_OK;
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (dst_pitch == width_bytes && src_pitch == width_bytes)
        LC_HIP(hipMemcpyAsync(dst, src, (size_t)(width_bytes * rows), k, lc::as_stream(stream)));
    else
        LC_HIP(hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)width_bytes, (size_t)rows, k,
                                lc::as_stream(stream)));
    return LC_OK;
}

// HOST code (no device involved): rows x cols block of a float64 host matrix -> float32, round to nearest even -- the
// cast of `torch.tensor(x, dtype=torch.float32)` (nested_cv.py:99-100) done while a block is staged into page-locked
// memory, so that 4 instead of 8 bytes per value cross PCIe.  Called from the staging threads through ctypes (which
// releases the interpreter lock for the duration of the call).
extern "C" int lc_host_cast_f64_f32(const double* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows,
                                    int64_t cols) {
    LC_REQUIRE(src && dst, LC_E_BADARG, "lc_host_cast_f64_f32: null pointer");
    LC_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, LC_E_SHAPE, "lc_host_cast_f64_f32: bad shape");
    for (int64_t r = 0; r < rows; ++r) {
        const double* __restrict__ s = src + r * ld_src;
        float* __restrict__ d = dst + r * ld_dst;
        for (int64_t c = 0; c < cols; ++c) d[c] = (float)s[c];
    }
    return LC_OK;
}

// HOST code: plain strided row copy of 4-byte values (float32 inputs staged into page-locked memory).
extern "C" int lc_host_copy_f32(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t cols) {
    LC_REQUIRE(src && dst, LC_E_BADARG, "lc_host_copy_f32: null pointer");
    LC_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, LC_E_SHAPE, "lc_host_copy_f32: bad shape");
    for (int64_t r = 0; r < rows; ++r) memcpy(dst + r * ld_dst, src + r * ld_src, (size_t)cols * sizeof(float));
    return LC_OK;
}

// hipMemset2DAsync: `rows` rows of `width_bytes` at byte pitch `pitch` (the zero padding columns of a matrix whose
// body arrives by copies).
extern "C" int lc_fill2d_bytes(void* d_ptr, int64_t pitch, int byte, int64_t width_bytes, int64_t rows, lc_stream_t stream) {
    LC_REQUIRE(d_ptr && pitch >= width_bytes && width_bytes >= 0 && rows >= 0, LC_E_BADARG, "lc_fill2d_bytes: bad argument");
    if (width_bytes == 0 || rows == 0) return LC_OK;
    LC_HIP(hipMemset2DAsync(d_ptr, (size_t)pitch, byte, (size_t)width_bytes, (size_t)rows, lc::as_stream(stream)));
    return LC_OK;
}
