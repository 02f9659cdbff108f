// Implementation of tools/sanitize/hip_stub/hip/hip_runtime.h (host-only stand-in for the HIP runtime; see there).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <thread>

struct StubStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t submitted = 0, done = 0;
    bool stop = false;
    std::thread th;
};

struct StubEvent {
    std::mutex mu;
    StubStream* stream = nullptr;
    uint64_t seq = 0;
    bool recorded = false;
    std::chrono::steady_clock::time_point when;
};

namespace {
std::atomic<long> g_fail_after{0}, g_copies{0};
std::atomic<int> g_jitter{0};

void run(StubStream* s) {
    std::mt19937 rng(12345);
    for (;;) {
        std::function<void()> op;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop || !s->q.empty(); });
            if (s->q.empty()) return;
            op = std::move(s->q.front());
            s->q.pop_front();
        }
        const int j = g_jitter.load();
        if (j > 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)(j + 1)));
        op();
        {
            std::lock_guard<std::mutex> lk(s->mu);
            ++s->done;
        }
        s->cv.notify_all();
    }
}

StubStream* default_stream() {
    static StubStream* s = [] {
        StubStream* p = new StubStream;
        p->th = std::thread(run, p);
        p->th.detach();                       // lives as long as the process
        return p;
    }();
    return s;
}

StubStream* of(hipStream_t s) { return s ? s : default_stream(); }

void submit(StubStream* s, std::function<void()> op) {
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->q.push_back(std::move(op));
        ++s->submitted;
    }
    s->cv.notify_all();
}

bool copy_fails() {
    g_copies.fetch_add(1);
    long n = g_fail_after.load();
    while (n > 0) {
        if (g_fail_after.compare_exchange_weak(n, n - 1)) return n == 1;
    }
    return false;
}
}  // namespace

void stub_fail_copy_after(long n) { g_fail_after.store(n); }
void stub_stream_jitter_us(int us) { g_jitter.store(us); }
long stub_copies_submitted() { return g_copies.load(); }

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "stub: injected failure"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int) { strcpy(prop->gcnArchName, "gfx950:stub"); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipStreamCreate(hipStream_t* out) {
    StubStream* s = new StubStream;
    s->th = std::thread(run, s);
    *out = s;
    return hipSuccess;
}
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreate(s); }

hipError_t hipStreamSynchronize(hipStream_t hs) {
    StubStream* s = of(hs);
    std::unique_lock<std::mutex> lk(s->mu);
    const uint64_t want = s->submitted;
    s->cv.wait(lk, [&] { return s->done >= want; });
    return hipSuccess;
}

hipError_t hipStreamDestroy(hipStream_t s) {
    if (!s) return hipErrorInvalidValue;
    hipStreamSynchronize(s);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    s->th.join();
    delete s;
    return hipSuccess;
}

hipError_t hipEventCreate(hipEvent_t* e) { *e = new StubEvent; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }

hipError_t hipEventRecord(hipEvent_t e, hipStream_t hs) {
    StubStream* s = of(hs);
    // the stamp is taken by the stream when it gets there (an operation of its own, like a real event record)
    submit(s, [e] {
        std::lock_guard<std::mutex> lk(e->mu);
        e->when = std::chrono::steady_clock::now();
    });
    std::lock_guard<std::mutex> lk(e->mu);
    std::lock_guard<std::mutex> lk2(s->mu);
    e->stream = s;
    e->seq = s->submitted;
    e->recorded = true;
    return hipSuccess;
}

hipError_t hipEventSynchronize(hipEvent_t e) {
    StubStream* s;
    uint64_t seq;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->recorded) return hipSuccess;
        s = e->stream;
        seq = e->seq;
    }
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [&] { return s->done >= seq; });
    return hipSuccess;
}

hipError_t hipStreamWaitEvent(hipStream_t hs, hipEvent_t e, unsigned) {
    submit(of(hs), [e] { hipEventSynchronize(e); });
    return hipSuccess;
}

hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    hipEventSynchronize(a);
    hipEventSynchronize(b);
    std::lock_guard<std::mutex> la(a->mu);
    std::lock_guard<std::mutex> lb(b->mu);
    *ms = std::chrono::duration<float, std::milli>(b->when - a->when).count();
    return hipSuccess;
}

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    if (copy_fails()) return hipErrorUnknown;
    submit(of(s), [=] { memcpy(dst, src, bytes); });
    return hipSuccess;
}

hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind, hipStream_t s) {
    if (copy_fails()) return hipErrorUnknown;
    submit(of(s), [=] {
        for (size_t r = 0; r < height; ++r)
            memcpy(static_cast<char*>(dst) + r * dpitch, static_cast<const char*>(src) + r * spitch, width);
    });
    return hipSuccess;
}

hipError_t hipMemset2DAsync(void* dst, size_t pitch, int value, size_t width, size_t height, hipStream_t s) {
    submit(of(s), [=] {
        for (size_t r = 0; r < height; ++r) memset(static_cast<char*>(dst) + r * pitch, value, width);
    });
    return hipSuccess;
}
