// Host -> device upload of the fit's inputs in column panels, on NATIVE threads (gfx950 box, host side).
//
// The reference's boundary is `torch.tensor(features / targets, dtype=torch.float32)` (nested_cv.py:99-100): float64
// numpy arrays in pageable host memory become float32 device tensors.  At cfg2 that is 1.92 GB of targets; staged
// through the driver a pageable copy runs at ~10 GB/s, longer than the whole fit.  Here the targets move as column
// panels of voxels (the first outer fold starts on a panel while the others are still crossing PCIe): a panel is cut
// into row chunks, worker threads cast each chunk float64 -> float32 (round to nearest even, the reference's host-side
// cast) straight into a ring of page-locked staging slots -- so 4 bytes per value cross the link -- and the thread that
// staged a chunk issues its 2-D copy into the panel's columns of the destination on the upload stream.
//
// Why native threads: the staging loop used to be Python threads calling small C functions.  While the main Python
// thread queues the fit's ~1000 launches the interpreter lock changes hands at every call of every thread, and both
// sides crawl (measured, round 3: a 24 576-column panel staged in 18 ms instead of 7, the main thread's launch loop
// 14 ms instead of 0.5).  One lc_upload_start call hands the whole job list to a native coordinator; Python only
// blocks (lock released) in lc_upload_wait when a phase needs a panel that has not been issued yet.
#include "lc_common.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Chunk {
    int job;
    int64_t r0, r1;        // rows of the job's source block
};

struct Slot {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    bool used = false;
};

}  // namespace

struct lc_upload {
    std::vector<lc_upload_job> jobs;
    std::vector<Chunk> chunks;
    std::vector<int> chunks_left;          // per job: chunks not yet issued
    std::vector<hipEvent_t> job_ev;        // recorded on the upload stream behind the job's last copy
    std::vector<char> job_issued;
    std::vector<void*> slot_ptr;
    std::vector<Slot> slots;
    int64_t slot_bytes = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    int n_threads = 1;
    std::atomic<size_t> next{0};
    std::mutex mu;                          // guards chunks_left / job_issued / error
    std::condition_variable cv;
    std::string error;
    bool failed = false;
    std::thread coordinator;

    explicit lc_upload(int n_slots) : slots(n_slots) {}
};

namespace {

void fail_upload(lc_upload* u, const std::string& what) {
    std::lock_guard<std::mutex> lk(u->mu);
    if (!u->failed) {
        u->failed = true;
        u->error = what;
    }
    u->cv.notify_all();
}

#define UP_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fail_upload(u, std::string(#call) + ": " + hipGetErrorString(e_));         \
            return;                                                                    \
        }                                                                              \
    } while (0)

void worker(lc_upload* u) {
    UP_HIP(hipSetDevice(u->device));
    for (;;) {
        const size_t k = u->next.fetch_add(1);
        if (k >= u->chunks.size()) return;
        {
            std::lock_guard<std::mutex> lk(u->mu);
            if (u->failed) return;
        }
        const Chunk& c = u->chunks[k];
        const lc_upload_job& j = u->jobs[c.job];
        const int64_t n = c.r1 - c.r0, w = j.c1 - j.c0;
        Slot& s = u->slots[k % u->slots.size()];
        std::lock_guard<std::mutex> hold(s.mu);          // one chunk at a time per staging slot
        if (s.used) UP_HIP(hipEventSynchronize(s.ev));   // the slot's previous copy has left it
        float* stage = static_cast<float*>(u->slot_ptr[k % u->slots.size()]);
        if (j.dtype == LC_F64) {
            const double* src = static_cast<const double*>(j.src) + c.r0 * j.ld_src + j.c0;
            for (int64_t r = 0; r < n; ++r) {
                const double* __restrict__ a = src + r * j.ld_src;
                float* __restrict__ d = stage + r * w;
                for (int64_t x = 0; x < w; ++x) d[x] = (float)a[x];
            }
        } else {
            const float* src = static_cast<const float*>(j.src) + c.r0 * j.ld_src + j.c0;
            for (int64_t r = 0; r < n; ++r) memcpy(stage + r * w, src + r * j.ld_src, (size_t)w * sizeof(float));
        }
        float* dst = static_cast<float*>(j.dst) + (j.dst_row0 + c.r0) * j.ld_dst + j.c0;
        if (j.ld_dst == w)
            UP_HIP(hipMemcpyAsync(dst, stage, (size_t)(n * w) * sizeof(float), hipMemcpyHostToDevice, u->stream));
        else
            UP_HIP(hipMemcpy2DAsync(dst, (size_t)j.ld_dst * sizeof(float), stage, (size_t)w * sizeof(float),
                                    (size_t)w * sizeof(float), (size_t)n, hipMemcpyHostToDevice, u->stream));
        UP_HIP(hipEventRecord(s.ev, u->stream));
        s.used = true;
        bool last;
        {
            std::lock_guard<std::mutex> lk(u->mu);
            last = --u->chunks_left[c.job] == 0;
        }
        if (last) {                                      // every copy of the job is in the stream: its event goes behind them
            UP_HIP(hipEventRecord(u->job_ev[c.job], u->stream));
            std::lock_guard<std::mutex> lk(u->mu);
            u->job_issued[c.job] = 1;
            u->cv.notify_all();
        }
    }
}

void coordinate(lc_upload* u) {
    std::vector<std::thread> pool;
    for (int t = 0; t < u->n_threads; ++t) pool.emplace_back(worker, u);
    for (auto& t : pool) t.join();
    // jobs without rows never had a chunk: mark them issued (their event is recorded here)
    if (hipSetDevice(u->device) == hipSuccess) {
        std::lock_guard<std::mutex> lk(u->mu);
        for (size_t j = 0; j < u->jobs.size(); ++j)
            if (!u->job_issued[j] && u->chunks_left[j] == 0 && !u->failed) {
                (void)hipEventRecord(u->job_ev[j], u->stream);
                u->job_issued[j] = 1;
            }
    }
    u->cv.notify_all();
}

}  // namespace

extern "C" int lc_upload_start(const lc_upload_job* jobs, int n_jobs, void* const* pinned_slots, int n_slots,
                               int64_t slot_bytes, int n_threads, int device, lc_stream_t stream, lc_upload_t** out) {
    LC_REQUIRE(jobs && pinned_slots && out && n_jobs > 0 && n_slots > 0 && slot_bytes > 0 && n_threads > 0, LC_E_BADARG,
               "lc_upload_start: bad argument");
    lc_upload* u = new lc_upload(n_slots);
    u->jobs.assign(jobs, jobs + n_jobs);
    u->slot_ptr.assign(pinned_slots, pinned_slots + n_slots);
    u->slot_bytes = slot_bytes;
    u->device = device;
    u->stream = lc::as_stream(stream);
    u->n_threads = n_threads < n_slots ? n_threads : n_slots;
    u->chunks_left.assign(n_jobs, 0);
    u->job_issued.assign(n_jobs, 0);
    u->job_ev.assign(n_jobs, nullptr);
    int rc = LC_OK;
    for (int j = 0; j < n_jobs && rc == LC_OK; ++j) {
        const lc_upload_job& b = jobs[j];
        const int64_t w = b.c1 - b.c0;
        if (!(b.src && b.dst && b.rows >= 0 && w > 0 && b.ld_src >= b.c1 && b.ld_dst >= b.c1 &&
              (b.dtype == LC_F32 || b.dtype == LC_F64) && w * 4 <= slot_bytes)) {
            rc = lc::fail(LC_E_SHAPE, "lc_upload_start: job %d: bad shape, or a row of the panel does not fit a staging slot", j);
            break;
        }
        // rows per chunk: what fits a staging slot, but no more than the job's share per thread -- the design (3000 x 3072)
        // fitted three 16 MB slots, so only three threads cast it (2.4 ms at the head of every fit); chunks stay >= 1 MB
        int64_t step = slot_bytes / (w * 4);
        const int64_t share = (b.rows + u->n_threads - 1) / (u->n_threads > 0 ? u->n_threads : 1);
        const int64_t floor_rows = ((1 << 20) + w * 4 - 1) / (w * 4);
        if (share < step) step = share > floor_rows ? share : (floor_rows < step ? floor_rows : step);
        if (step < 1) step = 1;
        for (int64_t r0 = 0; r0 < b.rows; r0 += step) {
            u->chunks.push_back({j, r0, r0 + step < b.rows ? r0 + step : b.rows});
            ++u->chunks_left[j];
        }
    }
    if (rc == LC_OK && hipSetDevice(device) != hipSuccess) rc = lc::fail(LC_E_HIP, "lc_upload_start: hipSetDevice failed");
    for (int j = 0; j < n_jobs && rc == LC_OK; ++j)
        if (hipEventCreateWithFlags(&u->job_ev[j], hipEventDisableTiming) != hipSuccess)
            rc = lc::fail(LC_E_HIP, "lc_upload_start: hipEventCreate failed");
    for (int s = 0; s < n_slots && rc == LC_OK; ++s)
        if (hipEventCreateWithFlags(&u->slots[s].ev, hipEventDisableTiming) != hipSuccess)
            rc = lc::fail(LC_E_HIP, "lc_upload_start: hipEventCreate failed");
    if (rc != LC_OK) {
        for (auto e : u->job_ev) if (e) (void)hipEventDestroy(e);
        for (auto& s : u->slots) if (s.ev) (void)hipEventDestroy(s.ev);
        delete u;
        return rc;
    }
    u->coordinator = std::thread(coordinate, u);
    *out = u;
    return LC_OK;
}

extern "C" int lc_upload_wait(lc_upload_t* u, int job, lc_stream_t consumer) {
    LC_REQUIRE(u && job >= 0 && job < (int)u->jobs.size(), LC_E_BADARG, "lc_upload_wait: bad argument");
    {
        std::unique_lock<std::mutex> lk(u->mu);
        u->cv.wait(lk, [&] { return u->failed || u->job_issued[job]; });
        if (u->failed) return lc::fail(LC_E_HIP, "upload failed: %s", u->error.c_str());
    }
    LC_HIP(hipStreamWaitEvent(lc::as_stream(consumer), u->job_ev[job], 0));
    return LC_OK;
}

extern "C" int lc_upload_finish(lc_upload_t* u) {
    LC_REQUIRE(u, LC_E_BADARG, "lc_upload_finish: null handle");
    if (u->coordinator.joinable()) u->coordinator.join();
    int rc = LC_OK;
    if (u->failed) rc = lc::fail(LC_E_HIP, "upload failed: %s", u->error.c_str());
    // the staging ring belongs to the process: leave it idle (every slot's last copy has left it)
    for (auto& s : u->slots)
        if (s.used && rc == LC_OK && hipEventSynchronize(s.ev) != hipSuccess) rc = lc::fail(LC_E_HIP, "lc_upload_finish: sync failed");
    return rc;
}

extern "C" int lc_upload_free(lc_upload_t* u) {
    LC_REQUIRE(u, LC_E_BADARG, "lc_upload_free: null handle");
    if (u->coordinator.joinable()) u->coordinator.join();
    for (auto& s : u->slots) if (s.ev) (void)hipEventDestroy(s.ev);
    for (auto e : u->job_ev) if (e) (void)hipEventDestroy(e);
    delete u;
    return LC_OK;
}
