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
#include <fstream>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Chunk {
    int job;
    int64_t r0, r1;        // rows of the job's source block
    int64_t x0, x1;        // columns of the job's panel, relative to c0 (z-scored jobs are cut by columns, all rows)
    int first_task, n_tasks;
};

// A chunk is ONE staging slot and ONE 2-D copy (wide rows: a copy engine takes ~1.5 us per row whatever its width, so a
// story's 350 rows should be tens of KB each, not 2 KB); its staging work is cut into TASKS -- column sub-ranges of a
// z-scored chunk -- that the threads take one by one, and whoever finishes a chunk's last task issues its copy.
struct Task {
    int chunk;
    int64_t m0, m1;        // columns relative to the chunk's x0
};

struct Slot {
    hipEvent_t ev = nullptr;
    bool recorded = false;  // written by the one thread that issues the slot's current chunk
};

// utils.zs (encoding/utils.py:23-29) on a (n rows) x (x1 - x0 columns) block of ONE story, written as float32 into the
// staging slot (row stride w_stage): per column  m = v - v.mean(0);  s = v.std(0) (population);  m /= s where s != 0 --
// in the block's own precision (float64 for float64 data, float32 for float32 data, as numpy computes it), then the
// cast of torch.tensor(..., dtype=float32) (nested_cv.py:99-100).  numpy reduces axis 0 of a C-ordered matrix row by
// row, one running sum per column: the loops below add in exactly that order, without fused multiply-adds, so the result
// is the reference's bit for bit.  Sub-tiles of ZS_TILE columns keep the block in cache between its three passes.
constexpr int64_t ZS_TILE = 1024;      // widest column sub-tile (the stack buffers); zs_tile() = the width in use
constexpr int64_t ZS_TASK = 1024;     // columns per staging task of a z-scored chunk (a multiple of ZS_TILE)

// columns per sub-tile: a story's sub-tile (rows x width x 8 bytes) should stay in the core's L2 between the three passes
inline int64_t zs_tile() {
    static const int64_t w = [] {
        const char* e = getenv("LITCODER_AMD_ZS_TILE");
        int64_t v = e ? atoll(e) : 128;
        return v < 8 ? 8 : (v > ZS_TILE ? ZS_TILE : v);
    }();
    return w;
}

template <typename T>
static inline __attribute__((always_inline)) void zscore_body(const T* src, int64_t ld_src, int64_t n, int64_t x0,
                                                              int64_t x1, float* stage, int64_t w_stage) {
#pragma clang fp contract(off)
    T mean[ZS_TILE], sdev[ZS_TILE];
    const T cnt = (T)n;
    const int64_t TW = zs_tile();
    for (int64_t t0 = x0; t0 < x1; t0 += TW) {
        const int64_t tw = (x1 - t0 < TW) ? x1 - t0 : TW;
        const T* a0 = src + t0;
        for (int64_t x = 0; x < tw; ++x) mean[x] = (T)0;
        for (int64_t r = 0; r < n; ++r) {
            const T* __restrict__ a = a0 + r * ld_src;
            for (int64_t x = 0; x < tw; ++x) mean[x] += a[x];
        }
        for (int64_t x = 0; x < tw; ++x) {
            mean[x] = mean[x] / cnt;
            sdev[x] = (T)0;
        }
        for (int64_t r = 0; r < n; ++r) {
            const T* __restrict__ a = a0 + r * ld_src;
            for (int64_t x = 0; x < tw; ++x) {
                const T d = a[x] - mean[x];
                sdev[x] += d * d;
            }
        }
        for (int64_t x = 0; x < tw; ++x) sdev[x] = std::sqrt(sdev[x] / cnt);
        for (int64_t r = 0; r < n; ++r) {
            const T* __restrict__ a = a0 + r * ld_src;
            float* __restrict__ d = stage + r * w_stage + (t0 - x0);
            for (int64_t x = 0; x < tw; ++x) {
                T m = a[x] - mean[x];
                if (sdev[x] != (T)0) m = m / sdev[x];
                d[x] = (float)m;
            }
        }
    }
}

// the same loops compiled for AVX2 (no FMA: target("avx2") does not enable it) and for the baseline ISA
template <typename T>
__attribute__((target("avx2"))) void zscore_avx2(const T* src, int64_t ld, int64_t n, int64_t x0, int64_t x1, float* st,
                                                 int64_t ws) {
    zscore_body<T>(src, ld, n, x0, x1, st, ws);
}

template <typename T>
void zscore_base(const T* src, int64_t ld, int64_t n, int64_t x0, int64_t x1, float* st, int64_t ws) {
    zscore_body<T>(src, ld, n, x0, x1, st, ws);
}

template <typename T>
void zscore_block(const T* src, int64_t ld, int64_t n, int64_t x0, int64_t x1, float* st, int64_t ws) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) zscore_avx2<T>(src, ld, n, x0, x1, st, ws);
    else zscore_base<T>(src, ld, n, x0, x1, st, ws);
}

}  // namespace

struct lc_upload {
    std::vector<lc_upload_job> jobs;
    std::vector<Chunk> chunks;
    std::vector<Task> tasks;
    std::unique_ptr<std::atomic<int>[]> tasks_left;     // per chunk: staging tasks not yet done
    std::unique_ptr<std::atomic<int>[]> chunk_issued;   // per chunk: its copy is in the stream, its slot event recorded
    std::vector<int> chunks_left;          // per job: chunks not yet issued
    std::vector<hipEvent_t> job_ev;        // recorded on the upload stream behind the job's last copy
    std::vector<char> job_issued;
    std::vector<void*> slot_ptr;
    std::vector<void*> dev_slot_ptr;       // optional device staging slots (one per host slot): see the worker
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
    cpu_set_t cpus;                         // CPUs of the source arrays' NUMA node (n_cpus > 0: the staging threads stay on it)
    int n_cpus = 0;

    explicit lc_upload(int n_slots) : slots(n_slots) { CPU_ZERO(&cpus); }
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
    const int n_slots = (int)u->slots.size();
    for (;;) {
        const size_t t = u->next.fetch_add(1);
        if (t >= u->tasks.size()) return;
        {
            std::lock_guard<std::mutex> lk(u->mu);
            if (u->failed) return;
        }
        const Task& tk = u->tasks[t];
        const int k = tk.chunk;
        const Chunk& c = u->chunks[k];
        const lc_upload_job& j = u->jobs[c.job];
        const int64_t n = c.r1 - c.r0, w = c.x1 - c.x0, mw = tk.m1 - tk.m0;
        Slot& s = u->slots[k % n_slots];
        if (k >= n_slots) {
            // the slot's previous chunk (k - n_slots): every task of it was taken before this one (tasks are handed out in
            // order) by a thread that finishes it without waiting for anything later -- so this wait ends -- and its copy
            // must have left the slot before the slot is written again
            while (!u->chunk_issued[k - n_slots].load(std::memory_order_acquire)) {
                {
                    std::lock_guard<std::mutex> lk(u->mu);
                    if (u->failed) return;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            UP_HIP(hipEventSynchronize(s.ev));
        }
        float* stage = static_cast<float*>(u->slot_ptr[k % n_slots]);
        static const int diag = getenv("LITCODER_AMD_UPLOAD_DIAG") ? atoi(getenv("LITCODER_AMD_UPLOAD_DIAG")) : 0;
        if (diag == 2) {
            // (diagnostic: no staging work, the copies alone)
        } else if (j.transform == LC_UPLOAD_ZSCORE) {
            if (j.dtype == LC_F64)
                zscore_block(static_cast<const double*>(j.src) + j.c0 + c.x0 + tk.m0, j.ld_src, n, 0, mw, stage + tk.m0, w);
            else
                zscore_block(static_cast<const float*>(j.src) + j.c0 + c.x0 + tk.m0, j.ld_src, n, 0, mw, stage + tk.m0, w);
        } else if (j.dtype == LC_F64) {
            const double* src = static_cast<const double*>(j.src) + c.r0 * j.ld_src + j.c0 + c.x0 + tk.m0;
            for (int64_t r = 0; r < n; ++r) {
                const double* __restrict__ a = src + r * j.ld_src;
                float* __restrict__ d = stage + r * w + tk.m0;
                for (int64_t x = 0; x < mw; ++x) d[x] = (float)a[x];
            }
        } else {
            const float* src = static_cast<const float*>(j.src) + c.r0 * j.ld_src + j.c0 + c.x0 + tk.m0;
            for (int64_t r = 0; r < n; ++r) memcpy(stage + r * w + tk.m0, src + r * j.ld_src, (size_t)mw * sizeof(float));
        }
        if (u->tasks_left[k].fetch_sub(1, std::memory_order_acq_rel) != 1) continue;
        // this thread staged the chunk's last piece: the chunk's ONE copy (rows of w floats), then its slot event
        float* dst = static_cast<float*>(j.dst) + (j.dst_row0 + c.r0) * j.ld_dst + j.c0 + c.x0;
        if (diag == 1) {
            // (diagnostic: staging alone, nothing crosses the link)
        } else if (j.ld_dst == w) {
            UP_HIP(hipMemcpyAsync(dst, stage, (size_t)(n * w) * sizeof(float), hipMemcpyHostToDevice, u->stream));
        } else if (!u->dev_slot_ptr.empty()) {
            // a strided (2-D) copy across PCIe pays per row: measured 30-38 GB/s for rows of 40-48 KB against 57 GB/s for
            // one contiguous copy of the same bytes (profiles/r04_upload_probe.txt).  So the chunk crosses the link as ONE
            // contiguous copy into a device staging slot and is laid out into the panel's columns by a device-to-device
            // 2-D copy (HBM to HBM: microseconds), both on the upload stream, in order -- the device slot is free again
            // when the next chunk that maps to it is copied
            void* ds = u->dev_slot_ptr[k % n_slots];
            UP_HIP(hipMemcpyAsync(ds, stage, (size_t)(n * w) * sizeof(float), hipMemcpyHostToDevice, u->stream));
            UP_HIP(hipMemcpy2DAsync(dst, (size_t)j.ld_dst * sizeof(float), ds, (size_t)w * sizeof(float),
                                    (size_t)w * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, u->stream));
        } else {
            UP_HIP(hipMemcpy2DAsync(dst, (size_t)j.ld_dst * sizeof(float), stage, (size_t)w * sizeof(float),
                                    (size_t)w * sizeof(float), (size_t)n, hipMemcpyHostToDevice, u->stream));
        }
        UP_HIP(hipEventRecord(s.ev, u->stream));
        s.recorded = true;
        u->chunk_issued[k].store(1, std::memory_order_release);
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

// The NUMA node most of the jobs' source pages live on (move_pages with a NULL node list only reports), -1 = unknown.
int source_node(const lc_upload* u) {
    std::vector<void*> pages;
    const long page = sysconf(_SC_PAGESIZE);
    if (page <= 0) return -1;
    const size_t per_job = u->jobs.size() >= 32 ? 2 : 64 / (u->jobs.size() ? u->jobs.size() : 1);
    for (const lc_upload_job& b : u->jobs) {
        if (b.rows <= 0) continue;
        const size_t es = b.dtype == LC_F64 ? 8 : 4;
        for (size_t k = 0; k < per_job && pages.size() < 128; ++k) {
            const int64_t r = (int64_t)((2 * k + 1) * (size_t)b.rows / (2 * per_job));
            const uintptr_t a = reinterpret_cast<uintptr_t>(b.src) + (uintptr_t)((r * b.ld_src + b.c0) * (int64_t)es);
            pages.push_back(reinterpret_cast<void*>(a / (uintptr_t)page * (uintptr_t)page));
        }
    }
    if (pages.empty()) return -1;
    std::vector<int> status(pages.size(), -1);
    if (syscall(SYS_move_pages, 0, (unsigned long)pages.size(), pages.data(), nullptr, status.data(), 0) != 0) return -1;
    int votes[64] = {0};
    for (int st : status)
        if (st >= 0 && st < 64) ++votes[st];
    int best = -1;
    for (int n = 0; n < 64; ++n)
        if (votes[n] > 0 && (best < 0 || votes[n] > votes[best])) best = n;
    return best;
}

// The CPUs of ONE NUMA node (/sys/devices/system/node/node<k>/cpulist) into u->cpus: the node the jobs' source arrays
// live on -- asked of the kernel page by page -- or, when that cannot be told, the node the calling thread runs on (its
// arrays were, as a rule, first touched there).  Staging threads scheduled on the other socket read the arrays across
// the socket link and the upload takes 1.5-2x as long, with stalls of tens of ms (measured on the 2-socket EPYC 9575F box:
// z-scored 6 GB 62-71 ms bound to the data's node, 93-148 ms unbound, profiles/r04_upload_probe_numa.txt; bound to
// the CALLER's node a fit was 40 ms slower whenever the scheduler had moved the Python thread to the other socket
// since it filled its arrays).  Any failure leaves the threads unbound.
void source_node_cpus(lc_upload* u) {
    if (getenv("LITCODER_AMD_UPLOAD_NO_AFFINITY")) return;
    const int want = source_node(u);
    const int cpu = sched_getcpu();
    if (want < 0 && cpu < 0) return;
    for (int node = 0; node < 64; ++node) {
        std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
        if (!f) break;
        std::string list;
        std::getline(f, list);
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = 0;
        bool mine = false;
        size_t i = 0;
        while (i < list.size()) {                       // "0-63,128-191"
            size_t e = list.find(',', i);
            if (e == std::string::npos) e = list.size();
            const std::string part = list.substr(i, e - i);
            const size_t dash = part.find('-');
            const int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
            for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) {
                CPU_SET(c, &set);
                ++n;
                mine |= c == cpu;
            }
            i = e + 1;
        }
        if (want >= 0 ? node == want : mine) {
            // only CPUs this process may use at all
            cpu_set_t allowed;
            if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
                CPU_AND(&set, &set, &allowed);
                n = CPU_COUNT(&set);
            }
            if (n > 0) {
                u->cpus = set;
                u->n_cpus = n;
            }
            if (getenv("LITCODER_AMD_UPLOAD_DIAG_NODE"))
                fprintf(stderr, "lc_upload: source node %d, caller cpu %d, staging threads on node %d (%d cpus)\n", want, cpu,
                        node, n);
            return;
        }
    }
}

void coordinate(lc_upload* u) {
    std::vector<std::thread> pool;
    for (int t = 0; t < u->n_threads; ++t) {
        pool.emplace_back(worker, u);
        if (u->n_cpus > 0) (void)pthread_setaffinity_np(pool.back().native_handle(), sizeof(u->cpus), &u->cpus);
    }
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
    return lc_upload_start_staged(jobs, n_jobs, pinned_slots, nullptr, n_slots, slot_bytes, n_threads, device, stream, out);
}

extern "C" int lc_upload_start_staged(const lc_upload_job* jobs, int n_jobs, void* const* pinned_slots,
                                      void* const* device_slots, int n_slots, int64_t slot_bytes, int n_threads, int device,
                                      lc_stream_t stream, lc_upload_t** out) {
    LC_REQUIRE(jobs && pinned_slots && out && n_jobs > 0 && n_slots > 0 && slot_bytes > 0 && n_threads > 0, LC_E_BADARG,
               "lc_upload_start: bad argument");
    lc_upload* u = new lc_upload(n_slots);
    u->jobs.assign(jobs, jobs + n_jobs);
    u->slot_ptr.assign(pinned_slots, pinned_slots + n_slots);
    if (device_slots) u->dev_slot_ptr.assign(device_slots, device_slots + n_slots);
    u->slot_bytes = slot_bytes;
    u->device = device;
    u->stream = lc::as_stream(stream);
    u->n_threads = n_threads;
    u->chunks_left.assign(n_jobs, 0);
    u->job_issued.assign(n_jobs, 0);
    u->job_ev.assign(n_jobs, nullptr);
    int rc = LC_OK;
    for (int j = 0; j < n_jobs && rc == LC_OK; ++j) {
        const lc_upload_job& b = jobs[j];
        const int64_t w = b.c1 - b.c0;
        const bool zs = b.transform == LC_UPLOAD_ZSCORE;
        if (!(b.src && b.dst && b.rows >= 0 && w > 0 && b.ld_src >= b.c1 && b.ld_dst >= b.c1 &&
              (b.dtype == LC_F32 || b.dtype == LC_F64) && (b.transform == LC_UPLOAD_CAST || zs) &&
              (zs ? b.rows * 4 : w * 4) <= slot_bytes)) {
            rc = lc::fail(LC_E_SHAPE, "lc_upload_start: job %d: bad shape or transform, or a row (z-scored jobs: a column) of "
                                      "the panel does not fit a staging slot", j);
            break;
        }
        if (zs) {
            // a story's column statistics need ALL its rows: a chunk is a column range (all rows), as WIDE as a staging
            // slot allows -- its copy then moves rows of tens of KB -- and its z-scoring is cut into tasks of ZS_TASK
            // columns that the threads share
            if (b.rows == 0) continue;
            int64_t step = slot_bytes / (b.rows * 4);
            if (step >= ZS_TILE) step = step / ZS_TILE * ZS_TILE;
            if (step < 1) step = 1;
            for (int64_t x0 = 0; x0 < w; x0 += step) {
                const int64_t x1 = x0 + step < w ? x0 + step : w;
                Chunk c{j, 0, b.rows, x0, x1, (int)u->tasks.size(), 0};
                for (int64_t m0 = 0; m0 < x1 - x0; m0 += ZS_TASK) {
                    u->tasks.push_back({(int)u->chunks.size(), m0, m0 + ZS_TASK < x1 - x0 ? m0 + ZS_TASK : x1 - x0});
                    ++c.n_tasks;
                }
                u->chunks.push_back(c);
                ++u->chunks_left[j];
            }
            continue;
        }
        // rows per chunk: what fits a staging slot, but no more than the job's share per thread -- the design (3000 x 3072)
        // fitted three 16 MB slots, so only three threads cast it (2.4 ms at the head of every fit); chunks stay >= 1 MB
        int64_t step = slot_bytes / (w * 4);
        // (threads per job: a few big jobs -- the design, the panels of one target matrix -- are each cut into a chunk per
        // thread; with many jobs -- the story blocks of every panel -- the jobs themselves run in parallel and a story is
        // one chunk, one copy: 1 MB chunks there ran the link at 23 GB/s, whole stories at 42, profiles/r04_upload_probe*)
        const int64_t par = 2 * n_jobs >= u->n_threads ? 1 : u->n_threads;
        const int64_t share = (b.rows + par - 1) / par;
        const int64_t floor_rows = ((1 << 20) + w * 4 - 1) / (w * 4);
        if (share < step) step = share > floor_rows ? share : (floor_rows < step ? floor_rows : step);
        if (step < 1) step = 1;
        for (int64_t r0 = 0; r0 < b.rows; r0 += step) {
            u->tasks.push_back({(int)u->chunks.size(), 0, w});                  // a plain chunk is one task
            u->chunks.push_back({j, r0, r0 + step < b.rows ? r0 + step : b.rows, 0, w, (int)u->tasks.size() - 1, 1});
            ++u->chunks_left[j];
        }
    }
    u->tasks_left.reset(new std::atomic<int>[u->chunks.size() + 1]);
    u->chunk_issued.reset(new std::atomic<int>[u->chunks.size() + 1]);
    for (size_t k = 0; k < u->chunks.size(); ++k) {
        u->tasks_left[k].store(u->chunks[k].n_tasks);
        u->chunk_issued[k].store(0);
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
    source_node_cpus(u);
    u->coordinator = std::thread(coordinate, u);
    *out = u;
    return LC_OK;
}

// HOST code (no device involved): what a z-scored upload job does to one story block, callable by itself -- the CPU test
// suite holds it to numpy's zs + float32 cast bit for bit (tests/test_host_logic.py).
extern "C" int lc_host_zscore_story(const void* src, int dtype, int64_t ld_src, int64_t rows, int64_t cols, float* out,
                                    int64_t ld_out) {
    LC_REQUIRE(src && out, LC_E_BADARG, "lc_host_zscore_story: null pointer");
    LC_REQUIRE((dtype == LC_F32 || dtype == LC_F64) && rows > 0 && cols >= 0 && ld_src >= cols && ld_out >= cols, LC_E_SHAPE,
               "lc_host_zscore_story: bad shape or dtype");
    if (dtype == LC_F64) zscore_block(static_cast<const double*>(src), ld_src, rows, 0, cols, out, ld_out);
    else zscore_block(static_cast<const float*>(src), ld_src, rows, 0, cols, out, ld_out);
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
        if (s.recorded && rc == LC_OK && hipEventSynchronize(s.ev) != hipSuccess) rc = lc::fail(LC_E_HIP, "lc_upload_finish: sync failed");
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
