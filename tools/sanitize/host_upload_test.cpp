// Sanitizer driver for the HOST side of liblitcoder_hip.so (tools/sanitize_host.sh builds it with g++
// -fsanitize=address,undefined and -fsanitize=thread against the HIP stand-in of tools/sanitize/hip_stub/): the native
// uploader of csrc/lc_upload.hip -- staging threads, the slot ring and its events, the coordinator, lc_upload_wait /
// _finish / _free -- through its job matrix:
//   plain     float64 and float32 sources, column panels, strided destination, contiguous destination, row blocks
//   zscored   LC_UPLOAD_ZSCORE story blocks (float64 and float32), chunks shared by several threads as tasks
//   lead      float32 lead jobs in front of the panels (harness.StoryPipeline's word features)
//   staged    with and without device staging slots; fewer slots than chunks (the ring turns many times)
//   abandoned finish / free without ever waiting for a job; an injected copy failure (every waiter must return an error,
//             finish must return, free must release everything)
// plus lc_host_zscore_story / lc_host_cast_f64_f32 / lc_host_copy_f32 / lc_memcpy2d_async on edge views (one row, one
// column, ld > cols).  Every destination byte is compared with a scalar restatement.  Test infrastructure, not product.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/litcoder_hip.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, "\n");                         \
            ++g_fail;                                      \
        }                                                  \
    } while (0)

template <typename T>
static void ref_zs(const T* src, int64_t ld, int64_t n, int64_t cols, float* out, int64_t ldo) {
    // utils.zs in numpy's order: running sums down the rows, population std, zero-std columns only de-meaned
    for (int64_t c = 0; c < cols; ++c) {
        volatile T mean = 0;
        for (int64_t r = 0; r < n; ++r) mean = mean + src[r * ld + c];
        mean = mean / (T)n;
        volatile T var = 0;
        for (int64_t r = 0; r < n; ++r) {
            volatile T d = src[r * ld + c] - mean;
            volatile T dd = d * d;
            var = var + dd;
        }
        volatile T q = var / (T)n;
        const T sd = std::sqrt((T)q);
        for (int64_t r = 0; r < n; ++r) {
            volatile T m = src[r * ld + c] - mean;
            if (sd != (T)0) m = m / sd;
            out[r * ldo + c] = (float)m;
        }
    }
}

struct Story {
    std::vector<double> d;
    std::vector<float> f;
    int64_t rows, ld;
};

static int run_matrix(int n_threads, int n_slots, int64_t slot_bytes, bool device_slots, bool zscore, bool f32, long fail_after,
                      bool wait_jobs, unsigned seed) {
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> nd(3.0, 2.0);
    const int64_t V = 700 + (int64_t)(rng() % 300), Vp = V + 37;     // destination wider than the panels (padding columns)
    const int n_blocks = zscore ? 5 : 2;
    std::vector<Story> blocks(n_blocks);
    int64_t T = 0;
    for (auto& b : blocks) {
        b.rows = zscore ? 40 + (int64_t)(rng() % 60) : 150 + (int64_t)(rng() % 100);
        b.ld = V + (int64_t)(rng() % 9);                              // source row stride > columns: a column view
        if (f32) {
            b.f.resize((size_t)(b.rows * b.ld));
            for (auto& x : b.f) x = (float)nd(rng);
            if (zscore) for (int64_t r = 0; r < b.rows; ++r) b.f[(size_t)(r * b.ld + 3)] = 1.25f;   // a constant column
        } else {
            b.d.resize((size_t)(b.rows * b.ld));
            for (auto& x : b.d) x = nd(rng);
            if (zscore) for (int64_t r = 0; r < b.rows; ++r) b.d[(size_t)(r * b.ld + 3)] = 1.25;
        }
        T += b.rows;
    }
    // lead job: a float32 matrix copied as is into its own destination
    const int64_t Lr = 90, Lc = 64;
    std::vector<float> lead((size_t)(Lr * Lc)), lead_dst((size_t)(Lr * Lc), -1.0f);
    for (auto& x : lead) x = (float)nd(rng);
    std::vector<float> dst((size_t)(T * Vp), -7.0f), want((size_t)(T * Vp), -7.0f);
    const int64_t edges[4] = {0, 256, 512, V};
    std::vector<lc_upload_job> jobs;
    jobs.push_back({lead.data(), Lc, LC_F32, Lr, 0, Lc, lead_dst.data(), Lc, 0, LC_UPLOAD_CAST});
    for (int pnl = 0; pnl < 3; ++pnl) {
        int64_t row0 = 0;
        for (auto& b : blocks) {
            const void* src = f32 ? (const void*)b.f.data() : (const void*)b.d.data();
            jobs.push_back({src, b.ld, f32 ? LC_F32 : LC_F64, b.rows, edges[pnl], edges[pnl + 1], dst.data(), Vp, row0,
                            zscore ? LC_UPLOAD_ZSCORE : LC_UPLOAD_CAST});
            row0 += b.rows;
        }
    }
    {   // expected bytes
        int64_t row0 = 0;
        for (auto& b : blocks) {
            if (zscore) {
                if (f32) ref_zs(b.f.data(), b.ld, b.rows, V, want.data() + row0 * Vp, Vp);
                else ref_zs(b.d.data(), b.ld, b.rows, V, want.data() + row0 * Vp, Vp);
            } else {
                for (int64_t r = 0; r < b.rows; ++r)
                    for (int64_t c = 0; c < V; ++c)
                        want[(size_t)((row0 + r) * Vp + c)] = f32 ? b.f[(size_t)(r * b.ld + c)] : (float)b.d[(size_t)(r * b.ld + c)];
            }
            row0 += b.rows;
        }
    }
    std::vector<std::vector<char>> slots(n_slots, std::vector<char>((size_t)slot_bytes)), dslots;
    std::vector<void*> sp, dp;
    for (auto& s : slots) sp.push_back(s.data());
    if (device_slots) {
        dslots.assign(n_slots, std::vector<char>((size_t)slot_bytes));
        for (auto& s : dslots) dp.push_back(s.data());
    }
    hipStream_t up = nullptr, consumer = nullptr;
    hipStreamCreate(&up);
    hipStreamCreate(&consumer);
    stub_fail_copy_after(fail_after);
    lc_upload_t* h = nullptr;
    int rc = lc_upload_start_staged(jobs.data(), (int)jobs.size(), sp.data(), device_slots ? dp.data() : nullptr, n_slots,
                                    slot_bytes, n_threads, 0, up, &h);
    CHECK(rc == LC_OK && h, "lc_upload_start_staged: %d %s", rc, lc_last_error());
    if (rc != LC_OK) return 1;
    int wait_errors = 0;
    if (wait_jobs) {
        // two host threads wait for jobs concurrently, in different orders (Python's main thread + a worker do that)
        std::thread other([&] {
            for (int j = (int)jobs.size() - 1; j >= 0; j -= 2)
                if (lc_upload_wait(h, j, consumer) != LC_OK) __atomic_fetch_add(&wait_errors, 1, __ATOMIC_RELAXED);
        });
        for (int j = 0; j < (int)jobs.size(); ++j)
            if (lc_upload_wait(h, j, consumer) != LC_OK) __atomic_fetch_add(&wait_errors, 1, __ATOMIC_RELAXED);
        other.join();
    }
    rc = lc_upload_finish(h);
    if (fail_after > 0) {
        CHECK(rc != LC_OK, "an injected copy failure must fail lc_upload_finish");
        if (wait_jobs) CHECK(wait_errors > 0, "an injected copy failure must fail the waits");
    } else {
        CHECK(rc == LC_OK, "lc_upload_finish: %d %s", rc, lc_last_error());
        CHECK(wait_errors == 0, "%d waits failed", wait_errors);
    }
    hipStreamSynchronize(up);
    hipStreamSynchronize(consumer);
    CHECK(lc_upload_free(h) == LC_OK, "lc_upload_free");
    stub_fail_copy_after(0);
    if (fail_after == 0) {
        size_t bad = 0;
        for (size_t i = 0; i < dst.size(); ++i) bad += memcmp(&dst[i], &want[i], 4) != 0;
        CHECK(bad == 0, "%zu of %zu destination values differ (threads %d slots %d zs %d f32 %d dev %d)", bad, dst.size(), n_threads,
              n_slots, (int)zscore, (int)f32, (int)device_slots);
        CHECK(memcmp(lead.data(), lead_dst.data(), lead.size() * 4) == 0, "lead job differs");
    }
    hipStreamDestroy(up);
    hipStreamDestroy(consumer);
    return 0;
}

static void edge_views() {
    // one row / one column / ld > cols through the plain host helpers and the 2-D copy
    std::vector<double> a = {1.5, 2.5, 1e-40, -3.25, 7.0, 1e39};
    std::vector<float> o(6, -1.f);
    CHECK(lc_host_cast_f64_f32(a.data(), 3, o.data(), 3, 2, 3) == LC_OK, "cast");
    for (int i = 0; i < 6; ++i) CHECK(o[i] == (float)a[i] || (std::isinf(o[i]) && std::isinf((float)a[i])), "cast value %d", i);
    CHECK(lc_host_cast_f64_f32(a.data(), 6, o.data(), 6, 1, 6) == LC_OK, "cast one row");
    CHECK(lc_host_cast_f64_f32(a.data(), 1, o.data(), 1, 6, 1) == LC_OK, "cast one column");
    CHECK(lc_host_cast_f64_f32(a.data(), 2, o.data(), 3, 2, 3) != LC_OK, "cast must refuse ld < cols");
    std::vector<float> f = {1, 2, 3, 4, 5, 6}, g(4, 0.f);
    CHECK(lc_host_copy_f32(f.data(), 3, g.data(), 2, 2, 2) == LC_OK && g[0] == 1 && g[1] == 2 && g[2] == 4 && g[3] == 5, "copy");
    std::vector<double> col = {2.0, 4.0, 6.0, 8.0};
    std::vector<float> z(4);
    CHECK(lc_host_zscore_story(col.data(), LC_F64, 1, 4, 1, z.data(), 1) == LC_OK, "zs one column");
    std::vector<float> zr(4);
    ref_zs(col.data(), 1, 4, 1, zr.data(), 1);
    CHECK(memcmp(z.data(), zr.data(), 16) == 0, "zs one column values");
    std::vector<double> one = {5.0, -1.0, 2.0};
    CHECK(lc_host_zscore_story(one.data(), LC_F64, 3, 1, 3, z.data(), 3) == LC_OK && z[0] == 0.f && z[1] == 0.f && z[2] == 0.f,
          "zs of a one-row story: zero std, de-meaned only");
    CHECK(lc_host_zscore_story(one.data(), LC_F64, 3, 0, 3, z.data(), 3) != LC_OK, "zs must refuse an empty story");
    // lc_memcpy2d_async on a strided view
    hipStream_t s = nullptr;
    hipStreamCreate(&s);
    std::vector<float> big(5 * 7, 9.f), panel(5 * 3, 0.f);
    for (size_t i = 0; i < big.size(); ++i) big[i] = (float)i;
    CHECK(lc_memcpy2d_async(panel.data(), 12, big.data() + 2, 28, 12, 5, 1, s) == LC_OK, "memcpy2d");
    hipStreamSynchronize(s);
    for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 3; ++c) CHECK(panel[r * 3 + c] == big[r * 7 + 2 + c], "memcpy2d value");
    CHECK(lc_memcpy2d_async(panel.data(), 8, big.data(), 28, 12, 5, 1, s) != LC_OK, "memcpy2d must refuse pitch < width");
    // the per-class event timers of lc_core.hip (mutex-guarded vectors touched from several threads)
    lc_timing_enable(1);
    std::thread t1([&] { for (int i = 0; i < 50; ++i) lc_memcpy2d_async(panel.data(), 12, big.data(), 28, 12, 5, 1, s); });
    for (int i = 0; i < 50; ++i) lc_memcpy2d_async(panel.data(), 12, big.data(), 28, 12, 5, 1, s);
    t1.join();
    hipStreamSynchronize(s);
    lc_timing_enable(0);
    hipStreamDestroy(s);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    edge_views();
    stub_stream_jitter_us(30);
    unsigned seed = 1;
    for (int rep = 0; rep < reps; ++rep)
        for (int zs = 0; zs < 2; ++zs)
            for (int f32 = 0; f32 < 2; ++f32)
                for (int dev = 0; dev < 2; ++dev) {
                    // few small slots: the ring turns many times and a z-scored chunk is split into several tasks
                    run_matrix(6, 3, zs ? (int64_t)(110 * 4 * 1100) : (int64_t)(64 << 10), dev, zs, f32, 0, true, seed++);
                    run_matrix(2, 12, (int64_t)(1 << 20), dev, zs, f32, 0, true, seed++);
                }
    // abandoned: nobody waits for a job; finish / free must still drain and release
    run_matrix(4, 3, (int64_t)(64 << 10), true, false, false, 0, false, seed++);
    run_matrix(4, 3, (int64_t)(110 * 4 * 1100), false, true, false, 0, false, seed++);
    // injected failures at different points of the job list
    for (long n : {1L, 3L, 7L, 12L}) {
        run_matrix(5, 3, (int64_t)(64 << 10), true, false, false, n, true, seed++);
        run_matrix(5, 3, (int64_t)(110 * 4 * 1100), false, true, true, n, true, seed++);
        run_matrix(5, 3, (int64_t)(64 << 10), false, false, true, n, false, seed++);
    }
    if (g_fail) {
        fprintf(stderr, "%d check(s) failed\n", g_fail);
        return 1;
    }
    printf("host upload matrix ok: %ld copies submitted through the stub\n", stub_copies_submitted());
    return 0;
}
