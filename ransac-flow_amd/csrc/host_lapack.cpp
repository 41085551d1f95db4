// host_lapack.cpp -- librfxhost.so: the host side of the exact RANSAC mode (include/rfx_host_api.h).
//
// utils/outil.py:68-87 solves every 4-point DLT system with np.linalg.svd, i.e. with dgesdd of the LAPACK numpy is linked
// to.  The rank-deficient systems the device flags are re-solved here by that very routine (resolved through numpy's own
// extension module, see rfx_host_lapack_bind), called like numpy's svd gufunc calls it, on a persistent pool of
// std::threads.  Plain host C++: no HIP, no Python.  Compiled with -ffp-contract=off: the float32 products of the system
// must round exactly like the reference's elementwise float32 multiplications.
#include "../../include/rfx_host_api.h"

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// dgesdd(JOBZ, M, N, A, LDA, S, U, LDU, VT, LDVT, WORK, LWORK, IWORK, INFO) with 32- or 64-bit Fortran INTEGERs
using gesdd32_t = void (*)(const char*, const int32_t*, const int32_t*, double*, const int32_t*, double*, double*, const int32_t*,
                           double*, const int32_t*, double*, const int32_t*, int32_t*, int32_t*);
using gesdd64_t = void (*)(const char*, const int64_t*, const int64_t*, double*, const int64_t*, double*, double*, const int64_t*,
                           double*, const int64_t*, double*, const int64_t*, int64_t*, int64_t*);

void* g_fn = nullptr;
int g_bits = 0;
std::string g_symbol;
int64_t g_lwork = 0;          // numpy's choice: the optimum a work-size query reports for (jobz 'A', 8, 9)
std::mutex g_bind_mutex;
// The BLAS's own thread pool: a threaded OpenBLAS serialises concurrent callers while its thread count is > 1 (measured: 8
// caller threads, 41 us per solve and thread instead of 11), and an 8x9 problem never uses more than one of its threads anyway.
// Where the library exports a get / set pair, a parallel region sets the count to 1 and restores it afterwards.
int (*g_blas_get_threads)() = nullptr;
void (*g_blas_set_threads)(int) = nullptr;

struct BlasSingleThread {
    int prev = 0;
    BlasSingleThread() {
        if (g_blas_get_threads && g_blas_set_threads) { prev = g_blas_get_threads(); if (prev > 1) g_blas_set_threads(1); }
    }
    ~BlasSingleThread() { if (prev > 1) g_blas_set_threads(prev); }
};

constexpr int M = 8, N = 9;

// one call, numpy's argument set (umath_linalg.cpp init_gesdd / call_gesdd): LDA = LDU = M, LDVT = N
inline int64_t call_gesdd(double* a, double* s, double* u, double* vt, double* work, int64_t lwork, void* iwork) {
    const char jobz = 'A';
    if (g_bits == 64) {
        const int64_t m = M, n = N, lda = M, ldu = M, ldvt = N;
        int64_t info = 0;
        reinterpret_cast<gesdd64_t>(g_fn)(&jobz, &m, &n, a, &lda, s, u, &ldu, vt, &ldvt, work, &lwork, static_cast<int64_t*>(iwork), &info);
        return info;
    }
    const int32_t m = M, n = N, lda = M, ldu = M, ldvt = N, lw = (int32_t)lwork;
    int32_t info = 0;
    reinterpret_cast<gesdd32_t>(g_fn)(&jobz, &m, &n, a, &lda, s, u, &ldu, vt, &ldvt, work, &lw, static_cast<int32_t*>(iwork), &info);
    return info;
}

// numpy keeps A | S | U | VT | IWORK in one allocation and WORK in a second one, reused over the matrices of a call; each
// worker thread keeps its own pair for its lifetime
struct Scratch {
    double* buf = nullptr;
    double* work = nullptr;
    double *a, *s, *u, *vt;
    void* iwork;
    Scratch() {
        const size_t nd = M * N + M + M * M + N * N;
        buf = static_cast<double*>(std::malloc(nd * sizeof(double) + 8 * M * sizeof(int64_t)));
        a = buf; s = a + M * N; u = s + M; vt = u + M * M; iwork = vt + N * N;
        work = static_cast<double*>(std::malloc((size_t)(g_lwork > 0 ? g_lwork : 1) * sizeof(double)));
    }
    ~Scratch() { std::free(buf); std::free(work); }
};

// utils/outil.py:72-84 for one sample: the system in column-major order (what numpy's linearize_matrix hands to LAPACK),
// products rounded to float32 first, then row 8 of Vh
inline int64_t solve_one(const float* r, Scratch& S, float* Hf, double* Hd) {
    double* a = S.a;
    for (int i = 0; i < M * N; ++i) a[i] = 0.0;
    for (int i = 0; i < 4; ++i) {
        const float u_ = r[2 * i], v_ = r[2 * i + 1];              // source point (X)
        const float u = r[8 + 2 * i], v = r[8 + 2 * i + 1];        // target point (Y)
        const float nu = -u, nv = -v, nu_ = -u_;
        const float p0 = v_ * u, p1 = v_ * v, p2 = nu_ * u, p3 = nu_ * v;
        const int r0 = 2 * i, r1 = 2 * i + 1;
        // row 2i:   [0, 0, 0, -u, -v, -1, v'u, v'v, v']
        a[r0 + M * 3] = (double)nu; a[r0 + M * 4] = (double)nv; a[r0 + M * 5] = -1.0;
        a[r0 + M * 6] = (double)p0; a[r0 + M * 7] = (double)p1; a[r0 + M * 8] = (double)v_;
        // row 2i+1: [u, v, 1, 0, 0, 0, -u'u, -u'v, -u']
        a[r1 + M * 0] = (double)u; a[r1 + M * 1] = (double)v; a[r1 + M * 2] = 1.0;
        a[r1 + M * 6] = (double)p2; a[r1 + M * 7] = (double)p3; a[r1 + M * 8] = (double)nu_;
    }
    const int64_t info = call_gesdd(a, S.s, S.u, S.vt, S.work, g_lwork, S.iwork);
    for (int j = 0; j < N; ++j) {
        const double h = S.vt[8 + N * j];                          // Vh[8, j] of the column-major VT
        if (Hd) Hd[j] = h;
        if (Hf) Hf[j] = (float)h;
    }
    return info;
}

// ---------------------------------------------------------------- a small persistent pool: parallel_for over item blocks
class Pool {
  public:
    explicit Pool(int n) { resize(n); }
    ~Pool() { stop(); }
    int size() const { return (int)workers_.size() + 1; }
    void resize(int n) {
        if (n < 1) n = 1;
        if (n == size()) return;
        stop();
        quit_ = false;
        for (int i = 0; i < n - 1; ++i) workers_.emplace_back([this] { loop(); });
    }
    // fn(begin, end) over [0, total) in blocks of `block`, dealt dynamically; the caller works too and returns when all is done
    void run(int64_t total, int64_t block, const std::function<void(int64_t, int64_t)>& fn) {
        if (total <= 0) return;
        std::lock_guard<std::mutex> serial(run_mutex_);            // one parallel region at a time
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; total_ = total; block_ = block; next_.store(0); pending_ = (int)workers_.size(); ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            const int64_t b = next_.fetch_add(block_);
            if (b >= total_) break;
            (*fn_)(b, b + block_ < total_ ? b + block_ : total_);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
            }
            work();
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_mutex_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int64_t, int64_t)>* fn_ = nullptr;
    std::atomic<int64_t> next_{0};
    int64_t total_ = 0, block_ = 1;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};

Pool* g_pool = nullptr;
std::mutex g_pool_mutex;

int default_threads() {
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 1;
    return (int)(hc < 64 ? hc : 64);
}

Pool& pool() {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (!g_pool) g_pool = new Pool(default_threads());
    return *g_pool;
}

inline uint64_t hash_row(const float* r) {
    uint64_t w[8];
    std::memcpy(w, r, 64);
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 8; ++i) {
        h ^= w[i] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
    }
    return h;
}

constexpr int64_t DEDUPE_BLOCK = 1024;

}  // namespace

extern "C" int rfx_host_lapack_bind(const char* module_path) {
    std::lock_guard<std::mutex> lk(g_bind_mutex);
    if (g_fn) return g_bits;
    if (!module_path) return RFX_HOST_E_ARG;
    void* h = dlopen(module_path, RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen(module_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return RFX_HOST_E_ARG;
    static const struct { const char* name; int bits; } cand[] = {
        {"scipy_dgesdd_64_", 64}, {"dgesdd_64_", 64}, {"dgesdd64_", 64}, {"scipy_dgesdd_", 32}, {"dgesdd_", 32}};
    for (const auto& c : cand) {
        void* f = dlsym(h, c.name);
        if (!f) continue;
        g_fn = f; g_bits = c.bits; g_symbol = c.name;
        break;
    }
    if (!g_fn) return RFX_HOST_E_ARG;
    static const struct { const char *get, *set; } thr[] = {
        {"scipy_openblas_get_num_threads64_", "scipy_openblas_set_num_threads64_"},
        {"openblas_get_num_threads64_", "openblas_set_num_threads64_"},
        {"scipy_openblas_get_num_threads", "scipy_openblas_set_num_threads"},
        {"openblas_get_num_threads", "openblas_set_num_threads"}};
    for (const auto& t : thr) {
        void *g = dlsym(h, t.get), *s = dlsym(h, t.set);
        if (!g || !s) continue;
        g_blas_get_threads = reinterpret_cast<int (*)()>(g);
        g_blas_set_threads = reinterpret_cast<void (*)(int)>(s);
        break;
    }
    // numpy's work-size query (LWORK = -1), truncated to an integer like its (fortran_int) cast
    double a[M * N] = {0}, s[M], u[M * M], vt[N * N], q = 0.0;
    int64_t iw[8 * M];
    const int64_t info = call_gesdd(a, s, u, vt, &q, -1, iw);
    g_lwork = info == 0 ? (int64_t)q : 0;
    if (g_lwork <= 0) g_lwork = 1;       // numpy: "fix a bug in lapack 3.0.0"
    return g_bits;
}

extern "C" const char* rfx_host_lapack_symbol(void) { return g_symbol.c_str(); }

extern "C" int rfx_host_set_threads(int n) {
    Pool& p = pool();
    if (n > 0) {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        p.resize(n);
    }
    return p.size();
}

extern "C" int rfx_host_dlt_null_vectors(const float* xy, int64_t k, float* H_out, double* hv_out, int dedupe,
                                         int64_t* n_solved_out) {
    if (k < 0 || (k > 0 && !xy) || (!H_out && !hv_out && k > 0)) return RFX_HOST_E_ARG;
    if (!g_fn) return RFX_HOST_E_UNBOUND;
    if (n_solved_out) *n_solved_out = 0;
    if (k == 0) return RFX_HOST_OK;
    std::atomic<int64_t> bad{0};
    auto out_f = [&](int64_t i) { return H_out ? H_out + i * 9 : nullptr; };
    auto out_d = [&](int64_t i) { return hv_out ? hv_out + i * 9 : nullptr; };
    if (!dedupe || k < 64) {
        auto body = [&](int64_t b, int64_t e) {
            thread_local Scratch S;
            int64_t nb = 0;
            for (int64_t i = b; i < e; ++i) nb += solve_one(xy + i * 16, S, out_f(i), out_d(i)) != 0;
            if (nb) bad += nb;
        };
        if (k < 64) body(0, k); else { BlasSingleThread one; pool().run(k, 16, body); }
        if (n_solved_out) *n_solved_out = k;
        return bad.load() ? RFX_HOST_E_LAPACK : RFX_HOST_OK;
    }
    BlasSingleThread one;
    // phase A: inside every block of DEDUPE_BLOCK rows, rep[i] = first row of the block with the same 64 bytes
    std::vector<int64_t> rep((size_t)k);
    const int64_t nblk = (k + DEDUPE_BLOCK - 1) / DEDUPE_BLOCK;
    std::vector<int64_t> uniq((size_t)k);                 // block b's representatives at uniq[b*DEDUPE_BLOCK ...]
    std::vector<int32_t> nuniq((size_t)nblk);
    pool().run(nblk, 1, [&](int64_t b0, int64_t b1) {
        constexpr int TS = 4096;                           // open addressing, load <= 0.25
        thread_local std::vector<int32_t> table;
        for (int64_t b = b0; b < b1; ++b) {
            table.assign(TS, -1);
            const int64_t lo = b * DEDUPE_BLOCK, hi = lo + DEDUPE_BLOCK < k ? lo + DEDUPE_BLOCK : k;
            int32_t nu = 0;
            for (int64_t i = lo; i < hi; ++i) {
                const float* r = xy + i * 16;
                uint32_t slot = (uint32_t)hash_row(r) & (TS - 1);
                for (;;) {
                    const int32_t t = table[slot];
                    if (t < 0) { table[slot] = (int32_t)(i - lo); rep[i] = i; uniq[lo + nu++] = i; break; }
                    if (std::memcmp(xy + (lo + t) * 16, r, 64) == 0) { rep[i] = lo + t; break; }
                    slot = (slot + 1) & (TS - 1);
                }
            }
            nuniq[b] = nu;
        }
    });
    // phase B: one dgesdd per representative, dealt over the pool
    std::vector<int64_t> todo;
    todo.reserve((size_t)k);
    for (int64_t b = 0; b < nblk; ++b)
        for (int32_t j = 0; j < nuniq[b]; ++j) todo.push_back(uniq[b * DEDUPE_BLOCK + j]);
    const int64_t nt = (int64_t)todo.size();
    pool().run(nt, 8, [&](int64_t b, int64_t e) {
        thread_local Scratch S;
        int64_t nb = 0;
        for (int64_t j = b; j < e; ++j) { const int64_t i = todo[j]; nb += solve_one(xy + i * 16, S, out_f(i), out_d(i)) != 0; }
        if (nb) bad += nb;
    });
    // phase C: the duplicates take their representative's bits
    pool().run(k, 4096, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            const int64_t r = rep[i];
            if (r == i) continue;
            if (H_out) std::memcpy(H_out + i * 9, H_out + r * 9, 9 * sizeof(float));
            if (hv_out) std::memcpy(hv_out + i * 9, hv_out + r * 9, 9 * sizeof(double));
        }
    });
    if (n_solved_out) *n_solved_out = nt;
    return bad.load() ? RFX_HOST_E_LAPACK : RFX_HOST_OK;
}
