// a-13: the two link files of the path as `write.table(x, file, append = T, quote = F, row.names = F, col.names = F,
// sep = '\t')` prints them (R/computePairwiseMI.R:140, :362; readers R/io_functions.R:32-66).  Host threads format
// disjoint row ranges into private buffers, one writer appends them in order.
//
// Number rule (R's formatReal with digits = 15, scipen = 0, one element at a time as write.table encodes a data.frame):
// an integer column prints its digits; a double prints the fewest significant digits (<= 15) that reproduce its 15-digit
// value, in fixed notation unless that is wider than the scientific form (so 100000 prints as 1e+05 and 0.0001 as 1e-04).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <vector>

#include "ldw_internal.h"

namespace {

// The 15 significant decimal digits of ax in [1e-5, 1e3), correctly rounded (ties to even on the exact binary value, like
// printf("%.14e")), by exact 128-bit integer arithmetic: ax = m 2^e, digits = round(m 10^(14 - e10) 2^e).  MI values live here;
// snprintf costs ~250 ns per value, this ~40.  Returns false when the shift does not fit (never in the stated range).
inline bool fast15(double ax, char *sig, int *e10_out) {
    static const unsigned long long P10[20] = {1ULL, 10ULL, 100ULL, 1000ULL, 10000ULL, 100000ULL, 1000000ULL, 10000000ULL, 100000000ULL,
                                               1000000000ULL, 10000000000ULL, 100000000000ULL, 1000000000000ULL, 10000000000000ULL,
                                               100000000000000ULL, 1000000000000000ULL, 10000000000000000ULL, 100000000000000000ULL,
                                               1000000000000000000ULL, 10000000000000000000ULL};
    int ex;
    const double fr = std::frexp(ax, &ex);                       // ax = fr 2^ex, fr in [0.5, 1)
    const unsigned long long m = (unsigned long long)std::ldexp(fr, 53);   // 53-bit integer mantissa
    const int e = ex - 53;                                        // ax = m 2^e, e < 0 here (ax < 1e3 < 2^53)
    int e10 = (int)std::floor(std::log10(ax));
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int k = 14 - e10;
        if (k < 0 || k > 19 || -e <= 0 || -e >= 126) return false;
        const unsigned __int128 prod = (unsigned __int128)m * P10[k];
        const int sh = -e;
        unsigned __int128 q = prod >> sh;
        // the decimal exponent is decided on the quotient BEFORE rounding (with floor(log10) one too high, a value just below
        // a power of ten rounded up to exactly 10^14 and passed for 1.00000000000000e(e10): 9.999999999999994e-05 printed as 1e-04)
        if (q >= (unsigned __int128)P10[15]) { ++e10; continue; }   // log10 was one low
        if (q < (unsigned __int128)P10[14]) { --e10; continue; }    // log10 was one high
        const unsigned __int128 rem = prod - (q << sh), half = (unsigned __int128)1 << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;
        if (q == (unsigned __int128)P10[15]) {                      // the rounding carried into a 16th digit: 9.99..95 -> 1.00..0 e(e10 + 1)
            q = (unsigned __int128)P10[14];
            ++e10;
        }
        unsigned long long d = (unsigned long long)q;
        for (int i = 14; i >= 0; --i) { sig[i] = (char)('0' + d % 10); d /= 10; }
        *e10_out = e10;
        return true;
    }
    return false;
}

// appends one double; buf has >= 40 bytes of room
inline char *fmt_double(char *p, double x) {
    if (x != x) { memcpy(p, "NA", 2); return p + 2; }
    if (std::isinf(x)) {
        if (x > 0) { memcpy(p, "Inf", 3); return p + 3; }
        memcpy(p, "-Inf", 4);
        return p + 4;
    }
    if (x == 0.0) { *p++ = '0'; return p; }
    const bool neg = x < 0;
    const double ax = neg ? -x : x;
    char sig[24];
    int nsig, e10;
    if (ax < 9.0e14 && ax == std::floor(ax)) {
        // integral value (every pos / len / cluster id): its decimal digits are exact, no rounding step
        unsigned long long v = (unsigned long long)ax;
        char tmp[24];
        int n = 0;
        while (v) { tmp[n++] = (char)('0' + v % 10); v /= 10; }
        for (int k = 0; k < n; ++k) sig[k] = tmp[n - 1 - k];
        e10 = n - 1;
        nsig = n;
        while (nsig > 1 && sig[nsig - 1] == '0') --nsig;
    } else if (ax < 1.0e13 && 2.0 * ax == std::floor(2.0 * ax)) {
        // k + 1/2 (every len when the genome length is odd): integer digits, then a 5
        unsigned long long v = (unsigned long long)ax;
        char tmp[24];
        int n = 0;
        while (v) { tmp[n++] = (char)('0' + v % 10); v /= 10; }
        for (int k = 0; k < n; ++k) sig[k] = tmp[n - 1 - k];
        sig[n] = '5';
        nsig = n + 1;
        e10 = n - 1;   // (n == 0: 0.5 = 5e-01)
    } else if (ax >= 1e-5 && ax < 1e3 && fast15(ax, sig, &e10)) {
        nsig = 15;
        while (nsig > 1 && sig[nsig - 1] == '0') --nsig;
    } else {
        char m[40];
        snprintf(m, sizeof(m), "%.14e", ax);   // d.dddddddddddddde[+-]xx
        sig[0] = m[0];
        memcpy(sig + 1, m + 2, 14);
        nsig = 15;
        while (nsig > 1 && sig[nsig - 1] == '0') --nsig;
        e10 = atoi(m + 17);
    }
    const int wexp = (e10 < 100 && e10 > -100) ? 2 : 3;
    const int w_sci = (neg ? 1 : 0) + (nsig > 1 ? nsig + 1 : 1) + 2 + wexp;
    const int rgt = nsig - e10 - 1 > 0 ? nsig - e10 - 1 : 0;
    const int left = e10 >= 0 ? e10 + 1 : 1;
    const int w_fix = (neg ? 1 : 0) + left + (rgt ? rgt + 1 : 0);
    if (neg) *p++ = '-';
    if (w_fix <= w_sci) {
        // fixed: the nsig digits placed around the point (the value rounded to rgt decimals has exactly these digits)
        if (e10 >= 15) {
            // more integer digits than the 15 significant ones: sprintf("%.0f") prints the double's own digits there
            p += snprintf(p, 36, "%.0f", ax);
        } else if (e10 >= 0) {
            for (int k = 0; k <= e10; ++k) *p++ = k < nsig ? sig[k] : '0';
            if (rgt) {
                *p++ = '.';
                for (int k = e10 + 1; k < nsig; ++k) *p++ = sig[k];
            }
        } else {
            *p++ = '0';
            *p++ = '.';
            for (int k = 0; k < -e10 - 1; ++k) *p++ = '0';
            for (int k = 0; k < nsig; ++k) *p++ = sig[k];
        }
        return p;
    }
    *p++ = sig[0];
    if (nsig > 1) {
        *p++ = '.';
        memcpy(p, sig + 1, (size_t)nsig - 1);
        p += nsig - 1;
    }
    *p++ = 'e';
    *p++ = e10 >= 0 ? '+' : '-';
    int ae = e10 >= 0 ? e10 : -e10;
    if (wexp == 3) { *p++ = (char)('0' + ae / 100); ae %= 100; }
    *p++ = (char)('0' + ae / 10);
    *p++ = (char)('0' + ae % 10);
    return p;
}

inline char *fmt_int(char *p, long long v) {
    if (v < 0) { *p++ = '-'; v = -v; }   // (INT64_MIN is not a value any column of the path can hold)
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

struct Col {
    int kind;          // LDW_COL_INT32 / INT64 / DOUBLE
    const void *data;
};

// default worker count: the machine's hardware threads, at most 16 (a container's CPU share is usually far below the host's count,
// and the formatting saturates the file write well before that)
// Big host buffers of the writer (thread buffers, derived columns) come from a small process-wide pool and go back to it instead of to the
// OS: see ldw_write_links_tsv on what an munmap next to GPU work can cost.  Blocks are kept for the life of the process (a C4 job: ~100 MB).
class HostPool {
public:
    void *get(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu_);
        size_t best = SIZE_MAX;
        // (best fit, but never a block more than twice the request: a 256-MB block handed to a small request forced a second large allocation later — ADVICE r04)
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= bytes && free_[i].second <= 2 * bytes + 4096 && (best == SIZE_MAX || free_[i].second < free_[best].second)) best = i;
        if (best != SIZE_MAX) {
            void *p = free_[best].first;
            used_.push_back(free_[best]);
            free_.erase(free_.begin() + (long)best);
            return p;
        }
        void *p = malloc(bytes);
        if (p) used_.push_back({p, bytes});
        return p;
    }
    void put(void *p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t i = 0; i < used_.size(); ++i)
            if (used_[i].first == p) {
                free_.push_back(used_[i]);
                used_.erase(used_.begin() + (long)i);
                return;
            }
    }
    // gives the idle blocks back to the OS (ldw_host_trim); returns the bytes released
    size_t trim() {
        std::lock_guard<std::mutex> lk(mu_);
        size_t n = 0;
        for (auto &b : free_) {
            n += b.second;
            free(b.first);
        }
        free_.clear();
        return n;
    }
private:
    std::mutex mu_;
    std::vector<std::pair<void *, size_t>> free_, used_;
};
HostPool &host_pool() {
    static HostPool *p = new HostPool;   // (never destroyed: threads may still hold blocks at exit)
    return *p;
}
struct PoolBlock {
    void *p = nullptr;
    bool pooled = true;
    // (blocks beyond 256 MB — the full short-range table of a 500k-SNP alignment would need 72 GB — are ordinary allocations: kept they would
    // pin that memory for the life of the process; the stall they may cost the next GPU call is the lesser evil there)
    explicit PoolBlock(size_t bytes) : pooled(bytes <= ((size_t)256 << 20)) { p = pooled ? host_pool().get(bytes) : malloc(bytes); }
    ~PoolBlock() {
        if (pooled) host_pool().put(p);
        else free(p);
    }
    PoolBlock(const PoolBlock &) = delete;
    PoolBlock &operator=(const PoolBlock &) = delete;
    template <class T> T *as() const { return static_cast<T *>(p); }
};

int default_threads(int nthreads) {
    if (nthreads > 0) return nthreads < 64 ? nthreads : 64;
    if (const char *e = getenv("LDW_TSV_THREADS")) {   // (experiments)
        const int k = atoi(e);
        if (k > 0) return k < 64 ? k : 64;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    int n = (int)(hw ? hw : 4);
    // r04: the cgroup's CPU quota, when there is one (a container sees the host's CPU count; workers beyond the quota get the whole process
    // throttled for the rest of the scheduling period — seen as a 15-25 ms stall in the NEXT call of a job)
    if (FILE *fh = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32];
        long long period = 0;
        if (fscanf(fh, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long share = atoll(q) / period;
            if (share >= 1 && share < n) n = (int)share;
        }
        fclose(fh);
    }
    if (n > 4) n -= 2;   // (room for the calling thread's other duties and the HIP runtime's own threads)
    return n < 16 ? n : 16;
}

int write_rows(const char *path, int append, int64_t nrows, const std::vector<Col> &cols, int nthreads, int64_t *bytes_out) {
    // r04: the file is written by the workers themselves with pwrite at offsets that follow from the chunk sizes (the one calling thread's
    // fwrite of 45 MB — a copy into the page cache at 2-3 GB/s — was most of the 19 ms an lr_links.tsv of 1e6 rows took: the formatting had long
    // been parallel).  Rounds of nt chunks: every worker formats its chunk into its own buffer, the last one to finish turns the
    // sizes of the round into offsets, every worker writes its own chunk and goes on to the next round.
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666);
    LDW_REQUIRE(fd >= 0, LDW_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
    int64_t base = 0;
    if (append) {
        base = (int64_t)lseek(fd, 0, SEEK_END);
        if (base < 0) {
            ldw::set_error("cannot seek in %s: %s", path, strerror(errno));
            close(fd);
            return LDW_ERR_ARG;
        }
    }
    int64_t total = 0;
    int rc = LDW_OK;
    if (nrows > 0) {
        // one chunk per worker and round: the whole share of a worker when its buffer stays below 64 MB (one barrier for a table of a
        // million rows instead of eight: the barrier waits were 3 of the 13 ms), at least 8192 rows
        const size_t row_max = cols.size() * 41 + 2;
        int nt = default_threads(nthreads);
        nt = (int)std::min<int64_t>(nt, (nrows + 8191) / 8192);
        if (nt < 1) nt = 1;
        const int64_t cap_rows = std::max<int64_t>(8192, (int64_t)(((size_t)64 << 20) / row_max));
        const int64_t chunk = std::min<int64_t>((nrows + nt - 1) / nt, cap_rows);
        const int64_t nchunks = (nrows + chunk - 1) / chunk;
        const int64_t rounds = (nchunks + nt - 1) / nt;
        std::vector<std::unique_ptr<PoolBlock>> bufs((size_t)nt);
        std::vector<size_t> used((size_t)nt, 0);
        std::vector<int64_t> at((size_t)nt, 0);
        std::mutex mu;
        std::condition_variable cv;
        int arrived = 0;           // workers that have formatted their chunk of the current round
        int64_t round_open = -1;   // the last round whose offsets are known
        int64_t next_off = base;
        bool failed = false;
        static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
        std::vector<double> t_fmt((size_t)nt, 0.0), t_wait((size_t)nt, 0.0), t_wr((size_t)nt, 0.0);
        auto worker = [&](int t) {
            bufs[(size_t)t].reset(new PoolBlock((size_t)chunk * row_max));   // (uninitialised; first touched by the thread that fills it)
            if (!bufs[(size_t)t]->p) {
                std::lock_guard<std::mutex> lk(mu);
                if (!failed) ldw::set_error("out of host memory for the tsv writer's buffers");
                failed = true;
            }
            for (int64_t r = 0; r < rounds; ++r) {
                const auto w0 = std::chrono::steady_clock::now();
                const int64_t c = r * nt + t;
                size_t u = 0;
                if (c < nchunks && bufs[(size_t)t]->p) {
                    const int64_t a = c * chunk, b = std::min(nrows, a + chunk);
                    char *p = bufs[(size_t)t]->as<char>();
                    for (int64_t i = a; i < b; ++i) {
                        for (size_t k = 0; k < cols.size(); ++k) {
                            if (k) *p++ = '\t';
                            const Col &cl = cols[k];
                            if (cl.kind == LDW_COL_DOUBLE) p = fmt_double(p, static_cast<const double *>(cl.data)[i]);
                            else if (cl.kind == LDW_COL_INT32) p = fmt_int(p, static_cast<const int32_t *>(cl.data)[i]);
                            else p = fmt_int(p, static_cast<const int64_t *>(cl.data)[i]);
                        }
                        *p++ = '\n';
                    }
                    u = (size_t)(p - bufs[(size_t)t]->as<char>());
                }
                const auto w1 = std::chrono::steady_clock::now();
                {
                    std::unique_lock<std::mutex> lk(mu);
                    used[(size_t)t] = u;
                    if (++arrived == nt) {   // the round is complete: offsets in chunk (= row) order
                        for (int k = 0; k < nt; ++k) {
                            at[(size_t)k] = next_off;
                            next_off += (int64_t)used[(size_t)k];
                        }
                        arrived = 0;
                        round_open = r;
                        cv.notify_all();
                    } else {
                        cv.wait(lk, [&] { return round_open >= r; });
                    }
                }
                const int64_t my_at = at[(size_t)t];
                const auto w2 = std::chrono::steady_clock::now();
                size_t w = 0;
                while (w < u) {
                    const ssize_t k = pwrite(fd, bufs[(size_t)t]->as<char>() + w, u - w, (off_t)(my_at + (int64_t)w));
                    if (k < 0) {
                        if (errno == EINTR) continue;
                        std::lock_guard<std::mutex> lk(mu);
                        if (!failed) ldw::set_error("short write to %s: %s", path, strerror(errno));
                        failed = true;
                        break;
                    }
                    w += (size_t)k;
                }
                // (at[] of round r + 1 is written only after every worker has arrived there, i.e. after it has read its offset of round r)
                if (host_timing) {
                    const auto w3 = std::chrono::steady_clock::now();
                    t_fmt[(size_t)t] += std::chrono::duration<double, std::milli>(w1 - w0).count();
                    t_wait[(size_t)t] += std::chrono::duration<double, std::milli>(w2 - w1).count();
                    t_wr[(size_t)t] += std::chrono::duration<double, std::milli>(w3 - w2).count();
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(worker, t);
        worker(0);
        for (auto &x : th) x.join();
        total = next_off - base;
        if (failed) rc = LDW_ERR_ARG;
        if (host_timing)
            fprintf(stderr, "[ldw] tsv writer: %d threads, %lld rounds; thread 0: format %.2f ms, barrier %.2f, pwrite %.2f; max over threads: %.2f / %.2f / %.2f\n", nt,
                    (long long)rounds, t_fmt[0], t_wait[0], t_wr[0], *std::max_element(t_fmt.begin(), t_fmt.end()), *std::max_element(t_wait.begin(), t_wait.end()),
                    *std::max_element(t_wr.begin(), t_wr.end()));
    }
    if (close(fd) != 0 && rc == LDW_OK) {
        ldw::set_error("closing %s: %s", path, strerror(errno));
        rc = LDW_ERR_ARG;
    }
    if (bytes_out) *bytes_out = total;
    return rc;
}

}  // namespace

extern "C" {

int ldw_format_number(double x, char *out, int capacity) {
    LDW_REQUIRE(out && capacity >= 48, LDW_ERR_SIZE, "ldw_format_number: capacity must be at least 48 bytes");
    char *e = fmt_double(out, x);
    *e = 0;
    return LDW_OK;
}

int ldw_write_table_tsv(const char *path, int append, int64_t nrows, int ncols, const int32_t *col_kind, const void *const *cols,
                        int nthreads, int64_t *bytes_out) {
    LDW_REQUIRE(path && nrows >= 0 && ncols > 0 && ncols <= 64 && col_kind && cols, LDW_ERR_ARG, "ldw_write_table_tsv: bad argument");
    std::vector<Col> cc((size_t)ncols);
    for (int k = 0; k < ncols; ++k) {
        LDW_REQUIRE(col_kind[k] >= LDW_COL_INT32 && col_kind[k] <= LDW_COL_DOUBLE, LDW_ERR_ARG, "ldw_write_table_tsv: column %d has kind %d", k,
                    col_kind[k]);
        LDW_REQUIRE(cols[k] || nrows == 0, LDW_ERR_ARG, "ldw_write_table_tsv: column %d is null", k);
        cc[(size_t)k] = Col{col_kind[k], cols[k]};
    }
    return write_rows(path, append, nrows, cc, nthreads, bytes_out);
}

// One lr_links.tsv / sr_links.tsv job: the synchronous half (count, fetch into the pinned arena) and the host half (derive the reference's
// columns, format, write), which ldw_write_links_tsv_begin runs on a thread of its own beside the caller's next GPU work.
struct LinksTsvJob {
    int64_t n = 0;
    int32_t *a = nullptr, *b = nullptr;
    double *mi = nullptr;
    std::unique_ptr<int32_t[]> a_own, b_own;
    std::unique_ptr<double[]> mi_own;
    std::unique_ptr<PoolBlock> derived;
    std::string path;
    int append = 0, nthreads = 0;
    double g = 0;
    const int32_t *POS = nullptr, *paint = nullptr;
    double fetch_ms = 0;
};

static int links_tsv_prepare(ldw_ctx *c, int which, const char *path, int append, int nthreads, LinksTsvJob &J) {
    if (int rc = ldw::check_gpu(c)) return rc;
    LDW_REQUIRE(path && (which == 0 || which == 1), LDW_ERR_ARG, "ldw_write_links_tsv: bad argument");
    LDW_REQUIRE(c->have_meta && (int64_t)c->h_POS.size() == c->L && (int64_t)c->h_paint.size() == c->L, LDW_ERR_STATE,
                "ldw_write_links_tsv: SNP meta data (POS, paint, g) not set");
    int64_t n = 0;
    if (int rc = ldw_links_count(c, which, &n)) return rc;
    J.n = n;
    J.path = path;
    J.append = append;
    J.nthreads = nthreads;
    J.g = c->g;
    J.POS = c->h_POS.data();
    J.paint = c->h_paint.data();
    if (n == 0) return LDW_OK;   // the reference writes nothing for an empty frame (R/computePairwiseMI.R:360)
    // (uninitialised arrays: a std::vector would clear 48 bytes per row on the calling thread before anything is written)
    // The table itself (a, b, MI: 16 bytes per row) is fetched into a PINNED arena that stays with the context.  r04, measured
    // (tools/job_profile.py --cold with two switches that leaked the arrays / slept before the next call): a D2H copy into pageable memory makes
    // the runtime register those pages with the GPU for the DMA, and giving such memory back to the OS afterwards (arrays of this size are
    // mmap'ed, so delete[] is munmap) stalls the process's NEXT GPU call by ~20 ms — the driver quiesces the queues to drop the registration
    // and restores them a moment later; the short-range model's first stream synchronisation after lr_links.tsv paid it in every job.
    const size_t fetch_bytes = (size_t)n * 16;
    if (fetch_bytes <= ((size_t)2 << 30)) {
        if (c->pin_fetch_cap < fetch_bytes) {
            if (c->pin_fetch) (void)hipHostFree(c->pin_fetch);
            c->pin_fetch = nullptr;
            c->pin_fetch_cap = 0;
            const size_t want = fetch_bytes + fetch_bytes / 8 + 4096;
            if (hipHostMalloc(&c->pin_fetch, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                c->pin_fetch = nullptr;
            } else {
                c->pin_fetch_cap = want;
            }
        }
    }
    if (c->pin_fetch && c->pin_fetch_cap >= fetch_bytes) {
        J.mi = static_cast<double *>(c->pin_fetch);
        J.a = reinterpret_cast<int32_t *>(J.mi + n);
        J.b = J.a + n;
    } else {   // (tables beyond 2 GB, or no pinned memory to be had: pageable arrays)
        J.a_own.reset(new int32_t[(size_t)n]);
        J.b_own.reset(new int32_t[(size_t)n]);
        J.mi_own.reset(new double[(size_t)n]);
        J.a = J.a_own.get();
        J.b = J.b_own.get();
        J.mi = J.mi_own.get();
    }
    const auto t_0 = std::chrono::steady_clock::now();
    if (int rc = ldw_links_fetch(c, which, J.a, J.b, J.mi, n, 0)) return rc;
    J.fetch_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count();
    return LDW_OK;
}

// ---- r05: lr_links.tsv appended WHILE the pass runs (the reference appends per block: R/computePairwiseMI.R:362) ------------------------------
// The long-range table grows in block order on the device (k_sel_scatter appends at the running count).  After every finished item the
// submitting thread queues a copy of that count and an event (lr_stream_push); a writer thread of the context waits for the event, fetches
// the rows that are new since its last batch on a stream of its own, derives pos1 pos2 clust1 clust2 len, formats and appends them.  Rows of
// finished blocks are final (the filter is per block: :352-358), so what is on disk at any time is a prefix of the reference's file in
// whole blocks — a pass that fails at block k leaves the rows of blocks 0..k-1, like the reference's loop.
struct LrStream {
    static constexpr int RING = 64;
    ldw_ctx *c = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    hipStream_t st = nullptr;           // the context's lr_st / lr_ev / lr_counts (ensure_streams)
    hipEvent_t *ev = nullptr;
    int64_t *counts = nullptr;          // pinned [RING]
    int64_t pushed = 0, consumed = 0;   // ring entries handed over / fully written
    bool busy = false, closing = false;
    int drain_req = 0;                  // > 0: someone waits in lr_stream_drain: write what there is, whatever the batch size
    static constexpr int BATCH_ITEMS = 8;   // items per batch while the pass runs (every HIP call of this thread contends with the submitting thread's
                                            // ~600 launches per pass for the runtime's locks: a batch per item cost the C4 pass 5-12 ms, r05)
    int64_t rows = 0, bytes = 0;        // written so far
    int64_t blocks_seen = 0;
    std::string path, err;
    int append = 1, nthreads = 0, rc = LDW_OK;
    bool first_batch = true;
    bool closing_seen = false;          // (the writer's own copy of `closing`, read outside the lock)
};

static int write_link_rows(const char *path, int append, int nthreads, int64_t n, const int32_t *a, const int32_t *b, double *mi, const int32_t *POS,
                           const int32_t *paint, double g, int64_t *bytes_out);

static void lr_stream_batch(LrStream *S, int64_t hi) {
    ldw_ctx *c = S->c;
    const int64_t lo = S->rows, n = hi - lo;
    if (n <= 0 || S->rc != LDW_OK) return;
    const size_t need = (size_t)n * 16;
    if (c->lr_pin_cap < need) {
        if (c->lr_pin) (void)hipHostFree(c->lr_pin);
        c->lr_pin = nullptr;
        c->lr_pin_cap = 0;
        const size_t want = std::max<size_t>(need + need / 2, (size_t)8 << 20);
        if (hipHostMalloc(&c->lr_pin, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            S->rc = LDW_ERR_HIP;
            S->err = "lr stream: no pinned staging memory";
            return;
        }
        c->lr_pin_cap = want;
    }
    double *mi = static_cast<double *>(c->lr_pin);
    int32_t *a = reinterpret_cast<int32_t *>(mi + n), *b = a + n;
    // (the table's buffers cannot move under this copy: a growth of the long-range table drains the stream first — ensure_links_capacity)
    hipError_t e = hipMemcpyAsync(a, c->lr_a.as<int32_t>() + lo, (size_t)n * 4, hipMemcpyDeviceToHost, S->st);
    if (e == hipSuccess) e = hipMemcpyAsync(b, c->lr_b.as<int32_t>() + lo, (size_t)n * 4, hipMemcpyDeviceToHost, S->st);
    if (e == hipSuccess) e = hipMemcpyAsync(mi, c->lr_mi.as<double>() + lo, (size_t)n * 8, hipMemcpyDeviceToHost, S->st);
    if (e == hipSuccess) e = hipStreamSynchronize(S->st);
    if (e != hipSuccess) {
        S->rc = LDW_ERR_HIP;
        S->err = std::string("lr stream: fetching rows failed: ") + hipGetErrorString(e);
        return;
    }
    int64_t wrote = 0;
    // (while the pass runs: few threads — the submitting thread and its helpers need the host more; once the stream is closing the pass is over)
    const int rc = write_link_rows(S->path.c_str(), S->first_batch ? S->append : 1, S->closing_seen ? 0 : S->nthreads, n, a, b, mi, c->h_POS.data(), c->h_paint.data(), c->g, &wrote);
    S->first_batch = false;
    if (rc != LDW_OK) {
        S->rc = rc;
        S->err = ldw_last_error();
        return;
    }
    S->rows = hi;
    S->bytes += wrote;
}

static void lr_stream_main(LrStream *S) {
    (void)hipSetDevice(S->c->device);
    for (;;) {
        int64_t take_to;
        {
            std::unique_lock<std::mutex> lk(S->mu);
            S->cv.wait(lk, [&] { return S->closing || S->drain_req > 0 ? (S->closing || S->pushed > S->consumed) : S->pushed - S->consumed >= LrStream::BATCH_ITEMS; });
            if (S->pushed == S->consumed) {
                if (S->closing) return;   // closing and nothing left
                continue;
            }
            take_to = S->pushed;                    // everything handed over so far: one batch (the newest entry's count covers the older ones)
            S->busy = true;
            S->closing_seen = S->closing;
        }
        const int idx = (int)((take_to - 1) % LrStream::RING);
        int64_t hi = -1;
        if (hipEventSynchronize(S->ev[idx]) == hipSuccess) hi = S->counts[idx];
        else {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(S->mu);
            if (S->rc == LDW_OK) {
                S->rc = LDW_ERR_HIP;
                S->err = "lr stream: waiting for a finished item failed";
            }
        }
        try {
            if (hi >= 0) lr_stream_batch(S, hi);
        } catch (const std::exception &e) {
            S->rc = LDW_ERR_ARG;
            S->err = std::string("lr stream: ") + e.what();
        }
        {
            std::lock_guard<std::mutex> lk(S->mu);
            S->consumed = take_to;
            S->busy = false;
        }
        S->cv.notify_all();
    }
}

extern "C++" {
namespace ldw {
// called by the item loop after the selection of an item has been QUEUED on `s`: the running long-range row count as it will be once that
// selection has run, and an event behind it
void lr_stream_push(ldw_ctx *c, hipStream_t s, const int64_t *d_lr_count, int64_t blocks_done) {
    LrStream *S = static_cast<LrStream *>(c->lr_stream);
    if (!S) return;
    {
        std::unique_lock<std::mutex> lk(S->mu);
        S->cv.wait(lk, [&] { return S->pushed - S->consumed < LrStream::RING; });   // (the writer is a whole ring behind: wait for a slot)
    }
    const int idx = (int)(S->pushed % LrStream::RING);
    if (hipMemcpyAsync(&S->counts[idx], d_lr_count, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(S->ev[idx], s) != hipSuccess) {
        (void)hipGetLastError();
        return;   // (this item's rows go out with the next one's, or with the final push of ldw_lr_stream_end)
    }
    {
        std::lock_guard<std::mutex> lk(S->mu);
        ++S->pushed;
        S->blocks_seen = blocks_done;
    }
    S->cv.notify_all();
}
// everything handed over so far is on disk when this returns (before the long-range table's buffers are reallocated, and at the end)
void lr_stream_drain(ldw_ctx *c) {
    LrStream *S = static_cast<LrStream *>(c->lr_stream);
    if (!S) return;
    std::unique_lock<std::mutex> lk(S->mu);
    ++S->drain_req;
    S->cv.notify_all();
    S->cv.wait(lk, [&] { return S->pushed == S->consumed && !S->busy; });
    --S->drain_req;
}
}  // namespace ldw
}  // extern "C++"

// the rows (a, b, MI) of a link table as the reference's MI_df rows `pos1 pos2 clust1 clust2 len MI` (R/computePairwiseMI.R:319-331), appended to `path`
static int write_link_rows(const char *path, int append, int nthreads, int64_t n, const int32_t *a, const int32_t *b, double *mi, const int32_t *POS,
                           const int32_t *paint, double g, int64_t *bytes_out) {
    if (bytes_out) *bytes_out = 0;
    if (n == 0) return LDW_OK;
    static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
    const auto t_1 = std::chrono::steady_clock::now();
    PoolBlock derived((size_t)n * 32);   // pos1, pos2 (int32), clust1, clust2, len (double)
    LDW_REQUIRE(derived.p != nullptr, LDW_ERR_ARG, "ldw_write_links_tsv: out of host memory");
    double *c1 = derived.as<double>(), *c2 = c1 + n, *len = c2 + n;
    int32_t *pos1 = reinterpret_cast<int32_t *>(len + n), *pos2 = pos1 + n;
    const double hg = 0.5 * g;
    int nt = (int)std::min<int64_t>(default_threads(nthreads), (n + 65535) / 65536);
    if (nt < 1) nt = 1;
    auto derive = [&](int t) {
        const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
        for (int64_t i = i0; i < i1; ++i) {
            // pos1 = POS_t[col], pos2 = POS_f[row]; len = 0.5 g - |((pos1 - pos2) %% g) - 0.5 g| with R's floored %% (R/computePairwiseMI.R:319-330)
            const int32_t p1 = POS[b[(size_t)i]], p2 = POS[a[(size_t)i]];
            pos1[(size_t)i] = p1;
            pos2[(size_t)i] = p2;
            c1[(size_t)i] = (double)paint[b[(size_t)i]];
            c2[(size_t)i] = (double)paint[a[(size_t)i]];
            const double d = (double)p1 - (double)p2;
            double m = std::fmod(d, g);
            if (m != 0.0 && ((m < 0) != (g < 0))) m += g;
            len[(size_t)i] = hg - std::fabs(m - hg);
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(derive, t);
        derive(0);
        for (auto &x : th) x.join();
    }
    // POS is an integer vector in the reference (src/getACGTNsites.cpp:97,173; R/extractSNPs.R:200), paint a double one
    // (R/estimateCDSDiversity.R:152), len and MI doubles
    std::vector<Col> cols = {{LDW_COL_INT32, pos1}, {LDW_COL_INT32, pos2}, {LDW_COL_DOUBLE, c1}, {LDW_COL_DOUBLE, c2}, {LDW_COL_DOUBLE, len}, {LDW_COL_DOUBLE, mi}};
    const auto t_2 = std::chrono::steady_clock::now();
    const int rc = write_rows(path, append, n, cols, nthreads, bytes_out);
    if (host_timing) {
        auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
        fprintf(stderr, "[ldw] links tsv (%lld rows): derive %.2f ms, format + write %.2f\n", (long long)n, ms(t_1, t_2), ms(t_2, std::chrono::steady_clock::now()));
    }
    return rc;
}

static int links_tsv_finish(LinksTsvJob &J, int64_t *bytes_out) {
    if (bytes_out) *bytes_out = 0;
    if (J.n == 0) return LDW_OK;
    static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
    if (host_timing) fprintf(stderr, "[ldw] links tsv: fetch %.2f ms\n", J.fetch_ms);
    const int rc = write_link_rows(J.path.c_str(), J.append, J.nthreads, J.n, J.a, J.b, J.mi, J.POS, J.paint, J.g, bytes_out);
    J.derived.reset();
    return rc;
}

// the asynchronous writer of a context, if one is running: joined by every call that needs what it holds (the pinned arena, h_POS / h_paint)
struct TsvAsync {
    std::thread th;
    LinksTsvJob job;
    int rc = LDW_OK;
    int64_t bytes = 0;
    std::string err;
};
static int tsv_async_join(ldw_ctx *c, int64_t *rows_out, int64_t *bytes_out) {
    TsvAsync *A = static_cast<TsvAsync *>(c->tsv_async);
    if (!A) return LDW_OK;
    if (A->th.joinable()) A->th.join();
    const int rc = A->rc;
    if (rows_out) *rows_out = A->job.n;
    if (bytes_out) *bytes_out = A->bytes;
    if (rc != LDW_OK) ldw::set_error("%s", A->err.c_str());
    delete A;
    c->tsv_async = nullptr;
    return rc;
}

int ldw_tsv_join(ldw_ctx *c) {
    LDW_REQUIRE(c != nullptr, LDW_ERR_ARG, "ldw_tsv_join: null context");
    return tsv_async_join(c, nullptr, nullptr);
}   // (ldw_ctx_destroy, ldw_set_snp_meta)

int ldw_write_links_tsv(ldw_ctx *c, int which, const char *path, int append, int nthreads, int64_t *rows_out, int64_t *bytes_out) {
    if (c && c->tsv_async) {
        if (int rc = tsv_async_join(c, nullptr, nullptr)) return rc;   // (an unfinished asynchronous table: its error, if any, first)
    }
    LinksTsvJob J;
    if (int rc = links_tsv_prepare(c, which, path, append, nthreads, J)) return rc;
    if (rows_out) *rows_out = J.n;
    return links_tsv_finish(J, bytes_out);
}

int ldw_write_links_tsv_begin(ldw_ctx *c, int which, const char *path, int append, int nthreads) {
    if (c && c->tsv_async) {
        if (int rc = tsv_async_join(c, nullptr, nullptr)) return rc;
    }
    TsvAsync *A = new TsvAsync;
    if (int rc = links_tsv_prepare(c, which, path, append, nthreads, A->job)) {
        delete A;
        return rc;
    }
    c->tsv_async = A;
    A->th = std::thread([A] {
        try {
            A->rc = links_tsv_finish(A->job, &A->bytes);
            if (A->rc != LDW_OK) A->err = ldw_last_error();   // (the error text is per thread: carried over to the joining one)
        } catch (const std::exception &e) {
            A->rc = LDW_ERR_ARG;
            A->err = std::string("ldw_write_links_tsv_begin: ") + e.what();
        } catch (...) {
            A->rc = LDW_ERR_ARG;
            A->err = "ldw_write_links_tsv_begin: unknown exception in the writer thread";
        }
    });
    return LDW_OK;
}

int ldw_host_trim(ldw_ctx *c, int64_t *bytes_out) {
    // a long-lived R / Python session keeps, after one big job, the writer's pooled thread buffers (process-wide) and the context's
    // pinned fetch arena (up to 2 GB x 1.125) until the context dies.  They are kept on purpose WHILE jobs run (an munmap next to GPU work
    // stalls the process's next GPU call by ~20 ms: docs/HISTORY.md 8); this gives them back between jobs, when the caller says so.
    int64_t n = (int64_t)host_pool().trim();
    n += (int64_t)ldw::device_pool_trim();   // r05: + the released DEVICE blocks kept for the next context (ldw_api.hip)
    if (c) {
        if (int rc = tsv_async_join(c, nullptr, nullptr)) return rc;
        if (c->lr_stream == nullptr && c->pin_fetch) {
            (void)hipSetDevice(c->device);
            (void)hipHostFree(c->pin_fetch);
            n += (int64_t)c->pin_fetch_cap;
            c->pin_fetch = nullptr;
            c->pin_fetch_cap = 0;
        }
    }
    if (bytes_out) *bytes_out = n;
    return LDW_OK;
}

int ldw_lr_stream_begin(ldw_ctx *c, const char *path, int append, int nthreads) {
    if (int rc = ldw::check_gpu(c)) return rc;
    LDW_REQUIRE(path != nullptr, LDW_ERR_ARG, "ldw_lr_stream_begin: null path");
    LDW_REQUIRE(c->lr_stream == nullptr, LDW_ERR_STATE, "ldw_lr_stream_begin: a stream is already open (ldw_lr_stream_end)");
    LDW_REQUIRE(c->have_meta && (int64_t)c->h_POS.size() == c->L && (int64_t)c->h_paint.size() == c->L, LDW_ERR_STATE,
                "ldw_lr_stream_begin: SNP meta data (POS, paint, g) not set");
    if (!append) {   // (truncate now: a pass that keeps no long-range row writes nothing, like the reference — :360 — but must not leave an old file behind)
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        LDW_REQUIRE(fd >= 0, LDW_ERR_ARG, "cannot open %s: %s", path, strerror(errno));
        close(fd);
    }
    std::unique_ptr<LrStream> S(new LrStream);
    S->c = c;
    S->path = path;
    S->append = 1;
    S->nthreads = nthreads > 0 ? nthreads : 4;   // (beside a running pass: its two helper threads and the submitting thread need the host's share more)
    if (int rc = ldw::ensure_streams(c)) return rc;   // (made with the context unless LDW_NO_PREPARE: the writer's stream, event ring, pinned counters)
    S->st = c->lr_st;
    S->ev = c->lr_ev;
    S->counts = c->lr_counts;
    LrStream *raw = S.release();
    raw->th = std::thread(lr_stream_main, raw);
    c->lr_stream = raw;
    return LDW_OK;
}

int ldw_lr_stream_end(ldw_ctx *c, int64_t *rows_out, int64_t *bytes_out, int64_t *blocks_out) {
    LDW_REQUIRE(c != nullptr, LDW_ERR_ARG, "ldw_lr_stream_end: null context");
    if (rows_out) *rows_out = 0;
    if (bytes_out) *bytes_out = 0;
    if (blocks_out) *blocks_out = 0;
    LrStream *S = static_cast<LrStream *>(c->lr_stream);
    if (!S) return LDW_OK;
    (void)hipSetDevice(c->device);
    {
        std::lock_guard<std::mutex> lk(S->mu);
        S->closing = true;
    }
    S->cv.notify_all();
    if (S->th.joinable()) S->th.join();
    const int rc = S->rc;
    if (rows_out) *rows_out = S->rows;
    if (bytes_out) *bytes_out = S->bytes;
    if (blocks_out) *blocks_out = S->blocks_seen;
    if (rc != LDW_OK) ldw::set_error("%s", S->err.c_str());
    delete S;
    c->lr_stream = nullptr;
    return rc;
}

int ldw_write_links_tsv_end(ldw_ctx *c, int64_t *rows_out, int64_t *bytes_out) {
    LDW_REQUIRE(c != nullptr, LDW_ERR_ARG, "ldw_write_links_tsv_end: null context");
    if (rows_out) *rows_out = 0;
    if (bytes_out) *bytes_out = 0;
    return tsv_async_join(c, rows_out, bytes_out);
}

}  // extern "C"
