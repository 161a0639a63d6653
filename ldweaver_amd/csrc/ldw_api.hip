// Context, residency, set-up kernels and the small element-wise twins of libldweaver_amd.so.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <thread>
#include "ldw_internal.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
namespace ldw {
static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return LDW_ERR_HIP;
}

// ---- r05: released device blocks are kept for the next taker (process-wide, per device).  tools/scratch/realloc_probe.cpp: device memory the runtime has to fetch from the driver
// costs up to 40 ms per GB on this stack (4 x 12 GB: 1.9 s, fresh or right after freeing the same amount; under ~16 GB the runtime re-uses what it just freed), so an engine created after
// another one was destroyed — every test, every job of a long session — waited about a second for the 13-19 GB its context reserves.  Blocks of >= 64 MB go to a free list on release and are
// handed out again best-fit (at most a quarter larger than asked for); LDW_DEVPOOL_GB (default 48, 0: off) caps what is kept; a failing hipMalloc flushes the list and tries again;
// ldw_host_trim gives everything back.  A block is given up only after the device has drained (what hipFree does implicitly), so its next owner cannot meet the last one's kernels.
namespace {
struct PoolBlock {
    void *p;
    size_t cap;
    int dev;
};
std::mutex g_pool_mtx;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MIN = (size_t)64 << 20;
size_t pool_limit() {
    static const size_t v = [] {
        const char *e = getenv("LDW_DEVPOOL_GB");
        const double gb = e ? atof(e) : 48.0;
        return gb > 0 ? (size_t)(gb * 1073741824.0) : (size_t)0;
    }();
    return v;
}
void *pool_take(size_t want, size_t &cap_out) {
    if (want < POOL_MIN || pool_limit() == 0) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mtx);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < g_pool.size(); ++i)
        if (g_pool[i].dev == dev && g_pool[i].cap >= want && g_pool[i].cap <= want + want / 4 && (best == (size_t)-1 || g_pool[i].cap < g_pool[best].cap)) best = i;
    if (best == (size_t)-1) return nullptr;
    void *p = g_pool[best].p;
    cap_out = g_pool[best].cap;
    g_pool_bytes -= cap_out;
    g_pool[best] = g_pool.back();
    g_pool.pop_back();
    return p;
}
// r06: a caller that has drained every stream that could touch its blocks (ldw_ctx_destroy: the context's own streams) says so for the
// duration of its releases, and the device-wide synchronisation — which stalls OTHER contexts of the device in the middle of their passes — is skipped
thread_local bool g_owner_drained = false;
std::atomic<int> g_live_ctx{0};
size_t pool_idle_limit() {   // what the free list may keep once the process holds no context at all (LDW_DEVPOOL_IDLE_GB, default 24; the live cap is LDW_DEVPOOL_GB)
    static const size_t v = [] {
        const char *e = getenv("LDW_DEVPOOL_IDLE_GB");
        const double gb = e ? atof(e) : 24.0;
        return gb > 0 ? (size_t)(gb * 1073741824.0) : (size_t)0;
    }();
    return v;
}
// a block goes back to the runtime: large ones zero-filled first (the runtime may hand the same memory to the next hipMalloc without clearing it;
// the pages are mapped, so this runs at HBM speed), so that whatever dev_alloc gets from the runtime reads as zero
void pool_free_zeroed(void *p, size_t cap) {
    if (cap >= POOL_MIN && hipMemsetAsync(p, 0, cap, nullptr) != hipSuccess) (void)hipGetLastError();
    (void)hipFree(p);   // (waits for the fill)
}
void pool_give(void *p, size_t cap) {
    if (!p) return;
    if (cap >= POOL_MIN && pool_limit() > 0) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (cur != at.device) (void)hipSetDevice(at.device);
            const bool drained = g_owner_drained || hipDeviceSynchronize() == hipSuccess;   // (the releasing context's kernels are done with it: what hipFree waits for as well)
            if (cur != at.device) (void)hipSetDevice(cur);
            if (drained) {
                std::lock_guard<std::mutex> lk(g_pool_mtx);
                if (g_pool_bytes + cap <= pool_limit()) {
                    g_pool.push_back(PoolBlock{p, cap, at.device});
                    g_pool_bytes += cap;
                    return;
                }
            }
        }
        (void)hipGetLastError();
    }
    pool_free_zeroed(p, cap);
}
// keep: bytes the list may hold afterwards (0: give everything back); the largest blocks go first
size_t pool_flush(size_t keep = 0) {
    std::vector<PoolBlock> all;
    {
        std::lock_guard<std::mutex> lk(g_pool_mtx);
        if (keep == 0) {
            all.swap(g_pool);
            g_pool_bytes = 0;
        } else {
            std::sort(g_pool.begin(), g_pool.end(), [](const PoolBlock &x, const PoolBlock &y) { return x.cap < y.cap; });
            while (g_pool_bytes > keep && !g_pool.empty()) {
                all.push_back(g_pool.back());
                g_pool_bytes -= g_pool.back().cap;
                g_pool.pop_back();
            }
        }
    }
    size_t n = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const PoolBlock &b : all) {
        (void)hipSetDevice(b.dev);
        pool_free_zeroed(b.p, b.cap);
        n += b.cap;
    }
    if (!all.empty()) (void)hipSetDevice(cur);
    return n;
}
// a block of at least `want` bytes: from the free list, else from the runtime (a second try after giving the free list back)
// LDW_POISON_ALLOC=1 (debugging aid, any build): every block handed out is filled with 0xA5 first — fresh device memory happens to be zero on this stack, a block from
// the free list (or from the runtime's own cache) is not, and code that reads what it never wrote must not depend on the difference (tools/fuzz_paths.py found one such read)
bool poison_on() {
    static const bool on = getenv("LDW_POISON_ALLOC") != nullptr;
    return on;
}
// LDW_POISON_ALLOC=1: bytes 0xA5 (an index read from it is wild: the kernel faults).  LDW_POISON_ALLOC=2: int32 words of 1 — indices stay in range, flags
// read "set", floats are denormals: a read-before-write shows as a wrong RESULT instead of a fault (the gentler first probe)
void poison_fill(void *p, size_t bytes) {
    static const int mode = getenv("LDW_POISON_ALLOC") ? atoi(getenv("LDW_POISON_ALLOC")) : 0;
    if (mode == 2) (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), 1, bytes / 4);
    else if (mode == 3) (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), 0x00000A5A, bytes / 4);   // a second in-range pattern: index 2650, flags set, float denormal
    else (void)hipMemset(p, 0xA5, bytes);
    (void)hipDeviceSynchronize();
}
hipError_t dev_alloc(void **out, size_t want, size_t &cap_out) {
    if (void *p = pool_take(want, cap_out)) {
        // a block from the free list holds its last owner's data; memory that comes from the driver is zero, and the engine has always been handed zeroed
        // blocks (nothing else was ever observed on this stack): keep it so — LDW_POISON_ALLOC (below) found kernels that read index arrays before writing
        // them, which zeroes make harmless and another buffer's contents would not.  The pages are mapped already: the fill runs at HBM speed (7 ms for 20 GB).
        hipError_t e = hipMemsetAsync(p, 0, cap_out, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // (the library's streams do not synchronise with the null stream; only the fill has to be done — no device-wide wait)
        if (e == hipSuccess && poison_on()) poison_fill(p, cap_out);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
        } else {
            *out = p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (pool_flush() > 0) e = hipMalloc(out, want);
    }
    cap_out = want;
    // r06 — "a fresh DevBuf reads as zero" as an invariant with a reason per source, instead of an accident of the stack:
    //   * from the free list: zero-filled above (mapped pages: HBM speed);
    //   * small blocks (< 64 MB, never pooled: the runtime caches and re-uses them DIRTY): zero-filled here, microseconds;
    //   * large blocks from the runtime: either fresh from the driver — the kernel driver clears VRAM it hands to a process — or a block this library
    //     gave back with hipFree, and those are zero-filled BEFORE they are freed (pool_free_zeroed).  Filling them here as well was measured: it maps
    //     every page of the 13-19 GB a context reserves (0.37 s on a job's first Hamming call against 6 ms: profiles/r06_alloc_zero_fill.txt).
    // It is NOT what makes stale data harmless: a DevBuf kept across problems holds the last problem's bytes; tools/fuzz_paths.py and tests/test_bounds.py
    // run ~500 problems one after the other on one context for that, LDW_POISON_ALLOC=2 / 3 fill every block with in-range garbage instead of zeroes.
    if (e == hipSuccess && want < POOL_MIN) {
        e = hipMemsetAsync(*out, 0, want, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(*out);
            *out = nullptr;
        }
    }
    if (e == hipSuccess && poison_on()) poison_fill(*out, want);
    return e;
}
}  // namespace

size_t device_pool_trim() { return pool_flush(); }
// the per-SNP state counts into ctx->counts ([L][5] int32) on the context's stream; no copy, no synchronisation (ldw_state_counts, ldw_hamming_weights)
int launch_state_counts(ldw_ctx *c);
void ctx_count(int d) {
    if (g_live_ctx.fetch_add(d) + d <= 0 && d < 0) (void)pool_flush(pool_idle_limit() ? pool_idle_limit() : 0);   // the last context of the process has gone
}
DrainedScope::DrainedScope() { g_owner_drained = true; }
DrainedScope::~DrainedScope() { g_owner_drained = false; }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap && p) return LDW_OK;
    if (p) {
        pool_give(p, cap);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes, got = 0;
    LDW_HIP(dev_alloc(&p, want, got));
    cap = got;
    return LDW_OK;
}

int DevBuf::reserve_keep(size_t bytes, size_t used, hipStream_t s) {
    if (bytes <= cap && p) return LDW_OK;
    size_t want = bytes < 256 ? 256 : bytes;
    if (want < cap + cap / 2) want = cap + cap / 2;  // geometric growth for the link tables
    void *np = nullptr;
    size_t got = 0;
    {
        hipError_t e = dev_alloc(&np, want, got);
        if (e != hipSuccess) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            set_error("hipMalloc of %zu bytes failed (%s); %zu of %zu bytes free on the device", want, hipGetErrorString(e), fr, tot);
            return LDW_ERR_HIP;
        }
    }
    if (p && used) {
        LDW_HIP(hipMemcpyAsync(np, p, used, hipMemcpyDeviceToDevice, s));
        LDW_HIP(hipStreamSynchronize(s));
    }
    if (p) pool_give(p, cap);
    p = np;
    cap = got;
    return LDW_OK;
}

void DevBuf::release() {
    if (p) pool_give(p, cap);
    p = nullptr;
    cap = 0;
}

int check_gpu(ldw_ctx *ctx) {
    LDW_REQUIRE(ctx != nullptr, LDW_ERR_ARG, "null context");
    LDW_HIP(hipSetDevice(ctx->device));
    return LDW_OK;
}
}  // namespace ldw

using namespace ldw;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
namespace ldw {

// states [L][N] (tight) -> padded [L][Npad], pad value 255 (matches no state)
__global__ void k_pad_states(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int64_t L, int64_t N,
                             int64_t Npad) {
    const int64_t a = blockIdx.x;
    for (int64_t s = threadIdx.x; s < Npad; s += blockDim.x)
        dst[a * Npad + s] = s < N ? src[a * N + s] : (uint8_t)255;
}

__global__ void k_unpad_states(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int64_t L, int64_t N,
                               int64_t Npad) {
    const int64_t a = blockIdx.x;
    for (int64_t s = threadIdx.x; s < N; s += blockDim.x)
        dst[a * N + s] = src[a * Npad + s];
}

// 5-state encoder (src/getACGTNsites.cpp:229-265): chars [N][L_total] -> states [n_pos][Npad]
__device__ __forceinline__ uint8_t encode_char(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

__global__ void k_encode(const char *__restrict__ chars, int64_t N, int64_t L_total, const int32_t *__restrict__ pos,
                         int64_t n_pos, uint8_t *__restrict__ states, int64_t Npad) {
    // tile transpose through LDS: 64 positions x 64 sequences per workgroup, coalesced on both sides
    // when the retained columns are dense; gathers when they are sparse.
    __shared__ uint8_t tile[64][65];
    const int64_t p0 = (int64_t)blockIdx.x * 64, s0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 64 x 4
    for (int i = ty; i < 64; i += 4) {                       // i: sequence within tile, tx: position
        const int64_t s = s0 + i, p = p0 + tx;
        uint8_t v = 255;
        if (s < N && p < n_pos) v = encode_char((unsigned char)chars[s * L_total + (pos[p] - 1)]);
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {  // i: position within tile, tx: sequence
        const int64_t p = p0 + i, s = s0 + tx;
        if (p < n_pos && s < Npad) states[p * Npad + s] = (s < N) ? tile[tx][i] : (uint8_t)255;
    }
}

// per-column allele counts of a raw alignment (src/getACGTNsites.cpp:58-70): one thread per column, coalesced along
// the sequence, five counters in registers
__global__ __launch_bounds__(256) void k_column_counts(const char *__restrict__ chars, int64_t N, int64_t L_total,
                                                       int32_t *__restrict__ counts) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= L_total) return;
    int c[5] = {0, 0, 0, 0, 0};
    for (int64_t s = 0; s < N; ++s) {
        const uint8_t st = encode_char((unsigned char)chars[s * L_total + j]);
#pragma unroll
        for (int x = 0; x < 5; ++x) c[x] += st == x;
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) counts[j * 5 + x] = c[x];
}

// per-SNP state counts and fixed-point weighted marginals.  One wave per SNP; each lane reads 4
// consecutive sequences (one dword) per step.
__global__ __launch_bounds__(256) void k_counts_marginals(const uint8_t *__restrict__ states, int64_t L, int64_t Npad,
                                                          const int64_t *__restrict__ vfixed,  // may be null
                                                          int32_t *__restrict__ counts, int64_t *__restrict__ pfix) {
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= L) return;
    int cnt[5] = {0, 0, 0, 0, 0};
    long long pf[5] = {0, 0, 0, 0, 0};
    const uint32_t *row = reinterpret_cast<const uint32_t *>(states + a * Npad);
    for (int64_t q = lane; q < Npad / 4; q += 64) {
        const uint32_t w = row[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t st = (w >> (8 * k)) & 0xFF;
            const long long v = vfixed ? vfixed[q * 4 + k] : 0;
#pragma unroll
            for (int x = 0; x < 5; ++x) {
                const bool hit = st == (uint32_t)x;
                cnt[x] += hit ? 1 : 0;
                pf[x] += hit ? v : 0;
            }
        }
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cnt[x] += __shfl_xor(cnt[x], off);
            pf[x] += __shfl_xor(pf[x], off);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            counts[a * 5 + x] = cnt[x];
            if (pfix) pfix[a * 5 + x] = pf[x];
        }
    }
}

// .ACGTN2num (src/ACGTN2num_parallel.cpp:10-43): zero the reference-allele entry of each 5-column
__global__ void k_acgtn2num(double *__restrict__ nv, const char *__restrict__ ref, int64_t L) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= L) return;
    const char cc = ref[c];
    int rowi = -1;
    if (cc == 'A') rowi = 0;
    else if (cc == 'C') rowi = 1;
    else if (cc == 'G') rowi = 2;
    else if (cc == 'T') rowi = 3;
    else if (cc == 'N' || cc == '-') rowi = 4;
    if (rowi >= 0) nv[c * 5 + rowi] = 0.0;
}

// .fastHadamard (src/computeMI.cpp:19), same association order
__global__ void k_fast_hadamard(double *__restrict__ MI, const double *__restrict__ den, const double *__restrict__ uq,
                                const double *__restrict__ pxy, const double *__restrict__ pxpy,
                                const double *__restrict__ RXY, const double *__restrict__ pXrX,
                                const double *__restrict__ pYrY, int64_t n) {
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
        const double d = ((pxpy[c] + RXY[c]) + pXrX[c]) + pYrY[c];
        MI[c] += ((uq[c] * pxy[c]) / den[c]) * log((pxy[c] / d) * den[c]);
    }
}

int launch_state_counts(ldw_ctx *c) {
    LDW_REQUIRE(c->L > 0, LDW_ERR_STATE, "state counts: no alignment resident");
    if (int rc = c->counts.reserve((size_t)c->L * 5 * 4)) return rc;
    hipLaunchKernelGGL(k_counts_marginals, dim3((unsigned)((c->L + 3) / 4)), dim3(256), 0, c->stream, c->states.as<uint8_t>(), c->L, c->Npad, (const int64_t *)nullptr,
                       c->counts.as<int32_t>(), (int64_t *)nullptr);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

}  // namespace ldw

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int ldw_version(void) { return 100; }

const char *ldw_last_error(void) { return ldw::g_err; }

int ldw_build_info(void) { return LDW_HAS_EXPERIMENTS ? 1 : 0; }

int ldw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ldw_ctx_create(int device, ldw_ctx **out) {
    LDW_REQUIRE(out != nullptr, LDW_ERR_ARG, "ldw_ctx_create: out is null");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("ldw_ctx_create: no HIP device visible; this library has no CPU fallback");
        return LDW_ERR_NOGPU;
    }
    LDW_REQUIRE(device >= 0 && device < n, LDW_ERR_ARG, "ldw_ctx_create: device %d out of range (0..%d)", device, n - 1);
    LDW_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    LDW_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("ldw_ctx_create: device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
        return LDW_ERR_NOGPU;
    }
    ldw_ctx *c = new ldw_ctx();
    ldw::ctx_count(+1);
    c->device = device;
    c->prune = getenv("LDW_NO_PRUNE") == nullptr;
    LDW_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    for (auto &e : c->ev) LDW_HIP(hipEventCreate(&e));
    // r04: the two extra streams of the all-pairs loop (hipStreamCreate: 12 ms each on this box), its events and pinned pick records are made
    // WITH the context, and the code objects of the pass's kernels (loaded at the first launch of a kernel of each translation unit
    // otherwise) by a side thread that starts with it — not inside the first pass of the job.  Joined by the entry points (join_prepare).
    static const bool no_prep = getenv("LDW_NO_PREPARE") != nullptr;
    if (!no_prep) {
        // (the streams here, synchronously: a side thread inside hipStreamCreate slowed the caller's upload of the alignment from 9.5 to 17 ms)
        if (int rc = ldw::ensure_streams(c)) {   // (a half-built context is not handed out — destroyed here, *out stays null)
            const std::string msg = ldw_last_error();
            ldw_ctx_destroy(c);
            *out = nullptr;
            ldw::set_error("%s", msg.c_str());
            return rc;
        }
        c->prep_thread = new std::thread([c] {
            int rc = LDW_OK;
            if (hipSetDevice(c->device) != hipSuccess) rc = LDW_ERR_HIP;
            if (rc == LDW_OK) {
                ldw::warm_mi();
                ldw::warm_apx();
                ldw::warm_gemm_bits();
                ldw::warm_srp();
                ldw::warm_post();
            }
            if (rc != LDW_OK) c->prep_err = ldw_last_error();
            c->prep_rc = rc;
        });
    }
    *out = c;
    return LDW_OK;
}

// r04: the pinned staging buffers of the all-pairs loop (hipHostMalloc: 0.2 ms per MB on this box; three of ~12 MB) and their device
// images, sized from the block geometry, by a second side thread — while the caller uploads the alignment and runs the Hamming GEMM.
// (tools/scratch/malloc_probe.cpp: hipMalloc itself is 0.02-0.25 ms whatever the size; what a first pass paid for was hipStreamCreate —
// now made with the context —, hipHostMalloc, the code-object loads and ensure_rows.)  Optional: everything is also made lazily.
int ldw_ctx_reserve(ldw_ctx *c, int64_t L, int64_t N, int64_t max_blk_sz) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(L > 0 && N > 0 && max_blk_sz > 0, LDW_ERR_ARG, "ldw_ctx_reserve: L, N and max_blk_sz must be positive");
    if (c->prep_thread2) {   // an earlier reservation: finish it first
        if (c->prep_thread2->joinable()) c->prep_thread2->join();
        delete c->prep_thread2;
        c->prep_thread2 = nullptr;
    }
    static const bool no_prep = getenv("LDW_NO_PREPARE") != nullptr;
    if (no_prep) return LDW_OK;
    const int64_t blk = std::min<int64_t>(L, max_blk_sz);
    const int64_t nblk = (L + blk - 1) / blk;
    const int64_t nseg = std::max<int64_t>(1, std::min<int64_t>(c->span_on ? c->span_max : 1, nblk - 2));
    // the packed staging image of a span (prep_block): index, row and permutation lists of both sides, 48 bytes of intervals per to-side SNP
    const int64_t nt = blk * nseg, rows_f = blk * 5 / 4 + 512, rows_t = nt * 5 / 4 + 512;
    const size_t stage = (size_t)((blk + nt) * 12 + (rows_f + rows_t) * 9 + nt * 52 + (blk + 64) * 16 + 3 * (rows_t / 128 + 1) * (rows_f / 64 + 1) + 65536);   // (r06: + the band's tile list, at most 2 bytes per tile of the mask)
    const int64_t Npad = (N + ldw::KSTEP - 1) / ldw::KSTEP * ldw::KSTEP;
    c->prep_thread2 = new std::thread([c, stage, Npad, blk, nseg] {
        int rc = LDW_OK;
        if (hipSetDevice(c->device) != hipSuccess) rc = LDW_ERR_HIP;
        if (rc == LDW_OK) rc = ldw::reserve_slot_buffers(c, Npad, blk, nseg);
        for (int k = 0; k < LDW_NSLOT && rc == LDW_OK; ++k) {
            if (c->pin_cap[k] >= stage) continue;
            if (c->pin[k]) (void)hipHostFree(c->pin[k]);
            c->pin[k] = nullptr;
            c->pin_cap[k] = 0;
            if (hipHostMalloc(&c->pin[k], stage * 2, hipHostMallocDefault) != hipSuccess) {
                ldw::set_error("ldw_ctx_reserve: hipHostMalloc of %zu bytes failed", stage * 2);
                rc = LDW_ERR_HIP;
                break;
            }
            c->pin_cap[k] = stage * 2;
            if (c->dstage[k].reserve(stage * 2) != LDW_OK) rc = LDW_ERR_HIP;
        }
        if (rc != LDW_OK) c->prep_err2 = ldw_last_error();
        c->prep_rc2 = rc;
    });
    return LDW_OK;
}

int ldw_ctx_destroy(ldw_ctx *c) {
    if (!c) return LDW_OK;
    (void)hipSetDevice(c->device);
    (void)ldw::join_prepare(c);
    (void)ldw_tsv_join(c);
    (void)ldw_lr_stream_end(c, nullptr, nullptr, nullptr);
    (void)hipStreamSynchronize(c->stream);
    if (c->gemm_stream) (void)hipStreamSynchronize(c->gemm_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->lr_st) (void)hipStreamSynchronize(c->lr_st);
    {
    ldw::DrainedScope drained;   // every stream that could touch this context's blocks is idle: no device-wide synchronisation per released block
    ldw::DevBuf *bufs[] = {&c->srm_tmp, &c->chars, &c->states, &c->digits, &c->vfixed, &c->r, &c->uqe, &c->POS, &c->paint, &c->Mbits, &c->row0,
                           &c->slot_meta, &c->slot_pfix, &c->apx_skip, &c->snp_sup, &c->counts, &c->pfix_state, &c->G, &c->MIblk, &c->rowlist_f, &c->rowlist_t,
                           &c->idx_f, &c->idx_t, &c->lrow_f, &c->lrow_t, &c->perm_f, &c->perm_t, &c->scr_units, &c->epi_rest, &c->slot_pfix_hi, &c->glo, &c->lo_rows, &c->packs, &c->colcnt,
                           &c->cand_key2, &c->cand_val2, &c->sel_bitmap, &c->sel_chunks, &c->sel_prefix, &c->scratch, &c->small, &c->sr_a, &c->sr_b,
                           &c->sr_mi, &c->lr_a, &c->lr_b, &c->lr_mi, &c->srm_pack, &c->srm_key, &c->srm_pack2, &c->srm_key2, &c->srm_pay, &c->srm_pay2, &c->srm_off,
                           &c->srm_q, &c->srm_n, &c->srm_md, &c->srm_part, &c->srm_shape, &c->srm_cnt, &c->red_row, &c->red_meta,
                           &c->red_srp, &c->pool_a, &c->pool_b, &c->pool_mi, &c->ar_key, &c->ar_val, &c->ar_key2, &c->ar_val2,
                           &c->ar_off, &c->ar_flags, &c->seq_perm, &c->dig_a, &c->dig_b, &c->apx_shift, &c->slot_papx, &c->pop_segs, &c->pop_wbeg, &c->pop_vpos,
                           &c->pair_sums, &c->tab11[0], &c->tab11[1], &c->G2, &c->G3, &c->miss_key, &c->miss_val, &c->srd_lower, &c->srd_cur, &c->srd_out, &c->srd_seg};
    for (auto *b : bufs) b->release();
    for (int k = 0; k < LDW_NSLOT; ++k)
        for (ldw::DevBuf *b : {&c->panel[k][0], &c->panel[k][1], &c->Gapx[k], &c->pairs[k], &c->apx_mini[k], &c->apx_units[k], &c->apx_packs[k], &c->apx_bins[k], &c->apx_clean[k], &c->scr_live[k], &c->sub_units[k], &c->sub_packs[k], &c->sub_bins[k], &c->sub_live[k],
                               &c->hist[k], &c->cand_key[k], &c->cand_val[k]})
            b->release();
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < LDW_NSLOT; ++k) {
        c->dstage[k].release();
        if (c->pin[k]) (void)hipHostFree(c->pin[k]);
        if (c->ev_up[k]) (void)hipEventDestroy(c->ev_up[k]);
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    }
    c->dstage[LDW_NSLOT].release();   // (the staging of a span segment redone on its own)
    }
    if (c->pin[LDW_NSLOT]) (void)hipHostFree(c->pin[LDW_NSLOT]);
    for (int k = 0; k < LDW_NSLOT; ++k) {
        if (c->pin_pick[k]) (void)hipHostFree(c->pin_pick[k]);
        if (c->ev_pick[k]) (void)hipEventDestroy(c->ev_pick[k]);
    }
    if (c->pin_fetch) (void)hipHostFree(c->pin_fetch);
    if (c->pin_lrc) (void)hipHostFree(c->pin_lrc);
    if (c->ev_lrc) (void)hipEventDestroy(c->ev_lrc);
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto &e : c->lr_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->lr_counts) (void)hipHostFree(c->lr_counts);
    if (c->lr_pin) (void)hipHostFree(c->lr_pin);
    if (c->lr_st) (void)hipStreamDestroy(c->lr_st);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->gemm_stream) (void)hipStreamDestroy(c->gemm_stream);
    for (auto &e : c->ev_gemm)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    ldw::ctx_count(-1);
    return LDW_OK;
}

int ldw_ctx_set_stream(ldw_ctx *c, void *s) {
    if (int rc = check_gpu(c)) return rc;
    LDW_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) LDW_HIP(hipStreamDestroy(c->stream));
    if (s) {
        c->stream = (hipStream_t)s;
        c->own_stream = false;
    } else {
        LDW_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return LDW_OK;
}

int ldw_ctx_sync(ldw_ctx *c) {
    if (int rc = check_gpu(c)) return rc;
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_ctx_last_timing(ldw_ctx *c, double ms_out[4]) {
    LDW_REQUIRE(c && ms_out, LDW_ERR_ARG, "ldw_ctx_last_timing: null argument");
    for (int i = 0; i < 4; ++i) ms_out[i] = c->last_ms[i];
    return LDW_OK;
}

int ldw_ctx_counters2(ldw_ctx *c, int64_t out[8]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_ctx_counters2: null argument");
    out[0] = c->spec_misses;
    out[1] = c->fused_blocks;
    out[2] = c->unfused_blocks;
    out[3] = c->screen_violations;
    out[4] = c->mixed_blocks;
    out[5] = c->apx_blocks;
    out[6] = c->apx_units_listed;
    out[7] = c->apx_pairs_listed;
    return LDW_OK;
}

int ldw_ctx_counters(ldw_ctx *c, int64_t out[4]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_ctx_counters: null argument");
    out[0] = c->spec_misses;
    out[1] = c->fused_blocks;
    out[2] = c->unfused_blocks;
    out[3] = c->screen_violations;
    return LDW_OK;
}

int ldw_reset_speculation(ldw_ctx *c) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    c->spec_B_next[0] = c->spec_B_next[1] = -1;
    c->spec_seen[0] = c->spec_seen[1] = false;
    c->spec_probed[0] = c->spec_probed[1] = false;
    c->spec_hist_n[0] = c->spec_hist_n[1] = 0;
    c->tab11_lo[0] = c->tab11_lo[1] = 0;
    c->maybe_off = false;
    return LDW_OK;
}

int ldw_path_report(ldw_ctx *c, int64_t out[8], char *gate, int capacity) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_path_report: null argument");
    out[0] = c->apx_blocks;
    out[1] = c->mixed_blocks;
    out[2] = c->unfused_blocks - c->apx_blocks - c->mixed_blocks + c->spec_misses + c->generic_blocks;   // (blocks in generic POS order always take the plain path)
    out[3] = c->fused_blocks;
    out[4] = c->spec_misses;
    out[5] = c->probe_blocks;
    out[6] = c->apx_pairs_listed;
    out[7] = c->apx_units_listed;
    if (gate && capacity > 0) snprintf(gate, (size_t)capacity, "%s", c->have_weights ? c->apx_gate.c_str() : "weights not set");
    return LDW_OK;
}

int ldw_set_prune(ldw_ctx *c, int on) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    c->prune = on != 0;
    return LDW_OK;
}

int ldw_prune_report(ldw_ctx *c, int64_t out[4]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_prune_report: null argument");
    out[0] = c->sorted_blocks;
    out[1] = c->apx_waves_skipped;
    out[2] = c->apx_waves_total;
    out[3] = c->prune ? 1 : 0;
    return LDW_OK;
}

int ldw_snp_bounds(ldw_ctx *c, double *out, int64_t capacity) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(out, LDW_ERR_ARG, "ldw_snp_bounds: null argument");
    if (int rc = ldw::ensure_rows(c)) return rc;
    LDW_REQUIRE(capacity >= 4 * c->L, LDW_ERR_ARG, "ldw_snp_bounds: capacity %lld < 4 L = %lld", (long long)capacity, (long long)(4 * c->L));
    LDW_HIP(hipMemcpyAsync(out, c->snp_sup.p, (size_t)c->L * 32, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_set_select(ldw_ctx *c, int mode) {
    LDW_REQUIRE(c && (mode == 0 || mode == 1), LDW_ERR_ARG, "ldw_set_select: mode must be 0 (auto) or 1 (radix sorts)");
    c->select_mode = mode;
    return LDW_OK;
}

int ldw_set_path(ldw_ctx *c, int mode) {
    LDW_REQUIRE(c && mode >= 0 && mode <= 2, LDW_ERR_ARG, "ldw_set_path: mode must be 0 (auto), 1 (limb GEMM paths) or 2 (approximate GEMM path)");
    c->path_mode = mode;
    return LDW_OK;
}

int ldw_apx_info(ldw_ctx *c, double out[6]) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_apx_info: null argument");
    out[0] = c->apx_ok ? 1.0 : 0.0;
    out[1] = c->apx_delta;
    out[2] = (double)c->n_classes;
    out[3] = (double)c->n_pop_segs;
    out[4] = (double)c->apx_transitions;
    out[5] = (double)c->apx_e_last;
    return LDW_OK;
}

int ldw_gemm_stats(ldw_ctx *c, double out[6], int reset) {
    LDW_REQUIRE(c && out, LDW_ERR_ARG, "ldw_gemm_stats: null argument");
    for (int k = 0; k < 6; ++k) {
        out[k] = c->gemm_stat[k];
        if (reset) c->gemm_stat[k] = 0;
    }
    return LDW_OK;
}

int ldw_set_engine(ldw_ctx *c, int engine) {
    LDW_REQUIRE(c, LDW_ERR_ARG, "null context");
    LDW_REQUIRE(engine == LDW_ENGINE_MFMA || engine == LDW_ENGINE_HIST || engine == LDW_ENGINE_HIST_STATES, LDW_ERR_ARG, "unknown engine %d", engine);
    LDW_REQUIRE(engine != LDW_ENGINE_HIST_STATES || LDW_HAS_EXPERIMENTS, LDW_ERR_STATE,
                "LDW_ENGINE_HIST_STATES (the first, byte-state histogram kernel: ~200x slower, kept as a cross-check) is only in the LDW_EXPERIMENTS build");
    c->engine = engine;
    return LDW_OK;
}

// ---- (1) ACGTN2num --------------------------------------------------------------------------------
int ldw_acgtn2num_dev(ldw_ctx *c, double *nv_dev, const char *ref_dev, int64_t L) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(nv_dev && ref_dev && L >= 0, LDW_ERR_ARG, "ldw_acgtn2num_dev: bad argument");
    if (L == 0) return LDW_OK;
    hipLaunchKernelGGL(k_acgtn2num, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, c->stream, nv_dev, ref_dev, L);
    LDW_HIP(hipGetLastError());
    return LDW_OK;
}

int ldw_acgtn2num(ldw_ctx *c, double *nv, const char *ref, int64_t L, int /*ncores*/) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(nv && ref && L >= 0, LDW_ERR_ARG, "ldw_acgtn2num: bad argument");
    if (L == 0) return LDW_OK;
    if (int rc = c->scratch.reserve((size_t)L * 41 + 64)) return rc;
    double *d_nv = c->scratch.as<double>();
    char *d_ref = reinterpret_cast<char *>(d_nv + 5 * L);
    LDW_HIP(hipMemcpyAsync(d_nv, nv, (size_t)L * 40, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(d_ref, ref, (size_t)L, hipMemcpyHostToDevice, c->stream));
    if (int rc = ldw_acgtn2num_dev(c, d_nv, d_ref, L)) return rc;
    LDW_HIP(hipMemcpyAsync(nv, d_nv, (size_t)L * 40, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

// ---- (3) fastHadamard ---------------------------------------------------------------------------
int ldw_fast_hadamard(ldw_ctx *c, double *MI, const double *den, const double *uq, const double *pxy,
                      const double *pxpy, const double *RXY, const double *pXrX, const double *pYrY, int64_t n,
                      int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(MI && den && uq && pxy && pxpy && RXY && pXrX && pYrY && n >= 0, LDW_ERR_ARG,
                "ldw_fast_hadamard: bad argument");
    if (n == 0) return LDW_OK;
    const double *src[8] = {MI, den, uq, pxy, pxpy, RXY, pXrX, pYrY};
    const double *dv[8];
    if (on_device) {
        for (int i = 0; i < 8; ++i) dv[i] = src[i];
    } else {
        if (int rc = c->scratch.reserve((size_t)n * 64)) return rc;
        for (int i = 0; i < 8; ++i) {
            double *d = c->scratch.as<double>() + (size_t)i * n;
            LDW_HIP(hipMemcpyAsync(d, src[i], (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
            dv[i] = d;
        }
    }
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_fast_hadamard, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0,
                       c->stream, const_cast<double *>(dv[0]), dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7], n);
    LDW_HIP(hipGetLastError());
    if (!on_device) {
        LDW_HIP(hipMemcpyAsync(MI, dv[0], (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
    }
    return LDW_OK;
}

// ---- alignment residency ------------------------------------------------------------------------
static int set_dims(ldw_ctx *c, int64_t L, int64_t N) {
    LDW_REQUIRE(L > 0 && N > 0, LDW_ERR_ARG, "alignment must be non-empty (L=%lld N=%lld)", (long long)L, (long long)N);
    LDW_REQUIRE(L < (int64_t)1 << 27, LDW_ERR_ARG, "L too large (%lld)", (long long)L);
    c->L = L;
    c->N = N;
    c->Npad = (N + KSTEP - 1) / KSTEP * KSTEP;
    c->KW = c->Npad / 64;
    c->rows_ready = false;
    c->have_weights = false;
    c->have_meta = false;
    return c->states.reserve((size_t)L * c->Npad);
}

int ldw_set_alignment(ldw_ctx *c, const uint8_t *states, int64_t L, int64_t N, int on_device) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(states, LDW_ERR_ARG, "ldw_set_alignment: states is null");
    if (int rc = set_dims(c, L, N)) return rc;
    const uint8_t *src = states;
    if (!on_device) {
        if (int rc = c->scratch.reserve((size_t)L * N)) return rc;
        LDW_HIP(hipMemcpyAsync(c->scratch.p, states, (size_t)L * N, hipMemcpyHostToDevice, c->stream));
        src = c->scratch.as<uint8_t>();
    }
    hipLaunchKernelGGL(k_pad_states, dim3((unsigned)L), dim3(256), 0, c->stream, src, c->states.as<uint8_t>(), L, N, c->Npad);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_get_alignment(ldw_ctx *c, uint8_t *out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(out && c->L > 0, LDW_ERR_STATE, "ldw_get_alignment: no alignment resident");
    if (int rc = c->scratch.reserve((size_t)c->L * c->N)) return rc;
    hipLaunchKernelGGL(k_unpad_states, dim3((unsigned)c->L), dim3(256), 0, c->stream, c->states.as<uint8_t>(), c->scratch.as<uint8_t>(),
                       c->L, c->N, c->Npad);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(out, c->scratch.p, (size_t)c->L * c->N, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_state_counts(ldw_ctx *c, int32_t *counts_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(counts_out && c->L > 0, LDW_ERR_STATE, "ldw_state_counts: no alignment resident");
    if (int rc = ldw::launch_state_counts(c)) return rc;
    // stored [L][5] row-major == 5 x L column-major (ACGTN_table layout)
    LDW_HIP(hipMemcpyAsync(counts_out, c->counts.p, (size_t)c->L * 20, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_alignment_scan(ldw_ctx *c, const char *chars, int64_t N, int64_t L_total, int32_t *allele_counts_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(chars && allele_counts_out && N > 0 && L_total > 0, LDW_ERR_ARG, "ldw_alignment_scan: bad argument");
    const size_t cb = (size_t)N * L_total;
    if (int rc = c->chars.reserve(cb)) return rc;
    if (int rc = c->scratch.reserve((size_t)L_total * 20)) return rc;
    LDW_HIP(hipMemcpyAsync(c->chars.p, chars, cb, hipMemcpyHostToDevice, c->stream));
    c->cN = N;
    c->cL = L_total;
    hipLaunchKernelGGL(k_column_counts, dim3((unsigned)((L_total + 255) / 256)), dim3(256), 0, c->stream, c->chars.as<char>(),
                       N, L_total, c->scratch.as<int32_t>());
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipMemcpyAsync(allele_counts_out, c->scratch.p, (size_t)L_total * 20, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    return LDW_OK;
}

int ldw_encode_alignment(ldw_ctx *c, const char *chars, int64_t N, int64_t L_total, const int32_t *pos, int64_t n_pos,
                         int32_t *acgtn_table_out) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(pos && N > 0 && L_total > 0 && n_pos > 0, LDW_ERR_ARG, "ldw_encode_alignment: bad argument");
    LDW_REQUIRE(chars || (c->chars.p && c->cN == N && c->cL == L_total), LDW_ERR_STATE,
                "ldw_encode_alignment: chars is NULL and no alignment of that shape was scanned");
    for (int64_t i = 0; i < n_pos; ++i)
        LDW_REQUIRE(pos[i] >= 1 && pos[i] <= L_total, LDW_ERR_ARG, "ldw_encode_alignment: pos[%lld]=%d outside 1..%lld",
                    (long long)i, pos[i], (long long)L_total);
    if (int rc = set_dims(c, n_pos, N)) return rc;
    const size_t cb = (size_t)N * L_total;
    const char *d_chars;
    if (chars) {
        if (int rc = c->chars.reserve(cb)) return rc;
        LDW_HIP(hipMemcpyAsync(c->chars.p, chars, cb, hipMemcpyHostToDevice, c->stream));
        c->cN = N;
        c->cL = L_total;
    }
    d_chars = c->chars.as<char>();
    if (int rc = c->scratch.reserve((size_t)n_pos * 4)) return rc;
    int32_t *d_pos = c->scratch.as<int32_t>();
    LDW_HIP(hipMemcpyAsync(d_pos, pos, (size_t)n_pos * 4, hipMemcpyHostToDevice, c->stream));
    dim3 grid((unsigned)((n_pos + 63) / 64), (unsigned)((c->Npad + 63) / 64));
    hipLaunchKernelGGL(k_encode, grid, dim3(256), 0, c->stream, d_chars, N, L_total, d_pos, n_pos,
                       c->states.as<uint8_t>(), c->Npad);
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipStreamSynchronize(c->stream));
    if (acgtn_table_out) return ldw_state_counts(c, acgtn_table_out);
    return LDW_OK;
}

// ---- weights / meta -------------------------------------------------------------------------------
int ldw_set_weights(ldw_ctx *c, const double *hdw, int64_t N, int nlimbs) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->L > 0, LDW_ERR_STATE, "ldw_set_weights: set the alignment first");
    LDW_REQUIRE(hdw && N == c->N, LDW_ERR_ARG, "ldw_set_weights: hdw has %lld entries, alignment has %lld sequences",
                (long long)N, (long long)c->N);
    if (nlimbs == 0) nlimbs = 5;
    LDW_REQUIRE(nlimbs >= 1 && nlimbs <= 6, LDW_ERR_ARG, "ldw_set_weights: nlimbs must be 1..6");
    if (const char *dz = getenv("LDW_DEBUG_ZERO")) {   // debugging aid: zero whole buffer groups when the weights change (which stale tail does a later, smaller problem read?)
        const int mask = atoi(dz);
        if (int rc = join_prepare(c)) return rc;
        LDW_HIP(hipDeviceSynchronize());
        std::vector<ldw::DevBuf *> g;
        if (mask & 1) g.insert(g.end(), {&c->dig_a, &c->dig_b, &c->apx_shift, &c->seq_perm, &c->digits, &c->vfixed});
        if (mask & 2) g.insert(g.end(), {&c->pop_segs, &c->pop_wbeg, &c->pop_vpos, &c->slot_papx});
        if (mask & 4)
            for (int k = 0; k < LDW_NSLOT; ++k) g.insert(g.end(), {&c->panel[k][0], &c->panel[k][1], &c->Gapx[k]});
        if (mask & 8)
            for (int k = 0; k < LDW_NSLOT; ++k)
                g.insert(g.end(), {&c->apx_bins[k], &c->apx_clean[k], &c->apx_mini[k], &c->apx_units[k], &c->apx_packs[k], &c->scr_live[k], &c->pairs[k], &c->sub_units[k], &c->sub_packs[k],
                                   &c->sub_bins[k], &c->sub_live[k]});
        for (int k = 0; k < LDW_NSLOT; ++k) {
            ldw::DevBuf *one[8] = {&c->apx_bins[k], &c->apx_clean[k], &c->apx_mini[k], &c->apx_units[k], &c->apx_packs[k], &c->scr_live[k], &c->pairs[k], &c->sub_units[k]};
            for (int q = 0; q < 8; ++q)
                if (mask & (64 << q)) g.push_back(one[q]);
        }
        if (mask & 16)
            g.insert(g.end(), {&c->Mbits, &c->row0, &c->slot_meta, &c->slot_pfix, &c->slot_pfix_hi, &c->counts, &c->pfix_state, &c->snp_sup, &c->packs, &c->glo, &c->lo_rows});
        if (mask & 32) {
            g.insert(g.end(), {&c->G, &c->G2, &c->G3, &c->MIblk, &c->rowlist_f, &c->rowlist_t, &c->idx_f, &c->idx_t, &c->lrow_f, &c->lrow_t, &c->perm_f, &c->perm_t, &c->scr_units, &c->epi_rest, &c->colcnt,
                               &c->cand_key2, &c->cand_val2, &c->sel_bitmap, &c->sel_chunks, &c->sel_prefix, &c->scratch, &c->small, &c->pair_sums, &c->tab11[0], &c->tab11[1], &c->miss_key,
                               &c->miss_val});
            for (int k = 0; k < LDW_NSLOT; ++k) g.insert(g.end(), {&c->hist[k], &c->cand_key[k], &c->cand_val[k], &c->dstage[k]});
        }
        for (ldw::DevBuf *b : g)
            if (b->p) LDW_HIP(hipMemsetAsync(b->p, 0, b->cap, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
    }
    std::vector<double> v((size_t)N);
    long double neff = 0.0L, vsum = 0.0L;
    double vmax = 0;
    for (int64_t s = 0; s < N; ++s) {
        LDW_REQUIRE(std::isfinite(hdw[s]) && hdw[s] >= 0, LDW_ERR_ARG, "ldw_set_weights: hdw[%lld] = %g is not a finite non-negative weight",
                    (long long)s, hdw[s]);
        const double sq = std::sqrt(hdw[s]);
        v[s] = sq * sq;  // the reference multiplies two sqrt(w)-scaled one-hots (R/computePairwiseMI.R:238,391)
        neff += (long double)hdw[s];
        vsum += (long double)v[s];
        if (v[s] > vmax) vmax = v[s];
    }
    LDW_REQUIRE(vmax > 0, LDW_ERR_ARG, "ldw_set_weights: all weights are zero");
    // fixed point: V_s = round(v_s * 2^F) must fit nlimbs balanced base-256 digits and every sum of V_s must
    // stay below 2^52 so that counts convert to double exactly.
    const double lim_digits = 0.99 * std::ldexp(1.0, 8 * nlimbs - 1) / vmax;
    const double lim_sum = std::ldexp(1.0, 52) / (double)vsum;
    int F = (int)std::floor(std::log2(std::fmin(lim_digits, lim_sum)));
    bool unit = true;  // all weights exactly 1 (unweighted counts): F = 0 is exact with one limb
    for (int64_t s = 0; s < N; ++s) unit = unit && (v[s] == 1.0);
    if (unit) F = 0;
    c->nlimbs = nlimbs;
    c->frac_bits = F;
    c->neff = (double)neff;
    c->h_vfixed.assign((size_t)c->Npad, 0);
    std::vector<int8_t> dseq((size_t)nlimbs * c->Npad, 0);   // digits by sequence
    int64_t total = 0;
    for (int64_t s = 0; s < N; ++s) {
        int64_t V = (int64_t)std::llround(std::ldexp(v[s], F));
        c->h_vfixed[s] = V;
        total += V;
        int64_t rem = V;
        for (int j = 0; j < nlimbs; ++j) {
            int64_t d = ((rem + 128) & 255) - 128;  // balanced digit in [-128, 127]
            dseq[(size_t)j * c->Npad + s] = (int8_t)d;
            rem = (rem - d) / 256;
        }
        LDW_REQUIRE(rem == 0, LDW_ERR_ARG, "ldw_set_weights: internal: weight %g does not fit %d limbs at F=%d", v[s], nlimbs, F);
    }
    c->total_fixed = total;
    // Mixed-precision path (5 limbs only): the block-wide GEMM carries the 3 high limbs, V = V_hi * 2^16 + V_lo.
    // lo_abs_sum = sum |V_lo| 2^-F bounds what the low limbs can add to any joint sum.
    c->h_vfixed_hi.assign((size_t)c->Npad, 0);
    c->total_fixed_hi = 0;
    c->lo_abs_sum = 0;
    if (nlimbs == 5) {
        long double lo_abs = 0;
        for (int64_t s = 0; s < N; ++s) {
            const int64_t lo = (int64_t)dseq[s] + 256 * (int64_t)dseq[(size_t)c->Npad + s];
            c->h_vfixed_hi[s] = (c->h_vfixed[s] - lo) / 65536;   // exact: the remainder is what limbs 2..4 encode
            c->total_fixed_hi += c->h_vfixed_hi[s];
            lo_abs += (long double)(lo < 0 ? -lo : lo);
        }
        c->lo_abs_sum = (double)(lo_abs * (long double)std::ldexp(1.0, -F));
    }
    // Position order of the bit rows: ascending weight (ties by sequence), padding last.  A sum over sequences does not
    // care about their order; this one makes the sequences of one weight CLASS contiguous (class-wise popcounts of the
    // approximate-GEMM path) and lets consecutive 128-position macro steps share a block exponent.
    {
        std::vector<int32_t> order((size_t)N);
        for (int64_t s = 0; s < N; ++s) order[(size_t)s] = (int32_t)s;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return c->h_vfixed[x] < c->h_vfixed[y]; });
        c->h_seq_perm.assign((size_t)c->Npad, -1);
        for (int64_t q = 0; q < N; ++q) c->h_seq_perm[(size_t)q] = order[(size_t)q];
    }
    std::vector<int8_t> dig((size_t)nlimbs * c->Npad, 0);   // digits by position
    for (int j = 0; j < nlimbs; ++j)
        for (int64_t q = 0; q < N; ++q) dig[(size_t)j * c->Npad + q] = dseq[(size_t)j * c->Npad + c->h_seq_perm[(size_t)q]];
    if (int rc = c->seq_perm.reserve((size_t)c->Npad * 4)) return rc;
    LDW_HIP(hipMemcpyAsync(c->seq_perm.p, c->h_seq_perm.data(), (size_t)c->Npad * 4, hipMemcpyHostToDevice, c->stream));
    if (int rc = prepare_apx_weights(c)) return rc;
    if (int rc = c->digits.reserve(dig.size())) return rc;
    if (int rc = c->vfixed.reserve((size_t)c->Npad * 8)) return rc;
    LDW_HIP(hipMemcpyAsync(c->digits.p, dig.data(), dig.size(), hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->vfixed.p, c->h_vfixed.data(), (size_t)c->Npad * 8, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    c->have_weights = true;
    c->rows_ready = false;
    c->tab11_lo[0] = c->tab11_lo[1] = 0;   // the threshold tables belong to the old weights
    c->tab11_on = ldw::exp_env("LDW_NO_TAB11") == nullptr;
    return LDW_OK;
}

int ldw_set_snp_meta(ldw_ctx *c, const double *r, const uint8_t *uqe, const int32_t *POS, const int32_t *paint, double g) {
    if (int rc = check_gpu(c)) return rc;
    LDW_REQUIRE(c->L > 0, LDW_ERR_STATE, "ldw_set_snp_meta: set the alignment first");
    if (int rc = ldw_tsv_join(c)) return rc;   // (a pending asynchronous table reads the positions this call replaces)
    LDW_REQUIRE(r && uqe && POS, LDW_ERR_ARG, "ldw_set_snp_meta: null argument");
    LDW_REQUIRE(g > 0, LDW_ERR_ARG, "ldw_set_snp_meta: genome length g must be positive (snp.dat$g)");
    const int64_t L = c->L;
    for (int64_t i = 0; i < L * 5; ++i) LDW_REQUIRE(uqe[i] <= 1, LDW_ERR_ARG, "ldw_set_snp_meta: uqe must be 0/1");
    if (int rc = c->r.reserve((size_t)L * 8)) return rc;
    if (int rc = c->uqe.reserve((size_t)L * 5)) return rc;
    if (int rc = c->POS.reserve((size_t)L * 4)) return rc;
    if (int rc = c->paint.reserve((size_t)L * 4)) return rc;
    LDW_HIP(hipMemcpyAsync(c->r.p, r, (size_t)L * 8, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->uqe.p, uqe, (size_t)L * 5, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->POS.p, POS, (size_t)L * 4, hipMemcpyHostToDevice, c->stream));
    if (paint) LDW_HIP(hipMemcpyAsync(c->paint.p, paint, (size_t)L * 4, hipMemcpyHostToDevice, c->stream));
    else LDW_HIP(hipMemsetAsync(c->paint.p, 0, (size_t)L * 4, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    c->h_r.assign(r, r + L);
    c->r_min = r[0];
    for (int64_t i = 1; i < L; ++i) c->r_min = r[i] < c->r_min ? r[i] : c->r_min;
    c->h_POS.assign(POS, POS + L);
    c->pos_sorted = true;
    for (int64_t i = 1; i < L && c->pos_sorted; ++i) c->pos_sorted = POS[i] >= POS[i - 1];
    if (paint) c->h_paint.assign(paint, paint + L);
    else c->h_paint.assign((size_t)L, 0);
    c->paint_min = c->paint_max = 0;
    if (paint) {
        c->paint_min = c->paint_max = paint[0];
        for (int64_t i = 1; i < L; ++i) {
            c->paint_min = paint[i] < c->paint_min ? paint[i] : c->paint_min;
            c->paint_max = paint[i] > c->paint_max ? paint[i] : c->paint_max;
        }
    }
    c->g = g;
    c->have_meta = true;
    c->rows_ready = false;
    c->sr_total = -1;
    c->sr_total_dist = -1;
    c->sr_share_rows = -1;
    // r04: with the alignment and the weights in place the row map (indicator rows, marginals, per-SNP bounds: ensure_rows, ~9 ms at C4) is
    // built HERE — it belongs to handing over the data — instead of lazily inside the first block loop; a later ldw_set_weights
    // invalidates it again and the next pass rebuilds it
    if (c->L > 0 && c->have_weights) return ldw::ensure_rows(c);
    return LDW_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// row map: which (SNP, state) pairs get an indicator row
// ------------------------------------------------------------------------------------------------
namespace ldw {

// ------------------------------------------------------------------------------------------------
// k_snp_sup: how large can the MI of SNP a get, whatever its partner looks like?  With the marginals p_x of a fixed, the MI of
// R/computePairwiseMI.R:390-398 is
//     F(n) = 1/den sum_xy q_xy ln(q_xy den / D_xy),   q = n + 1/2,   D_xy = p_x pY_y + RXY + p_x r_a/2 + pY_y r_b/2,   pY_y = sum_x n_xy
// over the joint tables n >= 0 with row sums p_x.  q and D are affine in n and q ln(q / D) is jointly convex (relative entropy),
// so F is convex on that polytope — a product of simplices — and takes its maximum at a vertex: every state of a sends ALL
// its weight to one state of the partner.  For a partner with k flagged states that is k^(states of a) tables: enumerate them.
// F falls as RXY grows; the reference's scrambled RXY (quirk Q1) is r r' / 4 of two other SNPs of the block, at least
// r_min^2 / 4: the m = 1 values use that floor.  Near-singleton sites (most of a real alignment) come out below any threshold
// a long-range link has to reach: the screen and the approximate GEMM drop their pairs without looking at them
// (apx_tile_prunable, k_mi_screen).  One thread per SNP, once per weighting.
// ------------------------------------------------------------------------------------------------
__device__ double mi_vertex_sup(const double *p, int ka, int kb, double neff, double rxy) {
    const double den = neff + 0.5 * ka * kb, rX = 0.5 * ka, rY = 0.5 * kb;
    int nv = 1;
    for (int x = 0; x < ka; ++x) nv *= kb;
    double best = -1e300;
    for (int v = 0; v < nv; ++v) {
        int phi[3];
        double pY[3] = {0.0, 0.0, 0.0};
        int t = v;
        for (int x = 0; x < ka; ++x) {
            phi[x] = t % kb;
            t /= kb;
            pY[phi[x]] += p[x];
        }
        double F = 0.0;
        for (int x = 0; x < ka; ++x)
            for (int y = 0; y < kb; ++y) {
                const double q = (phi[x] == y ? p[x] : 0.0) + 0.5;
                const double D = p[x] * pY[y] + rxy + p[x] * rX + pY[y] * rY;
                if (!(D > 0.0)) return 1e300;   // (degenerate r: no statement)
                F += q * log(q * den / D);
            }
        F /= den;
        if (!(F == F)) return 1e300;
        best = F > best ? F : best;
    }
    return best;
}

__global__ __launch_bounds__(256) void k_snp_sup(int64_t L, const uint32_t *__restrict__ slot_meta, const int64_t *__restrict__ slot_pfix,
                                                 const double *__restrict__ r, double scale, double neff, double r_min, double *__restrict__ sup) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= L) return;
    const uint32_t m = slot_meta[a];
    const int n = (int)(m & 7), ka = n + 1;
    double out[4] = {1e300, 1e300, 1e300, 1e300};
    const uint32_t full = (2u << n) - 1u;
    if ((ka == 2 || ka == 3) && ((m >> 3) & full) == full && r[a] == (double)ka) {
        double p[3] = {0.0, 0.0, 0.0};
        double minor = 0.0;
        for (int x = 0; x < ka; ++x) {
            p[x] = (double)slot_pfix[a * 5 + x] * scale;
            if (x < n) minor += p[x];
        }
        // a SNP with any sizeable minor state reaches every threshold of interest: not worth the enumeration (+inf is the safe answer)
        if (minor < 0.05 * neff + 4.0)
            for (int kb = 2; kb <= 3; ++kb) {
                out[kb - 2] = mi_vertex_sup(p, ka, kb, neff, 0.25 * ka * kb) * (1.0 + 1e-12) + 1e-12;
                const double rfloor = 0.25 * r_min * r_min;
                out[2 + kb - 2] = mi_vertex_sup(p, ka, kb, neff, rfloor < 0.25 * ka * kb ? rfloor : 0.25 * ka * kb) * (1.0 + 1e-12) + 1e-12;
            }
    }
    for (int k = 0; k < 4; ++k) sup[a * 4 + k] = out[k];
}

// Marginals of the high-limb weights, by slot, for the screen of the MIXED-precision path — made when a block first takes that path (r04: the
// default path never does, and building them eagerly cost 2.5 ms of every job's ldw_set_snp_meta).
int ensure_hi_marginals(ldw_ctx *c) {
    if (c->hi_ready || c->nlimbs != 5) return LDW_OK;
    const int64_t L = c->L, Npad = c->Npad;
    const std::vector<uint32_t> &meta = c->h_slot_meta;
    LDW_REQUIRE((int64_t)meta.size() == L, LDW_ERR_STATE, "ensure_hi_marginals: the row map has not been built");
    {
        ldw::DevBuf d_vhi, d_phi, d_cnt2;
        int rc = LDW_OK;
        if ((rc = d_vhi.reserve((size_t)Npad * 8)) || (rc = d_phi.reserve((size_t)L * 40)) || (rc = d_cnt2.reserve((size_t)L * 20)) ||
            (rc = c->slot_pfix_hi.reserve((size_t)L * 40))) {
            d_vhi.release(); d_phi.release(); d_cnt2.release();
            return rc;
        }
        hipError_t he = hipMemcpyAsync(d_vhi.p, c->h_vfixed_hi.data(), (size_t)Npad * 8, hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(k_counts_marginals, dim3((unsigned)((L + 3) / 4)), dim3(256), 0, c->stream, c->states.as<uint8_t>(), L, Npad,
                           d_vhi.as<int64_t>(), d_cnt2.as<int32_t>(), d_phi.as<int64_t>());
        std::vector<int64_t> phs((size_t)L * 5), sph((size_t)L * 5, 0);
        if (he == hipSuccess) he = hipMemcpyAsync(phs.data(), d_phi.p, (size_t)L * 40, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        d_vhi.release(); d_phi.release(); d_cnt2.release();
        if (he != hipSuccess) return ldw::hip_fail(he, "high-limb marginals", __FILE__, __LINE__);
        for (int64_t a = 0; a < L; ++a) {
            const uint32_t m = meta[a];
            const int n = (int)(m & 7);
            for (int i = 0; i <= n; ++i) sph[a * 5 + i] = phs[a * 5 + ((m >> (8 + 3 * i)) & 7)];
        }
        LDW_HIP(hipMemcpyAsync(c->slot_pfix_hi.p, sph.data(), (size_t)L * 40, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
    }
    c->hi_ready = true;
    return LDW_OK;
}

int ensure_rows(ldw_ctx *c) {
    if (c->rows_ready) return LDW_OK;
    static const bool host_timing = getenv("LDW_HOST_TIMING") != nullptr;
    const auto t_rows0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!host_timing) return;
        (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "[ldw] ensure_rows: %.2f ms at %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rows0).count(), what);
    };
    LDW_REQUIRE(c->L > 0 && c->have_weights && c->have_meta, LDW_ERR_STATE,
                "MI needs the alignment, the weights and the SNP meta data to be set first");
    const int64_t L = c->L, Npad = c->Npad;
    if (int rc = c->counts.reserve((size_t)L * 20)) return rc;
    if (int rc = c->slot_pfix.reserve((size_t)L * 40)) return rc;
    if (int rc = c->pfix_state.reserve((size_t)L * 40)) return rc;
    int64_t *d_pfix_state = c->pfix_state.as<int64_t>();  // [L][5] by state
    hipLaunchKernelGGL(k_counts_marginals, dim3((unsigned)((L + 3) / 4)), dim3(256), 0, c->stream,
                       c->states.as<uint8_t>(), L, Npad, c->vfixed.as<int64_t>(), c->counts.as<int32_t>(), d_pfix_state);
    LDW_HIP(hipGetLastError());
    c->h_counts.resize((size_t)L * 5);
    std::vector<int64_t> pfs((size_t)L * 5);
    std::vector<uint8_t> uqe((size_t)L * 5);
    LDW_HIP(hipMemcpyAsync(c->h_counts.data(), c->counts.p, (size_t)L * 20, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(pfs.data(), d_pfix_state, (size_t)L * 40, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipMemcpyAsync(uqe.data(), c->uqe.p, (size_t)L * 5, hipMemcpyDeviceToHost, c->stream));
    LDW_HIP(hipStreamSynchronize(c->stream));
    lap("counts + exact marginals fetched");

    // Slots of SNP a: one per state that is present (count > 0) or flagged in uqe.  The most frequent
    // present state is the "drop" slot: it gets no indicator row, its joint cells follow from the
    // marginals by exact integer subtraction.  Rows are the remaining slots in increasing state order.
    c->h_row0.assign((size_t)L + 1, 0);
    std::vector<uint32_t> meta((size_t)L);
    std::vector<int64_t> spf((size_t)L * 5, 0);
    std::vector<int32_t> rowinfo;
    rowinfo.reserve((size_t)L * 2);
    for (int64_t a = 0; a < L; ++a) {
        const int32_t *cnt = &c->h_counts[a * 5];
        int drop = -1;
        for (int x = 0; x < 5; ++x)
            if (cnt[x] > 0 && (drop < 0 || cnt[x] > cnt[drop])) drop = x;
        LDW_REQUIRE(drop >= 0, LDW_ERR_ARG, "SNP %lld has no valid state (all sequences outside 0..4)", (long long)a);
        uint32_t m = 0;
        int nrows = 0;
        for (int x = 0; x < 5; ++x) {
            if (x == drop) continue;
            if (cnt[x] > 0 || uqe[a * 5 + x]) {
                m |= (uint32_t)uqe[a * 5 + x] << (3 + nrows);
                m |= (uint32_t)x << (8 + 3 * nrows);
                spf[a * 5 + nrows] = pfs[a * 5 + x];
                rowinfo.push_back((int32_t)(a * 8 + x));
                ++nrows;
            }
        }
        m |= (uint32_t)uqe[a * 5 + drop] << (3 + nrows);
        m |= (uint32_t)drop << (8 + 3 * nrows);
        spf[a * 5 + nrows] = pfs[a * 5 + drop];
        m |= (uint32_t)nrows;
        meta[a] = m;
        c->h_row0[a + 1] = c->h_row0[a] + nrows;
    }
    c->R = c->h_row0[L];
    c->h_slot_meta = meta;
    c->h_minor_w.assign((size_t)L, INT64_MAX);   // order key of the tile pruning (prep_block): rows with a table bin first, by weight
    for (int64_t a = 0; a < L; ++a)
        if ((meta[a] & 7u) == 1u && ((meta[a] >> 3) & 3u) == 3u && c->h_r[(size_t)a] == 2.0) c->h_minor_w[(size_t)a] = spf[a * 5];
    c->order_cache.clear();
    c->h_span_bad.assign((size_t)L + 1, 0);
    for (int64_t a = 0; a < L; ++a) {
        const int n = (int)(meta[a] & 7u);
        const bool bad = n == 0 || ((n == 1 || n == 2) && (((meta[a] >> 3) & ((2u << n) - 1u)) != ((2u << n) - 1u)));
        c->h_span_bad[(size_t)a + 1] = c->h_span_bad[(size_t)a] + (bad ? 1 : 0);
    }
    const int64_t R = c->R;
    lap("slot maps built on the host");
    if (int rc = c->row0.reserve((size_t)(L + 1) * 4)) return rc;
    if (int rc = c->slot_meta.reserve((size_t)L * 4)) return rc;
    if (int rc = c->Mbits.reserve((size_t)(R + TILE) * c->KW * 8)) return rc;
    if (int rc = c->small.reserve((size_t)(R + 1) * 4)) return rc;
    LDW_HIP(hipMemcpyAsync(c->row0.p, c->h_row0.data(), (size_t)(L + 1) * 4, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->slot_meta.p, meta.data(), (size_t)L * 4, hipMemcpyHostToDevice, c->stream));
    LDW_HIP(hipMemcpyAsync(c->slot_pfix.p, spf.data(), (size_t)L * 40, hipMemcpyHostToDevice, c->stream));
    c->hi_ready = false;   // (the high-limb marginals of the mixed-precision path are made when a block first takes that path: ensure_hi_marginals)
    if (c->apx_ok) {   // marginals of the approximate weights V' = a b 2^e by slot, in the accumulators' final unit 2^e_last (floor)
        ldw::DevBuf d_v, d_p, d_cnt2;
        int rc = LDW_OK;
        if ((rc = d_v.reserve((size_t)Npad * 8)) || (rc = d_p.reserve((size_t)L * 40)) || (rc = d_cnt2.reserve((size_t)L * 20)) ||
            (rc = c->slot_papx.reserve((size_t)L * 40))) {
            d_v.release(); d_p.release(); d_cnt2.release();
            return rc;
        }
        hipError_t he = hipMemcpyAsync(d_v.p, c->h_vapx.data(), (size_t)Npad * 8, hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(k_counts_marginals, dim3((unsigned)((L + 3) / 4)), dim3(256), 0, c->stream, c->states.as<uint8_t>(), L, Npad,
                           d_v.as<int64_t>(), d_cnt2.as<int32_t>(), d_p.as<int64_t>());
        std::vector<int64_t> pas((size_t)L * 5), spa((size_t)L * 5, 0);
        if (he == hipSuccess) he = hipMemcpyAsync(pas.data(), d_p.p, (size_t)L * 40, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        d_v.release(); d_p.release(); d_cnt2.release();
        if (he != hipSuccess) return ldw::hip_fail(he, "approximate-weight marginals", __FILE__, __LINE__);
        for (int64_t a = 0; a < L; ++a) {
            const uint32_t m = meta[a];
            const int n = (int)(m & 7);
            for (int i = 0; i <= n; ++i) spa[a * 5 + i] = pas[a * 5 + ((m >> (8 + 3 * i)) & 7)] >> c->apx_e_last;
        }
        LDW_HIP(hipMemcpyAsync(c->slot_papx.p, spa.data(), (size_t)L * 40, hipMemcpyHostToDevice, c->stream));
        LDW_HIP(hipStreamSynchronize(c->stream));
    }
    // rows R .. R+TILE-1 stay zero: tile padding of the row lists points at row R
    LDW_HIP(hipMemsetAsync(c->Mbits.as<uint64_t>() + (size_t)R * c->KW, 0, (size_t)TILE * c->KW * 8, c->stream));
    lap("approximate marginals");
    if (R > 0) {
        LDW_HIP(hipMemcpyAsync(c->small.p, rowinfo.data(), (size_t)R * 4, hipMemcpyHostToDevice, c->stream));
        LDW_REQUIRE(R < 2147483647LL, LDW_ERR_ARG, "too many indicator rows");
        if (int rc = fill_rows_bits(c, c->small.as<int32_t>(), R)) return rc;
    }
    if (int rc = c->snp_sup.reserve((size_t)L * 32)) return rc;
    hipLaunchKernelGGL(k_snp_sup, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, c->stream, L, c->slot_meta.as<uint32_t>(), c->slot_pfix.as<int64_t>(),
                       c->r.as<double>(), std::ldexp(1.0, -c->frac_bits), c->neff, c->r_min, c->snp_sup.as<double>());
    LDW_HIP(hipGetLastError());
    LDW_HIP(hipStreamSynchronize(c->stream));
    lap("bit rows + per-SNP bounds");
    c->rows_ready = true;
    c->spec_B_next[0] = c->spec_B_next[1] = -1;   // bucket guesses of an earlier alignment / weighting say nothing about this one
    c->spec_seen[0] = c->spec_seen[1] = false;
    c->spec_probed[0] = c->spec_probed[1] = false;
    c->spec_hist_n[0] = c->spec_hist_n[1] = 0;
    return LDW_OK;
}

}  // namespace ldw
