// libdarsia_hip.so -- host side of the C ABI declared in include/darsia_hip.h.
// Written for gfx950 (MI355X) only.
#include <algorithm>
#include <chrono>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <utility>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "dxh_comm.h"
#include "dxh_kernels.h"
#include "dxh_multi.h"
#include "dxh_vol_multi.h"
#include "dxh_solvers.h"

namespace dxh {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// a device temporary that is released on every path out of the scope (hipFree waits for work that still uses it)
struct DevTmp {
    void* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
};

struct Ctx {
    dxh_tvd_desc desc{};
    Geom g{};
    size_t esz = 4;
    size_t plane_bytes = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    int tune = 0;                  // developer A/B bits (DXH_FUSED_TUNE)
    int fuse_k = 4;                // outer iterations fused per launch (1..4)
    int force_chunks = 0;          // developer knob DXH_NCHUNKS: row chunks of the K-iteration launches
    int waves_per_cu = 0;          // fused sweep: waves per CU the grid is sized for (0 = 8 for K = 1, 16 for K > 1)

    void* of = nullptr;            // omega * img
    void* u[2] = {nullptr, nullptr};
    void* q[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    void* rs = nullptr;            // rhs / diag (three-kernel path)
    void* dpl[3] = {nullptr, nullptr, nullptr};  // explicit d, b planes of a warm start (first iteration only)
    void* bpl[3] = {nullptr, nullptr, nullptr};
    void* cp_ell = nullptr;        // heterogeneous coefficient planes
    void* cp_diag = nullptr;
    void* cp_dsc = nullptr;
    void* cp_kq = nullptr;
    void* in_mu = nullptr;         // the coefficient arrays as given (heterogeneous fused sweep derives from these)
    void* in_om = nullptr;
    void* in_ell = nullptr;
    bool in_valid = false;         // cp_* still equal derive_coefs(in_*) (false after dxh_tvd_adapt_ell)
    int in_has_ell = 0;
    double in_h = 1.0;
    void* om_plane = nullptr;      // omega as a plane (array-valued omega only)
    void* mu_plane = nullptr;      // mu as a plane (array-valued mu; needed by the adaptive update)
    bool explicit_db = false;      // split variables live in dpl / bpl instead of the compressed q planes
    void* base_img = nullptr;      // ConcentrationAnalysis: dense baseline (H, W, base_channels)
    int base_channels = 0;
    int base_dtype = DXH_F32;      // stored type of base_img (dxh_tvd_set_base_typed)
    void* clean_filter = nullptr;  // dense (H, W) threshold cleaning filter
    void* probe_buf = nullptr;     // dense staging for the probe frame
    size_t probe_bytes = 0;
    void* stage = nullptr;         // dense staging buffer (dim * plane) for (*shape, dim) transfers
    double* partial = nullptr;
    double* red_out = nullptr;     // page-locked HOST memory the device writes into (no copy back: a pageable 8 - 48 B copy + its
                                   // synchronisation cost ~0.2 ms per stopping test), 16 doubles: [0, 8) single reductions (norms, energy), [8, 16) the per-iteration increments of a K-iteration launch
    double* inc_partial = nullptr; // per-wave squared increments of the fused iterations
    size_t inc_bytes = 0;
    int inc_nwaves = 0;
    int npartial = 1024;

    int cur = 0;                   // index of the current u / q buffers
    bool have_coefs = false, het = false, uploaded = false, from_db = false;
    bool pristine = true;   // no upload yet: q planes and u[1] still hold the zeros of their allocation
    bool last_iso = false, last_iso_prev = false;
    double mu_s = 0, omega_s = 1, ell_s = 0;  // scalar views (valid when !het)
    double c_ell = 0, c_diag = 0, c_dsc = 0, c_kq = 0;
    int c_div3 = 0;  // Coef::div3 of the current scalar coefficients (float only)
    bool het_cdiag = false;  // array-valued mu only: omega and ell (hence diag, dsc) are scalars
    double sq_img = -1.0;          // sum img^2 over this slab
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DeviceSolver* solver = nullptr;  // MG / CG in the place of the Jacobi sweeps (dxh_tvd_set_solver)
    int solver_kind = DXH_SOLVER_JACOBI;
    bool solver_params_set = false;

    // ---- the reference's NumPy >= 2 dtype flow for a float32 image (dxh_tvd_set_promotion; k_fused_np2) ----
    bool np2 = false;            // double context whose planes hold the reference's mixed float32 / float64 state
    void* of32 = nullptr;        // numpy2, scalar omega: omega * img once more as a float32 plane (read by the K-stage pipelines)
    bool of32_valid = false;
    // numpy2, three-launch flow: the previous iteration's fused pass (k_shrink_rhs_np2) has already formed this iteration's rs
    // and its first sweep; np2_pre_x = where that first sweep's result lives (u[cur ^ 1] or the sweep scratch plane)
    bool np2_pre = false;
    void* np2_pre_x = nullptr;
    bool np2_first = true;       // u still is the float32 image (first iteration after an upload)
    bool np2_mu_is_kq = false;   // the given mu array already is the threshold mu / ell (float32 quotient, widened)
    bool np2_mu_auto = false;    // ... because dxh_tvd_set_coeffs_typed formed that quotient itself from a float32 mu array
    bool ell_is_arr = false, om_is_arr = false;   // set_coeffs: ell (given, or 2 mu) / omega is an array
    int np2_sweeps = 0;          // > 0 while run_t launches the S-sweeps-per-iteration pipeline (dxh_np2_sweeps.h): S
    bool jm_plain = false;       // ... in its PLAIN form: K Jacobi sweeps from jm_in to jm_out (rs in c->rs), nothing else
    const void* jm_in = nullptr;
    void* jm_out = nullptr;
    bool np2_q32 = false;        // the stored q was formed in float32 (the solver handed the float32 iterate back): Np2Coef::q32
    bool np2_adapted = false;    // dxh_tvd_adapt_ell on a numpy2 context: cp_ell / cp_kq hold the adapted float32 ell and threshold

    // ---- row slab of a distributed problem (dxh_tvd_attach_comm) ----
    Comm* comm = nullptr;
    int64_t halo_top = 0, halo_bot = 0;  // the plan's halo rows (the same on every rank)
    int64_t top = 0, bot = 0;            // halo rows this slab really has above / below its owned rows
    int64_t halo_valid = 0;              // rows of the top-side halo budget still valid (a sweep consumes one)
    hipStream_t alt = nullptr;           // the context alternates between two streams: an exchange and everything after it
    hipStream_t alt_owned = nullptr;     //   go to the other one, so the launch behind an exchange follows it in stream order
    hipEvent_t eL = nullptr;             // "everything enqueued on the previous stream is done"
    bool signalling = false;             // the next fused launch puts the interface chunks first and counts them in *signal
    unsigned* signal = nullptr;          // device counter bumped by every finished interface wave
    unsigned signal_target = 0;          // its value once every signalling launch so far has finished its interface chunks
    unsigned signal_pending = 0;         // waves the launch being built adds to it
    int wait_mode = 0;                   // how the comm stream waits for it: 1 hipStreamWaitValue32, 2 polling kernel
    char* xbuf[4] = {nullptr, nullptr, nullptr, nullptr};  // staging: send up, send down, receive from below, receive from above
    size_t xbytes[4] = {0, 0, 0, 0};
    double global_sq_img = -1.0;         // ||img||^2 over all ranks' owned rows
    int64_t split_launches = 0;          // statistics: launches that carried an overlapped exchange
    int64_t row_lo = 0, row_hi = 0;      // rows a fused launch has to produce ([0, n0) when row_hi == 0): a slab launch skips
                                         //   the halo rows its own iterations would leave invalid (run_slab_t)
    // ---- diagnostics of the distributed sweep (dxh_tvd_slab_stats) ----
    // signal[0] is the interface-wave counter, signal[kTimeoutSlot] counts time-outs of the polling wait kernel: a
    // time-out means the exchange went ahead on rows that were never signalled, so the state is invalid from then on and
    // every later synchronising entry point returns DXH_EHIP until the next upload.
    static constexpr int kTimeoutSlot = 16;
    unsigned wait_timeout_ms = 10000;    // polling kernel: give up after this long (DXH_SLAB_WAIT_TIMEOUT_MS)
    bool test_lost_signal = false;       // DXH_SLAB_TEST_LOST_SIGNAL=1: wait for one wave more than will ever signal (tests)
    bool slab_fault = false;             // a time-out has been seen (sticky until upload)
    bool no_kernel = false;              // a launcher found no instantiated pipeline for what it was asked (checked after the launch)
    int64_t slab_launches = 0, slab_exchanges = 0, slab_timed = 0, slab_timeouts = 0;
    double slab_wait_ms = 0.0, slab_xchg_ms = 0.0;
    struct XEv { hipEvent_t a = nullptr, b = nullptr, c = nullptr; bool busy = false; };
    static constexpr int kXEv = 64;
    XEv xev[kXEv];
    int xev_next = 0;
};

// Row-chunk layout of one fused launch (FusedArgs::seg_*).  An ordinary launch is one run of equal chunks.  The slab
// launch in front of a halo exchange (Ctx::signalling) has three runs: short chunks over the rows next to the top
// interface [0, A), ordinary chunks over the interior [A, B), short chunks over [B, n0).  The short ("class A") chunks
// hold the rows the neighbours need (top .. top + halo_bot, and the last halo_top owned rows) and the halo rows
// themselves; their blocks are dispatched first and finish early, each wave bumps Ctx::signal, and the exchange starts
// on the comm stream as soon as the counter is complete -- while the interior chunks of the same launch are still
// running, in the CU slots the class A blocks have vacated.
struct ChunkLayout {
    int64_t seg_row[4];
    int64_t seg_rpc[3];
    int seg_chunk[4];
    int nchunks;    // all runs together
    int class_a;    // chunks of runs 0 and 2
};
static void slab_interface_rows(const Ctx* c, int64_t* A, int64_t* B) {
    *A = c->top > 0 ? c->top + c->halo_bot : 0;
    *B = c->bot > 0 ? c->g.n0 - c->bot - c->halo_top : c->g.n0;
}
// choose_chunks(rows, extra_blocks_per_xcd) -> number of equal chunks for `rows` rows when every XCD also hosts that
// many other blocks
template <typename F>
static ChunkLayout make_layout(const Ctx* c, int nbx, int K, F&& choose_chunks) {
    ChunkLayout l{};
    // rows the launch has to produce: all of them, or -- slab launches between two exchanges -- all but the halo rows
    // that this launch's iterations would leave invalid anyway (run_slab_t sets row_lo / row_hi)
    const int64_t lo = c->row_lo, hi = c->row_hi > 0 ? c->row_hi : c->g.n0;
    const int64_t n0 = hi - lo;
    int64_t A = lo, B = hi;
    if (c->signalling) {
        slab_interface_rows(c, &A, &B);
        A = std::max(A, lo);
        B = std::min(B, hi);
    }
    if (!c->signalling || B - A < 16 * K) {
        // one run (a slab too short to have an interior worth the name: the whole launch signals as class A below)
        const int64_t nc = choose_chunks(n0, 0);
        const int64_t rpc = (n0 + nc - 1) / nc;
        const int real = (int)((n0 + rpc - 1) / rpc);
        const bool all_a = c->signalling;
        l.seg_row[0] = lo;
        l.seg_row[1] = all_a ? hi : lo;
        l.seg_row[2] = l.seg_row[3] = hi;
        l.seg_rpc[0] = l.seg_rpc[1] = l.seg_rpc[2] = rpc;
        l.seg_chunk[0] = 0;
        l.seg_chunk[1] = all_a ? real : 0;
        l.seg_chunk[2] = l.seg_chunk[3] = real;
        l.nchunks = real;
        l.class_a = all_a ? real : 0;
        return l;
    }
    auto short_run = [&](int64_t rows, int64_t* rpc) -> int {  // chunks of at most ~48 rows
        if (rows <= 0) { *rpc = 1; return 0; }
        const int64_t n = (rows + 47) / 48;
        *rpc = (rows + n - 1) / n;
        return (int)((rows + *rpc - 1) / *rpc);
    };
    const int c0 = short_run(A - lo, &l.seg_rpc[0]);
    const int c2 = short_run(hi - B, &l.seg_rpc[2]);
    const int64_t extra = ((int64_t)nbx * (c0 + c2) + 7) / 8;
    const int64_t nc = choose_chunks(B - A, extra);
    l.seg_rpc[1] = (B - A + nc - 1) / nc;
    const int c1 = (int)((B - A + l.seg_rpc[1] - 1) / l.seg_rpc[1]);
    l.seg_row[0] = lo; l.seg_row[1] = A; l.seg_row[2] = B; l.seg_row[3] = hi;
    l.seg_chunk[0] = 0; l.seg_chunk[1] = c0; l.seg_chunk[2] = c0 + c1; l.seg_chunk[3] = c0 + c1 + c2;
    l.nchunks = c0 + c1 + c2;
    l.class_a = c0 + c2;
    return l;
}
template <typename T>
static unsigned apply_layout(Ctx* c, FusedArgs<T>& a, const ChunkLayout& l, int nbx) {
    for (int i = 0; i < 4; ++i) { a.seg_row[i] = l.seg_row[i]; a.seg_chunk[i] = l.seg_chunk[i]; }
    for (int i = 0; i < 3; ++i) a.seg_rpc[i] = l.seg_rpc[i];
    a.nchunks = l.nchunks;
    a.signal = c->signalling ? c->signal : nullptr;
    c->signal_pending = c->signalling ? (unsigned)l.class_a * (unsigned)a.nstrips : 0u;
    // grid of a launch whose blocks are dealt to tiles by block_to_tile_sub
    const int64_t ta = (int64_t)nbx * l.class_a, tb = (int64_t)nbx * (l.nchunks - l.class_a);
    return (unsigned)(8 * ((ta + 7) / 8 + (tb + 7) / 8));
}

// (Nothing has to be joined by the other entry points: after an exchange the context's work continues on the stream the
// exchange was enqueued on, so c->stream is always ordered behind everything the context has in flight, start_exchange.)

static int alloc_plane(Ctx* c, void** p) {
    if (*p) return DXH_OK;
    DXH_HIP(pool_malloc(p, c->plane_bytes));
    DXH_HIP(hipMemsetAsync(*p, 0, c->plane_bytes, c->stream));
    return DXH_OK;
}

static dim3 elem_grid(const Geom& g) {
    const int64_t bpr = (g.L + 255) / 256;
    return dim3((unsigned)(bpr * g.n0));
}


// Exhaustive test, one thread per float mantissa, of the three-operation division q0 = x r, e = x - d q0,
// q = q0 + e r against the IEEE quotient, for the two divisors of a scalar-coefficient solve.  Scaling x by a
// power of two scales q0, e and q exactly (the callers screen tiny numerators), and rounding to nearest is
// symmetric in sign, so the 2^23 values of [1, 2) decide it for every x.  Bit 0 / 1 of *bad: diag / dsc fails.
__global__ void __launch_bounds__(256) k_div3_check(float d0, float r0, float d1, float r1, unsigned* bad) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const float x = __uint_as_float(0x3f800000u | i);
    unsigned b = 0;
    if (div_by_const3(x, d0, r0) != x / d0) b |= 1u;
    if (div_by_const3(x, d1, r1) != x / d1) b |= 2u;
    if (b) atomicOr(bad, b);
}

// one launch of k_div3_check on `stream` and the read-back of its verdict
static int div3_run_check(hipStream_t stream, float d0, float r0, float d1, float r1, unsigned* bad) {
    unsigned* flag = nullptr;
    DXH_HIP(hipMalloc(&flag, sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned), stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_div3_check, dim3(1u << 15), dim3(256), 0, stream, d0, r0, d1, r1, flag);
        e = hipMemcpyAsync(bad, flag, sizeof(unsigned), hipMemcpyDefault, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(flag);
    if (e != hipSuccess) { set_error("division check failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
    return DXH_OK;
}

// Is the three-operation division exact for both divisors of this coefficient set?  float only (2^23 cases each, a
// ~10 us kernel plus a 4-byte read-back); answers are remembered per divisor pair.
static int div3_verified(Ctx* c, float diag, float dsc) {
    static std::mutex mtx;
    static std::map<std::pair<uint32_t, uint32_t>, int> known;
    uint32_t kd, ks;
    std::memcpy(&kd, &diag, 4);
    std::memcpy(&ks, &dsc, 4);
    {
        std::lock_guard<std::mutex> g(mtx);
        auto it = known.find({kd, ks});
        if (it != known.end()) return it->second;
    }
    unsigned bad = ~0u;
    const int ok = div3_run_check(c->stream, diag, 1.0f / diag, dsc, 1.0f / dsc, &bad) == DXH_OK && bad == 0;
    std::lock_guard<std::mutex> g(mtx);
    if (known.size() > 4096) known.clear();
    known[{kd, ks}] = ok;
    return ok;
}

// Exhaustive test of iso_split_vec's relaxations for ONE scalar threshold k >= 0: every float x in [2^-60, 2^60) through the
// fast chain WITHOUT the significand screen, the quotient after one (bit 1 of *bad) and after two (bit 0) residual
// corrections, against the IEEE evaluation of max(sqrt(x) - k, 0) / (sqrt(x) + 1e-18f).  ~1e9 values, a few milliseconds.
__global__ void __launch_bounds__(256) k_isolean_check(float k, unsigned* bad) {
    unsigned b = 0;
    const unsigned long long lo = __float_as_uint(kIsoLo), hi = __float_as_uint(kIsoHi);
    for (unsigned long long u = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < hi;
         u += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)u);
        const float y = __builtin_amdgcn_rsqf(fmaxf(x, kSqrtFastMin));
        const float g = x * y, h = 0.5f * y;
        const float s = fmaf(fmaf(-g, g, x), h, g);
        const float rho = fmaf(fmaf(-s, y, 1.0f), y, y);
        const float num = (s - k) > 0.0f ? (s - k) : 0.0f;
        const float f3 = div_by_const3(num, s, rho);
        const float f5 = div_refine(f3, num, s, rho);
        const float sr = sqrtf(x);
        const float want = t_max0(sr - k) / (sr + 1e-18f);
        if (__float_as_uint(s) != __float_as_uint(sr)) b |= 3u;
        if (__float_as_uint(f5) != __float_as_uint(want)) b |= 1u;
        if (__float_as_uint(f3) != __float_as_uint(want)) b |= 2u;
    }
    if (b) atomicOr(bad, b);
}

// bit 0: lean (no significand screen) with the five-operation quotient is exact for this k; bit 1: also with three operations.
// Remembered per threshold; the first call for a threshold costs one launch and a 4-byte read-back.
static unsigned iso_relaxations_verified(Ctx* c, float k) {
    if (!(k >= 0.0f) || !std::isfinite(k)) return 0;
    static std::mutex mtx;
    static std::map<uint32_t, unsigned> known;
    uint32_t key;
    std::memcpy(&key, &k, 4);
    {
        std::lock_guard<std::mutex> g(mtx);
        auto it = known.find(key);
        if (it != known.end()) return it->second;
    }
    unsigned* flag = nullptr;
    unsigned bad = 3u;
    if (hipMalloc(&flag, sizeof(unsigned)) == hipSuccess) {
        hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_isolean_check, dim3(8192), dim3(256), 0, c->stream, k, flag);
            e = hipMemcpyAsync(&bad, flag, sizeof(unsigned), hipMemcpyDefault, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) bad = 3u;
        (void)hipFree(flag);
    }
    const unsigned ok = ((bad & 1u) ? 0u : 1u) | ((bad & 3u) ? 0u : 2u);   // three operations count only when five are exact too
    std::lock_guard<std::mutex> g(mtx);
    if (known.size() > 4096) known.clear();
    known[key] = ok;
    return ok;
}

template <typename T>
static Coef<T> make_coef(const Ctx* c) {
    Coef<T> k;
    k.ell = (T)c->c_ell;
    k.diag = (T)c->c_diag;
    k.dsc = (T)c->c_dsc;
    k.kq = (T)c->c_kq;
    // reciprocals for div_by_const: only for divisors in a comfortable range, rounded in T by IEEE division
    k.rdiag = k.rdsc = T(0);
    k.div3 = 0;
    k.const_diag = 0;
    auto ok = [](T d) { return std::isfinite((double)d) && std::fabs((double)d) >= 0x1p-20 && std::fabs((double)d) <= 0x1p+20; };
    if ((!c->het || c->het_cdiag) && ok(k.diag) && ok(k.dsc)) {
        k.rdiag = T(1) / k.diag;
        k.rdsc = T(1) / k.dsc;
        k.div3 = c->c_div3;
        k.const_diag = c->het ? 1 : 0;
    }
    return k;
}
template <typename T>
static CoefPlanes<T> make_planes(const Ctx* c) {
    CoefPlanes<T> p;
    p.ell = (const T*)c->cp_ell;
    p.diag = (const T*)c->cp_diag;
    p.dsc = (const T*)c->cp_dsc;
    p.kq = (const T*)c->cp_kq;
    return p;
}

// The coefficient planes a one-thread-per-element or volume kernel really needs: with an array-valued mu only (scalar omega
// and ell -- what the TVD operator passes, e.g. examples/regularization.py:14-29: weight = mu array, regularization = 1.0)
// ell, diag and diag_scaled are the wave-invariant scalars of make_coef (set_coeffs_t fills them with derive_coefs' own
// values, so the bits are the planes' bits); only the threshold mu / ell varies.  The kernels read a scalar where the
// pointer is null: 4 - 12 bytes per element less per pass (DXH_FUSED_TUNE bit 2097152: all four planes, A/B).
template <typename T>
static CoefPlanes<T> make_planes_min(const Ctx* c) {
    CoefPlanes<T> p = make_planes<T>(c);
    if (c->het && c->het_cdiag && !(c->tune & 2097152)) p.ell = p.diag = p.dsc = nullptr;
    return p;
}

template <typename T>
static CoefIn<T> make_coefin(const Ctx* c) {
    CoefIn<T> in{};
    const double h2 = c->in_h * c->in_h;
    const int dim = c->desc.dim;
    in.mu = (const T*)c->in_mu;
    in.omega = (const T*)c->in_om;
    in.ell = (const T*)c->in_ell;
    in.mu_s = (T)c->mu_s;
    in.omega_s = (T)c->omega_s;
    in.ell_s = (T)c->ell_s;
    in.two_dim_over_h2_ell_s = (T)(c->ell_s * 2 * dim / h2);
    in.ell_over_h2_s = (T)(c->ell_s / h2);
    {
        const double diag_d = c->omega_s + c->ell_s * 2 * dim / h2;
        in.diag_s = (T)diag_d;
        in.dsc_s = (T)(diag_d / (c->ell_s / h2));
    }
    in.kq_s = (T)(c->mu_s / c->ell_s);
    in.h2 = (T)h2;
    in.two_dim = (T)(2 * dim);
    in.dim = dim;
    in.has_ell = c->in_has_ell;
    return in;
}

static int set_device(Ctx* c) {
    DXH_HIP(hipSetDevice(c->desc.device));
    return DXH_OK;
}

// dense host array (n0, L) <-> pitched device plane
static int h2d_plane(Ctx* c, void* dst, const void* src) {
    DXH_HIP(hipMemcpy2DAsync(dst, c->g.pitch * c->esz, src, c->g.L * c->esz, c->g.L * c->esz, c->g.n0,
                             hipMemcpyDefault, c->stream));
    return DXH_OK;
}
static int d2h_plane(Ctx* c, void* dst, const void* src) {
    DXH_HIP(hipMemcpy2DAsync(dst, c->g.L * c->esz, src, c->g.pitch * c->esz, c->g.L * c->esz, c->g.n0,
                             hipMemcpyDefault, c->stream));
    return DXH_OK;
}

// ---- reductions ---------------------------------------------------------------------------------
template <typename T>
static int sq_diff(Ctx* c, const void* a, const void* b, int64_t row0, int64_t nrows, double* out_host, int slot) {
    const int nb = c->npartial;
    hipLaunchKernelGGL((k_sqdiff_partial<T>), dim3(nb), dim3(256), 0, c->stream, c->g, (const T*)a, (const T*)b, row0,
                       nrows, c->partial);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->partial, nb, c->red_out + slot);
    DXH_HIP(hipGetLastError());
    if (out_host) {
        DXH_HIP(hipStreamSynchronize(c->stream));
        *out_host = reinterpret_cast<volatile double*>(c->red_out)[slot];
    }
    return DXH_OK;
}

// ---- the fused sweep launch ---------------------------------------------------------------------
template <typename T, int C, bool COUPLED, bool ISO, bool HET, bool DPP, bool ALIGN, bool NTS>
static void launch_fused_v(Ctx* c, int in, int out) {
    using SG = StripGeom<T, C, ALIGN>;
    FusedArgs<T> a;
    a.of = (const T*)c->of;
    a.u_in = (const T*)c->u[in];
    a.u_out = (T*)c->u[out];
    a.q0_in = (const T*)c->q[0][in];
    a.q1_in = (const T*)c->q[1][in];
    a.q0_out = (T*)c->q[0][out];
    a.q1_out = (T*)c->q[1][out];
    a.n0 = c->g.n0;
    a.L = c->g.L;
    a.pitch = c->g.pitch;
    a.nstrips = (int)((c->g.L + SG::STRIDE - 1) / SG::STRIDE);
    const int nbx = (a.nstrips + 3) / 4;
    // size the grid to the machine: c->waves_per_cu waves per CU, every wave the same number of rows
    const bool remap = !(c->tune & 4);
    auto choose = [&](int64_t rows, int64_t extra_per_xcd) -> int64_t {
        const int64_t target_blocks = (int64_t)c->num_cu * (c->waves_per_cu ? c->waves_per_cu : 8) / 4 - 8 * extra_per_xcd;
        int64_t nchunks = std::max<int64_t>(1, target_blocks / nbx);
        const int64_t rpc = std::max<int64_t>(8, (rows + nchunks - 1) / nchunks);
        return (rows + rpc - 1) / rpc;
    };
    const ChunkLayout lay = make_layout(c, nbx, 1, choose);
    const unsigned grid = apply_layout(c, a, lay, nbx);
    a.remap = remap ? 1 : 0;
    a.inc_partial = nullptr;
    a.nwaves_total = 0;
    a.c = make_coef<T>(c);
    a.cp = make_planes<T>(c);
    a.ci = make_coefin<T>(c);
    a.derive = (HET && c->in_valid && !(c->tune & 32)) ? 1 : 0;  // DXH_FUSED_TUNE bit 32: read the four derived planes
    hipLaunchKernelGGL((k_fused<T, C, COUPLED, ISO, HET, DPP, ALIGN, NTS>),
                       dim3(remap ? grid : (unsigned)(nbx * lay.nchunks)), dim3(256), 0, c->stream, a);
}

template <typename T, int C, bool COUPLED, bool ISO, bool HET, bool DPP>
static void launch_fused_t(Ctx* c, int in, int out) {
    // developer A/B switches (DXH_FUSED_TUNE: 1 = tight strips, 2 = plain stores, 4 = no XCD remap) exist
    // for the headline instantiation only
    if constexpr (std::is_same<T, float>::value && C == 3 && COUPLED && !ISO && !HET && DPP) {
        switch (c->tune & 3) {
            case 1: return launch_fused_v<T, C, COUPLED, ISO, HET, DPP, false, true>(c, in, out);
            case 2: return launch_fused_v<T, C, COUPLED, ISO, HET, DPP, true, false>(c, in, out);
            case 3: return launch_fused_v<T, C, COUPLED, ISO, HET, DPP, false, false>(c, in, out);
            default: break;
        }
    }
    launch_fused_v<T, C, COUPLED, ISO, HET, DPP, true, true>(c, in, out);
}

// K iterations per launch (scalar coefficients only)
template <typename T, int C, bool COUPLED, bool ISO, int K>
static void launch_fused_multi_t(Ctx* c, int in, int out, bool want_inc) {
    using SG = StripGeomK<T, C, K>;
    FusedArgs<T> a;
    a.of = (const T*)c->of;
    a.u_in = (const T*)c->u[in];
    a.u_out = (T*)c->u[out];
    a.q0_in = (const T*)c->q[0][in];
    a.q1_in = (const T*)c->q[1][in];
    a.q0_out = (T*)c->q[0][out];
    a.q1_out = (T*)c->q[1][out];
    a.n0 = c->g.n0;
    a.L = c->g.L;
    a.pitch = c->g.pitch;
    a.nstrips = (int)((c->g.L + SG::STRIDE - 1) / SG::STRIDE);
    // wave-pair variant (K = 4): a block of 4 waves covers 2 strips
#ifdef DXH_ENABLE_PAIR
    const bool pair = K == 4 && std::is_same<T, float>::value && (c->tune & 16);
#else
    const bool pair = false;
#endif
    const int nbx = pair ? (a.nstrips + 1) / 2 : (a.nstrips + 3) / 4;
    // Row chunks.  Array coefficients: grid sized for 24 waves per CU (measured best of 8..32).  Scalar coefficients:
    // all blocks take the same time, so the cost of a chunk count is (rows per chunk + pipeline fill) x the issue time of
    // the busiest CU's blocks, W of them co-resident (pipe_waves: 3 for float, 2 for double at K >= 3).  Any count
    // will do: the tile map (block_to_tile_sub) keeps neighbouring tiles on one XCD whatever the count.
    auto choose = [&](int64_t rows, int64_t extra_per_xcd) -> int64_t {
        int64_t nchunks;
        const bool kqv_pipe = (C == 1 || C == 3) && c->het && c->het_cdiag && c->cp_kq && !(c->tune & 65536) && make_coef<T>(c).const_diag;
        if (c->waves_per_cu || (c->het && !kqv_pipe)) {
            const int64_t target_blocks = (int64_t)c->num_cu * (c->waves_per_cu ? c->waves_per_cu : 24) / 4 - 8 * extra_per_xcd;
            nchunks = std::max<int64_t>(1, target_blocks / nbx);
        } else {
            // blocks resident per CU (one wave of each per SIMD); the numpy2 pipelines have their own register budget
            const int W = c->np2_sweeps ? (c->jm_plain ? np2_plain_waves(K) : c->np2 ? np2_sweep_waves(K) : 2)
                                          : c->np2 ? (kqv_pipe ? np2_kqv_waves(C) : (K > 4 ? DXH_NP2_WAVES_K6 : DXH_NP2_WAVES)) : pipe_waves<T, K <= 4 ? K : 4, false>();
            // issue cost of one VALU instruction per SIMD with 1 / 2 / 3 / 4 waves taking turns (tools/valubench.hip)
            static const double cpi[5] = {0.0, 6.6, 3.7, 3.5, 3.2};  // 3: fitted to chunk sweeps of this kernel (a third wave adds ~8 %)
            const int64_t cu_per_xcd = std::max(1, c->num_cu / 8);
            double best = 1e300;
            nchunks = 1;
            const bool any = !(c->tune & 1024);  // DXH_FUSED_TUNE bit 1024: multiples of 8 only (the round-1 rule)
            for (int64_t nc = 1; nc <= std::max<int64_t>(1, rows / (8 * K)); nc = (any || nc < 8) ? nc + 1 : nc + 8) {
                const int64_t rpc_c = (rows + nc - 1) / nc;
                const int64_t real = (rows + rpc_c - 1) / rpc_c;
                // The busiest CU decides.  An XCD gets ceil(tiles / 8) blocks, its busiest CU ceil(that / CUs per XCD):
                // whole rounds of W co-resident blocks, then a remainder round at lower occupancy (measured on
                // 4000 x 8000 x 3 float: 24 chunks = one full round beat 32 chunks by 11 % and 16 chunks by 6 %)
                const int64_t per_xcd = (real * nbx + 7) / 8 + extra_per_xcd;
                const int64_t per_cu = (per_xcd + cu_per_xcd - 1) / cu_per_xcd;
                const int64_t full = per_cu / W, rem = per_cu - full * W;
                // row steps of a chunk: rows R0 - (K - 1) .. R1 + (K - 1), in groups of four (fused_pipe_body)
                const int64_t steps = (rpc_c + 2 * (K - 1) + 1 + 3) / 4 * 4;
                const double cost = (double)steps * ((double)full * W * cpi[W] + (double)rem * cpi[rem]);
                if (cost < best * 0.999) { best = cost; nchunks = real; }
                if ((double)(real * nbx) / ((double)c->num_cu * W) > 4.5) break;
            }
        }
        if (c->force_chunks > 0) nchunks = c->force_chunks;  // developer knob DXH_NCHUNKS
        const int64_t rpc = std::max<int64_t>(8 * K, (rows + nchunks - 1) / nchunks);
        return (rows + rpc - 1) / rpc;
    };
    const ChunkLayout lay = make_layout(c, nbx, K, choose);
    const unsigned grid = apply_layout(c, a, lay, nbx);
    const int64_t nchunks = lay.nchunks;
    if (c->tune & 512) fprintf(stderr, "[dxh] K=%d strips=%d blocks/chunk=%d chunks=%d (%d interface-first) rpc=%lld div3=%d\n", K, a.nstrips, nbx, lay.nchunks, lay.class_a, (long long)lay.seg_rpc[1], c->c_div3);
    a.remap = 1;
    a.inc_partial = nullptr;
    a.nwaves_total = (int)(nchunks * a.nstrips);
    if (want_inc) {
        // [K][nwaves] doubles; (re)allocated on demand (every wave of the launch writes its slots)
        const size_t need = sizeof(double) * (size_t)K * a.nwaves_total;
        if (c->inc_bytes < need) {
            if (c->inc_partial) (void)hipFree(c->inc_partial);
            c->inc_partial = nullptr;
            c->inc_bytes = 0;
            if (hipMalloc((void**)&c->inc_partial, need) == hipSuccess) c->inc_bytes = need;
        }
        a.inc_partial = c->inc_partial;
        c->inc_nwaves = a.nwaves_total;
    }
    a.c = make_coef<T>(c);
    a.cp = make_planes<T>(c);
    a.ci = make_coefin<T>(c);
    // array-valued mu only, divisors usable through reciprocals: the scalar pipeline with a per-element shrink threshold
    // read from cp_kq (DXH_FUSED_TUNE bit 65536: the general array pipeline instead, A/B)
    a.kqv = ((C == 1 || C == 3) && c->het && c->het_cdiag && c->cp_kq && a.c.const_diag && !(c->tune & 65536)) ? 1 : 0;
    a.derive = (c->het && !a.kqv) ? 1 : 0;  // array coefficients (multi_supported: only with the given planes at hand, K = kHetFuseK)
    a.np2_first = (c->np2 && c->np2_first) ? 1 : 0;
    a.np2_div3 = 0;
    a.of32 = nullptr;
    a.iso_lean = a.iso_div3 = 0;
    if constexpr (ISO && std::is_same<T, float>::value) {
        if (!c->het && !(c->tune & 16777216)) {   // (DXH_FUSED_TUNE bit 16777216: the screened five-operation form, A/B)
            const unsigned r = iso_relaxations_verified(c, (float)c->c_kq);
            a.iso_lean = (r & 1u) ? 1 : 0;
            a.iso_div3 = (r & 2u) ? 1 : 0;
            if (c->tune & 512) fprintf(stderr, "[dxh] isotropic shrink factor, k = %g: lean %d, three-operation quotient %d\n", c->c_kq, a.iso_lean, a.iso_div3);
        }
    }
    if (c->np2_sweeps) {   // ONE iteration of K Jacobi sweeps per pass (dxh_np2_sweeps.h; objects np2_sw2 .. np2_sw6)
        if constexpr (std::is_same<T, double>::value && !ISO && (C == 1 || C == 3) && K >= kNp2SweepsMin && K <= kNp2SweepsMax) {
            if (c->jm_plain && c->np2_sweeps == K) {   // K plain sweeps of a Jacobi solve: x in, rs, x out
                a.u_in = (const T*)c->jm_in;
                a.u_out = (T*)c->jm_out;
                a.q0_in = (const T*)c->rs;
                a.inc_partial = nullptr;
                launch_np2_plain_sweeps<K>(C, COUPLED, a, grid, c->stream);
                return;
            }
            if (c->np2_sweeps == K && (c->np2 ? (c->of32_valid && (!c->het || a.kqv)) : !c->het)) {
                if (c->np2) {
                    a.of32 = (const float*)c->of32;
                    const float fd = (float)c->c_diag;
                    if (std::isfinite(fd) && std::fabs(fd) >= 0x1p-20f && std::fabs(fd) <= 0x1p+20f && !(c->tune & 128)) a.np2_div3 = div3_verified(c, fd, fd);
                }
                launch_np2_sweeps<K>(C, COUPLED, a.kqv != 0, !c->np2, a, grid, c->stream);
                return;
            }
        }
        c->no_kernel = true;
        return;
    }
    if (c->np2) {
        if (!c->of32_valid) { c->no_kernel = true; return; }   // (cannot happen: every upload of a scalar-omega numpy2 context fills it)
        a.of32 = (const float*)c->of32;
        // rhs / diag is a float32 division by the float32-rounded Python float: three operations through the reciprocal
        // where the exhaustive check says so for this divisor (remembered per divisor), five otherwise
        const float fd = (float)c->c_diag;
        if (std::isfinite(fd) && std::fabs(fd) >= 0x1p-20f && std::fabs(fd) <= 0x1p+20f && !(c->tune & 128))
            a.np2_div3 = div3_verified(c, fd, fd);
        // the reference's float32 dtype flow: PipeNp2 (dxh_np2_pipe.h; dxh_multi.hip objects np2_s / np2_q); pick_k only asks for
        // the combinations that exist
        if constexpr (std::is_same<T, double>::value && !ISO && (C == 1 || C == 3) && K == kNp2K) {
            if (a.kqv) launch_multi_kernel_np2_kqv(C, COUPLED, want_inc, a, grid, c->stream);
            else launch_multi_kernel_np2_scalar(C, COUPLED, want_inc, a, grid, c->stream);
            return;
        }
        if constexpr (std::is_same<T, double>::value && !ISO && (C == 2 || C == 4) && K == kNp2K) {
            if (!c->het) {   // two / four channels: the scalar-threshold pipeline (a threshold plane: one iteration per launch)
                launch_multi_kernel_np2_scalar_c24(C, COUPLED, want_inc, a, grid, c->stream);
                return;
            }
        }
        if constexpr (std::is_same<T, double>::value && !ISO && (C == 1 || C == 3) && (K == 2 || K == 3)) {
            if (!c->het) {   // the remainders of an iteration count: two / three iterations per pass
                if constexpr (K == 2) launch_multi_kernel_np2_scalar_k2(C, COUPLED, want_inc, a, grid, c->stream);
                else launch_multi_kernel_np2_scalar_k3(C, COUPLED, want_inc, a, grid, c->stream);
                return;
            }
        }
        if constexpr (std::is_same<T, double>::value && !ISO && C == 1 && K == kNp2GrayK) {
            if (!c->het) {   // single channel, scalar threshold: six iterations per pass (the K = 4 launch is HBM-bound)
                launch_multi_kernel_np2_gray6(COUPLED, want_inc, a, grid, c->stream);
                return;
            }
        }
        c->no_kernel = true;   // never a silent no-op: the callers turn this into DXH_ESTATE (launched_or_fail)
        return;
    }
    // float64 images, anisotropic, scalar coefficients: the numpy2 pipeline's structure with every quantity float64
    // (dxh_np2_pipe.h, F = double; DXH_FUSED_TUNE bit 4194304: the generic Pipe<double>, A/B)
    if constexpr (std::is_same<T, double>::value && !ISO && (C == 1 || C == 3) && K == kNp2K) {
        if (!c->het && !(c->tune & 4194304)) {
            launch_multi_kernel_f64p(C, COUPLED, want_inc, a, grid, c->stream);
            return;
        }
    }
    // the kernels are instantiated in dxh_multi.hip (one object per element type and K, built in parallel)
    if constexpr (K > 4) c->no_kernel = true;   // (six iterations per pass exist in the numpy2 flow only)
    else if constexpr (C == 1 || C == 3) launch_multi_kernel<T, K>(C, COUPLED, ISO, want_inc, pair, a, grid, c->stream);
    else launch_multi_kernel_c24<T, K>(C, COUPLED, ISO, want_inc, pair, a, grid, c->stream);
}

// channel counts the multi-iteration kernel is instantiated for
static bool multi_supported(const Ctx* c) {
    if (!(c->desc.dim == 2 && c->g.n2 >= 1 && c->g.n2 <= 4) || (c->tune & 8)) return false;
    if (c->g.n0 > (int64_t)INT32_MAX - 64) return false;  // the pipelines index rows with 32-bit integers
    if (c->het) return c->in_valid && !(c->tune & 32);  // array coefficients: derived in the kernel from the given planes
    // k_fused_multi divides through precomputed reciprocals: the divisors must be in a comfortable range
    auto ok = [](double d) { return std::isfinite(d) && std::fabs(d) >= 0x1p-20 && std::fabs(d) <= 0x1p+20; };
    return ok(c->c_diag) && ok(c->c_dsc) && c->c_kq >= 0;
}

// iterations of the next launch when `left` iterations remain (array coefficients: their pipeline exists for one K
// only, shorter remainders run as single-iteration sweeps)
static int pick_k(const Ctx* c, int left) {
    if (c->np2) {
        // the reference's float32 dtype flow: K = kNp2K pipelines -- scalar threshold for 1 - 4 channels, threshold plane for 1
        // and 3 --, remainders one iteration per launch (k_fused_np2)
        if (!(c->g.n2 == 1 || c->g.n2 == 3) && c->het) return 1;
        // single channel, scalar threshold: six iterations per pass where that leaves no worse a remainder than fours do
        // (DXH_FUSED_TUNE bit 8388608: fours only, A/B)
        if (c->g.n2 == 1 && !c->het && c->fuse_k >= kNp2K && !(c->tune & 8388608) &&
            (left == kNp2GrayK || left == kNp2GrayK + 1 || left >= kNp2GrayK + 4))
            return kNp2GrayK;
        if (c->fuse_k >= kNp2K && left >= kNp2K) return kNp2K;
        // remainders: two / three iterations per pass exist for the scalar threshold on 1 and 3 channels
        if (!c->het && (c->g.n2 == 1 || c->g.n2 == 3) && left >= 2) return std::min(std::min(c->fuse_k, left), 3);
        return 1;
    }
    if (c->het) return (c->fuse_k >= kHetFuseK && left >= kHetFuseK) ? kHetFuseK : 1;
    return std::max(1, std::min(c->fuse_k, left));
}

template <typename T, int C>
static void launch_fused_multi_c(Ctx* c, int in, int out, bool iso, int K, bool want_inc) {
    const bool coupled = c->desc.ndim == 3 && c->desc.nsum == DXH_NSUM_ALL_AXES;
#define DXH_M(CP_, ISO_)                                                              \
    switch (K) {                                                                      \
        case 2: launch_fused_multi_t<T, C, CP_, ISO_, 2>(c, in, out, want_inc); break;          \
        case 3: launch_fused_multi_t<T, C, CP_, ISO_, 3>(c, in, out, want_inc); break;          \
        case 5:                                                                                 \
            if constexpr ((C == 1 || C == 3) && !ISO_ && std::is_same<T, double>::value)        \
                launch_fused_multi_t<T, C, CP_, ISO_, 5>(c, in, out, want_inc);                 \
            else c->no_kernel = true;                                                           \
            break;                                                                              \
        case kNp2GrayK:                                                                         \
            if constexpr ((C == 1 || C == 3) && !ISO_ && std::is_same<T, double>::value)        \
                launch_fused_multi_t<T, C, CP_, ISO_, kNp2GrayK>(c, in, out, want_inc);         \
            else c->no_kernel = true;                                                           \
            break;                                                                              \
        default: launch_fused_multi_t<T, C, CP_, ISO_, 4>(c, in, out, want_inc); break;         \
    }
    if (coupled) { if (iso) { DXH_M(true, true) } else { DXH_M(true, false) } }
    else         { if (iso) { DXH_M(false, true) } else { DXH_M(false, false) } }
#undef DXH_M
}

template <typename T>
static void launch_fused_multi(Ctx* c, int in, int out, bool iso, int K, bool want_inc = false) {
    switch ((int)c->g.n2) {
        case 1: launch_fused_multi_c<T, 1>(c, in, out, iso, K, want_inc); break;
        case 2: launch_fused_multi_c<T, 2>(c, in, out, iso, K, want_inc); break;
        case 3: launch_fused_multi_c<T, 3>(c, in, out, iso, K, want_inc); break;
        default: launch_fused_multi_c<T, 4>(c, in, out, iso, K, want_inc); break;
    }
}

template <typename T, int C, bool COUPLED>
static void launch_fused_c(Ctx* c, int in, int out, bool iso, bool dpp) {
    const bool het = c->het;
#define DXH_F(ISO_, HET_, DPP_) launch_fused_t<T, C, COUPLED, ISO_, HET_, DPP_>(c, in, out)
    if (iso) {
        if (het) { if (dpp) DXH_F(true, true, true); else DXH_F(true, true, false); }
        else     { if (dpp) DXH_F(true, false, true); else DXH_F(true, false, false); }
    } else {
        if (het) { if (dpp) DXH_F(false, true, true); else DXH_F(false, true, false); }
        else     { if (dpp) DXH_F(false, false, true); else DXH_F(false, false, false); }
    }
#undef DXH_F
}

// channel count / coupling the fused kernel is instantiated for
static bool fused_supported(const Ctx* c) {
    if (c->desc.dim != 2) return false;
    const int64_t C = c->g.n2;
    if (C < 1 || C > 4) return false;
    // the scalar-coefficient fused kernels shrink through q - clamp(q, -k, k), which is the reference's
    // max(|q| - k, 0) sign(q) only for a threshold k = mu / ell >= 0
    if (!c->het && !(c->c_kq >= 0)) return false;
    return true;
}

template <typename T>
static void launch_fused(Ctx* c, int in, int out, bool iso, bool dpp) {
    const int C = (int)c->g.n2;
    // channel neighbours take part in the Jacobi sum only when the array has a third axis AND the
    // reference semantics (all axes) are requested
    const bool coupled = c->desc.ndim == 3 && c->desc.nsum == DXH_NSUM_ALL_AXES;
    switch (C) {
        case 1: coupled ? launch_fused_c<T, 1, true>(c, in, out, iso, dpp) : launch_fused_c<T, 1, false>(c, in, out, iso, dpp); break;
        case 2: coupled ? launch_fused_c<T, 2, true>(c, in, out, iso, dpp) : launch_fused_c<T, 2, false>(c, in, out, iso, dpp); break;
        case 3: coupled ? launch_fused_c<T, 3, true>(c, in, out, iso, dpp) : launch_fused_c<T, 3, false>(c, in, out, iso, dpp); break;
        case 4: coupled ? launch_fused_c<T, 4, true>(c, in, out, iso, dpp) : launch_fused_c<T, 4, false>(c, in, out, iso, dpp); break;
    }
}

// ---- the three-kernel path ----------------------------------------------------------------------
// grid of the vectorised volume kernels (k_jacobi_vol_x, k_rhs_vol, k_shrink_vol): tiles = (plane, row, 256-vector block)
struct VolTiles { int nbx; int64_t tiles, per_xcd; unsigned grid; };
template <typename T>
static bool vol_tiles(const Ctx* c, VolTiles* v) {
    constexpr int NV = VecOf<T>::N;
    if (!(c->g.ndim == 3 && c->desc.dim == 3 && c->g.n2 >= 64 && c->g.n2 % NV == 0) || (c->tune & (16384 | 262144))) return false;
    v->nbx = (int)((c->g.n2 / NV + 255) / 256);
    v->tiles = c->g.n0 * c->g.n1 * v->nbx;
    v->per_xcd = (v->tiles + 7) / 8;
    if (v->per_xcd * 8 >= (int64_t)INT32_MAX) return false;
    v->grid = (unsigned)(v->per_xcd * 8);
    return true;
}

template <typename T, int DIM>
static QPlanes<T, DIM> make_q(Ctx* c, int in, int out) {
    QPlanes<T, DIM> q;
    for (int j = 0; j < DIM; ++j) {
        q.in[j] = (const T*)c->q[j][in];
        q.out[j] = (T*)c->q[j][out];
    }
    return q;
}
template <typename T>
static DBPlanes<T> make_db(Ctx* c) {
    DBPlanes<T> db;
    for (int j = 0; j < 3; ++j) {
        db.d[j] = (const T*)c->dpl[j];
        db.b[j] = (const T*)c->bpl[j];
    }
    return db;
}

template <typename T, int DIM>
static void launch_rhs(Ctx* c, bool iso, bool from_db) {
    const Geom& gx = c->g;
    const dim3 grid = elem_grid(gx), blk(256);
    auto q = make_q<T, DIM>(c, c->cur, c->cur);
    auto db = make_db<T>(c);
    auto co = make_coef<T>(c);
    auto cp = make_planes_min<T>(c);
    if (c->solver) {  // MG / CG take the right-hand side itself, not rhs / diag: divide by exactly 1
        co.diag = T(1);
        cp.diag = nullptr;
    }
    if constexpr (DIM == 3) {
        VolTiles vt;
        if (!from_db && vol_tiles<T>(c, &vt)) {  // volumes: 16 bytes per lane, XCD-contiguous tiles (k_rhs_vol)
            if (iso) hipLaunchKernelGGL((k_rhs_vol<T, true>), dim3(vt.grid), blk, 0, c->stream, c->g, (const T*)c->of, q, co, cp, (T*)c->rs, vt.nbx, vt.tiles, vt.per_xcd);
            else hipLaunchKernelGGL((k_rhs_vol<T, false>), dim3(vt.grid), blk, 0, c->stream, c->g, (const T*)c->of, q, co, cp, (T*)c->rs, vt.nbx, vt.tiles, vt.per_xcd);
            return;
        }
    }
    if (from_db)
        hipLaunchKernelGGL((k_rhs<T, DIM, false, true>), grid, blk, 0, c->stream, gx, (const T*)c->of, q, db, co, cp, (T*)c->rs);
    else if (iso)
        hipLaunchKernelGGL((k_rhs<T, DIM, true, false>), grid, blk, 0, c->stream, gx, (const T*)c->of, q, db, co, cp, (T*)c->rs);
    else
        hipLaunchKernelGGL((k_rhs<T, DIM, false, false>), grid, blk, 0, c->stream, gx, (const T*)c->of, q, db, co, cp, (T*)c->rs);
}

template <typename T, int DIM>
static void launch_shrink(Ctx* c, const void* unew, bool iso, bool from_db) {
    const Geom& gx = c->g;
    const dim3 grid = elem_grid(gx), blk(256);
    auto q = make_q<T, DIM>(c, c->cur, c->cur);  // in place: each thread touches only its own element
    auto db = make_db<T>(c);
    auto co = make_coef<T>(c);
    auto cp = make_planes_min<T>(c);
    if constexpr (DIM == 3) {
        VolTiles vt;
        if (!from_db && vol_tiles<T>(c, &vt)) {  // volumes: k_shrink_vol
            if (iso) hipLaunchKernelGGL((k_shrink_vol<T, true>), dim3(vt.grid), blk, 0, c->stream, c->g, (const T*)unew, q, co, cp, vt.nbx, vt.tiles, vt.per_xcd);
            else hipLaunchKernelGGL((k_shrink_vol<T, false>), dim3(vt.grid), blk, 0, c->stream, c->g, (const T*)unew, q, co, cp, vt.nbx, vt.tiles, vt.per_xcd);
            return;
        }
    }
    if (from_db)
        hipLaunchKernelGGL((k_shrink<T, DIM, false, true>), grid, blk, 0, c->stream, gx, (const T*)unew, q, db, co, cp);
    else if (iso)
        hipLaunchKernelGGL((k_shrink<T, DIM, true, false>), grid, blk, 0, c->stream, gx, (const T*)unew, q, db, co, cp);
    else
        hipLaunchKernelGGL((k_shrink<T, DIM, false, false>), grid, blk, 0, c->stream, gx, (const T*)unew, q, db, co, cp);
}

// dim = 3: the whole iteration in one pass (k_fused3d)
template <typename T>
static void launch_fused3d(Ctx* c, int in, int out, bool iso) {
    QPlanes<T, 3> q;
    for (int j = 0; j < 3; ++j) {
        q.in[j] = (const T*)c->q[j][in];
        q.out[j] = (T*)c->q[j][out];
    }
    // float: the 2.5-D march; double: the brick kernel (the march needs 255 registers there and loses, 0.42 vs 0.39 ms on
    // 256^3); DXH_FUSED_TUNE bit 4096 forces the bricks, 8192 the march
    {
        constexpr int NV = VecOf<T>::N;
        if (c->g.n2 % NV == 0 && !(c->tune & (4096 | 8192 | 1048576))) {
            // volumes whose rows are 16-byte multiples: the march with NV = 16 B / sizeof(T) adjacent columns per thread
            // (k_march3d_v; DXH_FUSED_TUNE bit 1048576: the one-column-at-a-time march / the bricks below, A/B)
            // Threads per patch row (TPR; patch = 256 / TPR rows x NV TPR columns).  What a width costs depends on more than
            // the wasted rim and overhang columns (rim columns recompute t_2 from global memory, wide rows coalesce better, the
            // plane chunks must fill the resident-block slots), so for volumes of a million voxels and more the candidates
            // are TIMED once per (shape, variant) on the first launch -- the launch is idempotent (in and out planes differ) --
            // and the choice is remembered; small volumes take the width with the least waste.  Results never depend on it.
            auto launch_with = [&](int tpr) {
                const int o1 = 256 / tpr - 1, o2 = NV * tpr - NV;
                const int nt1 = (int)((c->g.n1 + o1 - 1) / o1), nt2 = (int)((c->g.n2 + o2 - 1) / o2);
                auto launch = [&](auto kernel) {
                    int bpc = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kernel, 256, 0) != hipSuccess || bpc < 1) bpc = 2;
                    if (c->waves_per_cu) bpc = std::max(1, c->waves_per_cu / 4);
                    const int64_t slots = (int64_t)c->num_cu * bpc, tiles = (int64_t)nt1 * nt2;
                    int64_t nchunk = std::max<int64_t>(1, slots / tiles);
                    int64_t ppc = std::max<int64_t>(std::min<int64_t>(c->g.n0, 8), (c->g.n0 + nchunk - 1) / nchunk);
                    if (c->force_chunks > 0) ppc = std::max<int64_t>(1, (c->g.n0 + c->force_chunks - 1) / c->force_chunks);
                    nchunk = (c->g.n0 + ppc - 1) / ppc;
                    const int64_t total = nchunk * tiles;
                    if (c->tune & 512) fprintf(stderr, "[dxh] march3d_v patch %dx%d tiles %lld chunks %lld (%lld planes) blocks/CU %d\n", o1 + 1, o2 + NV, (long long)tiles, (long long)nchunk, (long long)ppc, bpc);
                    hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 0, c->stream, c->g, (const T*)c->of, q,
                                       (const T*)c->u[in], (T*)c->u[out], make_coef<T>(c), make_planes_min<T>(c), nt1, nt2,
                                       (int)nchunk, (int)ppc, tpr);
                };
                if (iso) launch(k_march3d_v<T, true>); else launch(k_march3d_v<T, false>);
            };
            auto least_waste = [&]() {
                int tpr = 16;
                double best = -1.0;
                for (int t = 4; t <= 64; ++t) {
                    const int o1 = 256 / t - 1, o2 = NV * t - NV;
                    if (o1 < 1) continue;
                    const double tiles = (double)((c->g.n1 + o1 - 1) / o1) * (double)((c->g.n2 + o2 - 1) / o2);
                    const double eff = (double)c->g.n1 * (double)c->g.n2 / (tiles * 256.0 * NV);   // useful columns per thread slot
                    if (eff > best * 1.001) { best = eff; tpr = t; }
                }
                return tpr;
            };
            int tpr = 0;
            if (const char* e = getenv("DXH_MARCH_M2")) {  // developer knob: patch width in columns
                const int v = atoi(e);
                if (v >= 4 * NV && v <= 64 * NV && v % NV == 0) tpr = v / NV;
            }
            static const bool tune_on = [] { const char* e = getenv("DXH_MARCH_TUNE"); return !(e && atoi(e) == 0); }();
            if (!tpr && tune_on && c->g.n0 * c->g.n1 * c->g.n2 >= (int64_t(1) << 20) && !c->force_chunks && !c->waves_per_cu) {
                static std::mutex mtx;
                static std::map<std::array<int64_t, 5>, int> chosen;
                const std::array<int64_t, 5> key = {c->g.n0, c->g.n1, c->g.n2, (int64_t)iso + 2 * (int64_t)sizeof(T), (int64_t)c->het};
                {
                    std::lock_guard<std::mutex> lk(mtx);
                    auto it = chosen.find(key);
                    if (it != chosen.end()) tpr = it->second;
                }
                if (!tpr) {
                    std::vector<int> cand = {16, least_waste()};
                    for (int k = 1; k <= 4; ++k) {   // k patches exactly across axis 2
                        const int t = (int)((c->g.n2 + NV * k - 1) / (NV * k)) + 1;
                        if (t >= 8 && t <= 64) cand.push_back(t);
                    }
                    std::sort(cand.begin(), cand.end());
                    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
                    float best_ms = 1e30f;
                    hipEvent_t e0 = nullptr, e1 = nullptr;
                    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
                        for (int t : cand) {
                            float ms = 1e30f;
                            for (int rep = 0; rep < 2; ++rep) {
                                (void)hipEventRecord(e0, c->stream);
                                launch_with(t);
                                (void)hipEventRecord(e1, c->stream);
                                float m = 1e30f;
                                if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&m, e0, e1) == hipSuccess) ms = std::min(ms, m);
                            }
                            if (c->tune & 512) fprintf(stderr, "[dxh] march3d_v tuning: %d columns per patch row: %.4f ms\n", NV * t, ms);
                            if (ms < best_ms) { best_ms = ms; tpr = t; }
                        }
                    }
                    if (e0) (void)hipEventDestroy(e0);
                    if (e1) (void)hipEventDestroy(e1);
                    (void)hipGetLastError();
                    if (!tpr) tpr = least_waste();
                    std::lock_guard<std::mutex> lk(mtx);
                    chosen[key] = tpr;
                }
            }
            if (!tpr) tpr = least_waste();
            launch_with(tpr);
            return;
        }
    }
    if (((sizeof(T) == 4) && !(c->tune & 4096)) || (c->tune & 8192)) {
        // 2.5-D march.  Patch shape: the one of 4 x 256, 8 x 128, 16 x 64 (owning one row and one column less)
        // that wastes the least of its columns on this volume (ties: the widest, whose rims are the shortest); plane
        // chunks sized for ~4 blocks per CU.
        int best_m2 = 64;
        double best = -1.0;
        for (int m2 : {256, 128, 64}) {
            const int o1 = 1024 / m2 - 1, o2 = m2 - 1;  // owned
            const double eff = ((double)c->g.n1 / (double)(((c->g.n1 + o1 - 1) / o1) * (o1 + 1))) *
                               ((double)c->g.n2 / (double)(((c->g.n2 + o2 - 1) / o2) * (o2 + 1)));
            if (eff > best * 1.001) { best = eff; best_m2 = m2; }
        }
        if (const char* e = getenv("DXH_MARCH_M2")) {  // developer knob
            const int v = atoi(e);
            if (v == 64 || v == 128 || v == 256) best_m2 = v;
        }
        const int o1 = 1024 / best_m2 - 1, o2 = best_m2 - 1;
        const int nt1 = (int)((c->g.n1 + o1 - 1) / o1), nt2 = (int)((c->g.n2 + o2 - 1) / o2);
        // All blocks of a launch take about the same time (same number of planes each), so the launch should fill the
        // machine's resident-block slots once: as many plane chunks as fit without starting a second round.
        auto launch = [&](auto kernel) {
            int bpc = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kernel, 256, 0) != hipSuccess || bpc < 1) bpc = 2;
            if (c->waves_per_cu) bpc = std::max(1, c->waves_per_cu / 4);
            const int64_t slots = (int64_t)c->num_cu * bpc, tiles = (int64_t)nt1 * nt2;
            int64_t nchunk = std::max<int64_t>(1, slots / tiles);
            int64_t ppc = std::max<int64_t>(std::min<int64_t>(c->g.n0, 8), (c->g.n0 + nchunk - 1) / nchunk);
            if (c->force_chunks > 0) ppc = std::max<int64_t>(1, (c->g.n0 + c->force_chunks - 1) / c->force_chunks);
            nchunk = (c->g.n0 + ppc - 1) / ppc;
            const int64_t total = nchunk * tiles;
            if (c->tune & 512) fprintf(stderr, "[dxh] march3d patch %dx%d tiles %lld chunks %lld (%lld planes) blocks/CU %d\n", o1 + 1, o2 + 1, (long long)tiles, (long long)nchunk, (long long)ppc, bpc);
            hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 0, c->stream, c->g, (const T*)c->of, q,
                               (const T*)c->u[in], (T*)c->u[out], make_coef<T>(c), make_planes<T>(c), nt1, nt2, (int)nchunk,
                               (int)ppc);
        };
#define DXH_M3(ISO_, M2_) launch(k_march3d<T, ISO_, M2_>)
        if (iso) { if (best_m2 == 256) DXH_M3(true, 256); else if (best_m2 == 128) DXH_M3(true, 128); else DXH_M3(true, 64); }
        else     { if (best_m2 == 256) DXH_M3(false, 256); else if (best_m2 == 128) DXH_M3(false, 128); else DXH_M3(false, 64); }
#undef DXH_M3
        return;
    }
    const int nt0 = (int)((c->g.n0 + kB0 - 1) / kB0), nt1 = (int)((c->g.n1 + kB1 - 1) / kB1),
              nt2 = (int)((c->g.n2 + kB2 - 1) / kB2);
    const dim3 grid((unsigned)((int64_t)nt0 * nt1 * nt2)), blk(256);
    if (iso)
        hipLaunchKernelGGL((k_fused3d<T, true>), grid, blk, 0, c->stream, c->g, (const T*)c->of, q, (const T*)c->u[in],
                           (T*)c->u[out], make_coef<T>(c), make_planes<T>(c), nt1, nt2);
    else
        hipLaunchKernelGGL((k_fused3d<T, false>), grid, blk, 0, c->stream, c->g, (const T*)c->of, q, (const T*)c->u[in],
                           (T*)c->u[out], make_coef<T>(c), make_planes<T>(c), nt1, nt2);
}

static bool fused3d_supported(const Ctx* c) {
    if (c->desc.dim != 3 || c->desc.ndim != 3 || (c->tune & 2048)) return false;  // DXH_FUSED_TUNE bit 2048: three kernels
    const int64_t tiles = ((c->g.n0 + kB0 - 1) / kB0) * ((c->g.n1 + kB1 - 1) / kB1) * ((c->g.n2 + kB2 - 1) / kB2);
    return tiles < (int64_t)INT32_MAX / 4;  // either tiling (bricks, or patches x plane chunks) fits a 1-D grid
}

template <typename T>
static void launch_jacobi(Ctx* c, const void* xin, void* xout) {
    constexpr int NV = VecOf<T>::N;
    if (c->g.ndim == 3 && c->g.nsum_axes == 3 && c->g.n2 >= 64 && c->g.n2 % NV == 0 && !(c->tune & (16384 | 262144))) {
        // volumes, 16 bytes per lane, XCD x sweeps a contiguous run of (plane, row) tiles (DXH_FUSED_TUNE bit 262144: the
        // one-element-per-thread kernel below, A/B)
        const int nbx = (int)((c->g.n2 / NV + 255) / 256);
        const int64_t tiles = c->g.n0 * c->g.n1 * nbx, per_xcd = (tiles + 7) / 8;
        if (per_xcd * 8 < (int64_t)INT32_MAX) {
            hipLaunchKernelGGL((k_jacobi_vol_x<T>), dim3((unsigned)(per_xcd * 8)), dim3(256), 0, c->stream, c->g, (const T*)c->rs,
                               (const T*)xin, make_coef<T>(c), make_planes_min<T>(c), (T*)xout, nbx, tiles, per_xcd);
            return;
        }
    }
    if (c->g.ndim == 3 && c->g.nsum_axes == 3 && c->g.n2 >= 64 && c->g.n1 <= 65535 && c->g.n0 <= 65535 && !(c->tune & 16384)) {
        // volumes: 3-D grid, no per-element index divisions (DXH_FUSED_TUNE bit 16384: the generic kernel)
        const dim3 grid((unsigned)((c->g.n2 + 255) / 256), (unsigned)c->g.n1, (unsigned)c->g.n0);
        hipLaunchKernelGGL((k_jacobi_vol<T>), grid, dim3(256), 0, c->stream, c->g, (const T*)c->rs, (const T*)xin,
                           make_coef<T>(c), make_planes_min<T>(c), (T*)xout);
        return;
    }
    const Geom& gx = c->g;
    hipLaunchKernelGGL((k_jacobi<T>), elem_grid(gx), dim3(256), 0, c->stream, gx, (const T*)c->rs, (const T*)xin,
                       make_coef<T>(c), make_planes_min<T>(c), (T*)xout);
}


// volumes (dim = 3, all three axes in the sum): two or three sweeps per pass in tiles of the (axis 1, axis 2) plane
// (dxh_vol_multi.h; DXH_FUSED_TUNE bit 67108864: sweep by sweep, A/B)
static bool jacobi_vol_multi_supported(const Ctx* c) {
    return c->desc.dim == 3 && c->g.ndim == 3 && c->g.nsum_axes == 3 && (!c->het || c->het_cdiag || c->cp_dsc) && c->g.n1 >= 16 && c->g.n2 >= 32 &&
           c->g.n1 * c->g.n2 < (int64_t)INT32_MAX && !(c->tune & (64 | 67108864));
}

// float64 contexts, 2-D, 1 / 3 channels, scalar diag_scaled in the reciprocal's range: the LDS-row sweeps pipeline in its PLAIN
// form (dxh_np2_sweeps.h; DXH_FUSED_TUNE bit 134217728: k_jacobi_multi<double>, A/B)
static bool jacobi_plain_supported(const Ctx* c) {
    auto ok = [](double d) { return std::isfinite(d) && std::fabs(d) >= 0x1p-20 && std::fabs(d) <= 0x1p+20; };
    return c->desc.dtype == DXH_F64 && c->desc.dim == 2 && (c->g.n2 == 1 || c->g.n2 == 3) && (!c->het || c->het_cdiag) &&
           c->g.n0 <= (int64_t)INT32_MAX - 64 && ok(c->c_dsc) && !(c->tune & (8 | 134217728));
}

// sweeps of the next pass when `left` sweeps remain (0: no multi-sweep pass for this context).  2-D: fours, 5 = 3 + 2;
// volumes: threes, 4 = 2 + 2.  A last single sweep goes through the one-sweep kernels
static int jacobi_group(const Ctx* c, int left) {
    if (left < 2) return 0;
    if (c->desc.dim == 2) {
        if (!((c->g.n2 == 1 || c->g.n2 == 3) && (!c->het || c->het_cdiag) && !(c->tune & 64))) return 0;
        if (jacobi_plain_supported(c)) {   // float64 contexts: up to five sweeps per pass, never a single sweep left over
            const int k = std::min(5, left);
            return left - k == 1 ? k - 1 : k;
        }
        return left == 5 ? 3 : std::min(4, left);
    }
    if (!jacobi_vol_multi_supported(c)) return 0;
    if (c->het && !c->het_cdiag) return 2;   // a divisor plane: pairs
    return left == 4 ? 2 : std::min(3, left);
}

template <typename T>
static void launch_jacobi_vol_multi_t(Ctx* c, const void* xin, void* xout, int K) {
    VolMultiArgs<T> a{};
    a.rs = (const T*)c->rs;
    a.x_in = (const T*)xin;
    a.x_out = (T*)xout;
    a.n0 = c->g.n0; a.n1 = c->g.n1; a.n2 = c->g.n2; a.pitch = c->g.pitch;
    const int vy = 32 - 2 * K, vx = 64 - 2 * K;
    a.tiles_y = (int)((c->g.n1 + vy - 1) / vy);
    a.tiles_x = (int)((c->g.n2 + vx - 1) / vx);
    // two blocks per CU are resident: about one round of them, chunks of at least 8 K planes
    const int64_t tiles = (int64_t)a.tiles_y * a.tiles_x;
    int64_t nchunks = std::max<int64_t>(1, ((int64_t)c->num_cu * 2 + tiles / 2) / tiles);
    const int64_t ppc = std::max<int64_t>(8 * K, (c->g.n0 + nchunks - 1) / nchunks);
    nchunks = (c->g.n0 + ppc - 1) / ppc;
    a.planes_per_chunk = ppc;
    a.nchunks = (int)nchunks;
    a.dsc = (T)c->c_dsc;
    a.rdsc = T(0);
    if (std::isfinite((double)a.dsc) && std::fabs((double)a.dsc) >= 0x1p-20 && std::fabs((double)a.dsc) <= 0x1p+20) a.rdsc = T(1) / a.dsc;
    a.dsc_plane = make_planes_min<T>(c).dsc;
    const int64_t total = tiles * nchunks;
    launch_jacobi_vol_multi<T>(K, a, (unsigned)(((total + 7) / 8) * 8), c->stream);
}

template <typename T, int C, bool COUPLED, int K>
static void launch_jacobi_multi_k(Ctx* c, const void* xin, void* xout) {
    using SG = StripGeomK<T, C, K>;
    JacobiMultiArgs<T> a{};
    a.rs = (const T*)c->rs;
    a.x_in = (const T*)xin;
    a.x_out = (T*)xout;
    a.n0 = c->g.n0; a.L = c->g.L; a.pitch = c->g.pitch;
    a.nstrips = (int)((c->g.L + SG::STRIDE - 1) / SG::STRIDE);
    const int nbx = (a.nstrips + 3) / 4;
    int64_t nchunks = std::max<int64_t>(1, (int64_t)c->num_cu * 4 / nbx);
    if (nchunks >= 8) nchunks = nchunks / 8 * 8;
    const int64_t rpc = std::max<int64_t>(8 * K, (c->g.n0 + nchunks - 1) / nchunks);
    nchunks = (c->g.n0 + rpc - 1) / rpc;
    a.rows_per_chunk = rpc;
    a.nchunks = (int)nchunks;
    a.dsc = (T)c->c_dsc;
    a.rdsc = T(0);
    a.div3 = 0;
    if (std::isfinite((double)a.dsc) && std::fabs((double)a.dsc) >= 0x1p-20 && std::fabs((double)a.dsc) <= 0x1p+20) {
        a.rdsc = T(1) / a.dsc;
        if constexpr (std::is_same<T, float>::value)
            if (!(c->tune & 128)) a.div3 = div3_verified(c, a.dsc, a.dsc);  // remembered per divisor after the first call
    }
    hipLaunchKernelGGL((k_jacobi_multi<T, C, COUPLED, K>), dim3((unsigned)(nbx * nchunks)), dim3(256), 0, c->stream, a);
}

template <typename T>
static void launch_jacobi_multi(Ctx* c, const void* xin, void* xout, int K) {
    if (c->desc.dim == 3) { launch_jacobi_vol_multi_t<T>(c, xin, xout, K); return; }
    if constexpr (std::is_same<T, double>::value) {
        if (jacobi_plain_supported(c)) {
            c->np2_sweeps = K;
            c->jm_plain = true;
            c->jm_in = xin;
            c->jm_out = xout;
            launch_fused_multi<double>(c, c->cur, c->cur ^ 1, false, K);
            c->np2_sweeps = 0;
            c->jm_plain = false;
            if (!c->no_kernel) return;
            c->no_kernel = false;   // (cannot happen for K = 2 .. 5; fall through to the generic pass)
        }
    }
    const bool coupled = c->g.ndim == 3 && c->g.nsum_axes == 3;
#define DXH_JM(C_, CP_)                                                        \
    switch (K) {                                                               \
        case 2: launch_jacobi_multi_k<T, C_, CP_, 2>(c, xin, xout); break;     \
        case 3: launch_jacobi_multi_k<T, C_, CP_, 3>(c, xin, xout); break;     \
        default: launch_jacobi_multi_k<T, C_, CP_, 4>(c, xin, xout); break;    \
    }
    if (c->g.n2 == 1) {
        // an (H, W, 1) array still sums over its third axis: both channel neighbours are the element itself
        if (coupled) { DXH_JM(1, true) } else { DXH_JM(1, false) }
    }
    else if (coupled) { DXH_JM(3, true) }
    else { DXH_JM(3, false) }
#undef DXH_JM
}

// Jacobi.__call__ on device buffers: x0 in u[cur], rs ready; leaves the result in u[cur ^ 1]... or, when
// the reference returns x0 itself (tol met on the first sweep, or maxiter == 0), copies it there.
// Needs a third buffer for sweeps > 1: `tmp`.
// first_f32 (numpy2 flow, double context): x0 still is the float32 image, whose neighbours the reference sums in float32
// in the first sweep (k_jacobi_np2_first); every later sweep is the context's own.
// passes jacobi_sweeps needs for `sweeps` sweeps without a tolerance (its grouping rule: 5 = 3 + 2, else fours, a last single)
static int jacobi_passes(const Ctx* c, int sweeps, bool first_f32);

// started: the first sweep has been done already (by k_shrink_rhs_np2) and its result lives in `started` -- u[cur ^ 1] when the
// remaining passes are even in number, `tmp` when they are odd (np2_pre_target), so that the last pass lands in u[cur ^ 1]
template <typename T>
static int jacobi_sweeps(Ctx* c, int maxiter, double tol, void* tmp, int64_t* launches, int* sweeps_done, bool first_f32 = false,
                         const void* started = nullptr) {
    const int in = c->cur, out = c->cur ^ 1;
    const bool use_tol = tol >= 0.0 && !std::isnan(tol);
    double nrm0 = 0.0;
    if (use_tol) {
        int rc = sq_diff<T>(c, c->u[in], nullptr, 0, c->g.n0, &nrm0, 0);
        if (rc) return rc;
        nrm0 = std::sqrt(nrm0);
    }
    // x lives alternately in `out` and `tmp`; the first sweep reads u[in]
    const void* x = c->u[in];
    void* bufs[2] = {c->u[out], tmp};
    int w = 0, done = 0;
    bool stop = false;
    if (started) {
        x = started;
        w = started == c->u[out] ? 1 : 0;   // the next pass writes the other buffer
        done = 1;
        --maxiter;
    } else if (!use_tol && tmp) {
        // without a tolerance the number of passes is known: start in the buffer that makes the last one land in u[out]
        // (no plane copy at the end; with a tolerance the stop is not known in advance)
        int left = maxiter, passes = 0;
        if (first_f32 && left >= 1) { ++passes; --left; }
        for (int k = jacobi_group(c, left); k > 0; k = jacobi_group(c, left)) { left -= k; ++passes; }
        passes += left;
        if (passes >= 2 && passes % 2 == 0) w = 1;
    }
    auto single = [&](bool f32) -> int {
        void* xn = bufs[w];
        if (f32) {
            if constexpr (std::is_same<T, double>::value)
                hipLaunchKernelGGL(k_jacobi_np2_first, elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->rs,
                                   (const double*)x, c->c_dsc, (c->het && !c->het_cdiag) ? (const double*)c->cp_dsc : nullptr,
                                   (double*)xn);
        } else {
            launch_jacobi<T>(c, x, xn);
        }
        ++*launches;
        if (use_tol) {
            double inc = 0.0;
            int rc = sq_diff<T>(c, xn, x, 0, c->g.n0, &inc, 0);
            if (rc) return rc;
            if (std::sqrt(inc) / nrm0 < tol) { stop = true; return DXH_OK; }  // reference keeps the PREVIOUS iterate (jacobi.py:123-126)
        }
        x = xn;
        w ^= 1;
        ++done;
        return DXH_OK;
    };
    if (first_f32 && maxiter >= 1) {
        int rc = single(true);
        if (rc) return rc;
        --maxiter;
    }
    if (!stop && !use_tol && jacobi_group(c, maxiter) > 0) {
        // several sweeps per pass; a remainder of one sweep goes through the single-sweep kernel below
        int left = maxiter;
        for (int k = jacobi_group(c, left); k > 0; k = jacobi_group(c, left)) {
            void* xn = bufs[w];
            launch_jacobi_multi<T>(c, x, xn, k);
            ++*launches;
            x = xn;
            w ^= 1;
            left -= k;
            done += k;
        }
        maxiter = left;  // 0 or 1
    }
    for (int s = 0; s < maxiter && !stop; ++s) {
        int rc = single(false);
        if (rc) return rc;
    }
    if (x != c->u[out]) DXH_HIP(hipMemcpyAsync(c->u[out], x, c->plane_bytes, hipMemcpyDeviceToDevice, c->stream));
    if (sweeps_done) *sweeps_done = done;
    return DXH_OK;
}

static int jacobi_passes(const Ctx* c, int sweeps, bool first_f32) {
    int left = sweeps, passes = 0;
    if (first_f32 && left >= 1) { ++passes; --left; }
    for (int k = jacobi_group(c, left); k > 0; k = jacobi_group(c, left)) { left -= k; ++passes; }
    return passes + left;
}

// the u-update of one outer iteration: Jacobi sweeps, or the plugged-in solver on (u[cur], rhs) -> u[cur ^ 1]
template <typename T>
static int solve_u(Ctx* c, int jacobi_maxiter, double jacobi_tol, void* tmp, int64_t* launches) {
    if (!c->solver) return jacobi_sweeps<T>(c, jacobi_maxiter, jacobi_tol, tmp, launches, nullptr);
    const int64_t before = c->solver->launches;
    int rc = c->solver->solve(c->u[c->cur], c->rs, c->u[c->cur ^ 1], nullptr);
    *launches += c->solver->launches - before;
    return rc;
}

static Np2Coef np2_coef(const Ctx* c) {
    Np2Coef k{};
    k.dsc = c->c_dsc;
    k.kq = c->c_kq;
    k.ell = (float)c->c_ell;
    k.diag = (float)c->c_diag;
    k.kq_plane = c->het ? (const double*)c->cp_kq : nullptr;
    k.first = c->np2_first ? 1 : 0;
    k.ell_plane = c->np2_adapted ? (const double*)c->cp_ell : nullptr;
    k.q32 = c->np2_q32 ? 1 : 0;
    return k;
}

// numpy2 flow (double context): one outer iteration by k_fused_np2
static int launch_np2(Ctx* c, int in, int out) {
    FusedArgs<double> a{};
    a.of = (const double*)c->of;
    a.u_in = (const double*)c->u[in];
    a.u_out = (double*)c->u[out];
    a.q0_in = (const double*)c->q[0][in];
    a.q1_in = (const double*)c->q[1][in];
    a.q0_out = (double*)c->q[0][out];
    a.q1_out = (double*)c->q[1][out];
    a.n0 = c->g.n0;
    a.L = c->g.L;
    a.pitch = c->g.pitch;
    const int C = (int)c->g.n2;
    const int stride = C == 1 ? StripGeom<double, 1, true>::STRIDE : C == 2 ? StripGeom<double, 2, true>::STRIDE
                     : C == 3 ? StripGeom<double, 3, true>::STRIDE : StripGeom<double, 4, true>::STRIDE;
    a.nstrips = (int)((c->g.L + stride - 1) / stride);
    const int nbx = (a.nstrips + 3) / 4;
    auto choose = [&](int64_t rows, int64_t extra_per_xcd) -> int64_t {
        const int64_t target_blocks = (int64_t)c->num_cu * (c->waves_per_cu ? c->waves_per_cu : 8) / 4 - 8 * extra_per_xcd;
        const int64_t nchunks = std::max<int64_t>(1, target_blocks / nbx);
        const int64_t rpc = std::max<int64_t>(8, (rows + nchunks - 1) / nchunks);
        return (rows + rpc - 1) / rpc;
    };
    const ChunkLayout lay = make_layout(c, nbx, 1, choose);
    const unsigned grid = apply_layout(c, a, lay, nbx);
    a.remap = 1;
    const Np2Coef k = np2_coef(c);
    const bool coupled = c->desc.ndim == 3 && c->desc.nsum == DXH_NSUM_ALL_AXES;
    const bool arr = k.kq_plane != nullptr;
#define DXH_NP2(C_)                                                                                          \
    if (coupled) { if (arr) hipLaunchKernelGGL((k_fused_np2<C_, true, true>), dim3(grid), dim3(256), 0, c->stream, a, k);   \
                   else hipLaunchKernelGGL((k_fused_np2<C_, true, false>), dim3(grid), dim3(256), 0, c->stream, a, k); }    \
    else         { if (arr) hipLaunchKernelGGL((k_fused_np2<C_, false, true>), dim3(grid), dim3(256), 0, c->stream, a, k);  \
                   else hipLaunchKernelGGL((k_fused_np2<C_, false, false>), dim3(grid), dim3(256), 0, c->stream, a, k); }
    switch (C) {
        case 1: DXH_NP2(1) break;
        case 2: DXH_NP2(2) break;
        case 3: DXH_NP2(3) break;
        default: DXH_NP2(4) break;
    }
#undef DXH_NP2
    DXH_HIP(hipGetLastError());
    return DXH_OK;
}

// numpy2 flow outside the fused sweeps (dim = 3, more than four channels, several Jacobi sweeps): right-hand side, sweeps,
// shrinkage as three kinds of launches on the double context; u[cur] -> u[cur ^ 1], q updated in place
template <int DIM>
static int np2_generic_iteration(Ctx* c, int maxiter, double tol, void* tmp, int64_t* launches, bool more) {
    const int in = c->cur, out = c->cur ^ 1;
    const dim3 grid = elem_grid(c->g), blk(256);
    const Np2Coef k = np2_coef(c);
    auto q = make_q<double, DIM>(c, in, in);
    const bool fdb = c->from_db;   // first iteration of a warm start: d, b are explicit planes (float32 values)
    auto db = make_db<double>(c);
    int swept = 0;
    if (c->np2_pre) {
        // rs and the first sweep of this iteration came out of the previous iteration's fused pass
        c->np2_pre = false;
        int rc = jacobi_sweeps<double>(c, maxiter, tol, tmp, launches, &swept, false, c->np2_pre_x);
        if (rc) return rc;
    } else {
    if (c->het && !c->het_cdiag) {   // array-valued omega and / or ell: float64 enters the right-hand side (k_rhs_np2_het)
        Np2Het hk{};
        hk.ell_p = c->ell_is_arr ? (const double*)c->cp_ell : nullptr;
        hk.diag_p = (const double*)c->cp_diag;
        hk.kq_p = (const double*)c->cp_kq;
        hk.kq = c->c_kq;
        hk.diag_d = c->c_diag;
        hk.ell_f = (float)c->ell_s;
        hk.diag_f = (float)c->c_diag;
        hk.rhs64 = c->om_is_arr ? 1 : 0;
        if (fdb) hipLaunchKernelGGL((k_rhs_np2_het<DIM, true>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, hk, (double*)c->rs);
        else hipLaunchKernelGGL((k_rhs_np2_het<DIM, false>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, hk, (double*)c->rs);
    }
    else if (fdb) {
        if (k.kq_plane) hipLaunchKernelGGL((k_rhs_np2<DIM, true, true>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, k, (double*)c->rs);
        else hipLaunchKernelGGL((k_rhs_np2<DIM, false, true>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, k, (double*)c->rs);
    }
    else if (k.kq_plane) hipLaunchKernelGGL((k_rhs_np2<DIM, true>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, k, (double*)c->rs);
    else hipLaunchKernelGGL((k_rhs_np2<DIM, false>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, k, (double*)c->rs);
    ++*launches;
    int rc = jacobi_sweeps<double>(c, maxiter, tol, tmp, launches, &swept, c->np2_first);
    if (rc) return rc;
    }
    // the solver handed the float32 iterate itself back (tolerance met by the first sweep of a call): float32 D u + b
    const int u32 = (c->np2_first && swept == 0) ? 1 : 0;
    // Another iteration follows in this call and it is an ordinary one (scalar omega / ell, float64 iterate, no tolerance):
    // shrinkage, the next right-hand side and its first sweep in one pass (k_shrink_rhs_np2); q goes to the `out` buffers.
    // (DXH_FUSED_TUNE bit 524288: the three launches, A/B)
    const bool use_tol = tol >= 0.0 && !std::isnan(tol);
    // Measured (profiles/r04_np2_generic.txt): volumes gain at every sweep count (256^3, one sweep: 0.47 -> 0.34 ms per
    // iteration, 12 launches instead of 30); 2-D gains little (five sweeps 2.50 -> 2.43 ms: the K-sweep pass behind it is what
    // costs) and loses at exactly two sweeps, where the fused first sweep breaks up the two-sweep pass -- left unfused
    const bool fuse = more && !fdb && !u32 && !k.q32 && !use_tol && (!c->het || c->het_cdiag) && c->of32_valid && !(c->tune & 524288) &&
                      (maxiter == 1 || tmp) && (DIM == 3 || maxiter != 2);
    if (fuse) {
        // where the first sweep of the next iteration goes: its u[cur ^ 1] is this iteration's u[in]; the remaining sweeps
        // alternate between that plane and tmp and must end in it
        const int rest = jacobi_passes(c, maxiter - 1, false);
        void* x1 = (rest % 2 == 0) ? c->u[in] : tmp;
        auto q2 = make_q<double, DIM>(c, in, out);
        Np2Coef kf = k;
        auto okd = [](double d) { return std::isfinite(d) && std::fabs(d) >= 0x1p-20 && std::fabs(d) <= 0x1p+20; };
        if (okd(k.dsc) && okd((double)k.diag)) { kf.rdsc = 1.0 / k.dsc; kf.rdiag = 1.0f / k.diag; }
        const bool fastk = !k.kq_plane && k.kq >= 0.0;
#define DXH_SR(KQ_, WR_, FK_) hipLaunchKernelGGL((k_shrink_rhs_np2<DIM, KQ_, WR_, FK_>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], q2, (const float*)c->of32, kf, (double*)c->rs, (double*)x1)
        if (maxiter > 1) {
            if (k.kq_plane) DXH_SR(true, true, false);
            else if (fastk) DXH_SR(false, true, true);
            else DXH_SR(false, true, false);
        } else {
            if (k.kq_plane) DXH_SR(true, false, false);
            else if (fastk) DXH_SR(false, false, true);
            else DXH_SR(false, false, false);
        }
#undef DXH_SR
        c->np2_pre = true;
        c->np2_pre_x = x1;
        // the caller swaps q[.][in] and q[.][out] (in-place convention of the three launches): undo by swapping here
        for (int j = 0; j < DIM; ++j) std::swap(c->q[j][in], c->q[j][out]);
    }
    else if (fdb) {
        if (k.kq_plane) hipLaunchKernelGGL((k_shrink_np2<DIM, true, true>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], q, db, k, u32);
        else hipLaunchKernelGGL((k_shrink_np2<DIM, false, true>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], q, db, k, u32);
    }
    else if (k.kq_plane) hipLaunchKernelGGL((k_shrink_np2<DIM, true>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], q, db, k, u32);
    else hipLaunchKernelGGL((k_shrink_np2<DIM, false>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], q, db, k, u32);
    c->from_db = false;
    c->np2_q32 = u32 != 0;
    ++*launches;
    DXH_HIP(hipGetLastError());
    if (swept > 0) c->np2_first = false;   // (a tolerance met by the very first sweep hands back the float32 iterate itself)
    return DXH_OK;
}

// numpy2 flow with explicit d, b planes (the adaptive schedule): right-hand side from the planes, sweeps, shrinkage into the planes
template <int DIM>
static int np2_explicit_iteration(Ctx* c, int maxiter, double tol, void* tmp, int64_t* launches) {
    const int in = c->cur, out = c->cur ^ 1;
    const dim3 grid = elem_grid(c->g), blk(256);
    const Np2Coef k = np2_coef(c);
    DXH_REQUIRE(!c->het || c->het_cdiag || c->np2_adapted, "numpy2 promotion: explicit split variables need scalar omega and ell");
    auto q = make_q<double, DIM>(c, in, in);
    auto db = make_db<double>(c);
    if (k.kq_plane) hipLaunchKernelGGL((k_rhs_np2<DIM, true, true>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, k, (double*)c->rs);
    else hipLaunchKernelGGL((k_rhs_np2<DIM, false, true>), grid, blk, 0, c->stream, c->g, (const double*)c->of, q, db, k, (double*)c->rs);
    ++*launches;
    int swept = 0;
    int rc = jacobi_sweeps<double>(c, maxiter, tol, tmp, launches, &swept, c->np2_first);
    if (rc) return rc;
    const int u32 = (c->np2_first && swept == 0) ? 1 : 0;
    if (k.kq_plane) hipLaunchKernelGGL((k_shrink_db_np2<DIM, true>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], (double*)c->dpl[0], (double*)c->dpl[1], (double*)c->dpl[2], (double*)c->bpl[0], (double*)c->bpl[1], (double*)c->bpl[2], k, u32);
    else hipLaunchKernelGGL((k_shrink_db_np2<DIM, false>), grid, blk, 0, c->stream, c->g, (const double*)c->u[out], (double*)c->dpl[0], (double*)c->dpl[1], (double*)c->dpl[2], (double*)c->bpl[0], (double*)c->bpl[1], (double*)c->bpl[2], k, u32);
    ++*launches;
    DXH_HIP(hipGetLastError());
    if (swept > 0) c->np2_first = false;
    return DXH_OK;
}

// after a launcher: a (K, coefficient kind) combination without an instantiated pipeline is an error, not a skipped launch
static int launched_or_fail(Ctx* c, int K) {
    if (!c->no_kernel) return DXH_OK;
    c->no_kernel = false;
    set_error("no fused pipeline is instantiated for K = %d with these coefficients in this dtype flow (DXH_FUSED_TUNE / "
              "DXH_FUSE_K developer knobs can ask for one that does not exist); nothing was launched", K);
    return DXH_ESTATE;
}

template <typename T>
static void launch_any(Ctx* c, int in, int out, bool iso, bool dpp, int K) {
    if (c->np2) {
        if (K >= 2) launch_fused_multi<T>(c, in, out, iso, K);
        else if constexpr (std::is_same<T, double>::value) (void)launch_np2(c, in, out);
        c->np2_first = false;
        return;
    }
    if (K >= 2) launch_fused_multi<T>(c, in, out, iso, K);
    else launch_fused<T>(c, in, out, iso, dpp);
}

template <typename T>
static int run_t(Ctx* c, int n_iter, int jacobi_maxiter, double jacobi_tol, double eps, double norm_scale,
                 uint32_t flags, dxh_tvd_result* res) {
    const bool iso = flags & DXH_FLAG_ISOTROPIC;
    const bool use_eps = eps >= 0.0 && !std::isnan(eps);
    const bool use_tol = jacobi_tol >= 0.0 && !std::isnan(jacobi_tol);
    RoctxRange range("dxh_tvd_run: %lld TV iterations (%lld Jacobi sweeps each)", n_iter, jacobi_maxiter);
    c->signalling = false;  // only dxh_tvd_run_slab builds signalling launches
    c->row_lo = c->row_hi = 0;
    c->np2_pre = false;     // nothing prepared by an earlier call survives (halo exchanges, new coefficients, downloads between calls)
    const bool dpp = !(flags & DXH_FLAG_NO_DPP);
    bool fused_ok = fused_supported(c) && jacobi_maxiter == 1 && !use_tol && !(flags & DXH_FLAG_NO_FUSED) && !c->solver;
    const int dim = c->desc.dim;
    bool np2_generic = false;
    if (c->np2) {
        // the reference's float32-image flow: Jacobi sweeps, anisotropic shrinkage, scalar omega and
        // ell (mu: scalar or array), state as uploaded.  One sweep on a 2-D image of up to four channels runs in the fused
        // pipelines; everything else (dim = 3, more channels, several sweeps) as three kinds of launches
        DXH_REQUIRE(jacobi_maxiter >= 1 && !c->solver, "numpy2 promotion: Jacobi sweeps only");
        DXH_REQUIRE(!iso, "numpy2 promotion: anisotropic shrinkage only (the reference's isotropic branch is float64-only)");
        DXH_REQUIRE(!c->het || c->cp_kq, "numpy2 promotion: the coefficient planes are missing");
        if constexpr (!std::is_same<T, double>::value) { set_error("numpy2 promotion needs a float64 context"); return DXH_ESTATE; }
        np2_generic = !(fused_ok && dim == 2 && c->g.n2 >= 1 && c->g.n2 <= 4) || (c->het && !c->het_cdiag);
        if (np2_generic) fused_ok = false;
    }
    if (c->solver) {
        // Solver.update_params(mass_coeff=omega, diffusion_coeff=ell) (split_bregman_tvd.py:63-67, 154-158)
        if (c->solver_kind == DXH_SOLVER_MG)
            DXH_REQUIRE(!c->het, "MG inside the TV loop needs scalar omega / ell (the reference's MG() is not heterogeneous and "
                                 "fails to broadcast array coefficients on its coarse levels)");
        int rc = c->solver->set_params(c->omega_s, c->het ? c->om_plane : nullptr, c->ell_s, c->het ? c->cp_ell : nullptr);
        if (rc) return rc;
    }
    c->last_iso_prev = c->last_iso;
    c->last_iso = iso;

    DevTmp tmp_guard;
    void*& tmp = tmp_guard.p;
    const bool explicit_mode = (flags & DXH_FLAG_EXPLICIT) || c->explicit_db;
    const bool fused3d_ok = fused3d_supported(c) && jacobi_maxiter == 1 && !use_tol && !(flags & DXH_FLAG_NO_FUSED) &&
                            !c->solver && !explicit_mode && !c->np2;
    // numpy2, Jacobi(maxiter = 2 .. 6) without a tolerance, scalar omega / ell, 1 / 3 channels: one pass per iteration with
    // the sweeps as its stages (dxh_np2_sweeps.h; DXH_FUSED_TUNE bit 33554432: the three-launch flow instead, A/B)
    // (float64 images, anisotropic, scalar coefficients: the same pipeline with every quantity float64)
    const bool sweeps_ok = std::is_same<T, double>::value && dim == 2 && (c->g.n2 == 1 || c->g.n2 == 3) &&
                           (c->np2 ? (c->of32_valid && (!c->het || (c->het_cdiag && c->cp_kq && make_coef<T>(c).const_diag && !(c->tune & 65536))))
                                   : (!c->het && !iso && !c->solver)) &&
                           jacobi_maxiter >= kNp2SweepsMin && jacobi_maxiter <= kNp2SweepsMax && !use_tol && dpp &&
                           !(flags & DXH_FLAG_NO_FUSED) && !explicit_mode && fused_supported(c) && multi_supported(c) &&
                           !(c->tune & 33554432);
    const bool need_generic = (!fused_ok && !fused3d_ok && !sweeps_ok) || c->from_db || c->np2_q32;
    if (need_generic) {
        int rc = alloc_plane(c, &c->rs);
        if (rc) return rc;
        if (jacobi_maxiter > 1) {
            DXH_HIP(hipMalloc(&tmp, c->plane_bytes));
        }
    }
    if (explicit_mode) {
        fused_ok = false;
        int rc = alloc_plane(c, &c->rs);
        if (rc) return rc;
        if (jacobi_maxiter > 1 && !tmp) DXH_HIP(hipMalloc(&tmp, c->plane_bytes));
        if (!c->explicit_db) {
            if (!c->from_db) {
                // expand the compressed state into explicit planes
                for (int j = 0; j < dim; ++j)
                    if ((rc = alloc_plane(c, &c->dpl[j])) || (rc = alloc_plane(c, &c->bpl[j]))) return rc;
                const dim3 grid = elem_grid(c->g), blk(256);
#define DXH_XP(DIM_, ISO_) hipLaunchKernelGGL((k_expand_planes<T, DIM_, ISO_>), grid, blk, 0, c->stream, c->g, make_q<T, DIM_>(c, c->cur, c->cur), \
        make_coef<T>(c), make_planes<T>(c), (T*)c->dpl[0], (T*)c->dpl[1], (T*)c->dpl[2], (T*)c->bpl[0], (T*)c->bpl[1], (T*)c->bpl[2])
                if (c->np2) {   // float32 d, b re-derived from the stored q with the reference's operations
                    const Np2Coef k = np2_coef(c);
#define DXH_XPN(DIM_) do { auto qq = make_q<double, DIM_>(c, c->cur, c->cur);                                                              \
        if (k.kq_plane) hipLaunchKernelGGL((k_expand_planes_np2<DIM_, true>), grid, blk, 0, c->stream, c->g, qq, k, (double*)c->dpl[0], (double*)c->dpl[1], (double*)c->dpl[2], (double*)c->bpl[0], (double*)c->bpl[1], (double*)c->bpl[2]);  \
        else hipLaunchKernelGGL((k_expand_planes_np2<DIM_, false>), grid, blk, 0, c->stream, c->g, qq, k, (double*)c->dpl[0], (double*)c->dpl[1], (double*)c->dpl[2], (double*)c->bpl[0], (double*)c->bpl[1], (double*)c->bpl[2]); } while (0)
                    if (dim == 2) DXH_XPN(2); else DXH_XPN(3);
#undef DXH_XPN
                }
                else if (dim == 2) { if (c->last_iso_prev) DXH_XP(2, true); else DXH_XP(2, false); }
                else          { if (c->last_iso_prev) DXH_XP(3, true); else DXH_XP(3, false); }
#undef DXH_XP
                DXH_HIP(hipGetLastError());
            }
            c->from_db = false;
            c->explicit_db = true;
        }
    }
    double nrm_img = 0.0;
    if (use_eps) {
        if (norm_scale > 0.0)
            nrm_img = norm_scale;
        else
            nrm_img = std::sqrt(c->sq_img);
    }

    int64_t launches = 0;
    int done = 0, converged = 0;
    double rel = std::numeric_limits<double>::quiet_NaN();
    const bool async = (flags & DXH_FLAG_ASYNC) && !use_eps && !use_tol;
    if (!async) DXH_HIP(hipEventRecord(c->ev0, c->stream));
    for (int it = 0; it < n_iter; ++it) {
        const int in = c->cur, out = c->cur ^ 1;
        if (fused_ok && !c->from_db && !c->np2_q32 && use_eps && dpp && multi_supported(c) && pick_k(c, n_iter - it) >= 2) {
            // eps with temporal blocking: one launch does K iterations and also reduces every iteration's
            // squared increment; if the test fires at an inner iteration j, the K-iteration result is
            // discarded (the input buffers still hold iteration `it`) and exactly j+1 iterations are redone
            const int K = pick_k(c, n_iter - it);
            const bool first0 = c->np2_first;   // numpy2 flow: the redo starts from the same (float32-valued) iterate
            launch_fused_multi<T>(c, in, out, iso, K, true);
            if (int rc = launched_or_fail(c, K)) return rc;
            c->np2_first = false;
            ++launches;
            if (!c->inc_partial) { set_error("out of device memory for the increment partials"); return DXH_ENOMEM; }
            hipLaunchKernelGGL(k_sum_partials, dim3(K), dim3(256), 0, c->stream, c->inc_partial, K * c->inc_nwaves,
                               c->red_out + 8);
            double incs[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // K <= kNp2GrayK = 6 (four entries until round 4: two doubles past the array at K = 6)
            static_assert(kNp2GrayK <= 8 && kNp2K <= 8, "increments of one launch");
            DXH_HIP(hipStreamSynchronize(c->stream));
            for (int j = 0; j < K; ++j) incs[j] = reinterpret_cast<volatile double*>(c->red_out)[8 + j];
            int hit = -1;
            for (int j = 0; j < K; ++j) {
                rel = std::sqrt(incs[j]) / nrm_img;
                if (rel < eps) { hit = j; break; }
            }
            if (hit < 0) {
                c->cur = out;
                it += K - 1;
                done = it + 1;
                continue;
            }
            converged = 1;
            done = it + hit + 1;
            if (hit == K - 1) {
                c->cur = out;
            } else {
                // redo exactly hit+1 iterations from the untouched input state
                int left = hit + 1, a_in = in, a_out = out;
                c->np2_first = first0;
                while (left > 0) {
                    const int k2 = pick_k(c, left);
                    launch_any<T>(c, a_in, a_out, iso, dpp, k2);
                    if (int rc = launched_or_fail(c, k2)) return rc;
                    ++launches;
                    left -= k2;
                    std::swap(a_in, a_out);
                }
                c->cur = a_in;  // the buffer written last
            }
            break;
        } else if (fused_ok && !c->from_db && !c->np2_q32) {
            // temporal blocking: fuse_k iterations per launch while enough iterations remain and no
            // per-iteration host test is requested
            const int K = (!use_eps && dpp && multi_supported(c)) ? pick_k(c, n_iter - it) : 1;
            launch_any<T>(c, in, out, iso, dpp, K);
            if (int rc = launched_or_fail(c, K)) return rc;
            it += K - 1;
            ++launches;
            c->cur = out;
        } else if (sweeps_ok && !c->from_db && !c->np2_q32) {
            c->np2_sweeps = jacobi_maxiter;
            const bool fold = use_eps && np2_sweeps_fold_norm(jacobi_maxiter, c->np2);   // (with eps: the pass also reduces the increment)
            launch_fused_multi<T>(c, in, out, false, jacobi_maxiter, fold);
            c->np2_sweeps = 0;
            if (int rc = launched_or_fail(c, jacobi_maxiter)) return rc;
            c->np2_first = false;
            ++launches;
            c->cur = out;
            if (fold) {
                if (!c->inc_partial) { set_error("out of device memory for the increment partials"); return DXH_ENOMEM; }
                hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, c->inc_partial, c->inc_nwaves, c->red_out + 8);
                DXH_HIP(hipStreamSynchronize(c->stream));
                rel = std::sqrt(reinterpret_cast<volatile double*>(c->red_out)[8]) / nrm_img;
                done = it + 1;
                if (rel < eps) { converged = 1; break; }
                continue;
            }
        } else if (fused3d_ok && !c->from_db) {
            launch_fused3d<T>(c, in, out, iso);
            ++launches;
            c->cur = out;
        } else if (explicit_mode && c->np2) {
            int rc = dim == 2 ? np2_explicit_iteration<2>(c, jacobi_maxiter, jacobi_tol, tmp, &launches)
                              : np2_explicit_iteration<3>(c, jacobi_maxiter, jacobi_tol, tmp, &launches);
            if (rc) return rc;
            c->cur = out;
        } else if (explicit_mode) {
            // rhs from explicit planes, Jacobi, shrink back into the planes
            if (dim == 2) launch_rhs<T, 2>(c, iso, true); else launch_rhs<T, 3>(c, iso, true);
            ++launches;
            int rc = solve_u<T>(c, jacobi_maxiter, jacobi_tol, tmp, &launches);
            if (rc) return rc;
            {
                const dim3 grid = elem_grid(c->g), blk(256);
                auto co = make_coef<T>(c);
                auto cp = make_planes<T>(c);
#define DXH_SDB(DIM_, ISO_) hipLaunchKernelGGL((k_shrink_db<T, DIM_, ISO_>), grid, blk, 0, c->stream, c->g, (const T*)c->u[out], \
        (T*)c->dpl[0], (T*)c->dpl[1], (T*)c->dpl[2], (T*)c->bpl[0], (T*)c->bpl[1], (T*)c->bpl[2], co, cp)
                if (dim == 2) { if (iso) DXH_SDB(2, true); else DXH_SDB(2, false); }
                else          { if (iso) DXH_SDB(3, true); else DXH_SDB(3, false); }
#undef DXH_SDB
            }
            ++launches;
            c->cur = out;
        } else if (c->np2) {   // np2_generic, or the first iteration of a warm start
            // (with an eps test the previous iterate must survive the iteration: its plane is where the fused pass would put
            // the next first sweep)
            const bool more = it + 1 < n_iter && !use_eps && !sweeps_ok;   // (the sweeps pipeline starts from the plain state)
            int rc = dim == 2 ? np2_generic_iteration<2>(c, jacobi_maxiter, jacobi_tol, tmp, &launches, more)
                              : np2_generic_iteration<3>(c, jacobi_maxiter, jacobi_tol, tmp, &launches, more);
            if (rc) return rc;
            for (int j = 0; j < dim; ++j) std::swap(c->q[j][in], c->q[j][out]);
            c->cur = out;
        } else {
            const bool fdb = c->from_db;
            if (dim == 2) launch_rhs<T, 2>(c, iso, fdb); else launch_rhs<T, 3>(c, iso, fdb);
            ++launches;
            int rc = solve_u<T>(c, jacobi_maxiter, jacobi_tol, tmp, &launches);
            if (rc) return rc;
            if (dim == 2) launch_shrink<T, 2>(c, c->u[out], iso, fdb); else launch_shrink<T, 3>(c, c->u[out], iso, fdb);
            ++launches;
            // the q planes were updated in place in buffer `cur`; move them along with u
            for (int j = 0; j < dim; ++j) std::swap(c->q[j][in], c->q[j][out]);
            c->cur = out;
            c->from_db = false;
        }
        done = it + 1;
        if (use_eps) {
            double inc = 0.0;
            int rc = sq_diff<T>(c, c->u[c->cur], c->u[c->cur ^ 1], 0, c->g.n0, &inc, 0);
            if (rc) return rc;
            rel = std::sqrt(inc) / nrm_img;
            if (rel < eps) { converged = 1; break; }
        }
    }
    DXH_HIP(hipGetLastError());
    float ms = std::numeric_limits<float>::quiet_NaN();
    if (!async) {
        DXH_HIP(hipEventRecord(c->ev1, c->stream));
        DXH_HIP(hipStreamSynchronize(c->stream));
        DXH_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    }
    if (tmp && async) DXH_HIP(hipStreamSynchronize(c->stream));  // tmp_guard frees it
    if (res) {
        res->iters_done = done;
        res->converged = converged;
        res->rel_increment = rel;
        res->loop_ms = ms;
        res->launches = launches;
    }
    return DXH_OK;
}

// ---- the distributed sweep ----------------------------------------------------------------------------------------
static int64_t slab_own0(const Ctx* c) { return c->top; }
static int64_t slab_own1(const Ctx* c) { return c->g.n0 - c->bot; }

// Enqueue one halo exchange on the comm stream, to start when `after` has happened: rows are contiguous in every state
// plane, so the row blocks go out of / come into the planes directly.  What I send up is what the upper slab keeps as
// its bottom halo (my first halo_bot owned rows); what I send down is the lower slab's top halo (my last halo_top owned
// rows).
// Up to eight device-to-device block copies in one launch (gather / scatter of halo rows); sizes are multiples of 16 B
// (row pitch is a multiple of 128 B).
struct CopySegs {
    char* dst[8];
    const char* src[8];
    unsigned long long vec[8];  // 16-byte vectors per segment
    unsigned long long total;
    int n;
    void add(void* d, const void* s_, size_t bytes) {
        dst[n] = static_cast<char*>(d);
        src[n] = static_cast<const char*>(s_);
        vec[n] = bytes / 16;
        total += bytes / 16;
        ++n;
    }
};
__global__ void __launch_bounds__(256) k_copy_segs(CopySegs a) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.total; i += stride) {
        unsigned long long j = i;
        int sgm = 0;
        while (sgm < a.n - 1 && j >= a.vec[sgm]) { j -= a.vec[sgm]; ++sgm; }
        reinterpret_cast<float4*>(a.dst[sgm])[j] = reinterpret_cast<const float4*>(a.src[sgm])[j];
    }
}
static void launch_copy_segs(const CopySegs& a, hipStream_t st) {
    if (a.n == 0 || a.total == 0) return;
    // 2048 blocks: measured best of 64 / 256 / 2048 beside the interior chunks of the signalling launch (9.9 / 9.25 / 9.17 ms
    // per step on the N = 8 slab of configs[4]); DXH_COPY_BLOCKS: developer knob
    static const unsigned cap = [] { const char* e = getenv("DXH_COPY_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? (unsigned)v : 2048u; }();
    const unsigned blocks = (unsigned)std::min<unsigned long long>((a.total + 255) / 256, cap);
    hipLaunchKernelGGL(k_copy_segs, dim3(blocks), dim3(256), 0, st, a);
}

// Polls the interface-wave counter (fallback when hipStreamWaitValue32 is not available).  One lane; ends when the
// counter has reached `target` (cyclic compare).  So that the stream can never hang it gives up after `ticks` of the
// 100 MHz wall clock -- and says so in *timeouts: the exchange behind it then packs rows nobody has signalled, so the
// host turns a non-zero count into DXH_EHIP at the next synchronising call (slab_check_fault).
__global__ void k_wait_counter(const unsigned* counter, unsigned target, long long ticks, unsigned* timeouts) {
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > ticks) {
            __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

// The context has drained (the caller synchronised): has a polling wait given up since the last upload?
static int slab_check_fault(Ctx* c) {
    if (!c->signal) return DXH_OK;
    if (!c->slab_fault && c->wait_mode == 2) {
        unsigned n = 0;
        DXH_HIP(hipMemcpy(&n, c->signal + Ctx::kTimeoutSlot, sizeof n, hipMemcpyDeviceToHost));
        if (n) { c->slab_fault = true; c->slab_timeouts += n; }
    }
    if (c->slab_fault) {
        set_error("halo exchange: the wait for the interface rows timed out after %u ms (%lld time-out(s)); the exchange "
                  "went ahead on rows that had not been signalled, the slab state is invalid -- upload again",
                  c->wait_timeout_ms, (long long)c->slab_timeouts);
        return DXH_EHIP;
    }
    return DXH_OK;
}

// fold the event triples of finished exchanges into the running sums; `all`: the context has drained
static void slab_fold_events(Ctx* c, bool all) {
    for (auto& e : c->xev) {
        if (!e.busy) continue;
        if (!all && hipEventQuery(e.c) != hipSuccess) continue;
        float w = 0.f, x = 0.f;
        if (hipEventElapsedTime(&w, e.a, e.b) == hipSuccess && hipEventElapsedTime(&x, e.b, e.c) == hipSuccess) {
            c->slab_wait_ms += w;
            c->slab_xchg_ms += x;
            ++c->slab_timed;
        }
        e.busy = false;
    }
    (void)hipGetLastError();
}
static Ctx::XEv* slab_take_events(Ctx* c) {
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < Ctx::kXEv; ++i) {
            Ctx::XEv& e = c->xev[(c->xev_next + i) % Ctx::kXEv];
            if (e.busy) continue;
            if (!e.a && (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess ||
                         hipEventCreate(&e.c) != hipSuccess)) {
                (void)hipGetLastError();
                return nullptr;
            }
            c->xev_next = (c->xev_next + i + 1) % Ctx::kXEv;
            return &e;
        }
        slab_fold_events(c, false);
    }
    return nullptr;  // every triple still in flight: this exchange goes untimed (exchanges_timed < exchanges)
}

static int start_exchange(Ctx* c, bool after_signal) {
    RoctxRange range(after_signal ? "halo exchange: enqueue (starts on the interface rows' signal)" : "halo exchange: enqueue (stream order)");
    Comm* cm = c->comm;
    if (c->from_db) { set_error("halo exchange before the first iteration of a warm start is not supported"); return DXH_ESTATE; }
    // Overlapped exchange: it -- and every launch after it -- goes to the context's OTHER stream (a different hardware
    // queue: one of the two streams is created with high priority), so that it can start while the signalling launch is
    // still running, and what follows an exchange follows it in stream order (an event between two streams cost ~80 us
    // from the end of the scatter to the start of the next launch).  The only cross-stream dependence left is "the
    // previous stream has drained", which as a rule has happened long before it is looked at.  Serialised exchange: same
    // stream, plain stream order, no event at all.
    hipStream_t from = c->stream, to = after_signal ? c->alt : c->stream;
    // three events per exchange (dxh_tvd_slab_stats): a -> b the exchange stream idles until the interface waves have
    // signalled, b -> c gather + send / receive + scatter
    Ctx::XEv* xe = slab_take_events(c);
    if (xe) DXH_HIP(hipEventRecord(xe->a, to));
    if (after_signal) {
        // start as soon as every interface wave of the launch just enqueued has released its rows
        const unsigned target = c->signal_target + (c->test_lost_signal ? 1u : 0u);
        if (c->wait_mode == 1)
            DXH_HIP(hipStreamWaitValue32(to, c->signal, target, hipStreamWaitValueGte, 0xffffffffu));
        else
            hipLaunchKernelGGL(k_wait_counter, dim3(1), dim3(1), 0, to, c->signal, target,
                               (long long)c->wait_timeout_ms * 100000LL, c->signal + Ctx::kTimeoutSlot);
    }
    if (xe) DXH_HIP(hipEventRecord(xe->b, to));
    // The rows of the 1 + dim state planes are gathered into one staging block per direction, so that an exchange is four
    // messages (RCCL packs up to eight point-to-point operations into one kernel: twelve row blocks took two kernels, the
    // second of which started only when the first had ended), and scattered into the halo rows on arrival.
    const size_t row_bytes = (size_t)c->g.pitch * c->esz;
    void* fields[4] = {c->u[c->cur], c->q[0][c->cur], c->q[1][c->cur], c->q[2][c->cur]};
    const int nf = 1 + c->desc.dim;
    const int64_t o0 = slab_own0(c), o1 = slab_own1(c);
    auto rows = [&](int f, int64_t r0) { return static_cast<char*>(fields[f]) + (size_t)r0 * row_bytes; };
    const size_t up_b = (size_t)c->halo_bot * row_bytes, dn_b = (size_t)c->halo_top * row_bytes;  // per field
    CopySegs pack{}, unpack{};
    if (c->top > 0)
        for (int f = 0; f < nf; ++f) {
            pack.add(c->xbuf[0] + f * up_b, rows(f, o0), up_b);                       // my first rows -> upper's bottom halo
            unpack.add(rows(f, 0), c->xbuf[3] + f * dn_b, (size_t)c->top * row_bytes);  // upper's last rows -> my top halo
        }
    if (c->bot > 0)
        for (int f = 0; f < nf; ++f) {
            pack.add(c->xbuf[1] + f * dn_b, rows(f, o1 - c->halo_top), dn_b);          // my last rows -> lower's top halo
            unpack.add(rows(f, o1), c->xbuf[2] + f * up_b, (size_t)c->bot * row_bytes);  // lower's first rows -> my bottom halo
        }
    launch_copy_segs(pack, to);
    // A single rank whose slab has halos on both sides is its own neighbour (periodic in axis 0): sends first, then the
    // receives in the order that pairs them with those sends (what goes up fills a bottom halo).
    const int up = (cm->rank - 1 + cm->world) % cm->world, down = (cm->rank + 1) % cm->world;
    P2POp ops[4];
    int n = 0;
    if (c->top > 0) ops[n++] = P2POp{up, c->xbuf[0], nf * up_b, 1};
    if (c->bot > 0) ops[n++] = P2POp{down, c->xbuf[1], nf * dn_b, 1};
    if (c->bot > 0) ops[n++] = P2POp{down, c->xbuf[2], nf * up_b, 0};
    if (c->top > 0) ops[n++] = P2POp{up, c->xbuf[3], nf * dn_b, 0};
    int rc = comm_exchange(cm, ops, n, to);
    if (rc) return rc;
    launch_copy_segs(unpack, to);
    DXH_HIP(hipGetLastError());
    if (xe) {
        DXH_HIP(hipEventRecord(xe->c, to));
        xe->busy = true;
    }
    ++c->slab_exchanges;
    if (after_signal) {
        // the interior chunks of the signalling launch must be done as well before anything overwrites their input
        DXH_HIP(hipEventRecord(c->eL, from));
        DXH_HIP(hipStreamWaitEvent(to, c->eL, 0));
        c->alt = from;
        c->stream = to;
    }
    c->halo_valid = c->halo_top;
    return DXH_OK;
}

// exchange in stream order behind everything in flight (no overlap)
static int exchange_now(Ctx* c) { return start_exchange(c, false); }


template <typename T>
static int run_slab_t(Ctx* c, int n_iter, int m, double eps, uint32_t flags, dxh_tvd_result* res) {
    RoctxRange range("dxh_tvd_run_slab: %lld TV iterations on a row slab (halos valid for %lld)", n_iter, (long long)c->halo_valid);
    Comm* cm = c->comm;
    const bool dist = cm && (c->top > 0 || c->bot > 0);
    const bool iso = flags & DXH_FLAG_ISOTROPIC;
    const bool use_eps = eps >= 0.0 && !std::isnan(eps);
    const bool dpp = !(flags & DXH_FLAG_NO_DPP);
    const bool fused_ok = fused_supported(c) && m == 1 && !(flags & DXH_FLAG_NO_FUSED) && !(flags & DXH_FLAG_EXPLICIT) &&
                          !c->solver && !c->explicit_db && !c->from_db;
    const bool overlap = dist && fused_ok && !(flags & DXH_FLAG_NO_OVERLAP);
    const bool async = (flags & DXH_FLAG_ASYNC) && !use_eps;
    if (c->np2) {   // the reference's float32 dtype flow on slabs: one iteration per launch (k_fused_np2), same exchanges
        constexpr bool is_double = std::is_same<T, double>::value;
        DXH_REQUIRE(fused_ok && m == 1 && !iso && is_double && (!c->het || (c->het_cdiag && c->cp_kq)),
                    "numpy2 promotion on slabs: dim = 2, one Jacobi sweep, anisotropic, scalar omega and ell");
    }
    if (dist) DXH_REQUIRE(c->halo_top >= m, "halos of %lld rows cannot carry an iteration of %d Jacobi sweeps",
                          (long long)c->halo_top, m);
    if (dist && c->from_db) { set_error("a warm start with explicit split variables is not supported on row slabs"); return DXH_ESTATE; }
    int rc;
    double nrm_img = 0.0;
    if (use_eps) {
        if (c->global_sq_img < 0.0) {
            double v = 0.0;
            if (dist) {
                DXH_REQUIRE(!std::isnan(c->omega_s) && c->omega_s != 0.0, "the global eps test needs a scalar, non-zero omega");
                rc = sq_diff<T>(c, c->of, nullptr, slab_own0(c), slab_own1(c) - slab_own0(c), &v, 1);
                if (rc) return rc;
                v /= c->omega_s * c->omega_s;
                if ((rc = comm_allreduce(cm, &v, 1, 0))) return rc;
            } else {
                v = c->sq_img;
            }
            c->global_sq_img = v;
        }
        nrm_img = std::sqrt(c->global_sq_img);
    }
    if (fused_ok) { c->last_iso_prev = c->last_iso; c->last_iso = iso; }
    if (!async) {
        DXH_HIP(hipEventRecord(c->ev0, c->stream));
    }
    int64_t launches = 0;
    int done = 0, converged = 0;
    double rel = std::numeric_limits<double>::quiet_NaN();
    while (done < n_iter) {
        if (dist && c->halo_valid < m) {
            if ((rc = exchange_now(c))) return rc;  // halos used up by an earlier call: nothing to hide this one behind
        }
        const int left = n_iter - done;
        const int cap = dist ? (int)std::min<int64_t>(left, c->halo_valid / m) : left;
        int K;
        if (fused_ok) {
            K = use_eps ? 1 : std::min(pick_k(c, left), cap);
            if (K >= 2 && !(dpp && multi_supported(c))) K = 1;
            if (K >= 2 && c->het && !c->np2 && K != kHetFuseK) K = 1;
            if (K >= 2 && c->np2 && K != kNp2K && K != kNp2GrayK) {   // numpy2 flow: the K its pipelines exist for
                const bool small_k = !c->het && (c->g.n2 == 1 || c->g.n2 == 3);   // two / three iterations per pass
                K = K > kNp2K ? kNp2K : (small_k ? K : 1);
            }
            const int in = c->cur, out = c->cur ^ 1;
            const bool xchg_after = dist && (c->halo_valid - (int64_t)K * m < m) &&
                                    (done + K < n_iter || (flags & DXH_FLAG_CONTINUE)) && !use_eps;
            // the launch in front of an exchange runs its interface chunks first and counts them (make_layout)
            c->signalling = overlap && xchg_after;
            if (c->signalling && c->signal_target > 0x40000000u) {
                // far from wrapping, but cheap to rule out: drain and restart the counter
                DXH_HIP(hipStreamSynchronize(c->stream));
                DXH_HIP(hipStreamSynchronize(c->alt));
                DXH_HIP(hipMemset(c->signal, 0, 256));
                c->signal_target = 0;
            }
            if (dist && !(c->tune & 131072)) {
                // After these K iterations, j sweeps after the last exchange, a non-boundary top has lost its first j rows
                // and a non-boundary bottom the last j rows of u and q1 and the last j + 1 rows of q0 (file header of
                // darsia_amd/parallel.py; the one row of q0 more is why the bottom halo has k + 1 rows): the launch need
                // not produce rows outside [j, n0 - j) -- a launch stores u, q1 of its rows and q0 of the same rows, the
                // last of which is the invalid one.  What is skipped are halo rows, all rewritten by the next exchange.
                // (DXH_FUSED_TUNE bit 131072: whole slab every time, A/B)
                const int64_t j = c->halo_top - c->halo_valid + (int64_t)K * m;
                c->row_lo = c->top > 0 ? std::min(j, c->top) : 0;
                c->row_hi = c->bot > 0 ? c->g.n0 - std::min(j, c->bot) : c->g.n0;
            }
            launch_any<T>(c, in, out, iso, dpp, K);
            c->row_lo = c->row_hi = 0;
            if ((rc = launched_or_fail(c, K))) return rc;
            if (c->signalling) {
                c->signal_target += c->signal_pending;
                ++c->split_launches;
            }
            DXH_HIP(hipGetLastError());
            ++launches;
            c->cur = out;
            c->halo_valid -= (int64_t)K * m;
            if (xchg_after) {
                rc = c->signalling ? start_exchange(c, true) : exchange_now(c);
                c->signalling = false;
                if (rc) return rc;
            }
        } else {
            // three-kernel / solver / multi-sweep paths: whole-slab launches, exchange in between
            K = use_eps ? 1 : cap;
            dxh_tvd_result r{};
            rc = run_t<T>(c, K, m, -1.0, -1.0, 0.0, (flags & (DXH_FLAG_ISOTROPIC | DXH_FLAG_NO_FUSED | DXH_FLAG_NO_DPP | DXH_FLAG_EXPLICIT)) | DXH_FLAG_ASYNC, &r);
            if (rc) return rc;
            launches += r.launches;
            c->halo_valid -= (int64_t)K * m;
        }
        done += K;
        if (use_eps) {
            double inc = 0.0;
            rc = sq_diff<T>(c, c->u[c->cur], c->u[c->cur ^ 1], slab_own0(c), slab_own1(c) - slab_own0(c), &inc, 0);
            if (rc) return rc;
            if (dist && (rc = comm_allreduce(cm, &inc, 1, 0))) return rc;
            rel = std::sqrt(inc) / nrm_img;
            if (rel < eps) { converged = 1; break; }
        }
    }
    float ms = std::numeric_limits<float>::quiet_NaN();
    c->slab_launches += launches;
    if (!async) {
        DXH_HIP(hipEventRecord(c->ev1, c->stream));
        DXH_HIP(hipStreamSynchronize(c->stream));
        DXH_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    }
    if (res) {
        res->iters_done = done;
        res->converged = converged;
        res->rel_increment = rel;
        res->loop_ms = ms;
        res->launches = launches;
    }
    if (!async && (rc = slab_check_fault(c))) return rc;
    return DXH_OK;
}

template <typename T>
static int set_coeffs_t(Ctx* c, double mu, const void* mu_arr, double omega, const void* omega_arr, int has_ell,
                        double ell, const void* ell_arr, double h, bool want_kq) {
    const int dim = c->desc.dim;
    const double h2 = h * h;
    c->het = mu_arr || omega_arr || ell_arr;
    c->ell_is_arr = c->om_is_arr = false;
    c->np2_adapted = false;
    c->mu_s = mu;
    c->omega_s = omega;
    for (void** p : {&c->in_mu, &c->in_om, &c->in_ell}) {
        if (*p) { DXH_HIP(hipFree(*p)); *p = nullptr; }
    }
    c->in_valid = false;
    c->het_cdiag = false;
    c->c_div3 = 0;
    if (!c->het) {
        // all-scalar: fold in double like Python floats, round once where the value meets an array
        const double ell_d = has_ell ? ell : 2.0 * mu;
        c->ell_s = ell_d;
        c->c_ell = ell_d;
        c->c_diag = omega + ell_d * 2 * dim / h2;
        c->c_dsc = c->c_diag / (ell_d / h2);
        c->c_kq = mu / ell_d;
        c->c_div3 = 0;
        if constexpr (std::is_same<T, float>::value) {
            const float fd = (float)c->c_diag, fs = (float)c->c_dsc;
            auto ok = [](float d) { return std::isfinite(d) && std::fabs(d) >= 0x1p-20f && std::fabs(d) <= 0x1p+20f; };
            if (ok(fd) && ok(fs) && !(c->tune & 128)) c->c_div3 = div3_verified(c, fd, fs);
        }
        for (void** p : {&c->cp_ell, &c->cp_diag, &c->cp_dsc, &c->cp_kq}) {
            if (*p) { DXH_HIP(hipStreamSynchronize(c->stream)); pool_free(*p); *p = nullptr; }
        }
        return DXH_OK;
    }
    // heterogeneous: build the four planes on the device with NumPy's evaluation order
    // (uploaded straight into the context's slots, released above / in dxh_tvd_destroy: nothing leaks on an error return)
    void *&d_mu = c->in_mu, *&d_om = c->in_om, *&d_ell = c->in_ell;
    c->in_valid = false;
    auto up = [&](const void* src, void** dst) -> int {
        if (!src) return DXH_OK;
        DXH_HIP(hipMalloc(dst, c->plane_bytes));
        return h2d_plane(c, *dst, src);
    };
    int rc;
    if ((rc = up(mu_arr, &d_mu)) || (rc = up(omega_arr, &d_om)) || (rc = up(ell_arr, &d_ell))) return rc;
    if ((rc = alloc_plane(c, &c->cp_ell)) || (rc = alloc_plane(c, &c->cp_diag)) || (rc = alloc_plane(c, &c->cp_dsc)) ||
        (rc = alloc_plane(c, &c->cp_kq)))
        return rc;
    CoefIn<T> in;
    in.mu = (const T*)d_mu;
    in.omega = (const T*)d_om;
    in.ell = (const T*)d_ell;
    in.mu_s = (T)mu;
    in.omega_s = (T)omega;
    const double ell_d = has_ell ? ell : 2.0 * mu;  // only meaningful when ell is scalar
    in.ell_s = (T)ell_d;
    in.two_dim_over_h2_ell_s = (T)(ell_d * 2 * dim / h2);
    in.ell_over_h2_s = (T)(ell_d / h2);
    {
        const double diag_d = omega + ell_d * 2 * dim / h2;
        in.diag_s = (T)diag_d;
        in.dsc_s = (T)(diag_d / (ell_d / h2));
    }
    in.kq_s = (T)(mu / ell_d);
    in.h2 = (T)h2;
    in.two_dim = (T)(2 * dim);
    in.dim = dim;
    in.has_ell = has_ell;
    hipLaunchKernelGGL((k_coef_planes<T>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, in, (T*)c->cp_ell,
                       (T*)c->cp_diag, (T*)c->cp_dsc, want_kq ? (T*)c->cp_kq : nullptr);
    DXH_HIP(hipGetLastError());
    DXH_HIP(hipStreamSynchronize(c->stream));
    // the given arrays stay on the device: the fused heterogeneous sweep derives its coefficients from them
    c->ell_s = ell_d;
    c->in_has_ell = has_ell;
    c->in_h = h;
    c->in_valid = want_kq;  // (mass, diffusion) solves without mu never use the fused sweep
    // only mu varies (the TVD operator always passes a scalar ell): the Jacobi divisors are the Python floats of
    // derive_coefs' diag_s / dsc_s, and the fused sweep divides by them through reciprocals like the scalar pipeline
    const bool ell_is_arr = ell_arr || (!has_ell && mu_arr);
    c->ell_is_arr = ell_is_arr;
    c->om_is_arr = omega_arr != nullptr;
    if (!ell_is_arr && !omega_arr) {
        c->c_diag = omega + ell_d * 2 * dim / h2;
        c->c_dsc = c->c_diag / (ell_d / h2);
        c->c_ell = ell_d;   // the scalar ell of the per-element-threshold pipeline (k_fused_multi<..., KQV>)
        c->het_cdiag = true;
        if constexpr (std::is_same<T, float>::value) {
            const float fd = (float)c->c_diag, fs = (float)c->c_dsc;
            auto okf = [](float d) { return std::isfinite(d) && std::fabs(d) >= 0x1p-20f && std::fabs(d) <= 0x1p+20f; };
            if (okf(fd) && okf(fs) && !(c->tune & 128)) c->c_div3 = div3_verified(c, fd, fs);
        }
    }
    return DXH_OK;
}

}  // namespace dxh

using namespace dxh;

// dense array conversion between the two float types (round to nearest, as ndarray.astype)
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_convert_dense(const TS* __restrict__ src, TD* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (TD)src[i];
}

// dense array times a scalar rounded to the array's type first (what NumPy does for ``python_float * array``)
template <typename T>
__global__ void __launch_bounds__(256) k_scale_dense(const T* __restrict__ src, T s, T* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = s * src[i];
}

// dense (n0, L) array of one float type <-> pitched plane of the other (round to nearest, as ndarray.astype)
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_dense_to_plane(Geom g, const TS* __restrict__ src, TD* __restrict__ plane) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    plane[e.off] = (TD)src[e.row * g.L + e.x];
}
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_plane_to_dense(Geom g, const TS* __restrict__ plane, TD* __restrict__ dst) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    dst[e.row * g.L + e.x] = (TD)plane[e.off];
}

namespace {
// k_signal<T, CH, S, B> for the types the arguments name
template <typename T, int CH, typename S, typename B, typename O = T>
void launch_signal(Ctx* c, int diff_option, int reduction, const ModelOps& pre) {
    hipLaunchKernelGGL((k_signal<T, CH, S, B, O>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const S*)c->probe_buf,
                       (const B*)c->base_img, (const T*)c->clean_filter, diff_option, reduction, pre, (O*)c->u[0]);
}

template <typename T, int CH, typename S, typename O = T>
int signal_by_base(Ctx* c, int diff_option, int reduction, const ModelOps& pre) {
    using F = typename SigLoad<S>::F;
    // without a baseline B only sizes the difference type: take the probe's own float type
    const int bdt = c->base_img ? c->base_dtype : (sizeof(F) == 4 ? DXH_F32 : DXH_F64);
    if (bdt == DXH_F32) {
        if constexpr (sizeof(T) >= sizeof(F)) { launch_signal<T, CH, S, float, O>(c, diff_option, reduction, pre); return DXH_OK; }
    } else {
        if constexpr (sizeof(T) == 8) { launch_signal<T, CH, S, double, O>(c, diff_option, reduction, pre); return DXH_OK; }
    }
    set_error("the context's dtype is narrower than probe - baseline (NumPy would promote to float64)");
    return DXH_EINVAL;
}

template <typename T, int CH, typename O = T>
int signal_by_probe(Ctx* c, int probe_dtype, int diff_option, int reduction, const ModelOps& pre) {
    switch (probe_dtype) {
        case DXH_F32: return signal_by_base<T, CH, float, O>(c, diff_option, reduction, pre);
        case DXH_F64: return signal_by_base<T, CH, double, O>(c, diff_option, reduction, pre);
        case DXH_U8: return signal_by_base<T, CH, unsigned char, O>(c, diff_option, reduction, pre);
        default: return signal_by_base<T, CH, unsigned short, O>(c, diff_option, reduction, pre);
    }
}
}  // namespace

extern "C" {

const char* dxh_last_error(void) { return g_err.c_str(); }
int dxh_version(void) { return 100; }

int dxh_device_count(int* count) {
    if (!count) return DXH_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return DXH_OK;
}

int dxh_tvd_create(dxh_tvd_t* out, const dxh_tvd_desc* d) {
    DXH_REQUIRE(out && d, "null argument");
    DXH_REQUIRE(d->struct_size == (int32_t)sizeof(dxh_tvd_desc), "dxh_tvd_desc size mismatch (%d vs %zu)", d->struct_size,
                sizeof(dxh_tvd_desc));
    DXH_REQUIRE(d->dtype == DXH_F32 || d->dtype == DXH_F64, "dtype must be float32 or float64");
    DXH_REQUIRE(d->ndim == 2 || d->ndim == 3, "ndim must be 2 or 3 (got %d)", d->ndim);
    DXH_REQUIRE(d->dim == 2 || d->dim == 3, "dim must be 2 or 3 (got %d)", d->dim);
    DXH_REQUIRE(d->dim <= d->ndim, "dim (%d) exceeds ndim (%d)", d->dim, d->ndim);
    DXH_REQUIRE(d->n0 >= 1 && d->n1 >= 1 && d->n2 >= 1, "empty image");
    DXH_REQUIRE(d->ndim == 3 || d->n2 == 1, "n2 must be 1 for a 2-D array");
    DXH_REQUIRE(d->nsum == DXH_NSUM_ALL_AXES || d->nsum == DXH_NSUM_SPATIAL, "bad nsum");
    int ndev = 0;
    DXH_HIP(hipGetDeviceCount(&ndev));
    DXH_REQUIRE(d->device >= 0 && d->device < ndev, "device %d not available (%d devices)", d->device, ndev);
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) { set_error("out of host memory"); return DXH_ENOMEM; }
    c->desc = *d;
    c->esz = d->dtype == DXH_F32 ? 4 : 8;
    c->g.n0 = d->n0;
    c->g.n1 = d->n1;
    c->g.n2 = d->n2;
    c->g.L = d->n1 * d->n2;
    c->g.pitch = round_up(c->g.L, 32);
    c->g.ndim = d->ndim;
    c->g.dim = d->dim;
    c->g.nsum_axes = d->nsum == DXH_NSUM_ALL_AXES ? d->ndim : d->dim;
    c->plane_bytes = (size_t)c->g.pitch * (size_t)c->g.n0 * c->esz;
    auto fail = [&](int rc) { dxh_tvd_destroy(reinterpret_cast<dxh_tvd_t>(c)); return rc; };
    if (hipSetDevice(d->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", d->device); return fail(DXH_EHIP); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (const char* e = getenv("DXH_FUSED_TUNE")) c->tune = atoi(e);
    if (const char* e = getenv("DXH_NCHUNKS")) c->force_chunks = atoi(e);
    if (const char* e = getenv("DXH_FUSE_K")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4) c->fuse_k = v;
    }
    if (const char* e = getenv("DXH_WAVES_PER_CU")) {  // developer knob
        const int v = atoi(e);
        if (v >= 1 && v <= 32) c->waves_per_cu = v;
    }
    if (d->stream) {
        c->stream = (hipStream_t)d->stream;
    } else {
        // the thread's cached stream: contexts of one host thread are used one after the other (header: "one ctx per
        // host thread, not re-entrant"), and creating a stream costs 10-15 ms
        if (cached_stream(c->desc.device, &c->stream) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(DXH_EHIP); }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { set_error("hipEventCreate failed"); return fail(DXH_EHIP); }
    int rc;
    if ((rc = alloc_plane(c, &c->of))) return fail(rc);
    for (int k = 0; k < 2; ++k) {
        if ((rc = alloc_plane(c, &c->u[k]))) return fail(rc);
        for (int j = 0; j < d->dim; ++j)
            if ((rc = alloc_plane(c, &c->q[j][k]))) return fail(rc);
    }
    if (hipMalloc((void**)&c->partial, sizeof(double) * c->npartial) != hipSuccess ||
        hipHostMalloc((void**)&c->red_out, sizeof(double) * 16, hipHostMallocDefault) != hipSuccess) {
        set_error("hipMalloc failed");
        return fail(DXH_ENOMEM);
    }
    *out = reinterpret_cast<dxh_tvd_t>(c);
    return DXH_OK;
}

int dxh_tvd_destroy(dxh_tvd_t h) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    if (!c) return DXH_OK;
    (void)hipSetDevice(c->desc.device);
    if (c->comm && c->comm->stream) (void)hipStreamSynchronize(c->comm->stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->alt) (void)hipStreamSynchronize(c->alt);
    if (c->alt_owned) (void)hipStreamDestroy(c->alt_owned);
    if (c->eL) (void)hipEventDestroy(c->eL);
    for (auto& e : c->xev)
        for (hipEvent_t ev : {e.a, e.b, e.c})
            if (ev) (void)hipEventDestroy(ev);
    if (c->signal) (void)hipFree(c->signal);
    for (char* b : c->xbuf)
        if (b) (void)hipFree(b);
    comm_release(c->comm);
    delete c->solver;
    auto fr = [](void* p) { pool_free(p); };   // planes go back to the cache, anything else to the runtime
    fr(c->of); fr(c->u[0]); fr(c->u[1]); fr(c->rs); fr(c->of32);
    for (int j = 0; j < 3; ++j) { fr(c->q[j][0]); fr(c->q[j][1]); fr(c->dpl[j]); fr(c->bpl[j]); }
    fr(c->in_mu); fr(c->in_om); fr(c->in_ell);
    fr(c->cp_ell); fr(c->cp_diag); fr(c->cp_dsc); fr(c->cp_kq); fr(c->om_plane); fr(c->mu_plane); fr(c->base_img); fr(c->clean_filter); fr(c->probe_buf); fr(c->stage); fr(c->partial); fr(c->inc_partial);
    if (c->red_out) (void)hipHostFree(c->red_out);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
    return DXH_OK;
}

int dxh_tvd_set_solver(dxh_tvd_t h, int kind, int depth, int smoother_iterations, int maxiter, double tol) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    DXH_REQUIRE(kind == DXH_SOLVER_JACOBI || kind == DXH_SOLVER_MG || kind == DXH_SOLVER_CG, "unknown solver kind %d", kind);
    int rc = set_device(c);
    if (rc) return rc;
    DXH_HIP(hipStreamSynchronize(c->stream));
    delete c->solver;
    c->solver = nullptr;
    c->solver_kind = DXH_SOLVER_JACOBI;
    if (kind == DXH_SOLVER_JACOBI) return DXH_OK;
    Geom g = c->g;
    g.nsum_axes = g.ndim;  // the smoother sums over every array axis (jacobi.py:41)
    if (kind == DXH_SOLVER_MG)
        c->solver = make_mg_solver(c->desc.dtype, g, depth, smoother_iterations, maxiter, tol, false, c->stream, &rc);
    else
        c->solver = make_cg_solver(c->desc.dtype, g, maxiter, tol, -1.0, c->stream, &rc);
    if (!c->solver) return rc ? rc : DXH_EHIP;
    c->solver_kind = kind;
    return DXH_OK;
}

int dxh_tvd_set_coeffs(dxh_tvd_t h, double mu, const void* mu_arr, double omega, const void* omega_arr, int has_ell,
                       double ell, const void* ell_arr) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    int rc = set_device(c);
    if (rc) return rc;
    if (!ell_arr && has_ell) DXH_REQUIRE(ell != 0.0, "ell must be non-zero");
    rc = c->desc.dtype == DXH_F32 ? set_coeffs_t<float>(c, mu, mu_arr, omega, omega_arr, has_ell, ell, ell_arr, 1.0, true)
                                  : set_coeffs_t<double>(c, mu, mu_arr, omega, omega_arr, has_ell, ell, ell_arr, 1.0, true);
    if (rc) return rc;
    if (c->np2 && c->np2_mu_is_kq && mu_arr && c->cp_kq) {
        // the caller formed mu / ell in the array's own float32 (what NumPy does for a float32 mu) and widened it
        DXH_HIP(hipMemcpyAsync(c->cp_kq, c->in_mu, c->plane_bytes, hipMemcpyDeviceToDevice, c->stream));
        DXH_HIP(hipStreamSynchronize(c->stream));
    }
    // omega array: kept as a plane, upload multiplies it into the image
    if (omega_arr) {
        if ((rc = alloc_plane(c, &c->om_plane))) return rc;
        if ((rc = h2d_plane(c, c->om_plane, omega_arr))) return rc;
        DXH_HIP(hipStreamSynchronize(c->stream));
    } else if (c->om_plane) {
        DXH_HIP(hipStreamSynchronize(c->stream));
        pool_free(c->om_plane);
        c->om_plane = nullptr;
    }
    if (mu_arr) {
        if ((rc = alloc_plane(c, &c->mu_plane))) return rc;
        if ((rc = h2d_plane(c, c->mu_plane, mu_arr))) return rc;
        DXH_HIP(hipStreamSynchronize(c->stream));
    } else if (c->mu_plane) {
        DXH_HIP(hipStreamSynchronize(c->stream));
        pool_free(c->mu_plane);
        c->mu_plane = nullptr;
    }
    c->explicit_db = false;
    c->have_coefs = true;
    c->uploaded = false;  // omega*img must be rebuilt
    c->omega_s = omega_arr ? std::numeric_limits<double>::quiet_NaN() : omega;
    return DXH_OK;
}

// float32 mu array / float32(ell), widened: what NumPy makes of  float32_array / python_float  (the quotient is float32)
__global__ void __launch_bounds__(256) k_convert_div_dense(const float* __restrict__ src, float ell, double* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)(src[i] / ell);
}

int dxh_tvd_set_coeffs_typed(dxh_tvd_t h, double mu, const void* mu_arr, int mu_dtype, double omega, const void* omega_arr,
                             int omega_dtype, int has_ell, double ell, const void* ell_arr, int ell_dtype) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t n = (size_t)c->g.n0 * (size_t)c->g.L;
    DevTmp raw[3], conv[3];
    const void* arrs[3] = {mu_arr, omega_arr, ell_arr};
    const int dts[3] = {mu_dtype, omega_dtype, ell_dtype};
    // numpy2 context, float32 mu array (host or device) next to scalar omega and ell: the reference's threshold mu / ell is a
    // float32 quotient (split_bregman_tvd.py:144 with a float32 array and a Python float) -- formed here, on the device, and
    // handed on as the threshold itself (DXH_PROMOTION_MU_IS_THRESHOLD's meaning, set for this coefficient set only)
    const bool mu_div = c->np2 && mu_arr && mu_dtype == DXH_F32 && !omega_arr && !ell_arr && has_ell &&
                        !(c->np2_mu_is_kq && !c->np2_mu_auto);   // (not when the caller said the array already is the threshold)
    if (!mu_div && c->np2_mu_auto) { c->np2_mu_is_kq = false; c->np2_mu_auto = false; }
    for (int k = 0; k < 3; ++k) {
        if (k == 0 && mu_div) {
            const void* src = arrs[0];
            if (!is_device_ptr(src)) {
                DXH_HIP(hipMalloc(&raw[0].p, n * 4));
                DXH_HIP(hipMemcpyAsync(raw[0].p, src, n * 4, hipMemcpyDefault, c->stream));
                src = raw[0].p;
            }
            DXH_HIP(hipMalloc(&conv[0].p, n * 8));
            hipLaunchKernelGGL(k_convert_div_dense, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float*)src, (float)ell,
                               (double*)conv[0].p, n);
            DXH_HIP(hipGetLastError());
            arrs[0] = conv[0].p;
            c->np2_mu_is_kq = true;
            c->np2_mu_auto = true;
            continue;
        }
        if (!arrs[k] || dts[k] == c->desc.dtype) continue;
        DXH_REQUIRE(dts[k] == DXH_F32 || dts[k] == DXH_F64, "coefficient arrays must be float32 or float64");
        const size_t ssz = dts[k] == DXH_F32 ? 4 : 8;
        const void* src = arrs[k];
        if (!is_device_ptr(src)) {
            DXH_HIP(hipMalloc(&raw[k].p, n * ssz));
            DXH_HIP(hipMemcpyAsync(raw[k].p, src, n * ssz, hipMemcpyDefault, c->stream));
            src = raw[k].p;
        }
        DXH_HIP(hipMalloc(&conv[k].p, n * c->esz));
        const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
        if (dts[k] == DXH_F64) hipLaunchKernelGGL((k_convert_dense<double, float>), grid, blk, 0, c->stream, (const double*)src, (float*)conv[k].p, n);
        else hipLaunchKernelGGL((k_convert_dense<float, double>), grid, blk, 0, c->stream, (const float*)src, (double*)conv[k].p, n);
        DXH_HIP(hipGetLastError());
        arrs[k] = conv[k].p;
    }
    rc = dxh_tvd_set_coeffs(h, mu, arrs[0], omega, arrs[1], has_ell, ell, arrs[2]);
    (void)hipStreamSynchronize(c->stream);   // the temporaries go out of scope
    return rc;
}

// common tail of the two upload paths: the image is in u[0]; builds ||img||^2, of = omega*img, zeroes the state
// lean: enqueue only -- no ||img||^2 (the caller runs without a stopping criterion) and no stream synchronisation
static int finish_upload(Ctx* c, const void* u0, const void* d0, const void* b0, bool lean = false) {
    int rc;
    const bool f32 = c->desc.dtype == DXH_F32;
    const int dim = c->desc.dim;
    c->cur = 0;
    if (lean) {
        c->sq_img = std::numeric_limits<double>::quiet_NaN();
    } else {
        rc = f32 ? sq_diff<float>(c, c->u[0], nullptr, 0, c->g.n0, &c->sq_img, 1)
                 : sq_diff<double>(c, c->u[0], nullptr, 0, c->g.n0, &c->sq_img, 1);
        if (rc) return rc;
    }
    const bool om_arr = std::isnan(c->omega_s);
    if (f32)
        hipLaunchKernelGGL((k_scale<float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->u[0],
                           (float)c->omega_s, om_arr ? (const float*)c->om_plane : nullptr, (float*)c->of);
    else if (c->np2 && !om_arr) {  // np.multiply(omega, img) on the float32 image: a float32 product, kept in the float64 plane
        if (!c->of32) {             // ... and as a float32 plane for the K-stage pipelines
            DXH_HIP(pool_malloc(&c->of32, c->plane_bytes / 2));
            DXH_HIP(hipMemsetAsync(c->of32, 0, c->plane_bytes / 2, c->stream));
        }
        hipLaunchKernelGGL(k_scale_np2, elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[0],
                           (float)c->omega_s, (double*)c->of, (float*)c->of32);
        c->of32_valid = true;
    }
    else
        hipLaunchKernelGGL((k_scale<double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[0],
                           c->omega_s, om_arr ? (const double*)c->om_plane : nullptr, (double*)c->of);
    DXH_HIP(hipGetLastError());
    if (!(c->np2 && !om_arr)) c->of32_valid = false;
    c->np2_first = true;
    c->np2_q32 = false;
    if (c->np2) DXH_REQUIRE(!u0 && !d0, "numpy2 promotion: no warm start");   // (a float64 omega array: a float64 product, the branch above)
    if (u0 && (rc = h2d_plane(c, c->u[0], u0))) return rc;
    if (!c->pristine) {   // a fresh context's planes were cleared when they were allocated
        for (int j = 0; j < dim; ++j) {
            DXH_HIP(hipMemsetAsync(c->q[j][0], 0, c->plane_bytes, c->stream));
            DXH_HIP(hipMemsetAsync(c->q[j][1], 0, c->plane_bytes, c->stream));
        }
        DXH_HIP(hipMemsetAsync(c->u[1], 0, c->plane_bytes, c->stream));
    }
    c->pristine = false;
    c->from_db = false;
    c->explicit_db = false;
    if (c->signal && c->slab_fault) {  // a fresh state: the time-out of an earlier sequence no longer applies
        DXH_HIP(hipMemsetAsync(c->signal + Ctx::kTimeoutSlot, 0, sizeof(unsigned), c->stream));
        c->slab_fault = false;
    }
    if (d0) {
        // explicit split variables: keep them as planes for the first iteration
        const size_t dense = (size_t)c->g.n0 * c->g.L * dim * c->esz;
        if (!c->stage) DXH_HIP(hipMalloc(&c->stage, dense));
        for (int pass = 0; pass < 2; ++pass) {
            void** pl = pass == 0 ? c->dpl : c->bpl;
            for (int j = 0; j < dim; ++j)
                if ((rc = alloc_plane(c, &pl[j]))) return rc;
            DXH_HIP(hipMemcpyAsync(c->stage, pass == 0 ? d0 : b0, dense, hipMemcpyDefault, c->stream));
            if (f32) {
                if (dim == 2) hipLaunchKernelGGL((k_deinterleave<float, 2>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->stage, (float*)pl[0], (float*)pl[1], (float*)pl[2]);
                else hipLaunchKernelGGL((k_deinterleave<float, 3>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->stage, (float*)pl[0], (float*)pl[1], (float*)pl[2]);
            } else {
                if (dim == 2) hipLaunchKernelGGL((k_deinterleave<double, 2>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->stage, (double*)pl[0], (double*)pl[1], (double*)pl[2]);
                else hipLaunchKernelGGL((k_deinterleave<double, 3>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->stage, (double*)pl[0], (double*)pl[1], (double*)pl[2]);
            }
            DXH_HIP(hipGetLastError());
            DXH_HIP(hipStreamSynchronize(c->stream));
        }
        c->from_db = true;
    }
    if (!lean) DXH_HIP(hipStreamSynchronize(c->stream));
    c->uploaded = true;
    c->halo_valid = c->halo_top;  // a slab is uploaded with valid halos
    c->global_sq_img = -1.0;
    return DXH_OK;
}

int dxh_tvd_upload(dxh_tvd_t h, const void* img, const void* u0, const void* d0, const void* b0) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && img, "null argument");
    if (!c->have_coefs) { set_error("dxh_tvd_set_coeffs must be called before dxh_tvd_upload"); return DXH_ESTATE; }
    DXH_REQUIRE((d0 == nullptr) == (b0 == nullptr), "d0 and b0 must be given together");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = h2d_plane(c, c->u[0], img))) return rc;
    return finish_upload(c, u0, d0, b0);
}

static int to_ops(const dxh_model_op* ops, int n, ModelOps* out) {
    DXH_REQUIRE(n >= 0 && n <= 8 && (ops || n == 0), "0..8 model operations supported");
    out->n = n;
    for (int i = 0; i < n; ++i) {
        DXH_REQUIRE(ops[i].kind >= 0 && ops[i].kind <= 2, "unknown model operation %d", ops[i].kind);
        out->kind[i] = ops[i].kind;
        out->flags[i] = ops[i].flags;
        out->a[i] = ops[i].a;
        out->b[i] = ops[i].b;
    }
    return DXH_OK;
}

static int set_dense(Ctx* c, void** slot, const void* host, size_t bytes) {
    if (*slot) { DXH_HIP(hipFree(*slot)); *slot = nullptr; }
    if (!host) return DXH_OK;
    DXH_HIP(hipMalloc(slot, bytes));
    DXH_HIP(hipMemcpyAsync(*slot, host, bytes, hipMemcpyDefault, c->stream));
    DXH_HIP(hipStreamSynchronize(c->stream));
    return DXH_OK;
}

int dxh_tvd_set_base_typed(dxh_tvd_t h, const void* base, int channels, int dtype) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    DXH_REQUIRE(c->desc.ndim == 2, "the signal pipeline needs a scalar (H, W) context");
    DXH_REQUIRE(base == nullptr || channels == 1 || channels == 3, "channels must be 1 or 3");
    DXH_REQUIRE(dtype == DXH_F32 || dtype == DXH_F64, "the baseline must be float32 or float64");
    DXH_REQUIRE(dtype <= c->desc.dtype, "a float64 baseline needs a float64 context (NumPy promotes probe - baseline)");
    int rc = set_device(c);
    if (rc) return rc;
    c->base_channels = base ? channels : 0;
    c->base_dtype = dtype;
    return set_dense(c, &c->base_img, base, (size_t)c->g.n0 * c->g.L * (base ? channels : 0) * (dtype == DXH_F32 ? 4 : 8));
}

int dxh_tvd_set_base(dxh_tvd_t h, const void* base, int channels) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    return dxh_tvd_set_base_typed(h, base, channels, c->desc.dtype);
}

int dxh_tvd_set_cleaning_filter(dxh_tvd_t h, const void* filter) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    DXH_REQUIRE(c->desc.ndim == 2, "the signal pipeline needs a scalar (H, W) context");
    int rc = set_device(c);
    if (rc) return rc;
    // (numpy2 context: the signal is a float32 array, and so is a filter that leaves it one)
    return set_dense(c, &c->clean_filter, filter, (size_t)c->g.n0 * c->g.L * (c->np2 ? 4 : c->esz));
}

int dxh_tvd_upload_signal_typed(dxh_tvd_t h, const void* probe, int probe_dtype, int channels, int diff_option, int reduction,
                                const dxh_model_op* pre_ops, int n_pre) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && probe, "null argument");
    if (!c->have_coefs) { set_error("dxh_tvd_set_coeffs must be called before dxh_tvd_upload_signal"); return DXH_ESTATE; }
    DXH_REQUIRE(c->desc.ndim == 2, "the signal pipeline needs a scalar (H, W) context");
    DXH_REQUIRE(channels == 1 || channels == 3, "channels must be 1 or 3");
    DXH_REQUIRE(probe_dtype == DXH_F32 || probe_dtype == DXH_F64 || probe_dtype == DXH_U8 || probe_dtype == DXH_U16,
                "probe dtype %d not supported (float32, float64, uint8, uint16)", probe_dtype);
    DXH_REQUIRE(diff_option >= 0 && diff_option <= 3, "diff_option must be 0..3");
    DXH_REQUIRE((channels == 1 && reduction == 0) || (channels == 3 && reduction >= 1 && reduction <= 6),
                "reduction %d does not fit %d channel(s)", reduction, channels);
    DXH_REQUIRE(c->base_img == nullptr || c->base_channels == channels, "baseline has %d channel(s), probe %d",
                c->base_channels, channels);
    ModelOps pre;
    int rc = to_ops(pre_ops, n_pre, &pre);
    if (rc) return rc;
    if ((rc = set_device(c))) return rc;
    static const size_t width[4] = {4, 8, 1, 2};
    const size_t bytes = (size_t)c->g.n0 * c->g.L * channels * width[probe_dtype];
    if (c->probe_bytes < bytes) {
        if (c->probe_buf) { DXH_HIP(hipFree(c->probe_buf)); c->probe_buf = nullptr; c->probe_bytes = 0; }
        DXH_HIP(hipMalloc(&c->probe_buf, bytes));
        c->probe_bytes = bytes;
    }
    DXH_HIP(hipMemcpyAsync(c->probe_buf, probe, bytes, hipMemcpyDefault, c->stream));
    if (c->np2)   // a float32 signal (formed in float32, as NumPy forms it) into the float64 planes of the reference's float32 flow
        rc = channels == 1 ? signal_by_probe<float, 1, double>(c, probe_dtype, diff_option, reduction, pre)
                           : signal_by_probe<float, 3, double>(c, probe_dtype, diff_option, reduction, pre);
    else if (c->desc.dtype == DXH_F32)
        rc = channels == 1 ? signal_by_probe<float, 1>(c, probe_dtype, diff_option, reduction, pre)
                           : signal_by_probe<float, 3>(c, probe_dtype, diff_option, reduction, pre);
    else
        rc = channels == 1 ? signal_by_probe<double, 1>(c, probe_dtype, diff_option, reduction, pre)
                           : signal_by_probe<double, 3>(c, probe_dtype, diff_option, reduction, pre);
    if (rc) return rc;
    DXH_HIP(hipGetLastError());
    return finish_upload(c, nullptr, nullptr, nullptr);
}

int dxh_tvd_upload_signal(dxh_tvd_t h, const void* probe, int channels, int diff_option, int reduction,
                          const dxh_model_op* pre_ops, int n_pre) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null argument");
    return dxh_tvd_upload_signal_typed(h, probe, c->desc.dtype, channels, diff_option, reduction, pre_ops, n_pre);
}

int dxh_tvd_download_model(dxh_tvd_t h, void* out, const dxh_model_op* post_ops, int n_post) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && out, "null argument");
    if (!c->uploaded) { set_error("nothing to download"); return DXH_ESTATE; }
    ModelOps post;
    int rc = to_ops(post_ops, n_post, &post);
    if (rc) return rc;
    if ((rc = set_device(c))) return rc;
    const size_t dense = (size_t)c->g.n0 * c->g.L * (c->np2 ? 4 : c->esz);   // numpy2: the float32 array the reference returns
    HostPrefault pf;  // map the output pages while the device works (the iterations may still be running: DXH_FLAG_ASYNC)
    pf.start(out, dense);
    if (!c->stage || dense > (size_t)c->g.n0 * c->g.L * c->desc.dim * c->esz) {
        if (c->stage) { DXH_HIP(hipFree(c->stage)); c->stage = nullptr; }
        DXH_HIP(hipMalloc(&c->stage, (size_t)c->g.n0 * c->g.L * c->desc.dim * c->esz));
    }
    if (c->np2)
        hipLaunchKernelGGL((k_model<float, double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[c->cur], post, (float*)c->stage);
    else if (c->desc.dtype == DXH_F32)
        hipLaunchKernelGGL((k_model<float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->u[c->cur], post, (float*)c->stage);
    else
        hipLaunchKernelGGL((k_model<double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[c->cur], post, (double*)c->stage);
    DXH_HIP(hipGetLastError());
    pf.join();
    DXH_HIP(hipMemcpyAsync(out, c->stage, dense, hipMemcpyDefault, c->stream));
    DXH_HIP(hipStreamSynchronize(c->stream));
    return DXH_OK;
}

int dxh_tvd_run(dxh_tvd_t h, int n_iter, int jacobi_maxiter, double jacobi_tol, double eps, double norm_scale,
                uint32_t flags, dxh_tvd_result* result) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    if (!c->uploaded) { set_error("dxh_tvd_upload must be called before dxh_tvd_run"); return DXH_ESTATE; }
    DXH_REQUIRE(n_iter >= 0, "n_iter must be >= 0");
    DXH_REQUIRE(jacobi_maxiter >= 0, "jacobi_maxiter must be >= 0");
    int rc = set_device(c);
    if (rc) return rc;
    return c->desc.dtype == DXH_F32 ? run_t<float>(c, n_iter, jacobi_maxiter, jacobi_tol, eps, norm_scale, flags, result)
                                    : run_t<double>(c, n_iter, jacobi_maxiter, jacobi_tol, eps, norm_scale, flags, result);
}

// ---- whole call, host array in / host array out, in row bands -----------------------------------------------------
// n_iter iterations move information by at most n_iter * m rows upwards and n_iter * m + 1 rows downwards (the same
// dependence cone as the multi-GPU slabs: jacobi.py:43-54, split_bregman_tvd.py:84-89,144).  A band of rows with that many
// extra rows on its open sides therefore reproduces the whole-image result on its own rows, bit for bit, without ever
// talking to its neighbours -- and the bands can go through the device one after the other, band b + 1 being uploaded and
// band b - 1 downloaded while band b iterates.  The price is the recomputed halo rows.
static int band_streams(int device, hipStream_t out[3]) {
    constexpr int kMax = 16;
    static std::mutex mtx;
    static hipStream_t streams[kMax][3] = {};
    if (device < 0 || device >= kMax) { set_error("device %d out of range", device); return DXH_EINVAL; }
    std::lock_guard<std::mutex> lock(mtx);
    for (int i = 0; i < 3; ++i) {
        if (!streams[device][i]) DXH_HIP(hipStreamCreateWithFlags(&streams[device][i], hipStreamNonBlocking));
        out[i] = streams[device][i];
    }
    return DXH_OK;
}

// Band cuts: S bands over H rows; the first and the last band get `edge` times the height of the others (edge < 1: the
// first upload and the last download, which nothing hides, get shorter).  cuts has S + 1 entries.
static void band_cuts(int64_t H, int S, double edge, std::vector<int64_t>* cuts) {
    cuts->assign(S + 1, 0);
    const double total = S >= 3 ? (S - 2) + 2 * edge : (double)S;
    double acc = 0;
    for (int b = 0; b < S; ++b) {
        acc += (S >= 3 && (b == 0 || b == S - 1)) ? edge : 1.0;
        (*cuts)[b + 1] = b + 1 == S ? H : std::min<int64_t>(H, std::max<int64_t>((*cuts)[b] + 1, (int64_t)std::llround(H * acc / total)));
    }
}

// finishing time of the bands through upload -> iterate -> download (three engines, each takes the bands in order);
// sec_up / sec_comp / sec_down per row, per_band: fixed cost of a band's launches
static double banded_time(const std::vector<int64_t>& cuts, int64_t H, int64_t ht, int64_t hb, double sec_up, double sec_comp,
                          double sec_down, double per_band) {
    double tu = 0, tc = 0, td = 0;
    for (size_t b = 0; b + 1 < cuts.size(); ++b) {
        const int64_t lo = std::max<int64_t>(0, cuts[b] - ht), hi = std::min<int64_t>(H, cuts[b + 1] + hb);
        tu += (hi - lo) * sec_up;
        tc = std::max(tc, tu) + (hi - lo) * sec_comp + per_band;
        td = std::max(td, tc) + (cuts[b + 1] - cuts[b]) * sec_down;
    }
    return td;
}

// host_dtype: element type of img / out (the coefficient arrays have the context's); promotion: DXH_PROMOTION_* of the
// band contexts (numpy2: a float32 host image through float64 contexts, scalar coefficients)
static int solve_banded_impl(const dxh_tvd_desc* desc, const void* img, void* out, int host_dtype, int promotion, double mu,
                             const void* mu_arr, double omega, const void* omega_arr, int has_ell, double ell,
                             const void* ell_arr, int n_iter, int jacobi_maxiter, uint32_t flags, int nbands,
                             int* nbands_used) {
    DXH_REQUIRE(desc && img && out, "null argument");
    DXH_REQUIRE(desc->struct_size == (int32_t)sizeof(dxh_tvd_desc), "dxh_tvd_desc size mismatch");
    DXH_REQUIRE(desc->dim == 2, "row bands need dim == 2 (axis 0 is the band axis; a volume's planes are coupled along it "
                "by the same stencil, but its halos would be whole planes)");
    DXH_REQUIRE(desc->top_is_boundary && desc->bottom_is_boundary, "the descriptor must describe the whole image");
    DXH_REQUIRE(n_iter >= 0 && jacobi_maxiter >= 1, "n_iter >= 0 and jacobi_maxiter >= 1 required");
    DXH_REQUIRE(nbands >= 0 && nbands <= 64, "nbands must be 0 (choose) .. 64");
    DXH_REQUIRE(!is_device_ptr(img) && !is_device_ptr(out), "host arrays expected (device arrays: dxh_tvd_run)");
    const int64_t H = desc->n0, L = desc->n1 * desc->n2;
    const size_t esz = host_dtype == DXH_F32 ? 4 : 8, csz = desc->dtype == DXH_F32 ? 4 : 8;
    const bool typed = host_dtype != desc->dtype;
    const bool np2 = (promotion & DXH_PROMOTION_NUMPY2) != 0;
    const int64_t ht = (int64_t)n_iter * jacobi_maxiter, hb = ht + 1;
    int S = nbands;
    double edge = 1.0;
    std::vector<int64_t> cuts;
    if (S == 0) {
        // rates measured on MI355X + PCIe 5 from pageable memory: 50 GB/s each way; per value
        // and iteration 1.6 ps (float) / 3.3 ps (double) / 3.0 ps (numpy2) in one piece, isotropic and array-valued coefficients ~1.5x;
        // bands iterate ~20 % slower than their share of the whole image (fewer row chunks, the copies alongside)
        const double sec_copy = (double)L * esz / 50e9;
        double per = desc->dtype == DXH_F32 ? 1.6e-12 : 3.3e-12;
        if (np2) per = 3.0e-12;   // 0.285 ms per iteration at 96 M elements (round 3: 3.9e-12)
        if ((flags & DXH_FLAG_ISOTROPIC) || mu_arr || omega_arr || ell_arr) per *= 1.5;
        const double sec_comp = (double)L * per * n_iter * jacobi_maxiter;
        band_cuts(H, 1, 1.0, &cuts);
        double best = banded_time(cuts, H, ht, hb, sec_copy, sec_comp, sec_copy, 0.0);
        S = 1;
        for (int k = 2; k <= 8; ++k) {
            if ((H / k) < 64) break;
            for (double e : {1.0, 0.7, 0.45}) {
                if (k < 3 && e != 1.0) continue;
                band_cuts(H, k, e, &cuts);
                // (the numpy2 pipelines are arithmetic-bound: the copies alongside cost them less -- 1.1, measured)
                const double t = banded_time(cuts, H, ht, hb, sec_copy, (np2 ? 1.1 : 1.2) * sec_comp, sec_copy, 0.3e-3);
                if (t < 0.97 * best) { best = t; S = k; edge = e; }
            }
        }
    }
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, H / 8));
    if (const char* e = getenv("DXH_BAND_EDGE")) edge = std::min(1.0, std::max(0.1, atof(e)));   // developer knob
    band_cuts(H, S, edge, &cuts);
    if (nbands_used) *nbands_used = S;
    DXH_HIP(hipSetDevice(desc->device));
    hipStream_t streams[3];
    int rc = band_streams(desc->device, streams);
    if (rc) return rc;
    HostPrefault pf;  // map the result's pages while the first band goes up
    pf.start(out, (size_t)H * L * esz, {{img, (size_t)H * L * esz}});

    struct Band { int64_t own0, own1, lo, hi; dxh_tvd_t ctx = nullptr; hipEvent_t up = nullptr, done = nullptr; std::atomic<int> state{0};
                  void* raw = nullptr; };   // raw: the band's rows as the host stores them (typed), converted on the device
    std::atomic<int> downloaded{0};   // bands whose rows are back on the host (or given up)
    std::vector<Band> bands(S);
    for (int b = 0; b < S; ++b) {
        Band& bd = bands[b];
        bd.own0 = cuts[b];
        bd.own1 = cuts[b + 1];
        bd.lo = std::max<int64_t>(0, bd.own0 - ht);
        bd.hi = std::min<int64_t>(H, bd.own1 + hb);
    }
    std::string worker_err;
    const bool trace = getenv("DXH_BAND_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<std::array<double, 6>> tr(S);
    auto at = [&](const void* base, int64_t row) -> const void* {
        return base ? static_cast<const char*>(base) + (size_t)row * L * esz : nullptr;
    };
    auto coef_at = [&](const void* base, int64_t row) -> const void* {
        return base ? static_cast<const char*>(base) + (size_t)row * L * csz : nullptr;
    };
    // the uploader: creates band b's context, sends its rows up and enqueues its iterations
    std::thread uploader([&]() {
        for (int b = 0; b < S; ++b) {
            Band& bd = bands[b];
            dxh_tvd_desc d = *desc;
            d.n0 = bd.hi - bd.lo;
            d.top_is_boundary = bd.lo == 0;
            d.bottom_is_boundary = bd.hi == H;
            d.stream = streams[b % 3];
            // three streams in turn: band b re-uses the stream of band b - 3, whose download must have been issued
            // before anything of band b is enqueued behind it
            while (downloaded.load(std::memory_order_acquire) < b - 2) std::this_thread::yield();
            RoctxRange range("row band %lld: create, upload, enqueue %lld iterations", b, n_iter);
            tr[b][0] = now_ms();
            int r = dxh_tvd_create(&bd.ctx, &d);
            if (!r && promotion) r = dxh_tvd_set_promotion(bd.ctx, promotion);
            if (!r) r = dxh_tvd_set_coeffs(bd.ctx, mu, coef_at(mu_arr, bd.lo), omega, coef_at(omega_arr, bd.lo), has_ell, ell, coef_at(ell_arr, bd.lo));
            tr[b][1] = now_ms();
            if (!r) {
                Ctx* c = reinterpret_cast<Ctx*>(bd.ctx);
                // the uploads go over the bus one after the other (in flight together they would all arrive late)
                hipError_t e = hipEventCreateWithFlags(&bd.up, hipEventDisableTiming);
                if (e == hipSuccess && b > 0) e = hipStreamWaitEvent(c->stream, bands[b - 1].up, 0);
                if (e != hipSuccess) { set_error("band event failed: %s", hipGetErrorString(e)); r = DXH_EHIP; }
                if (!r && !typed) r = h2d_plane(c, c->u[0], at(img, bd.lo));
                if (!r && typed) {   // the rows as stored, then converted straight into the pitched image plane
                    const size_t nb = (size_t)(bd.hi - bd.lo) * L * esz;
                    e = pool_malloc(&bd.raw, nb);
                    if (e == hipSuccess) e = hipMemcpyAsync(bd.raw, at(img, bd.lo), nb, hipMemcpyDefault, c->stream);
                    if (e != hipSuccess) { set_error("band upload failed: %s", hipGetErrorString(e)); r = DXH_EHIP; }
                }
                if (!r && hipEventRecord(bd.up, c->stream) != hipSuccess) { set_error("hipEventRecord failed"); r = DXH_EHIP; }
                if (!r && typed) {
                    if (host_dtype == DXH_F32) hipLaunchKernelGGL((k_dense_to_plane<float, double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)bd.raw, (double*)c->u[0]);
                    else hipLaunchKernelGGL((k_dense_to_plane<double, float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)bd.raw, (float*)c->u[0]);
                    if (hipGetLastError() != hipSuccess) { set_error("band conversion failed"); r = DXH_EHIP; }
                }
                if (!r) r = finish_upload(c, nullptr, nullptr, nullptr, /*lean=*/true);
            }
            tr[b][2] = now_ms();
            if (!r) {
                // ... and the bands iterate one after the other: side by side they would share the CUs and all finish
                // late, with nothing left to hide the downloads behind
                Ctx* c = reinterpret_cast<Ctx*>(bd.ctx);
                hipError_t e = hipEventCreateWithFlags(&bd.done, hipEventDisableTiming);
                if (e == hipSuccess && b > 0) e = hipStreamWaitEvent(c->stream, bands[b - 1].done, 0);
                if (e != hipSuccess) { set_error("band event failed: %s", hipGetErrorString(e)); r = DXH_EHIP; }
            }
            if (!r) r = dxh_tvd_run(bd.ctx, n_iter, jacobi_maxiter, -1.0, -1.0, 0.0, flags | DXH_FLAG_ASYNC, nullptr);
            if (!r && hipEventRecord(bd.done, reinterpret_cast<Ctx*>(bd.ctx)->stream) != hipSuccess) { set_error("hipEventRecord failed"); r = DXH_EHIP; }
            tr[b][3] = now_ms();
            if (r) {
                worker_err = dxh_last_error();
                for (int k = b; k < S; ++k) bands[k].state.store(r, std::memory_order_release);  // negative: failed
                return;
            }
            bd.state.store(1, std::memory_order_release);
        }
    });
    pf.join();
    rc = DXH_OK;
    for (int b = 0; b < S && rc == DXH_OK; ++b) {
        Band& bd = bands[b];
        int st;
        while ((st = bd.state.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
        if (st < 0) { rc = st; break; }
        Ctx* c = reinterpret_cast<Ctx*>(bd.ctx);
        RoctxRange range("row band %lld: download of %lld owned rows", b, (long long)(bd.own1 - bd.own0));
        tr[b][4] = now_ms();
        const char* src = static_cast<const char*>(c->u[c->cur]) + (size_t)(bd.own0 - bd.lo) * c->g.pitch * c->esz;
        hipError_t e = hipSuccess;
        void* conv = nullptr;
        if (!typed) {
            e = hipMemcpy2DAsync(static_cast<char*>(out) + (size_t)bd.own0 * L * esz, L * esz, src, c->g.pitch * c->esz,
                                 L * esz, bd.own1 - bd.own0, hipMemcpyDefault, c->stream);
        } else {   // the owned rows, rounded to the host's type (ndarray.astype) into a dense block, then down in one copy
            Geom g2 = c->g;
            g2.n0 = bd.own1 - bd.own0;
            const size_t nb = (size_t)g2.n0 * L * esz;
            e = pool_malloc(&conv, nb);
            if (e == hipSuccess) {
                if (host_dtype == DXH_F32) hipLaunchKernelGGL((k_plane_to_dense<double, float>), elem_grid(g2), dim3(256), 0, c->stream, g2, (const double*)src, (float*)conv);
                else hipLaunchKernelGGL((k_plane_to_dense<float, double>), elem_grid(g2), dim3(256), 0, c->stream, g2, (const float*)src, (double*)conv);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(static_cast<char*>(out) + (size_t)bd.own0 * L * esz, conv, nb, hipMemcpyDefault, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (conv) pool_free(conv);
        if (e != hipSuccess) { set_error("band download failed: %s", hipGetErrorString(e)); rc = DXH_EHIP; }
        tr[b][5] = now_ms();
        downloaded.store(b + 1, std::memory_order_release);
    }
    downloaded.store(S + 3, std::memory_order_release);   // on an error: let the uploader run out
    uploader.join();
    const double t_joined = now_ms();
    if (rc != DXH_OK && !worker_err.empty()) set_error("%s", worker_err.c_str());
    for (auto& bd : bands)
        if (bd.ctx) {
            (void)hipStreamSynchronize(reinterpret_cast<Ctx*>(bd.ctx)->stream);
            dxh_tvd_destroy(bd.ctx);
        }
    for (auto& bd : bands) {
        if (bd.raw) pool_free(bd.raw);
        if (bd.up) (void)hipEventDestroy(bd.up);
        if (bd.done) (void)hipEventDestroy(bd.done);
    }
    if (trace) {
        for (int b = 0; b < S; ++b)
            fprintf(stderr, "[dxh band %d rows %lld..%lld (+%lld/%lld)] create %.2f-%.2f upload -%.2f enqueue -%.2f | download %.2f-%.2f\n",
                    b, (long long)bands[b].own0, (long long)bands[b].own1, (long long)(bands[b].own0 - bands[b].lo),
                    (long long)(bands[b].hi - bands[b].own1), tr[b][0], tr[b][1], tr[b][2], tr[b][3], tr[b][4], tr[b][5]);
        fprintf(stderr, "[dxh bands] joined %.2f, contexts destroyed %.2f ms\n", t_joined, now_ms());
    }
    return rc;
}

int dxh_tvd_solve_banded(const dxh_tvd_desc* desc, const void* img, void* out, double mu, const void* mu_arr, double omega,
                         const void* omega_arr, int has_ell, double ell, const void* ell_arr, int n_iter,
                         int jacobi_maxiter, uint32_t flags, int nbands, int* nbands_used) {
    DXH_REQUIRE(desc, "null argument");
    return solve_banded_impl(desc, img, out, desc->dtype, DXH_PROMOTION_NATIVE, mu, mu_arr, omega, omega_arr, has_ell, ell, ell_arr,
                             n_iter, jacobi_maxiter, flags, nbands, nbands_used);
}

int dxh_tvd_solve_banded_typed(const dxh_tvd_desc* desc, const void* img, void* out, int host_dtype, int promotion, double mu,
                               double omega, int has_ell, double ell, int n_iter, uint32_t flags, int nbands,
                               int* nbands_used) {
    DXH_REQUIRE(desc, "null argument");
    DXH_REQUIRE(host_dtype == DXH_F32 || host_dtype == DXH_F64, "host dtype must be float32 or float64");
    DXH_REQUIRE(promotion == DXH_PROMOTION_NATIVE || promotion == DXH_PROMOTION_NUMPY2, "promotion must be native or numpy2");
    if (promotion == DXH_PROMOTION_NUMPY2)
        DXH_REQUIRE(host_dtype == DXH_F32 && desc->dtype == DXH_F64 && !(flags & DXH_FLAG_ISOTROPIC),
                    "numpy2 promotion: a float32 host image through float64 contexts, anisotropic");
    return solve_banded_impl(desc, img, out, host_dtype, promotion, mu, nullptr, omega, nullptr, has_ell, ell, nullptr, n_iter, 1,
                             flags, nbands, nbands_used);
}

int dxh_tvd_download(dxh_tvd_t h, void* u, void* d, void* b) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    if (!c->uploaded) { set_error("nothing to download"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    if (c->signal) {  // a slab: a timed-out halo wait makes the state invalid (slab_check_fault)
        DXH_HIP(hipStreamSynchronize(c->stream));
        if ((rc = slab_check_fault(c))) return rc;
    }
    if (u && (rc = d2h_plane(c, u, c->u[c->cur]))) return rc;
    if (d || b) {
        const int dim = c->desc.dim;
        const size_t dense = (size_t)c->g.n0 * c->g.L * dim * c->esz;
        if (c->from_db || c->explicit_db) {
            // explicit planes: interleave them into the reference layout
            void* tmpd = nullptr;
            DXH_HIP(hipMalloc(&tmpd, dense));
            const dim3 grid2 = elem_grid(c->g), blk2(256);
            for (int pass = 0; pass < 2; ++pass) {
                void* host = pass == 0 ? d : b;
                if (!host) continue;
                void** pl = pass == 0 ? c->dpl : c->bpl;
                if (c->desc.dtype == DXH_F32) {
                    if (dim == 2) hipLaunchKernelGGL((k_interleave<float, 2>), grid2, blk2, 0, c->stream, c->g, (const float*)pl[0], (const float*)pl[1], (const float*)pl[2], (float*)tmpd);
                    else hipLaunchKernelGGL((k_interleave<float, 3>), grid2, blk2, 0, c->stream, c->g, (const float*)pl[0], (const float*)pl[1], (const float*)pl[2], (float*)tmpd);
                } else {
                    if (dim == 2) hipLaunchKernelGGL((k_interleave<double, 2>), grid2, blk2, 0, c->stream, c->g, (const double*)pl[0], (const double*)pl[1], (const double*)pl[2], (double*)tmpd);
                    else hipLaunchKernelGGL((k_interleave<double, 3>), grid2, blk2, 0, c->stream, c->g, (const double*)pl[0], (const double*)pl[1], (const double*)pl[2], (double*)tmpd);
                }
                hipError_t e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(host, tmpd, dense, hipMemcpyDefault, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) { (void)hipFree(tmpd); set_error("download failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
            }
            DXH_HIP(hipFree(tmpd));
            DXH_HIP(hipStreamSynchronize(c->stream));
            return DXH_OK;
        }
        DevTmp dd_guard, bb_guard;
        void *&dd = dd_guard.p, *&bb = bb_guard.p;
        if (d) DXH_HIP(hipMalloc(&dd, dense));
        if (b) DXH_HIP(hipMalloc(&bb, dense));
        const bool iso = c->last_iso;
        const dim3 grid = elem_grid(c->g), blk(256);
#define DXH_EXPAND(T, DIM, ISO) \
    hipLaunchKernelGGL((k_expand<T, DIM, ISO>), grid, blk, 0, c->stream, c->g, make_q<T, DIM>(c, c->cur, c->cur), \
                       make_coef<T>(c), make_planes<T>(c), (T*)dd, (T*)bb)
        if (c->desc.dtype == DXH_F32) {
            if (dim == 2) { if (iso) DXH_EXPAND(float, 2, true); else DXH_EXPAND(float, 2, false); }
            else          { if (iso) DXH_EXPAND(float, 3, true); else DXH_EXPAND(float, 3, false); }
        } else {
            if (dim == 2) { if (iso) DXH_EXPAND(double, 2, true); else DXH_EXPAND(double, 2, false); }
            else          { if (iso) DXH_EXPAND(double, 3, true); else DXH_EXPAND(double, 3, false); }
        }
#undef DXH_EXPAND
        DXH_HIP(hipGetLastError());
        if (d) DXH_HIP(hipMemcpyAsync(d, dd, dense, hipMemcpyDefault, c->stream));
        if (b) DXH_HIP(hipMemcpyAsync(b, bb, dense, hipMemcpyDefault, c->stream));
        DXH_HIP(hipStreamSynchronize(c->stream));  // dd / bb are released by their guards
    }
    DXH_HIP(hipStreamSynchronize(c->stream));
    return DXH_OK;
}

// ---- halo plumbing -------------------------------------------------------------------------------
int dxh_tvd_halo_bytes(dxh_tvd_t h, int64_t nrows, size_t* bytes) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && bytes && nrows >= 0, "bad argument");
    *bytes = (size_t)(1 + c->desc.dim) * (size_t)nrows * (size_t)c->g.pitch * c->esz;
    return DXH_OK;
}

static int halo_copy(Ctx* c, int64_t row0, int64_t nrows, void* buf, bool pack) {
    DXH_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= c->g.n0, "row range [%lld, %lld) outside the slab",
                (long long)row0, (long long)(row0 + nrows));
    if (c->from_db) { set_error("halo exchange before the first iteration of a warm start is not supported"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const size_t row_bytes = (size_t)c->g.pitch * c->esz;
    const size_t blk = (size_t)nrows * row_bytes;
    if (blk == 0) return DXH_OK;
    char* p = static_cast<char*>(buf);
    void* fields[4] = {c->u[c->cur], c->q[0][c->cur], c->q[1][c->cur], c->q[2][c->cur]};
    for (int f = 0; f < 1 + c->desc.dim; ++f) {
        char* fld = static_cast<char*>(fields[f]) + (size_t)row0 * row_bytes;
        if (pack)
            DXH_HIP(hipMemcpyAsync(p + f * blk, fld, blk, hipMemcpyDeviceToDevice, c->stream));
        else
            DXH_HIP(hipMemcpyAsync(fld, p + f * blk, blk, hipMemcpyDeviceToDevice, c->stream));
    }
    return DXH_OK;
}

int dxh_tvd_pack_rows(dxh_tvd_t h, int64_t row0, int64_t nrows, void* dst) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && dst, "null argument");
    return halo_copy(c, row0, nrows, dst, true);
}
int dxh_tvd_unpack_rows(dxh_tvd_t h, int64_t row0, int64_t nrows, const void* src) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && src, "null argument");
    return halo_copy(c, row0, nrows, const_cast<void*>(src), false);
}

int dxh_tvd_attach_comm(dxh_tvd_t h, dxh_comm_t comm, int64_t halo_top_rows, int64_t halo_bottom_rows) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    Comm* cm = reinterpret_cast<Comm*>(comm);
    DXH_REQUIRE(c && cm, "null argument");
    DXH_REQUIRE(cm->device == c->desc.device, "communicator lives on device %d, the context on %d", cm->device, c->desc.device);
    DXH_REQUIRE(halo_top_rows >= 1 && halo_bottom_rows >= halo_top_rows + 1,
                "halo rows (%lld above, %lld below): need k >= 1 above and at least k + 1 below",
                (long long)halo_top_rows, (long long)halo_bottom_rows);
    // one rank with halos on both sides: the slab is its own neighbour, i.e. the problem is periodic along axis 0 (lets
    // the whole exchange / overlap machinery run on a single GPU)
    const bool periodic = cm->world == 1 && !c->desc.top_is_boundary && !c->desc.bottom_is_boundary;
    if (!periodic) {
        DXH_REQUIRE((cm->rank == 0) == (c->desc.top_is_boundary != 0), "rank %d of %d: top_is_boundary = %d does not fit",
                    cm->rank, cm->world, c->desc.top_is_boundary);
        DXH_REQUIRE((cm->rank == cm->world - 1) == (c->desc.bottom_is_boundary != 0),
                    "rank %d of %d: bottom_is_boundary = %d does not fit", cm->rank, cm->world, c->desc.bottom_is_boundary);
    }
    const int64_t top = c->desc.top_is_boundary ? 0 : halo_top_rows;
    const int64_t bot = c->desc.bottom_is_boundary ? 0 : halo_bottom_rows;
    // a slab must own the rows both neighbours ask for
    DXH_REQUIRE(c->g.n0 - top - bot >= halo_bottom_rows || cm->world == 1,
                "a slab of %lld owned rows is smaller than the halo (%lld rows): use fewer ranks or shorter halos",
                (long long)(c->g.n0 - top - bot), (long long)halo_bottom_rows);
    int rc = set_device(c);
    if (rc) return rc;
    if (!c->signal) {
        DXH_HIP(hipEventCreateWithFlags(&c->eL, hipEventDisableTiming));
        int lo = 0, hi = 0;
        DXH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        DXH_HIP(hipStreamCreateWithPriority(&c->alt_owned, hipStreamNonBlocking, hi));  // its own hardware queue
        c->alt = c->alt_owned;
        DXH_HIP(hipMalloc((void**)&c->signal, 256));
        DXH_HIP(hipMemset(c->signal, 0, 256));
        c->signal_target = 0;
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->desc.device) != hipSuccess) can = 0;
        c->wait_mode = can ? 1 : 2;
        if (const char* e = getenv("DXH_SLAB_WAIT")) {  // developer knob: 1 stream wait-value, 2 polling kernel
            const int v = atoi(e);
            if (v == 1 || v == 2) c->wait_mode = v;
        }
        if (const char* e = getenv("DXH_SLAB_WAIT_TIMEOUT_MS")) {  // polling kernel: time-out (default 10 s)
            const long v = atol(e);
            if (v > 0 && v <= 3600000) c->wait_timeout_ms = (unsigned)v;
        }
        if (const char* e = getenv("DXH_SLAB_TEST_LOST_SIGNAL")) c->test_lost_signal = atoi(e) == 1;  // tests only
    }
    if (c->comm != cm) {
        comm_release(c->comm);
        ++cm->refs;
    }
    c->comm = cm;
    c->halo_top = halo_top_rows;
    c->halo_bot = halo_bottom_rows;
    c->top = top;
    c->bot = bot;
    {
        const size_t row_bytes = (size_t)c->g.pitch * c->esz, nf = 1 + (size_t)c->desc.dim;
        const size_t need[4] = {top ? nf * halo_bottom_rows * row_bytes : 0, bot ? nf * halo_top_rows * row_bytes : 0,
                                bot ? nf * halo_bottom_rows * row_bytes : 0, top ? nf * halo_top_rows * row_bytes : 0};
        for (int i = 0; i < 4; ++i) {
            if (c->xbytes[i] >= need[i]) continue;
            if (c->xbuf[i]) { DXH_HIP(hipFree(c->xbuf[i])); c->xbuf[i] = nullptr; c->xbytes[i] = 0; }
            DXH_HIP(hipMalloc((void**)&c->xbuf[i], need[i]));
            c->xbytes[i] = need[i];
        }
    }
    c->halo_valid = halo_top_rows;  // an uploaded slab carries valid halos
    c->global_sq_img = -1.0;
    return DXH_OK;
}

int dxh_tvd_exchange(dxh_tvd_t h) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    if (!c->comm) { set_error("dxh_tvd_attach_comm must be called first"); return DXH_ESTATE; }
    if (!c->uploaded) { set_error("nothing uploaded"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    if (c->top == 0 && c->bot == 0) return DXH_OK;
    return exchange_now(c);
}

int dxh_tvd_run_slab(dxh_tvd_t h, int n_iter, int jacobi_maxiter, double eps, uint32_t flags, dxh_tvd_result* result) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    if (!c->uploaded) { set_error("dxh_tvd_upload must be called before dxh_tvd_run_slab"); return DXH_ESTATE; }
    DXH_REQUIRE(n_iter >= 0, "n_iter must be >= 0");
    DXH_REQUIRE(jacobi_maxiter >= 1, "jacobi_maxiter must be >= 1");
    DXH_HIP(hipSetDevice(c->desc.device));  // no join here: an exchange left in flight by DXH_FLAG_CONTINUE stays overlapped
    return c->desc.dtype == DXH_F32 ? run_slab_t<float>(c, n_iter, jacobi_maxiter, eps, flags, result)
                                    : run_slab_t<double>(c, n_iter, jacobi_maxiter, eps, flags, result);
}

int dxh_tvd_partial_norms(dxh_tvd_t h, int64_t row0, int64_t nrows, double* sq_inc, double* sq_img) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    DXH_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= c->g.n0, "row range outside the slab");
    if (!c->uploaded) { set_error("nothing uploaded"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const bool f32 = c->desc.dtype == DXH_F32;
    if (sq_inc) {
        rc = f32 ? sq_diff<float>(c, c->u[c->cur], c->u[c->cur ^ 1], row0, nrows, sq_inc, 0)
                 : sq_diff<double>(c, c->u[c->cur], c->u[c->cur ^ 1], row0, nrows, sq_inc, 0);
        if (rc) return rc;
    }
    if (sq_img) {
        // img = of / omega is not kept; the slab norm over a row range is recomputed from `of` only for
        // scalar omega (exact up to the rounding of omega*img); callers with array omega pass row0=0,nrows=n0.
        if (row0 == 0 && nrows == c->g.n0) {
            *sq_img = c->sq_img;
        } else {
            DXH_REQUIRE(!std::isnan(c->omega_s) && c->omega_s != 0.0, "partial image norm needs a scalar, non-zero omega");
            double s = 0.0;
            rc = f32 ? sq_diff<float>(c, c->of, nullptr, row0, nrows, &s, 1) : sq_diff<double>(c, c->of, nullptr, row0, nrows, &s, 1);
            if (rc) return rc;
            *sq_img = s / (c->omega_s * c->omega_s);
        }
    }
    return DXH_OK;
}

int dxh_tvd_energy(dxh_tvd_t h, int which, double* value) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && value, "null argument");
    DXH_REQUIRE(which == 0 || which == 1, "which: 0 = current iterate, 1 = previous iterate");
    if (!c->uploaded) { set_error("nothing uploaded"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const dim3 grid = elem_grid(c->g), blk(256);
    const unsigned nb = grid.x;
    double* part = nullptr;
    DXH_HIP(hipMalloc((void**)&part, sizeof(double) * nb));
    const void* x = c->u[which == 0 ? c->cur : c->cur ^ 1];
    const double om = std::isnan(c->omega_s) ? 0.0 : c->omega_s;
#define DXH_EN(T, DIM) hipLaunchKernelGGL((k_energy<T, DIM>), grid, blk, 0, c->stream, c->g, (const T*)x, (const T*)c->of, om, \
                                          (const T*)c->om_plane, make_coef<T>(c), make_planes<T>(c), part)
    if (c->desc.dtype == DXH_F32) { if (c->desc.dim == 2) DXH_EN(float, 2); else DXH_EN(float, 3); }
    else                          { if (c->desc.dim == 2) DXH_EN(double, 2); else DXH_EN(double, 3); }
#undef DXH_EN
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, c->stream, part, (int)nb, c->red_out + 6);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) *value = reinterpret_cast<volatile double*>(c->red_out)[6];
    (void)hipFree(part);
    if (e != hipSuccess) { set_error("energy evaluation failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
    return DXH_OK;
}

int dxh_tvd_adapt_ell(dxh_tvd_t h) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    if (!c->uploaded || !c->explicit_db) { set_error("dxh_tvd_adapt_ell needs a context in explicit mode (run with DXH_FLAG_EXPLICIT first)"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const dim3 grid = elem_grid(c->g), blk(256);
    const bool f32 = c->desc.dtype == DXH_F32;
    if (c->np2) {
        // the reference's float32 flow: ell becomes a float32 array, the threshold mu / ell float32 (scalar mu) or float64
        // (float64 mu array); the Jacobi diagonals keep their cached scalar values (quirk Q3)
        DXH_REQUIRE(!c->het || c->het_cdiag || c->np2_adapted, "numpy2 promotion: the adaptive schedule needs scalar omega and ell");
        DXH_REQUIRE(!c->np2_mu_is_kq, "numpy2 promotion: the adaptive schedule needs mu as a scalar or a float64 array");
        const bool mu_arr = c->het && c->mu_plane;
        if ((rc = alloc_plane(c, &c->cp_ell)) || (rc = alloc_plane(c, &c->cp_kq))) return rc;
        if (!c->het) {   // scalar coefficients so far: planes of the same scalars for the kernels that want planes
            if ((rc = alloc_plane(c, &c->cp_diag)) || (rc = alloc_plane(c, &c->cp_dsc))) return rc;
            hipLaunchKernelGGL((k_fill_coef<double>), grid, blk, 0, c->stream, c->g, make_coef<double>(c), (double*)c->cp_ell, (double*)c->cp_diag, (double*)c->cp_dsc, (double*)c->cp_kq);
            DXH_HIP(hipGetLastError());
        }
        const void* un = c->u[c->cur];
        if (c->desc.dim == 2) hipLaunchKernelGGL((k_adapt_ell_np2<2>), grid, blk, 0, c->stream, c->g, (const double*)un, c->mu_s, mu_arr ? (const double*)c->mu_plane : nullptr, (double*)c->cp_ell, (double*)c->cp_kq);
        else hipLaunchKernelGGL((k_adapt_ell_np2<3>), grid, blk, 0, c->stream, c->g, (const double*)un, c->mu_s, mu_arr ? (const double*)c->mu_plane : nullptr, (double*)c->cp_ell, (double*)c->cp_kq);
        DXH_HIP(hipGetLastError());
        c->het = true;
        c->het_cdiag = false;
        c->np2_adapted = true;
        c->in_valid = false;
        return DXH_OK;
    }
    if (!c->het) {
        // materialise the four coefficient planes with exactly the scalar path's values
        if ((rc = alloc_plane(c, &c->cp_ell)) || (rc = alloc_plane(c, &c->cp_diag)) || (rc = alloc_plane(c, &c->cp_dsc)) ||
            (rc = alloc_plane(c, &c->cp_kq)))
            return rc;
        if (f32) hipLaunchKernelGGL((k_fill_coef<float>), grid, blk, 0, c->stream, c->g, make_coef<float>(c), (float*)c->cp_ell, (float*)c->cp_diag, (float*)c->cp_dsc, (float*)c->cp_kq);
        else hipLaunchKernelGGL((k_fill_coef<double>), grid, blk, 0, c->stream, c->g, make_coef<double>(c), (double*)c->cp_ell, (double*)c->cp_diag, (double*)c->cp_dsc, (double*)c->cp_kq);
        DXH_HIP(hipGetLastError());
        c->het = true;
    }
    c->in_valid = false;  // the coefficient planes no longer follow from the given arrays
    c->het_cdiag = false; // ell varies per element from here on: nobody may take it for the scalar it was (make_planes_min)
    const void* u = c->u[c->cur];
#define DXH_AD(T, DIM) hipLaunchKernelGGL((k_adapt_ell<T, DIM>), grid, blk, 0, c->stream, c->g, (const T*)u, (T)c->mu_s, \
                                          (const T*)c->mu_plane, (T*)c->cp_ell, (T*)c->cp_kq)
    if (f32) { if (c->desc.dim == 2) DXH_AD(float, 2); else DXH_AD(float, 3); }
    else     { if (c->desc.dim == 2) DXH_AD(double, 2); else DXH_AD(double, 3); }
#undef DXH_AD
    DXH_HIP(hipGetLastError());
    return DXH_OK;
}

int dxh_scale(int device, int dtype, size_t n, const void* src, double factor, void* dst) {
    DXH_REQUIRE(src && dst, "null argument");
    DXH_REQUIRE(dtype == DXH_F32 || dtype == DXH_F64, "dtype must be float32 or float64");
    DXH_REQUIRE(is_device_ptr(src) && is_device_ptr(dst), "dxh_scale works on device arrays (dxh_mem_alloc)");
    DXH_HIP(hipSetDevice(device));
    hipStream_t st = nullptr;
    DXH_HIP(cached_stream(device, &st));
    if (n == 0) return DXH_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (dtype == DXH_F32) hipLaunchKernelGGL((k_scale_dense<float>), grid, blk, 0, st, (const float*)src, (float)factor, (float*)dst, n);
    else hipLaunchKernelGGL((k_scale_dense<double>), grid, blk, 0, st, (const double*)src, factor, (double*)dst, n);
    DXH_HIP(hipGetLastError());
    DXH_HIP(hipStreamSynchronize(st));
    return DXH_OK;
}

int dxh_tvd_set_promotion(dxh_tvd_t h, int mode) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    DXH_REQUIRE(mode == DXH_PROMOTION_NATIVE || mode == DXH_PROMOTION_NUMPY2 || mode == (DXH_PROMOTION_NUMPY2 | DXH_PROMOTION_MU_IS_THRESHOLD),
                "unknown promotion mode %d", mode);
    if (mode != DXH_PROMOTION_NATIVE)
        DXH_REQUIRE(c->desc.dtype == DXH_F64, "numpy2 promotion needs a float64 context (it holds the reference's float64 iterate "
                    "of a float32 image)");
    c->np2 = (mode & DXH_PROMOTION_NUMPY2) != 0;
    c->np2_mu_is_kq = (mode & DXH_PROMOTION_MU_IS_THRESHOLD) != 0;
    c->np2_mu_auto = false;
    c->uploaded = false;
    c->have_coefs = false;
    return DXH_OK;
}

// image of another float type than the context's: converted on the device (float32 -> float64 is exact; the other way
// rounds to nearest like ndarray.astype)
int dxh_tvd_upload_typed(dxh_tvd_t h, const void* img, int img_dtype) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && img, "null argument");
    DXH_REQUIRE(img_dtype == DXH_F32 || img_dtype == DXH_F64, "image dtype must be float32 or float64");
    if (img_dtype == c->desc.dtype) return dxh_tvd_upload(h, img, nullptr, nullptr, nullptr);
    if (!c->have_coefs) { set_error("dxh_tvd_set_coeffs must be called before dxh_tvd_upload"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const size_t n = (size_t)c->g.n0 * (size_t)c->g.L, ssz = img_dtype == DXH_F32 ? 4 : 8;
    void* raw = nullptr;   // the image as stored, in HBM (a cached block: exact-size re-use from call to call)
    const void* src = img;
    if (!is_device_ptr(src)) {
        DXH_HIP(pool_malloc(&raw, n * ssz));
        hipError_t e = hipMemcpyAsync(raw, src, n * ssz, hipMemcpyDefault, c->stream);
        if (e != hipSuccess) { pool_free(raw); set_error("upload failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
        src = raw;
    }
    // converted straight into the context's (pitched) image plane
    if (img_dtype == DXH_F64) hipLaunchKernelGGL((k_dense_to_plane<double, float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)src, (float*)c->u[0]);
    else hipLaunchKernelGGL((k_dense_to_plane<float, double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)src, (double*)c->u[0]);
    rc = hipGetLastError() == hipSuccess ? finish_upload(c, nullptr, nullptr, nullptr) : DXH_EHIP;
    if (raw) {
        (void)hipStreamSynchronize(c->stream);
        pool_free(raw);
    }
    return rc;
}

// Warm start of a numpy2 context (split_bregman_tvd.py:92-96 with float32 arrays): img float32; u0 float32 (the first sweep
// sums its neighbours in float32, like the image's) or float64, or NULL (= img); d0, b0 float32 (..., dim) arrays, kept as
// explicit planes for the first iteration.  Host pointers.
int dxh_tvd_upload_warm_np2(dxh_tvd_t h, const void* img, const void* u0, int u0_dtype, const void* d0, const void* b0) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && img && d0 && b0, "null argument");
    DXH_REQUIRE(c->np2, "the context is not in numpy2 promotion");
    DXH_REQUIRE(!u0 || u0_dtype == DXH_F32 || u0_dtype == DXH_F64, "u0 must be float32 or float64");
    // (host or device pointers: every copy below resolves its direction from the pointers)
    int rc = dxh_tvd_upload_typed(h, img, DXH_F32);   // of = omega img (float32 product), u = img, planes cleared
    if (rc) return rc;
    const int dim = c->desc.dim;
    const size_t n = (size_t)c->g.n0 * (size_t)c->g.L;
    const dim3 grid = elem_grid(c->g), blk(256);
    void* raw = nullptr;
    DXH_HIP(pool_malloc(&raw, n * (size_t)dim * 4 > n * 8 ? n * (size_t)dim * 4 : n * 8));
    auto fail = [&](int r) { (void)hipStreamSynchronize(c->stream); pool_free(raw); return r; };
    if (u0) {
        if (u0_dtype == DXH_F64) {
            if ((rc = h2d_plane(c, c->u[0], u0))) return fail(rc);
            c->np2_first = false;   // a float64 iterate: float64 neighbour sums from the first sweep on
        } else {
            if (hipMemcpyAsync(raw, u0, n * 4, hipMemcpyDefault, c->stream) != hipSuccess) { set_error("upload failed"); return fail(DXH_EHIP); }
            hipLaunchKernelGGL((k_dense_to_plane<float, double>), grid, blk, 0, c->stream, c->g, (const float*)raw, (double*)c->u[0]);
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        void** pl = pass == 0 ? c->dpl : c->bpl;
        for (int j = 0; j < dim; ++j)
            if ((rc = alloc_plane(c, &pl[j]))) return fail(rc);
        if (hipMemcpyAsync(raw, pass == 0 ? d0 : b0, n * (size_t)dim * 4, hipMemcpyDefault, c->stream) != hipSuccess) { set_error("upload failed"); return fail(DXH_EHIP); }
        if (dim == 2) hipLaunchKernelGGL((k_deinterleave_cvt<float, double, 2>), grid, blk, 0, c->stream, c->g, (const float*)raw, (double*)pl[0], (double*)pl[1], (double*)pl[2]);
        else hipLaunchKernelGGL((k_deinterleave_cvt<float, double, 3>), grid, blk, 0, c->stream, c->g, (const float*)raw, (double*)pl[0], (double*)pl[1], (double*)pl[2]);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { set_error("warm-start upload failed"); return fail(DXH_EHIP); }
    }
    c->from_db = true;
    return fail(DXH_OK);
}

// d, b of a numpy2 context as float32 (..., dim) host arrays (either may be NULL)
int dxh_tvd_download_split_np2(dxh_tvd_t h, void* d, void* b) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    DXH_REQUIRE(c->np2, "the context is not in numpy2 promotion");
    if (!c->uploaded) { set_error("nothing to download"); return DXH_ESTATE; }
    DXH_REQUIRE(!c->het || c->cp_kq, "numpy2 promotion: the threshold plane is missing");   // (array-valued omega / ell: the threshold mu / ell is cp_kq, as in k_rhs_np2_het)
    int rc = set_device(c);
    if (rc) return rc;
    const int dim = c->desc.dim;
    const size_t dense = (size_t)c->g.n0 * (size_t)c->g.L * (size_t)dim * 4;
    const dim3 grid = elem_grid(c->g), blk(256);
    void *dd = nullptr, *bb = nullptr;
    if (d) DXH_HIP(pool_malloc(&dd, dense));
    if (b && pool_malloc(&bb, dense) != hipSuccess) { if (dd) pool_free(dd); set_error("out of device memory"); return DXH_ENOMEM; }
    if (c->from_db || c->explicit_db) {   // explicit planes (a warm start before its first iteration; the adaptive schedule)
        for (int pass = 0; pass < 2; ++pass) {
            void* dst = pass == 0 ? dd : bb;
            if (!dst) continue;
            void** pl = pass == 0 ? c->dpl : c->bpl;
            if (dim == 2) hipLaunchKernelGGL((k_interleave_cvt<double, float, 2>), grid, blk, 0, c->stream, c->g, (const double*)pl[0], (const double*)pl[1], (const double*)pl[2], (float*)dst);
            else hipLaunchKernelGGL((k_interleave_cvt<double, float, 3>), grid, blk, 0, c->stream, c->g, (const double*)pl[0], (const double*)pl[1], (const double*)pl[2], (float*)dst);
        }
    } else {
        const Np2Coef k = np2_coef(c);
#define DXH_XNP2(DIM_) do { auto q = make_q<double, DIM_>(c, c->cur, c->cur);                                                        \
        if (k.kq_plane) hipLaunchKernelGGL((k_expand_np2<DIM_, true>), grid, blk, 0, c->stream, c->g, q, k, (float*)dd, (float*)bb);  \
        else hipLaunchKernelGGL((k_expand_np2<DIM_, false>), grid, blk, 0, c->stream, c->g, q, k, (float*)dd, (float*)bb); } while (0)
        if (dim == 2) DXH_XNP2(2); else DXH_XNP2(3);
#undef DXH_XNP2
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && d) e = hipMemcpyAsync(d, dd, dense, hipMemcpyDefault, c->stream);
    if (e == hipSuccess && b) e = hipMemcpyAsync(b, bb, dense, hipMemcpyDefault, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else (void)hipStreamSynchronize(c->stream);
    if (dd) pool_free(dd);
    if (bb) pool_free(bb);
    if (e != hipSuccess) { set_error("download failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
    return DXH_OK;
}

int dxh_tvd_download_typed(dxh_tvd_t h, void* u, int out_dtype) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && u, "null argument");
    DXH_REQUIRE(out_dtype == DXH_F32 || out_dtype == DXH_F64, "output dtype must be float32 or float64");
    if (out_dtype == c->desc.dtype) return dxh_tvd_download(h, u, nullptr, nullptr);
    if (!c->uploaded) { set_error("nothing to download"); return DXH_ESTATE; }
    int rc = set_device(c);
    if (rc) return rc;
    const size_t n = (size_t)c->g.n0 * (size_t)c->g.L, osz = out_dtype == DXH_F32 ? 4 : 8;
    HostPrefault pf;   // map the output pages while the device converts (and, after an asynchronous run, still iterates)
    pf.start(u, n * osz);
    if (c->signal) {   // a slab: a timed-out halo wait makes the state invalid (slab_check_fault)
        DXH_HIP(hipStreamSynchronize(c->stream));
        if ((rc = slab_check_fault(c))) return rc;
    }
    const bool to_device = is_device_ptr(u);
    void* conv = to_device ? u : nullptr;   // a device-resident result is written where it goes
    if (!conv) DXH_HIP(pool_malloc(&conv, n * osz));
    if (out_dtype == DXH_F32) hipLaunchKernelGGL((k_plane_to_dense<double, float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[c->cur], (float*)conv);
    else hipLaunchKernelGGL((k_plane_to_dense<float, double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->u[c->cur], (double*)conv);
    hipError_t e = hipGetLastError();
    pf.join();
    if (e == hipSuccess && !to_device) e = hipMemcpyAsync(u, conv, n * osz, hipMemcpyDefault, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (!to_device) pool_free(conv);
    if (e != hipSuccess) { set_error("download failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
    return DXH_OK;
}

int dxh_tvd_sync(dxh_tvd_t h) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c, "null context");
    int rc = set_device(c);
    if (rc) return rc;
    DXH_HIP(hipStreamSynchronize(c->stream));
    return slab_check_fault(c);
}

int dxh_tvd_slab_stats(dxh_tvd_t h, dxh_slab_stats* out) {
    Ctx* c = reinterpret_cast<Ctx*>(h);
    DXH_REQUIRE(c && out, "null argument");
    DXH_REQUIRE(out->struct_size == (int32_t)sizeof(dxh_slab_stats), "dxh_slab_stats size mismatch (%d vs %zu)",
                out->struct_size, sizeof(dxh_slab_stats));
    int rc = set_device(c);
    if (rc) return rc;
    DXH_HIP(hipStreamSynchronize(c->stream));
    if (c->alt) DXH_HIP(hipStreamSynchronize(c->alt));
    slab_fold_events(c, true);
    (void)slab_check_fault(c);  // refreshes the time-out count; the verdict belongs to sync / run / download
    out->wait_mode = c->wait_mode;
    out->launches = c->slab_launches;
    out->signalling_launches = c->split_launches;
    out->exchanges = c->slab_exchanges;
    out->exchanges_timed = c->slab_timed;
    out->wait_timeouts = c->slab_timeouts;
    out->wait_signal_ms = c->slab_wait_ms;
    out->exchange_ms = c->slab_xchg_ms;
    return DXH_OK;
}

// ---- one-shot helpers ----------------------------------------------------------------------------
// every float bit pattern: t_sqrt (the five-instruction chain with its rarely taken IEEE branch, dxh_common.h) against
// the compiler's correctly rounded sqrtf; NaN results compare equal to NaN
__global__ void __launch_bounds__(256) k_sqrt_check(unsigned long long* mismatches, unsigned long long* chain_inputs) {
    unsigned long long bad = 0, fast = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32);
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const float want = sqrtf(x), got = t_sqrt(x);
        const bool same = (want != want) ? (got != got) : (__float_as_uint(want) == __float_as_uint(got));
        bad += same ? 0 : 1;
        const unsigned u = (unsigned)b;
        fast += ((u - 0x0D800000u) < (0x7F800000u - 0x0D800000u) || u == 0u) ? 1 : 0;
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(chain_inputs, fast);
}

// every float in [2^-100, 2^100]: t_rcp_rn (v_rcp_f32 + one Newton step, dxh_common.h) against RN(1 / d)
__global__ void __launch_bounds__(256) k_rcp_check(unsigned long long* mismatches, unsigned long long* tested) {
    unsigned long long bad = 0, n = 0;
    const unsigned long long lo = 0x0D800000ull, hi = 0x71800000ull + 1ull;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < hi;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((unsigned)b);
        bad += (__float_as_uint(t_rcp_rn(d)) == __float_as_uint(1.0f / d)) ? 0 : 1;
        ++n;
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(tested, n);
}

// div_shared (three quotients through one reciprocal, dxh_common.h) against the IEEE division on pseudo-random operands:
// raw bit patterns (every class of float: zeros, denormals, NaN, inf, both signs), operands in the ranges images have, and
// operands straddling the screen's thresholds.  Statistical, unlike the two exhaustive checks above: the exactness of the
// refinement given RN(1/d) is Markstein's theorem, this guards the screening logic around it.
__device__ __forceinline__ unsigned mix32(unsigned long long v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
    return (unsigned)v;
}
__global__ void __launch_bounds__(256) k_div_shared_check(unsigned long long* mismatches, unsigned long long* tested) {
    unsigned long long bad = 0, n = 0;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long it = 0; it < 4096; ++it) {
        const unsigned long long base = (tid * 4096 + it) * 8;
        float x[3], d;
        const unsigned mode = mix32(base + 7) % 3;
        auto gen = [&](unsigned long long k, bool positive) -> float {
            const unsigned r = mix32(base + k);
            if (mode == 0) return __uint_as_float(positive ? (r & 0x7FFFFFFFu) : r);            // any bit pattern
            if (mode == 1) {                                                                     // image-like magnitudes
                const float m = 1.0f + (float)(r & 0xFFFFFF) * 0x1p-24f;
                const int e = (int)((r >> 24) & 31) - 20;
                const float v = ldexpf(m, e);
                return (!positive && (r >> 31)) ? -v : ((r >> 29) == 0 ? 0.0f * v : v);
            }
            const int e = ((r >> 24) & 1) ? 62 : -62;                                              // around the thresholds
            const float v = ldexpf(1.0f + (float)(r & 0xFFFFFF) * 0x1p-24f, e + (int)((r >> 25) & 3) - 2);
            return (!positive && (r >> 31)) ? -v : v;
        };
        for (int i = 0; i < 3; ++i) x[i] = gen(i, false);
        d = gen(3, true);
        float q[3];
        div_shared<float, 3>(x, d, q);
        for (int i = 0; i < 3; ++i) {
            // numerically equal (a zero quotient may come out as +0 where the IEEE division gives -0: the refinement adds
            // +0 and -0; no later operation of the algorithm can tell them apart) or both NaN
            const float want = x[i] / d;
            const bool same = (want != want) ? (q[i] != q[i]) : (want == q[i]);
            bad += same ? 0 : 1;
        }
        n += 3;
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(tested, n);
}

static int run_probe(int device, void (*kernel)(unsigned long long*, unsigned long long*), unsigned long long* a,
                     unsigned long long* b) {
    DXH_REQUIRE(a && b, "null argument");
    int ndev = 0;
    DXH_HIP(hipGetDeviceCount(&ndev));
    DXH_REQUIRE(device >= 0 && device < ndev, "device %d not available", device);
    DXH_HIP(hipSetDevice(device));
    DevTmp guard;
    DXH_HIP(hipMalloc(&guard.p, 16));
    unsigned long long* d = static_cast<unsigned long long*>(guard.p);
    DXH_HIP(hipMemset(d, 0, 16));
    hipLaunchKernelGGL(kernel, dim3(8192), dim3(256), 0, 0, d, d + 1);
    DXH_HIP(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    DXH_HIP(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    *a = h[0];
    *b = h[1];
    return DXH_OK;
}

int dxh_rcp_probe(int device, unsigned long long* mismatches, unsigned long long* tested) {
    int rc = run_probe(device, k_rcp_check, mismatches, tested);
    if (rc || *mismatches) return rc;
    unsigned long long bad2 = 0, n2 = 0;
    rc = run_probe(device, k_div_shared_check, &bad2, &n2);   // 8192 x 256 x 4096 operand sets
    *mismatches += bad2;
    return rc;
}

// The isotropic shrink factor (iso_s_fac_fast, dxh_common.h).  (a) every float x in [2^-60, 2^60): the rsq-seeded reciprocal
// against RN(1 / sqrt(x)), and s + 1e-18f == s;  (b) the whole routine against the IEEE expressions on pseudo-random (x, k):
// raw bit patterns, image-like ranges, k just below / above s, zeros.
__global__ void __launch_bounds__(256) k_isofac_rcp_check(unsigned long long* mismatches, unsigned long long* tested) {
    unsigned long long bad = 0, n = 0;
    const unsigned long long lo = __float_as_uint(kIsoLo), hi = __float_as_uint(kIsoHi);
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < hi;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const float y = __builtin_amdgcn_rsqf(fmaxf(x, kSqrtFastMin));
        const float g = x * y, h = 0.5f * y;
        const float s = fmaf(fmaf(-g, g, x), h, g);
        const float rho = fmaf(fmaf(-s, y, 1.0f), y, y);
        const bool excepted = (__float_as_uint(s) & 0x7FFFFFu) == 0x7FFFFFu;   // iso_s_fac_fast sends these to the IEEE branch
        const bool same = (excepted || __float_as_uint(rho) == __float_as_uint(1.0f / s)) &&
                          __float_as_uint(s + 1e-18f) == __float_as_uint(s) && __float_as_uint(s) == __float_as_uint(sqrtf(x));
        bad += same ? 0 : 1;
        ++n;
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(tested, n);
}
__global__ void __launch_bounds__(256) k_isofac_check(unsigned long long* mismatches, unsigned long long* tested) {
    unsigned long long bad = 0, n = 0;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long it = 0; it < 4096; ++it) {
        const unsigned long long base = (tid * 4096 + it) * 4;
        const unsigned mode = mix32(base + 3) % 4;
        float x, k;
        const unsigned r0 = mix32(base), r1 = mix32(base + 1);
        if (mode == 0) { x = __uint_as_float(r0); k = __uint_as_float(r1); }                               // any bit patterns
        else if (mode == 1) { x = ldexpf(1.0f + (r0 & 0x7FFFFF) * 0x1p-23f, (int)((r0 >> 23) % 40) - 30);   // image-like magnitudes
                              k = ldexpf(1.0f + (r1 & 0x7FFFFF) * 0x1p-23f, (int)((r1 >> 23) % 24) - 20); }
        else if (mode == 2) { x = ldexpf(1.0f + (r0 & 0x7FFFFF) * 0x1p-23f, (int)((r0 >> 23) % 30) - 20);   // k within a few ulps of s
                              k = __uint_as_float(__float_as_uint(sqrtf(x)) + (int)(r1 % 9) - 4); }
        else { x = (r0 & 1) ? 0.0f : __uint_as_float(r0 & 0x7FFFFFFFu); k = (r1 & 1) ? 0.0f : ldexpf(1.0f, (int)((r1 >> 1) % 200) - 100); }
        float s, fac;
        iso_s_fac<float>(x, k, s, fac);
        const float s_ref = t_sqrt(x);
        const float fac_ref = t_max0(s_ref - k) / (s_ref + 1e-18f);
        auto eq = [](float a, float b) { return (a != a) ? (b != b) : (__float_as_uint(a) == __float_as_uint(b)); };
        bad += (eq(s, s_ref) && eq(fac, fac_ref)) ? 0 : 1;
        ++n;
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(tested, n);
}

int dxh_isofac_probe(int device, unsigned long long* mismatches_rcp, unsigned long long* mismatches_fac, unsigned long long* tested) {
    DXH_REQUIRE(mismatches_rcp && mismatches_fac && tested, "null argument");
    unsigned long long n1 = 0, n2 = 0;
    int rc = run_probe(device, k_isofac_rcp_check, mismatches_rcp, &n1);
    if (rc) return rc;
    rc = run_probe(device, k_isofac_check, mismatches_fac, &n2);
    *tested = n1 + n2;
    return rc;
}

int dxh_sqrt_probe(int device, unsigned long long* mismatches, unsigned long long* chain_inputs) {
    return run_probe(device, k_sqrt_check, mismatches, chain_inputs);
}

int dxh_div3_probe(int device, float d0, float r0, float d1, float r1, unsigned* bad) {
    DXH_REQUIRE(bad, "null argument");
    DXH_HIP(hipSetDevice(device));
    hipStream_t stream = nullptr;
    DXH_HIP(cached_stream(device, &stream));
    return div3_run_check(stream, d0, r0 != 0.0f ? r0 : 1.0f / d0, d1, r1 != 0.0f ? r1 : 1.0f / d1, bad);
}

int dxh_jacobi_solve(int device, int dtype, int ndim, int dim, int nsum, int64_t n0, int64_t n1, int64_t n2,
                     const void* x0, const void* rhs, void* out, double mass, const void* mass_arr, double diffusion,
                     const void* diffusion_arr, double h, int maxiter, double tol, int* sweeps_done) {
    DXH_REQUIRE(x0 && out, "null argument");
    DXH_REQUIRE(rhs || !mass_arr, "rhs = NULL (rhs = mass * x0) needs a scalar mass");
    DXH_REQUIRE(maxiter >= 0, "maxiter must be >= 0");
    DXH_REQUIRE(h > 0.0, "h must be positive");
    dxh_tvd_desc d{};
    d.struct_size = sizeof d;
    d.device = device;
    d.dtype = dtype;
    d.ndim = ndim;
    d.dim = dim;
    d.nsum = nsum;
    d.n0 = n0; d.n1 = n1; d.n2 = n2;
    d.top_is_boundary = d.bottom_is_boundary = 1;
    dxh_tvd_t hctx = nullptr;
    int rc = dxh_tvd_create(&hctx, &d);
    if (rc) return rc;
    Ctx* c = reinterpret_cast<Ctx*>(hctx);
    HostPrefault pf;  // map the output pages while the device works
    {
        const size_t nb = (size_t)n0 * (size_t)n1 * (size_t)n2 * (dtype == DXH_F32 ? 4 : 8);
        pf.start(out, nb, {{x0, nb}, {rhs, rhs ? nb : 0}});
    }
    auto body = [&]() -> int {
        int r;
        // mass plays omega, diffusion plays ell; mu is unused (kq not needed)
        r = dtype == DXH_F32 ? set_coeffs_t<float>(c, 0.0, nullptr, mass, mass_arr, 1, diffusion, diffusion_arr, h, false)
                             : set_coeffs_t<double>(c, 0.0, nullptr, mass, mass_arr, 1, diffusion, diffusion_arr, h, false);
        if (r) return r;
        if ((r = alloc_plane(c, &c->rs))) return r;
        if ((r = h2d_plane(c, c->u[0], x0))) return r;
        if (rhs) {
            if ((r = h2d_plane(c, c->u[1], rhs))) return r;
        } else {
            // H1 regularisation: rhs = omega * img (h1_regularization.py:73), the scalar rounded to the image dtype first
            // as NumPy does for a Python float
            if (dtype == DXH_F32)
                hipLaunchKernelGGL((k_scale<float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->u[0],
                                   (float)mass, (const float*)nullptr, (float*)c->u[1]);
            else
                hipLaunchKernelGGL((k_scale<double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[0],
                                   mass, (const double*)nullptr, (double*)c->u[1]);
            DXH_HIP(hipGetLastError());
        }
        // rs = rhs / diag
        if (dtype == DXH_F32)
            hipLaunchKernelGGL((k_divide<float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->u[1],
                               (float)c->c_diag, (const float*)c->cp_diag, (float*)c->rs);
        else
            hipLaunchKernelGGL((k_divide<double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[1],
                               c->c_diag, (const double*)c->cp_diag, (double*)c->rs);
        DXH_HIP(hipGetLastError());
        void* tmp = nullptr;
        if (maxiter > 1) DXH_HIP(hipMalloc(&tmp, c->plane_bytes));
        c->cur = 0;
        int64_t launches = 0;
        r = dtype == DXH_F32 ? jacobi_sweeps<float>(c, maxiter, tol, tmp, &launches, sweeps_done)
                             : jacobi_sweeps<double>(c, maxiter, tol, tmp, &launches, sweeps_done);
        pf.join();
        if (!r) r = d2h_plane(c, out, c->u[1]);
        if (!r && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("stream sync failed"); r = DXH_EHIP; }
        if (tmp) (void)hipFree(tmp);
        return r;
    };
    rc = body();
    dxh_tvd_destroy(hctx);
    return rc;
}

// float32 rhs -> rs = float32(rhs / diag) in the double plane (jacobi.py:107 on a float32 array with a Python-float diagonal)
static __global__ void __launch_bounds__(256) k_divide_np2(Geom g, const double* __restrict__ a, float diag, double* __restrict__ out) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    out[e.off] = (double)((float)a[e.off] / diag);
}

int dxh_jacobi_solve_np2(int device, int ndim, int dim, int nsum, int64_t n0, int64_t n1, int64_t n2, const void* x0,
                         const void* rhs, void* out, int out_dtype, double mass, double diffusion, double h, int maxiter,
                         double tol, int* sweeps_done) {
    DXH_REQUIRE(x0 && out, "null argument");
    DXH_REQUIRE(maxiter >= 0, "maxiter must be >= 0");
    DXH_REQUIRE(h > 0.0, "h must be positive");
    DXH_REQUIRE(out_dtype == DXH_F32 || out_dtype == DXH_F64, "output dtype must be float32 or float64");
    dxh_tvd_desc d{};
    d.struct_size = sizeof d;
    d.device = device;
    d.dtype = DXH_F64;
    d.ndim = ndim;
    d.dim = dim;
    d.nsum = nsum;
    d.n0 = n0; d.n1 = n1; d.n2 = n2;
    d.top_is_boundary = d.bottom_is_boundary = 1;
    dxh_tvd_t hctx = nullptr;
    int rc = dxh_tvd_create(&hctx, &d);
    if (rc) return rc;
    Ctx* c = reinterpret_cast<Ctx*>(hctx);
    const size_t n = (size_t)n0 * (size_t)n1 * (size_t)n2;
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, n * (out_dtype == DXH_F32 ? 4 : 8), {{x0, n * 4}, {rhs, rhs ? n * 4 : 0}});
    void *raw0 = nullptr, *raw1 = nullptr, *conv = nullptr, *tmp = nullptr;
    auto body = [&]() -> int {
        int r = set_coeffs_t<double>(c, 0.0, nullptr, mass, nullptr, 1, diffusion, nullptr, h, false);
        if (r) return r;
        if ((r = alloc_plane(c, &c->rs))) return r;
        const dim3 grid = elem_grid(c->g), blk(256);
        // float32 operands into the double planes (exact)
        auto up = [&](const void* src, void** raw, void* plane) -> int {
            if (!is_device_ptr(src)) {
                DXH_HIP(pool_malloc(raw, n * 4));
                DXH_HIP(hipMemcpyAsync(*raw, src, n * 4, hipMemcpyDefault, c->stream));
                src = *raw;
            }
            hipLaunchKernelGGL((k_dense_to_plane<float, double>), grid, blk, 0, c->stream, c->g, (const float*)src, (double*)plane);
            DXH_HIP(hipGetLastError());
            return DXH_OK;
        };
        if ((r = up(x0, &raw0, c->u[0]))) return r;
        if (rhs) {
            if ((r = up(rhs, &raw1, c->u[1]))) return r;
        } else {
            // H1 regularisation: rhs = omega * img (h1_regularization.py:73), a float32 product
            hipLaunchKernelGGL(k_scale_np2, grid, blk, 0, c->stream, c->g, (const double*)c->u[0], (float)mass, (double*)c->u[1]);
        }
        hipLaunchKernelGGL(k_divide_np2, grid, blk, 0, c->stream, c->g, (const double*)c->u[1], (float)c->c_diag, (double*)c->rs);
        DXH_HIP(hipGetLastError());
        if (maxiter > 1) DXH_HIP(hipMalloc(&tmp, c->plane_bytes));
        c->cur = 0;
        int64_t launches = 0;
        r = jacobi_sweeps<double>(c, maxiter, tol, tmp, &launches, sweeps_done, /*first_f32=*/true);
        pf.join();
        if (r) return r;
        if (out_dtype == DXH_F64) {
            r = d2h_plane(c, out, c->u[1]);
        } else {   // convert_dtype(out, float32) (h1_regularization.py:76): rounded on the device
            void* dst = out;
            if (!is_device_ptr(out)) {
                DXH_HIP(pool_malloc(&conv, n * 4));
                dst = conv;
            }
            hipLaunchKernelGGL((k_plane_to_dense<double, float>), grid, blk, 0, c->stream, c->g, (const double*)c->u[1], (float*)dst);
            DXH_HIP(hipGetLastError());
            if (conv) DXH_HIP(hipMemcpyAsync(out, conv, n * 4, hipMemcpyDefault, c->stream));
        }
        if (!r && hipStreamSynchronize(c->stream) != hipSuccess) { set_error("stream sync failed"); r = DXH_EHIP; }
        return r;
    };
    rc = body();
    (void)hipStreamSynchronize(c->stream);
    if (raw0) pool_free(raw0);
    if (raw1) pool_free(raw1);
    if (conv) pool_free(conv);
    if (tmp) (void)hipFree(tmp);
    dxh_tvd_destroy(hctx);
    return rc;
}

int dxh_diff(int device, int dtype, int64_t n0, int64_t n1, int64_t n2, int axis, int mode, double h, const void* in,
             void* out) {
    DXH_REQUIRE(in && out, "null argument");
    DXH_REQUIRE(axis >= 0 && axis < 3, "axis must be 0, 1 or 2");
    DXH_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (backward) or 1 (forward)");
    dxh_tvd_desc d{};
    d.struct_size = sizeof d;
    d.device = device;
    d.dtype = dtype;
    d.ndim = 3;
    d.dim = 2;
    d.nsum = DXH_NSUM_ALL_AXES;
    d.n0 = n0; d.n1 = n1; d.n2 = n2;
    d.top_is_boundary = d.bottom_is_boundary = 1;
    dxh_tvd_t hctx = nullptr;
    int rc = dxh_tvd_create(&hctx, &d);
    if (rc) return rc;
    Ctx* c = reinterpret_cast<Ctx*>(hctx);
    auto body = [&]() -> int {
        int r;
        if ((r = h2d_plane(c, c->u[0], in))) return r;
        if (dtype == DXH_F32)
            hipLaunchKernelGGL((k_diff<float>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const float*)c->u[0], axis,
                               mode, (float)h, h > 0.0 ? 1 : 0, (float*)c->u[1]);
        else
            hipLaunchKernelGGL((k_diff<double>), elem_grid(c->g), dim3(256), 0, c->stream, c->g, (const double*)c->u[0],
                               axis, mode, h, h > 0.0 ? 1 : 0, (double*)c->u[1]);
        DXH_HIP(hipGetLastError());
        if ((r = d2h_plane(c, out, c->u[1]))) return r;
        DXH_HIP(hipStreamSynchronize(c->stream));
        return DXH_OK;
    };
    rc = body();
    dxh_tvd_destroy(hctx);
    return rc;
}

}  // extern "C"
