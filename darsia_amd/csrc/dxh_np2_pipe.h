// K outer iterations per pass in the reference's OWN dtype flow for a float32 image under NumPy >= 2 ("numpy2" promotion;
// restoration/split_bregman_tvd.py:84-101,129-146, utils/linear_solvers/jacobi.py:101-114) -- the product's DEFAULT flow.
//
// Same decomposition as k_fused_multi (dxh_kernels.h: a wave owns a column strip of 64 lanes x 16 B and a chunk of rows and
// marches down the rows with a K-stage register pipeline; column / channel neighbours by DPP wave shifts, no barriers), but
// every quantity lives in the type the reference gives it, so that nothing is converted back and forth:
//
//   float64 (registers: double rings; HBM: double planes)   u, q_j = dub_j, the neighbour sum N(u), N(u) / diag_scaled
//   float32 (registers: FLOAT rings; LDS: float rows)       d_j, b_j, t_j = ell (b_j - d_j), omega*img, the right-hand side,
//                                                           rhs / diag
//
// Per element and stage: the two float64 shrinkages (clamp form) with their roundings to float32 (d, then b = q - d), the
// float32 right-hand side (the column difference of t_1 with the lane shift folded into the subtraction on interior
// strips), rhs / diag as a float32 division by a wave-invariant divisor through the host-rounded reciprocal (three
// operations where an exhaustive device check of all 2^23 mantissas says that is exact for this diag, five otherwise;
// dxh_common.h: div_by_const), N(u) / diag_scaled as a float64 division through the reciprocal with two residual
// corrections, and the new q_j in float64.  Stage 0 of the first launch after an upload sums N(u) in float32 (u still is
// the float32 image, exactly held by the doubles).
//
// The first version of this flow (round 3) ran inside the generic Pipe<double, ...>: the float32-valued state sat in double
// rings and was converted on every use (8 v_cvt per element and stage), rhs / diag was the compiler's IEEE division
// sequence, and the kernel needed 176 VGPRs (2 waves per SIMD): 1.47 - 1.55 ms per K = 4 launch at 4000 x 8000 x 3
// (profiles/r03_sq_counters.csv: SQ_INSTS_VALU 7.64e8).
#pragma once

#include "dxh_kernels.h"

// DXH_NP2_WAVES (dxh_multi.h): waves per SIMD the pipeline is compiled for (register cap 168 / 256)
#ifndef DXH_NP2_INC_REGS
#define DXH_NP2_INC_REGS 0   // eps instantiation: the running sums of squared increments in registers (measured: the allocator spills them to
                             // scratch at 168 registers -- a load / add / store chain per stage) or (0) in LDS through ds_add_f64, which returns nothing
#endif
#ifndef DXH_NP2_PFD
#define DXH_NP2_PFD 1     // prefetch distance in row steps (2: staging registers, needs DXH_NP2_WAVES = 2)
#endif

namespace dxh {

// F: the type of the quantities that are float32 in the reference's mixed flow (d, b, t, omega*img, the right-hand side) --
// float for a numpy2 context; double for a float64 IMAGE, whose flow is float64 throughout: the same pipeline then serves
// the native float64 contexts (no conversions at all; both divisions through the reciprocal with two corrections).
template <int K, typename F>
struct Np2State {
    static constexpr int N = 2;
    double U[K + 1][4][N];   // rows of level l, slot = (row - rs) & 3 (level K: parity)
    F t0[K][2][N];           // ell (b_0 - d_0) of each stage's current / previous row
    F b0[K][2][N];
    double q0l[2][N], q1l[2][N];  // loaded level-0 q rows (current / prefetched)
    F of0[N];                // omega*img of stage 0's row
    double pf[2][4][N];      // prefetch distance 2: rows in flight two steps ahead (u, -, q0, q1)
    F pff[2][N];             // ... and omega*img
    double kqv[1][N];        // KQV: the float64 threshold mu / ell of stage 0's current row (the later stages: LDS rows)
    double kqn[N];
    double inc[K];           // eps instantiation (DXH_NP2_INC_REGS): this lane's running sum of squared increments per stage
};

// LDS of one wave.  Rows of the levels 1 .. K - 1 go from their producer (stage l - 1) to their consumer (stage l, one step
// later) through LDS, and the consumer reads its column (+-C) and channel (+-1) neighbours straight out of the row at
// per-lane addresses -- no v_mov_b32_dpp pairs per double and, because the image-border / channel-border replicate rules are
// folded into the (loop-invariant) addresses, no v_bfi selects either: 14 of the 71 VALU instructions per element and stage of
// the first version.  A wave's LDS operations execute in order, so no barrier is involved.
//   U rows     K - 1 slots (one per level; two without DXH_NP2_PREFETCH), each the row's EVEN elements (element 0 of every lane: 64 x 8 B) followed by its ODD elements,
//              16 B of zeros in front of each half and behind the last (lanes that look past the wave's ends read zeros, as
//              bound_ctrl DPP gives them).  De-interleaved so that a wave's 8-byte reads at consecutive lanes are consecutive
//              in LDS: conflict-free (a 16-byte lane stride is a 2-way bank conflict for ds_read_b64: measured, 5.4e7
//              conflict cycles per launch).  Single-buffered: stage s issues stage s + 1's reads of the row before it writes the
//              next one (round 4 started with two slots per level; the eps instantiation's 8 KB of running sums then left two
//              blocks per CU instead of three: +25 % per launch)
//   b_1 rows   K - 1 slots of 64 lanes x 2 F       float32 (F) b_1 of the same rows (the consumer forms q_1 = D_1 u + b_1 itself)
//   omega*img  4 slots of 64 lanes x 2 F          rows r .. r - 3, written by stage 0
// K > 4 (Np2Lds::FWD; the row loop is unrolled by four, so a ring of K rows has no compile-time slot): omega*img travels down
// the stages WITH b_1 -- one row of 64 lanes x 4 F per level -- and there is ONE slot per level: every stage's row of the
// previous step is fetched at the start of the step, before any stage writes (4 (K - 1) registers held through the step)
template <int K, typename F, bool KQV = false>
struct Np2Lds {
    static constexpr int PAD = 16;
    static constexpr int UHALF = PAD + kWave * 8;           // one parity of a row: pad + 64 x 8 B (the next half's pad follows)
    static constexpr int USLOT = 2 * UHALF;
    // DXH_NP2_PREFETCH: ONE slot per level -- stage s fetches stage s + 1's operands (the row it wrote one step ago) before it
    // writes the new row, and a wave's LDS operations execute in order; without the prefetch two (parity of the reading step)
    static constexpr int SLOTS_PER_LEVEL = DXH_NP2_PREFETCH ? 1 : 2;
    static constexpr int NU = SLOTS_PER_LEVEL * (K - 1);
    static constexpr int U_BYTES = NU * USLOT + PAD;
    static constexpr int B1_OFF = U_BYTES;
    static constexpr bool FWD = K > 4;
    static constexpr int FSLOT = kWave * (FWD ? 4 : 2) * (int)sizeof(F);   // (float64 images: 16 B per lane, 16.6 KB per wave, 2 blocks per CU)
    static constexpr int OF_OFF = B1_OFF + (FWD ? (K - 1) : NU) * FSLOT;
    // per wave: K = 4 float 6352 + 3072 + 2048 = 11472; K = 6 float (FWD) 10576 + 5120 = 15696 (two blocks of four waves per CU)
    // KQV: the threshold pair of a row travels down the stages through one more row per level (64 lanes x 16 B) instead of a
    // register delay line of K x 4 registers (three waves per SIMD without spills)
    static constexpr int KQ_OFF = OF_OFF + (FWD ? 0 : 4 * FSLOT);
    static constexpr int KQSLOT = kWave * 16;
    static constexpr int BYTES = KQ_OFF + (KQV ? (K - 1) * KQSLOT : 0);
    // byte offset, within U slot 0, of element e of the wave's window (e = 2 lane + j; e in [-4, 132) stays inside the pads)
    static __device__ __forceinline__ int elem_off(int e) {
        const int par = e & 1;
        return PAD + par * UHALF + ((e - par) >> 1) * 8;
    }
    static_assert(U_BYTES % 16 == 0 && USLOT % 16 == 0, "16-byte rows");
    static_assert(PAD >= 16, "C <= 4: a lane looks at most 4 elements (2 lanes x 8 B per half) past the wave's ends");
};

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)p;
    // (no memory clobber: the zeroing stores and the final reads of these slots are volatile accesses, which keep their order
    // relative to a volatile asm; everything else may move across it)
    asm volatile("ds_add_f64 %0, %1" : : "v"(addr), "v"(v));
}

template <int DELTA, typename F>
__device__ __forceinline__ void shifted2f(const F (&v)[2], F (&w)[2]) {
    w[0] = shifted_elem<DELTA, 0, true>(v);
    w[1] = shifted_elem<DELTA, 1, true>(v);
}

// x - clamp(x, -k, k) pieces as single instructions: fmax / fmin of the compiler canonicalise (v_max_f64 x, x, x) every
// operand they cannot prove quiet -- one more double-rate instruction per value that comes out of memory or LDS.  The
// hardware instructions themselves return the other operand for a NaN, which is what fmax / fmin do.
__device__ __forceinline__ double np2_clamp(double x, double negk, double k) {
    double t, r;
    asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(x), "v"(negk));
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(t), "v"(k));
    return r;
}
__device__ __forceinline__ float np2_absmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double np2_absmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// d = float32(shrink(q, k)), b = float32(q - d) with the clamp form (scalar threshold k >= 0, np2_split<true>)
__device__ __forceinline__ void np2_split_fast(double q, double negk, double k, float& d, float& b) {
    d = (float)(q - np2_clamp(q, negk, k));
    b = (float)(q - (double)d);
}
// float64 image: d = q - clamp(q, -k, k), b = q - d (split_vars<double, 2, false, true>)
__device__ __forceinline__ void np2_split_fast(double q, double negk, double k, double& d, double& b) {
    d = q - np2_clamp(q, negk, k);
    b = q - d;
}
template <typename F> struct Np2Vec2;
template <> struct Np2Vec2<float> { using type = float2; };
template <> struct Np2Vec2<double> { using type = double2; };

template <int C, bool COUPLED, int K, bool NTS, bool EDGE, bool NORM, bool KQV, typename F = float>
struct PipeNp2 {
    static constexpr int N = 2;
    static constexpr bool F32 = std::is_same<F, float>::value;
    static_assert(F32 || !KQV, "float64 images: scalar threshold only (the threshold plane keeps the generic pipeline)");
    static constexpr int PFD = KQV ? 1 : (F32 ? (K > 4 ? DXH_NP2_PFD_K6 : DXH_NP2_PFD) : DXH_F64P_PFD);   // the per-element threshold is prefetched at distance 1
    using L = Np2Lds<K, F, KQV>;
    using F2 = typename Np2Vec2<F>::type;
    const F* of_p;       // omega * img: the float32 plane of a numpy2 context (FusedArgs::of32) / the float64 plane of a float64 image
    const double* u_in;
    const double* q0_in;
    const double* q1_in;
    const double* kq_p;
    double* u_out;
    double* q0_out;
    double* q1_out;
    int64_t P, nrow, R0, R1;
    int nrow32, R0_32, R1_32;
    double dsc, rdsc, kq, negkq;
    F ell_f, diag_f, rdiag_f;
    bool div3f;         // rhs / diag: one residual correction is exact for every numerator (k_div3_check on this diag)
    bool np2_first;
    bool store_lane;
    uint32_t edgeL[N], edgeR[N], ch0[N], chLast[N];
    bool inL[N];
    uint32_t voff;
    char* lw[N];        // this lane's own elements in U slot 0 of the wave's LDS region
    char* aL[N];        // where element j's left / right column neighbour and left / right channel neighbour live in U slot 0
    char* aR[N];        // (the element itself where the reference replicates: image border, first / last channel)
    char* aCL[N];
    char* aCR[N];
    F* lb1;             // this lane's pair in b_1 slot 0
    F* lof;             // this lane's pair in omega*img slot 0
    char* lkq;          // KQV: this lane's 16 B in threshold-row slot 0
    double* linc;       // eps instantiation: running sums of squared increments, [stage][lane] in LDS

    __device__ __forceinline__ void u_put(int slot, const double (&v)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) *reinterpret_cast<double*>(lw[j] + slot * L::USLOT) = v[j];
    }
    __device__ __forceinline__ double u_get(const char* a, int slot) const {
        return *reinterpret_cast<const double*>(a + slot * L::USLOT);
    }
    __device__ __forceinline__ void f_put(F* base, int slot, const F (&v)[N]) const {
        F2 t;
        t.x = v[0];
        t.y = v[1];
        *reinterpret_cast<F2*>(base + slot * (kWave * N)) = t;
    }
    __device__ __forceinline__ void f_get(const F* base, int slot, F (&v)[N]) const {
        const F2 t = *reinterpret_cast<const F2*>(base + slot * (kWave * N));
        v[0] = t.x;
        v[1] = t.y;
    }
    // FWD rows: (b_1 pair, omega*img pair) of one level, 4 F per lane
    __device__ __forceinline__ void fw_put(int slot, const F (&b1)[N], const F (&of)[N]) const {
        F2* p = reinterpret_cast<F2*>(lb1 + slot * (kWave * 4));
        F2 t, o;
        t.x = b1[0];
        t.y = b1[1];
        o.x = of[0];
        o.y = of[1];
        p[0] = t;
        p[1] = o;
    }
    __device__ __forceinline__ void fw_get(int slot, F (&b1)[N], F (&of)[N]) const {
        const F2* p = reinterpret_cast<const F2*>(lb1 + slot * (kWave * 4));
        const F2 t = p[0], o = p[1];
        b1[0] = t.x;
        b1[1] = t.y;
        of[0] = o.x;
        of[1] = o.y;
    }
    __device__ __forceinline__ int64_t clampr(int64_t r) const { return r < 0 ? 0 : (r > nrow - 1 ? nrow - 1 : r); }
    __device__ __forceinline__ const double* at(const double* plane, int64_t row) const {
        return reinterpret_cast<const double*>(reinterpret_cast<const char*>(plane + row * P) + voff);
    }
    __device__ __forceinline__ double* at(double* plane, int64_t row) const {
        return reinterpret_cast<double*>(reinterpret_cast<char*>(plane + row * P) + voff);
    }
    // the lane's two values of a row of the omega * img plane (same pitch in elements; float32: half the bytes)
    __device__ __forceinline__ void load_of(int64_t row, F (&v)[N]) const {
        const F2 t = *reinterpret_cast<const F2*>(reinterpret_cast<const char*>(of_p + row * P) + (F32 ? (voff >> 1) : voff));
        v[0] = t.x;
        v[1] = t.y;
    }
    __device__ __forceinline__ void split(double q, double kqj, F& d, F& b) const {
        if constexpr (KQV) np2_split_fast(q, -kqj, kqj, d, b);
        else np2_split_fast(q, negkq, kq, d, b);
    }

    // un = float64( float32(rhs) / float32(diag) ) + N(u) / diag_scaled        (jacobi.py:107,110-114)
    __device__ __forceinline__ void divide(const F (&rhs)[N], const double (&acc)[N], double (&un)[N]) const {
        F rs[N];
#pragma unroll
        for (int j = 0; j < N; ++j) rs[j] = div_by_const3(rhs[j], diag_f, rdiag_f);
        if (!F32 || !div3f) {   // (float64: no exhaustive check exists -- always both corrections)
            DXH_UNIFORM_BRANCH();
#pragma unroll
            for (int j = 0; j < N; ++j) rs[j] = div_refine(rs[j], rhs[j], diag_f, rdiag_f);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) un[j] = (double)rs[j] + div_by_const<double>(acc[j], dsc, rdsc);
        // numerators outside the reciprocal chains' proven range (non-zero and tiny): the IEEE divisions.  One cheap
        // screen per lane first (exact zeros trip it and are sorted out by the precise test)
        const F lo_f = np2_absmin(rhs[0], rhs[1]);
        const double lo_d = np2_absmin(acc[0], acc[1]);
        if (__builtin_expect(__any(lo_f < DivSafe<F>::tiny || lo_d < DivSafe<double>::tiny), 0)) {
            bool unsafe = false;
#pragma unroll
            for (int j = 0; j < N; ++j) unsafe = unsafe || div_unsafe(rhs[j]) || div_unsafe(acc[j]);
            if (__any(unsafe)) {
#pragma unroll
                for (int j = 0; j < N; ++j) un[j] = (double)(rhs[j] / diag_f) + acc[j] / dsc;
            }
        }
    }

    // ACT: stages 0 .. ACT - 1 run (the first steps of a chunk, fused_np2_body)
    template <int PH, int ACT = K>
    __device__ __forceinline__ void step(Np2State<K, F>& st, const int64_t r) const {
        F frn[N];
        if constexpr (PFD == 1) {
            const int64_t rn = clampr(r + 1);
            load_vec(at(u_in, clampr(r + 2)), st.U[0][(PH + 2) & 3]);
            load_of(rn, frn);
            load_vec(at(q0_in, rn), st.q0l[(PH + 1) & 1]);
            load_vec(at(q1_in, rn), st.q1l[(PH + 1) & 1]);
            if constexpr (KQV) load_vec(at(kq_p, rn), st.kqn);
        } else {
            const int64_t rn = clampr(r + 2);
            load_vec(at(u_in, clampr(r + 3)), st.pf[PH & 1][0]);
            load_of(rn, st.pff[PH & 1]);
            load_vec(at(q0_in, rn), st.pf[PH & 1][2]);
            load_vec(at(q1_in, rn), st.pf[PH & 1][3]);
        }
        F hb1[L::FWD ? K : 1][N], hof[L::FWD ? K : 1][N];
        if constexpr (L::FWD) {
            // every stage's (b_1, omega*img) row of the previous step comes out of LDS before any stage of this step writes
            static_for<1, ACT>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                fw_get(s - 1, hb1[s], hof[s]);
            });
        } else if constexpr (K > 1) {
            f_put(lof, PH & 3, st.of0);   // omega*img of row r for the later stages
        }
        double q0h[N];   // Q0 of the next level at the next stage's row, handed from stage to stage
        // DXH_NP2_PREFETCH: stage s issues the LDS reads of stage s + 1 (rows written in the PREVIOUS step, so they may be
        // fetched at any time) before its own tail -- the divisions' consumers, the q updates, the stores -- instead of
        // stage s + 1 waiting for them at its head
        double pnl[N], pnr[N], pcl[N], pcr[N];
        F pb1[N], pof[N];
        double pkq[N];
        static_assert(!KQV || DXH_NP2_PREFETCH, "the threshold rows are single-buffered: the prefetch order is what makes that safe");
        auto fetch = [&](auto sn) {
            constexpr int s1 = decltype(sn)::value;           // the stage whose operands are fetched
            constexpr int slot1 = (s1 - 1) * L::SLOTS_PER_LEVEL + (L::SLOTS_PER_LEVEL == 2 ? (PH & 1) : 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                pnl[j] = u_get(aL[j], slot1);
                pnr[j] = u_get(aR[j], slot1);
                if constexpr (COUPLED) {
                    pcl[j] = u_get(aCL[j], slot1);
                    pcr[j] = u_get(aCR[j], slot1);
                }
            }
            if constexpr (!L::FWD) {
                f_get(lb1, slot1, pb1);
                f_get(lof, (PH - s1) & 3, pof);
            }
            if constexpr (KQV) {
                const double2 t = *reinterpret_cast<const double2*>(lkq + (s1 - 1) * L::KQSLOT);
                pkq[0] = t.x;
                pkq[1] = t.y;
            }
        };
        static_for<0, ACT>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int c = (PH - s) & 3;
            constexpr int cm = (c + 3) & 3, cp = (c + 1) & 3;
            constexpr int cur = PH & 1, prv = cur ^ 1;
            constexpr bool last = (s == K - 1);
            const int64_t p = r - s;
            const int p32 = (int)p;
            double(&um)[N] = st.U[s][cm];
            double(&uc)[N] = st.U[s][c];
            double(&up)[N] = st.U[s][cp];
            // KQV: this stage's thresholds -- stage 0's row came from HBM, the later stages' out of the row the stage before wrote
            double kqs[N];
            if constexpr (KQV) {
#pragma unroll
                for (int j = 0; j < N; ++j) kqs[j] = (s == 0) ? st.kqv[0][j] : pkq[j];
            }
            F(&t0m)[N] = st.t0[s][prv];
            F(&t0c)[N] = st.t0[s][cur];
            F(&b0m)[N] = st.b0[s][prv];
            F(&b0c)[N] = st.b0[s][cur];
            double(&un)[N] = last ? st.U[K][cur] : st.U[s + 1][c];
            const double(&unm)[N] = last ? st.U[K][prv] : st.U[s + 1][cm];

            // neighbours of this stage's row: nl / nr (column -+C), cl / cr (channel -+1), replicate rules applied
            double nl[N], nr[N], cl[N], cr[N], q1c[N];
            F ofv[N];
            if constexpr (s == 0) {
                // level 0 comes from HBM into registers: lane shifts by DPP, selects for the replicate rules
                double ul[N], ur[N], ucl[N], ucr[N];
                shifted<-C, true>(uc, ul);
                shifted<+C, true>(uc, ur);
                if constexpr (COUPLED) {
                    shifted<-1, true>(uc, ucl);
                    shifted<+1, true>(uc, ucr);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    nl[j] = EDGE ? sel_bits(edgeL[j], uc[j], ul[j]) : ul[j];
                    nr[j] = EDGE ? sel_bits(edgeR[j], uc[j], ur[j]) : ur[j];
                    if constexpr (COUPLED) {
                        cl[j] = sel_bits(ch0[j], uc[j], ucl[j]);
                        cr[j] = sel_bits(chLast[j], uc[j], ucr[j]);
                    }
                    q1c[j] = st.q1l[cur][j];
                    ofv[j] = st.of0[j];
                }
            } else {
                // levels >= 1: out of the row stage s - 1 put into LDS one step ago; q_1 = D_1 u + b_1 is formed here
                constexpr int slot = (s - 1) * L::SLOTS_PER_LEVEL + (L::SLOTS_PER_LEVEL == 2 ? cur : 0);
                F b1p[N];
                if constexpr (DXH_NP2_PREFETCH) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        nl[j] = pnl[j];
                        nr[j] = pnr[j];
                        if constexpr (COUPLED) {
                            cl[j] = pcl[j];
                            cr[j] = pcr[j];
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        nl[j] = u_get(aL[j], slot);
                        nr[j] = u_get(aR[j], slot);
                        if constexpr (COUPLED) {
                            cl[j] = u_get(aCL[j], slot);
                            cr[j] = u_get(aCR[j], slot);
                        }
                    }
                }
                if constexpr (L::FWD) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        b1p[j] = hb1[s][j];
                        ofv[j] = hof[s][j];
                    }
                } else if constexpr (DXH_NP2_PREFETCH) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        b1p[j] = pb1[j];
                        ofv[j] = pof[j];
                    }
                } else {
                    f_get(lb1, slot, b1p);
                    f_get(lof, c, ofv);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) q1c[j] = (nr[j] - uc[j]) + (double)b1p[j];
            }

            // d_j = float32(shrink(q_j)), b_j = float32(q_j - d_j), t_j = ell (b_j - d_j) in float32   (:84-89,143-146)
            F b1c[N], t1c[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double kqj = KQV ? kqs[j] : kq;
                F d0f, b0f, d1f, b1f;
                split((s == 0) ? st.q0l[cur][j] : q0h[j], kqj, d0f, b0f);
                split(q1c[j], kqj, d1f, b1f);
                t0c[j] = ell_f * (b0f - d0f);
                t1c[j] = ell_f * (b1f - d1f);
                b0c[j] = b0f;
                b1c[j] = b1f;
            }
            if constexpr (KQV) {
                // wave-uniform: a negative or NaN threshold in this row: np.maximum(|x| - k, 0) * np.sign(x) as written
                if (__any(!(kqs[0] >= 0.0) || !(kqs[1] >= 0.0))) {
                    DXH_UNIFORM_BRANCH();
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        float d0f, b0f, d1f, b1f;
                        np2_split<false>((s == 0) ? st.q0l[cur][j] : q0h[j], kqs[j], d0f, b0f);
                        np2_split<false>(q1c[j], kqs[j], d1f, b1f);
                        t0c[j] = ell_f * (b0f - d0f);
                        t1c[j] = ell_f * (b1f - d1f);
                        b0c[j] = b0f;
                        b1c[j] = b1f;
                    }
                }
            }
            if (p == 0) {  // replicate the row above the image
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    um[j] = uc[j];
                    t0m[j] = t0c[j];
                }
            }
            F rhs[N];
            double acc[N];
            if constexpr (EDGE) {
                F t1l[N];
                shifted2f<-C>(t1c, t1l);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const F rf = ofv[j] + (t0c[j] - t0m[j]);
                    rhs[j] = rf + (t1c[j] - sel_bits(edgeL[j], t1c[j], t1l[j]));
                }
            } else {
                // interior strips: the lane shift rides on the subtraction where the neighbour is one lane away
                static_for<0, N>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const F rf = ofv[j] + (t0c[j] - t0m[j]);
                    rhs[j] = rf + add_shifted_elem<-C, j, true>(t1c[j], t1c);
                });
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double a = um[j] + up[j];
                a = a + nl[j];
                a = a + nr[j];
                if constexpr (COUPLED) {
                    a = a + cl[j];
                    a = a + cr[j];
                }
                acc[j] = a;
            }
            if constexpr (s == 0 && F32) {
                if (np2_first) {  // u is the float32 image: the reference's accumulator is float32 too (wave-uniform)
                    DXH_UNIFORM_BRANCH();
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        float af = (float)um[j] + (float)up[j];
                        af = af + (float)nl[j];
                        af = af + (float)nr[j];
                        if constexpr (COUPLED) {
                            af = af + (float)cl[j];
                            af = af + (float)cr[j];
                        }
                        acc[j] = (double)af;
                    }
                }
            }
            divide(rhs, acc, un);
            if constexpr (DXH_NP2_PREFETCH && s + 1 < ACT) fetch(IC<s + 1>{});
            if (p32 >= nrow32) {  // below the image: replicate the new iterate
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) un[j] = unm[j];
            }
            if constexpr (NORM) {
                const bool in_chunk = p32 >= R0_32 && p32 < R1_32;  // wave-uniform; rows of this chunk only (each element once)
                // one running sum per lane and stage in LDS; the lane mask (strip halo) is applied when the sums are read back
                // (fused_np2_body)
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const double dlt = un[j] - uc[j];
                    const double sq = (in_chunk && inL[j]) ? dlt * dlt : 0.0;   // (a select, not a branch: the stages stay one block)
                    t = j == 0 ? sq : t + sq;
                }
                if constexpr (DXH_NP2_INC_REGS) {
                    st.inc[s] = st.inc[s] + t;
                } else {
                    // fire and forget: no result comes back, so nothing waits on the LDS round trip (a volatile read-modify-write here
                    // cost 0.6 ms per K = 4 launch); a lane's adds reach its slot in program order
                    lds_add_f64(linc + s * kWave, t);
                }
            }
            double q0o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) q0o[j] = (un[j] - unm[j]) + (double)b0m[j];
            if constexpr (last) {
                double unr[N], q1o[N];
                shifted<+C, true>(un, unr);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if constexpr (EDGE)
                        q1o[j] = (sel_bits(edgeR[j], un[j], unr[j]) - un[j]) + (double)b1c[j];
                    else
                        q1o[j] = (unr[j] - un[j]) + (double)b1c[j];
                }
                if (store_lane) {
                    if (p32 >= R0_32 && p32 < R1_32) {
                        if constexpr (NTS) {
                            store_vec_nt(at(q1_out, p), q1o);
                            store_vec_nt(at(u_out, p), un);
                        } else {
                            store_vec(at(q1_out, p), q1o);
                            store_vec(at(u_out, p), un);
                        }
                    }
                    if (p32 - 1 >= R0_32 && p32 - 1 < R1_32) {
                        if constexpr (NTS)
                            store_vec_nt(at(q0_out, p - 1), q0o);
                        else
                            store_vec(at(q0_out, p - 1), q0o);
                    }
                }
            } else {
                // the new row and its b_1 for stage s + 1, which reads them at the next step (its parity "cur")
                u_put(s * L::SLOTS_PER_LEVEL + (L::SLOTS_PER_LEVEL == 2 ? prv : 0), un);
                if constexpr (L::FWD) fw_put(s, b1c, ofv);
                else f_put(lb1, s * L::SLOTS_PER_LEVEL + (L::SLOTS_PER_LEVEL == 2 ? prv : 0), b1c);
                if constexpr (KQV) {
                    double2 t;
                    t.x = kqs[0];
                    t.y = kqs[1];
                    *reinterpret_cast<double2*>(lkq + s * L::KQSLOT) = t;   // (stage s + 1's read of this slot was issued above)
                }
#pragma unroll
                for (int j = 0; j < N; ++j) q0h[j] = q0o[j];
            }
        });
        if constexpr (KQV) {
#pragma unroll
            for (int j = 0; j < N; ++j) st.kqv[0][j] = st.kqn[j];
        }
        if constexpr (PFD == 2) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                st.U[0][(PH + 2) & 3][j] = st.pf[(PH + 1) & 1][0][j];
                frn[j] = st.pff[(PH + 1) & 1][j];
                st.q0l[(PH + 1) & 1][j] = st.pf[(PH + 1) & 1][2][j];
                st.q1l[(PH + 1) & 1][j] = st.pf[(PH + 1) & 1][3][j];
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) st.of0[j] = frn[j];
    }
};

template <int C, bool COUPLED, int K, bool NTS, bool EDGE, bool NORM, bool KQV, typename F>
__device__ __forceinline__ void fused_np2_body(const FusedArgs<double>& a, const int strip, const int chunk, const int lane,
                                               char* lds_wave, double* linc_lane) {
    constexpr int N = 2;
    constexpr bool F32 = std::is_same<F, float>::value;
    using SG = StripGeomK<double, C, K>;
    using PipeT = PipeNp2<C, COUPLED, K, NTS, EDGE, NORM, KQV, F>;
    using L = Np2Lds<K, F, KQV>;
    PipeT pp;
    pp.linc = linc_lane;
    chunk_rows(a, chunk, pp.R0, pp.R1);
    if (pp.R0 >= a.n0) return;
    const int64_t x0 = (int64_t)strip * SG::STRIDE + (int64_t)(lane - SG::LH) * N;
    int64_t xl = x0;
    if (xl < 0) xl = 0;
    if (xl > a.pitch - N) xl = a.pitch - N;
    pp.store_lane = lane >= SG::LH && lane < SG::LH + SG::VALID && x0 < a.L;
    pp.lb1 = reinterpret_cast<F*>(lds_wave + L::B1_OFF) + lane * (L::FWD ? 4 : N);
    pp.lof = reinterpret_cast<F*>(lds_wave + L::OF_OFF) + lane * N;
    pp.lkq = lds_wave + L::KQ_OFF + lane * 16;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int64_t x = x0 + j;
        const bool eL = x < C, eR = x >= a.L - C;
        const int cc = (int)(((x % C) + C) % C);
        pp.edgeL[j] = lane_mask_bits(eL);
        pp.edgeR[j] = lane_mask_bits(eR);
        pp.ch0[j] = lane_mask_bits(cc == 0);
        pp.chLast[j] = lane_mask_bits(cc == C - 1);
        pp.inL[j] = x >= 0 && x < a.L;
        // LDS addresses of the element and of its neighbours (+-C / +-1 elements along the row); the element itself where the
        // reference replicates
        const int e = 2 * lane + j;
        pp.lw[j] = lds_wave + L::elem_off(e);
        pp.aL[j] = lds_wave + L::elem_off(eL ? e : e - C);
        pp.aR[j] = lds_wave + L::elem_off(eR ? e : e + C);
        pp.aCL[j] = lds_wave + L::elem_off(cc == 0 ? e : e - 1);
        pp.aCR[j] = lds_wave + L::elem_off(cc == C - 1 ? e : e + 1);
    }
    if constexpr (F32) pp.of_p = a.of32;
    else pp.of_p = a.of;
    pp.u_in = a.u_in;
    pp.q0_in = a.q0_in;
    pp.q1_in = a.q1_in;
    pp.u_out = a.u_out;
    pp.q0_out = a.q0_out;
    pp.q1_out = a.q1_out;
    pp.kq_p = a.cp.kq;
    pp.voff = (uint32_t)(xl * (int64_t)sizeof(double));
    pp.P = a.pitch;
    pp.nrow = a.n0;
    pp.nrow32 = (int)a.n0;  // the host keeps n0 below 2^31 - 64 for these kernels (multi_supported)
    pp.R0_32 = (int)pp.R0;
    pp.R1_32 = (int)pp.R1;
    pp.dsc = a.c.dsc;
    pp.rdsc = a.c.rdsc;
    pp.kq = a.c.kq;
    pp.negkq = -a.c.kq;
    pp.ell_f = (F)a.c.ell;
    pp.diag_f = (F)a.c.diag;
    pp.rdiag_f = F(1) / pp.diag_f;     // RN(1 / diag): IEEE division, once per wave
    pp.div3f = F32 && a.np2_div3 != 0;
    pp.np2_first = F32 && a.np2_first != 0;

    Np2State<K, F> st;
#pragma unroll
    for (int l = 0; l <= K; ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) st.U[l][i][j] = 0.0;
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) st.t0[l][i][j] = st.b0[l][i][j] = F(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) st.q0l[i][j] = st.q1l[i][j] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) st.kqv[0][j] = 0.0;   // (the not yet started stages read zeroed LDS rows: shrink by nothing)
#pragma unroll
    for (int j = 0; j < N; ++j) {
        st.of0[j] = F(0);
        st.kqn[j] = 0.0;
    }
    {
        // the wave's LDS region starts zeroed (pads included): the leading (discarded) rows must not see NaN bit patterns,
        // and lanes that look past the wave's ends read zeros
        static_assert(L::BYTES % 16 == 0, "whole 16-byte words");
        for (int off = lane * 16; off < L::BYTES; off += kWave * 16)
            *reinterpret_cast<float4*>(lds_wave + off) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // prologue: phase 0 at row rs; rows rs - 1, rs, rs + 1 of u in ring slots 3, 0, 1, the carried (t0, b0) of row rs - 1
    // recomputed from the stored q0
    const int64_t rs = pp.R0 - (K - 1);
    {
        load_vec(pp.at(pp.u_in, pp.clampr(rs - 1)), st.U[0][3]);
        load_vec(pp.at(pp.u_in, pp.clampr(rs)), st.U[0][0]);
        load_vec(pp.at(pp.u_in, pp.clampr(rs + 1)), st.U[0][1]);
        pp.load_of(pp.clampr(rs), st.of0);
        load_vec(pp.at(pp.q0_in, pp.clampr(rs)), st.q0l[0]);
        load_vec(pp.at(pp.q1_in, pp.clampr(rs)), st.q1l[0]);
        if constexpr (PipeT::PFD == 2) {
            load_vec(pp.at(pp.u_in, pp.clampr(rs + 2)), st.pf[1][0]);
            pp.load_of(pp.clampr(rs + 1), st.pff[1]);
            load_vec(pp.at(pp.q0_in, pp.clampr(rs + 1)), st.pf[1][2]);
            load_vec(pp.at(pp.q1_in, pp.clampr(rs + 1)), st.pf[1][3]);
        }
        const int64_t rm = pp.clampr(rs - 1);
        double q0v[N], kqm[N];
        load_vec(pp.at(pp.q0_in, rm), q0v);
        if constexpr (KQV) {
            load_vec(pp.at(pp.kq_p, rm), kqm);
            load_vec(pp.at(pp.kq_p, pp.clampr(rs)), st.kqv[0]);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            F df, bf;
            if constexpr (KQV) np2_split<false>(q0v[j], kqm[j], df, bf);   // (once per chunk: the general expression)
            else pp.split(q0v[j], pp.kq, df, bf);
            st.t0[0][1][j] = pp.ell_f * (bf - df);   // parity "prv" of phase 0
            st.b0[0][1][j] = bf;
        }
    }
    const int64_t r_end = pp.R1 + (K - 1);
#pragma unroll
    for (int i = 0; i < K; ++i) st.inc[i] = 0.0;
    if constexpr (NORM && !DXH_NP2_INC_REGS) {
#pragma unroll
        for (int i = 0; i < K; ++i) *(volatile double*)(linc_lane + i * kWave) = 0.0;
    }
    // Start-up: stage s first matters at step 2 s - 1 of the chunk (fused_pipe_body states the argument)
    int64_t r = rs;
#ifndef DXH_NO_STAGE_SKIP
    {
        constexpr int NG = (2 * K - 4) / 4 + 1;
        static_for<0, NG>([&](auto gc) {
            constexpr int t0 = 4 * decltype(gc)::value;
            constexpr auto act = [](int t) { return (t + 1) / 2 + 1 < K ? (t + 1) / 2 + 1 : K; };
            pp.template step<0, act(t0)>(st, r);
            pp.template step<1, act(t0 + 1)>(st, r + 1);
            pp.template step<2, act(t0 + 2)>(st, r + 2);
            pp.template step<3, act(t0 + 3)>(st, r + 3);
            r += 4;
        });
    }
#endif
    for (; r <= r_end; r += 4) {
        pp.template step<0>(st, r);
        pp.template step<1>(st, r + 1);
        pp.template step<2>(st, r + 2);
        pp.template step<3>(st, r + 3);
    }
    if constexpr (NORM) {
        // a wave's LDS operations complete in order: the sums are final here.  Lanes outside the strip's stored part drop out
        double inc[K][N];
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const double v = DXH_NP2_INC_REGS ? st.inc[s] : *(volatile double*)(linc_lane + s * kWave);
            inc[s][0] = pp.store_lane ? v : 0.0;
            inc[s][1] = 0.0;
        }
        pipe_write_partials<K, 0>(a, inc, chunk * a.nstrips + strip, lane);
    }
}

template <int C, bool COUPLED, int K, bool NORM, bool KQV, bool NTS = true, typename F = float>
__global__ void __launch_bounds__(256, (std::is_same<F, double>::value ? DXH_F64P_WAVES : (KQV ? np2_kqv_waves(C) : (K > 4 ? DXH_NP2_WAVES_K6 : DXH_NP2_WAVES))))
k_fused_multi_np2(FusedArgs<double> a) {
    constexpr int N = 2;
    using SG = StripGeomK<double, C, K>;
    using L = Np2Lds<K, F, KQV>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int WPB = 4;
    const int nbx = (a.nstrips + WPB - 1) / WPB;
    int bx, chunk;
    if (!block_to_tile_sub((int)blockIdx.x, nbx, a, bx, chunk)) return;  // block-uniform
    const int strip = bx * WPB + wave;
    if (strip >= a.nstrips || chunk >= a.nchunks) return;  // wave-uniform
    const int64_t w0 = (int64_t)strip * SG::STRIDE - SG::LH * N;  // first element of the wave's window
    const bool edge = (w0 < C) || (w0 + kWave * N > a.L - C);     // wave-uniform
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * L::BYTES];
    __shared__ double lds_inc[(NORM && !DXH_NP2_INC_REGS) ? WPB * K * kWave : 1];
    char* lds_wave = lds_all + (size_t)wave * L::BYTES;
    double* linc_lane = (NORM && !DXH_NP2_INC_REGS) ? lds_inc + (size_t)wave * (K * kWave) + lane : lds_inc;
    if (edge)
        fused_np2_body<C, COUPLED, K, NTS, true, NORM, KQV, F>(a, strip, chunk, lane, lds_wave, linc_lane);
    else
        fused_np2_body<C, COUPLED, K, NTS, false, NORM, KQV, F>(a, strip, chunk, lane, lds_wave, linc_lane);
    signal_chunk_done(a, chunk, lane);
}

}  // namespace dxh
