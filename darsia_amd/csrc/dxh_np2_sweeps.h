// ONE outer iteration with S = K Jacobi sweeps per pass in the reference's float32 dtype flow ("numpy2" promotion) -- the
// caller's solver is Jacobi(maxiter=S), 2 <= S <= 6 (restoration/split_bregman_tvd.py:129-146 with
// utils/linear_solvers/jacobi.py:101-114:  rs = rhs / diag once, then S times  x <- rs + N(x) / diag_scaled).
//
// The decomposition is PipeNp2's (dxh_np2_pipe.h): a wave owns a column strip of 64 lanes x 16 B and a chunk of rows and
// marches down the rows with a K-stage register pipeline, rows of the levels >= 1 handed from stage to stage through LDS
// (neighbours at per-lane addresses that carry the replicate rules).  Here the stages are the SWEEPS of one iteration:
//
//   stage 0        the iteration's head: d_j, b_j = the shrinkage of the stored q_j (float32), t_j = ell (b_j - d_j), the float32
//                  right-hand side, rs = rhs / diag (float32), and the first sweep on the stored iterate
//   stages 1..K-1  plain sweeps x_{s+1} = float64(rs) + N(x_s) / diag_scaled; rs of a row travels down the stages in registers
//   stage K-1      also the iteration's tail: q_0 = D_0 u + b_0, q_1 = D_1 u + b_1 of the new iterate, stored with it
//
// b_0, b_1 of a row are produced by stage 0 and consumed K - 1 (b_1) and K (b_0, by the row below) steps later by the last
// stage: they wait in a ring of eight rows in LDS (the row loop is unrolled by eight, so every slot is a compile-time offset).
// The flow this replaces streamed 60 B per element through a fused shrinkage / right-hand-side / first-sweep pass and 24 B per
// element through every K = 4 pass of the remaining sweeps (k_shrink_rhs_np2 + k_jacobi_multi<double>: 1.43 + 1.03 ms per
// iteration at 4000 x 8000 x 3, five sweeps); this pass reads u, q_0, q_1 (float64) and omega*img (float32) and writes u, q_0,
// q_1: 52 B per element and iteration whatever S.
// Scalar omega and ell; mu scalar or a float64 array (KQV: the threshold mu / ell read per element by stage 0 -- 8 B more per
// element; rows with a negative or NaN threshold take the shrinkage as written); no solver tolerance: everything else keeps
// the three-launch flow.
// F = double: float64 IMAGES (every quantity float64, the native flow of such images; scalar coefficients).  The tail then
// re-reads the stored q_0, q_1 of its rows (L2: stage 0 read them K - 1 steps ago) and repeats the head's shrinkage instead of
// holding b_0, b_1 in LDS -- a ring of float64 rows would leave one block per CU.
// PLAIN (F = double): K Jacobi sweeps and nothing else -- every stage a plain sweep, rs = rhs / diag read from its plane by
// stage 0, only the new iterate stored: the multi-sweep passes of a Jacobi solve on a float64 context (H1_regularization, the
// `Jacobi` operator, TV loops with more than six sweeps), bit for bit k_jacobi_multi<double> (dxh_kernels.h), which shifts
// every level by DPP pairs and selects the replicate cases per element (1.03 ms per four sweeps at 4000 x 8000 x 3).
#pragma once

#include "dxh_np2_pipe.h"

namespace dxh {

template <int K, typename F = float>
struct Np2SweepLds {
    static constexpr bool FIFO = std::is_same<F, float>::value;   // b_0, b_1 wait in LDS (float) / are recomputed by the tail (double)
    static constexpr int PAD = 16;
    static constexpr int UHALF = PAD + kWave * 8;
    static constexpr int USLOT = 2 * UHALF;
    // ONE slot per level 1 .. K - 1: stage s fetches stage s + 1's operands (the row it wrote one step ago) before it writes
    // the new row, and a wave's LDS operations execute in order
    static constexpr int NU = K - 1;
    static constexpr int U_BYTES = NU * USLOT + PAD;
    static constexpr int FIFO_OFF = U_BYTES;
    static constexpr int FHALF = kWave * 2 * (int)sizeof(float);   // a row's b_0 pairs, then its b_1 pairs: 8 B per lane each
    static constexpr int FSLOT = 2 * FHALF;                        // (consecutive lanes 8 B apart: conflict-free ds_read_b64)
    static constexpr int NF = K + 1 <= 4 ? 4 : 8;                  // rows between stage 0 and the last stage, a power of two
    // eps: this lane's running sum of squared increments.  Not at K = 5 in float: its 168 registers and 49.7 KB of LDS per block
    // have no room for it (36 - 80 B of scratch and 51.8 KB: +9 % per pass without the criterion) -- the host reduces the
    // increment with a kernel of its own there (np2_sweeps_fold_norm)
    static constexpr bool HAS_NORM = !(FIFO && K == 5);
    static constexpr int INC_OFF = FIFO ? FIFO_OFF + NF * FSLOT : U_BYTES;
    static constexpr int BYTES = INC_OFF + (HAS_NORM ? kWave * 8 : 0);            // K = 5: 4240 + 8192 = 12432 per wave (three blocks per CU)
    static __device__ __forceinline__ int elem_off(int e) {
        const int par = e & 1;
        return PAD + par * UHALF + ((e - par) >> 1) * 8;
    }
    static_assert(K >= 2 && K + 1 <= NF, "the ring holds the rows between stage 0 and the last stage");
    static_assert(U_BYTES % 16 == 0 && USLOT % 16 == 0, "16-byte rows");
};

template <int K, typename F = float>
struct Np2SweepState {
    static constexpr int N = 2;
    double U[K + 1][4][N];        // rows of level l, slot = (row - rs) & 3 (level K: parity)
    F t0[2][N];                   // stage 0: ell (b_0 - d_0) of the current / previous row
    double q0l[2][N], q1l[2][N];  // loaded q rows (current / prefetched)
    F of0[N];                     // omega*img of stage 0's row
    double kql[2][N];             // KQV: the threshold of stage 0's row (current / prefetched)
    F rs[K][2][N];                // rs[s][parity of the step it is read in]: rhs / diag of stage s's row (s >= 1)
};

template <int C, bool COUPLED, int K, bool NTS, bool EDGE, bool KQV, typename F = float, bool PLAIN = false>
struct PipeNp2Sweeps {
    static constexpr int N = 2;
    static constexpr bool F32 = std::is_same<F, float>::value;
    static_assert(F32 || !KQV, "float64 images: scalar threshold");
    static_assert(!PLAIN || !F32, "plain sweeps: float64 contexts");
    using L = Np2SweepLds<K, F>;
    using F2 = typename Np2Vec2<F>::type;
    const F* of_p;
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
    bool div3f, np2_first, store_lane;
    uint32_t edgeL[N], edgeR[N], ch0[N], chLast[N];
    uint32_t voff;
    char* lw[N];
    char* aL[N];
    char* aR[N];
    char* aCL[N];
    char* aCR[N];
    char* lfifo;   // this lane's b_0 pair in slot 0 of the (b_0, b_1) ring (its b_1 pair: FHALF further)
    double* linc;  // eps (norm): this lane's running sum of squared increments of the iteration, in LDS (ds_add_f64)
    bool norm;     // wave-uniform: the launch also reduces || u_new - u_old ||^2 (FusedArgs::inc_partial set)
    bool inL[N];

    __device__ __forceinline__ void u_put(int slot, const double (&v)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) *reinterpret_cast<double*>(lw[j] + slot * L::USLOT) = v[j];
    }
    __device__ __forceinline__ double u_get(const char* a, int slot) const {
        return *reinterpret_cast<const double*>(a + slot * L::USLOT);
    }
    __device__ __forceinline__ int64_t clampr(int64_t r) const { return r < 0 ? 0 : (r > nrow - 1 ? nrow - 1 : r); }
    __device__ __forceinline__ const double* at(const double* plane, int64_t row) const {
        return reinterpret_cast<const double*>(reinterpret_cast<const char*>(plane + row * P) + voff);
    }
    __device__ __forceinline__ double* at(double* plane, int64_t row) const {
        return reinterpret_cast<double*>(reinterpret_cast<char*>(plane + row * P) + voff);
    }
    __device__ __forceinline__ void load_of(int64_t row, F (&v)[N]) const {
        const F2 t = *reinterpret_cast<const F2*>(reinterpret_cast<const char*>(of_p + row * P) + (F32 ? (voff >> 1) : voff));
        v[0] = t.x;
        v[1] = t.y;
    }
    __device__ __forceinline__ void split(double q, double kqj, F& d, F& b) const {
        if constexpr (KQV) np2_split_fast(q, -kqj, kqj, d, b);
        else np2_split_fast(q, negkq, kq, d, b);
    }

    // rs = float32(rhs) / float32(diag) through the host-rounded reciprocal (three operations where the exhaustive check of
    // this divisor allows, five otherwise); non-zero tiny numerators: the IEEE division
    __device__ __forceinline__ void div_rhs(const F (&rhs)[N], F (&rs)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) rs[j] = div_by_const3(rhs[j], diag_f, rdiag_f);
        if (!F32 || !div3f) {   // (float64: no exhaustive check exists -- always both corrections)
            DXH_UNIFORM_BRANCH();
#pragma unroll
            for (int j = 0; j < N; ++j) rs[j] = div_refine(rs[j], rhs[j], diag_f, rdiag_f);
        }
        if (__builtin_expect(__any(np2_absmin(rhs[0], rhs[1]) < DivSafe<F>::tiny), 0)) {
            if (__any(div_unsafe(rhs[0]) || div_unsafe(rhs[1]))) {
#pragma unroll
                for (int j = 0; j < N; ++j) rs[j] = rhs[j] / diag_f;
            }
        }
    }
    // N(x) / diag_scaled in float64 through the reciprocal with two residual corrections
    __device__ __forceinline__ void div_acc(const double (&acc)[N], double (&qa)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) qa[j] = div_by_const<double>(acc[j], dsc, rdsc);
        if (__builtin_expect(__any(np2_absmin(acc[0], acc[1]) < DivSafe<double>::tiny), 0)) {
            if (__any(div_unsafe(acc[0]) || div_unsafe(acc[1]))) {
#pragma unroll
                for (int j = 0; j < N; ++j) qa[j] = acc[j] / dsc;
            }
        }
    }

    // PH = step number mod 8; ACT: stages 0 .. ACT - 1 run (the first steps of a chunk)
    template <int PH, int ACT = K>
    __device__ __forceinline__ void step(Np2SweepState<K, F>& st, const int64_t r) const {
        constexpr int cur = PH & 1, prv = cur ^ 1;
        F frn[N];
        {
            const int64_t rn = clampr(r + 1);
            load_vec(at(u_in, clampr(r + 2)), st.U[0][(PH + 2) & 3]);
            load_vec(at(q0_in, rn), st.q0l[prv]);   // (PLAIN: q0_in is the rs plane)
            if constexpr (!PLAIN) {
                load_of(rn, frn);
                load_vec(at(q1_in, rn), st.q1l[prv]);
                if constexpr (KQV) load_vec(at(kq_p, rn), st.kql[prv]);
            }
        }
        // the last stage's b_1 of its row (written K - 1 steps ago) and b_0 of the row above it (K steps ago)
        F tb0m[N], tb1c[N];
        double tq0[N], tq1[N], tu0[N];
        if constexpr (ACT == K && !PLAIN && L::HAS_NORM) {
            if (norm) {   // the iterate this iteration started from, at the tail's row (read by stage 0 K - 1 steps ago: L2)
                DXH_UNIFORM_BRANCH();
                load_vec(at(u_in, clampr(r - (K - 1))), tu0);
            }
        }
        if constexpr (ACT == K && !L::FIFO && !PLAIN) {
            // float64: the stored q_1 of the tail's row and q_0 of the row above it (read by stage 0 K - 1 / K steps ago: L2)
            load_vec(at(q1_in, clampr(r - (K - 1))), tq1);
            load_vec(at(q0_in, clampr(r - K)), tq0);
        }
        if constexpr (ACT == K && L::FIFO && !PLAIN) {
            const float2 vb1 = *reinterpret_cast<const float2*>(lfifo + ((PH - (K - 1)) & (L::NF - 1)) * L::FSLOT + L::FHALF);
            const float2 vb0 = *reinterpret_cast<const float2*>(lfifo + ((PH - K) & (L::NF - 1)) * L::FSLOT);
            tb1c[0] = vb1.x;
            tb1c[1] = vb1.y;
            tb0m[0] = vb0.x;
            tb0m[1] = vb0.y;
        }
        // stage s issues the LDS reads of stage s + 1 (the row stage s wrote one step ago) before it overwrites that row
        double pnl[N], pnr[N], pcl[N], pcr[N];
        auto fetch = [&](auto sn) {
            constexpr int slot1 = decltype(sn)::value - 1;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                pnl[j] = u_get(aL[j], slot1);
                pnr[j] = u_get(aR[j], slot1);
                if constexpr (COUPLED) {
                    pcl[j] = u_get(aCL[j], slot1);
                    pcr[j] = u_get(aCR[j], slot1);
                }
            }
        };
        static_for<0, ACT>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int c = (PH - s) & 3;
            constexpr int cm = (c + 3) & 3, cp = (c + 1) & 3;
            constexpr bool last = (s == K - 1);
            const int64_t p = r - s;
            const int p32 = (int)p;
            double(&um)[N] = st.U[s][cm];
            double(&uc)[N] = st.U[s][c];
            double(&up)[N] = st.U[s][cp];
            double(&un)[N] = last ? st.U[K][cur] : st.U[s + 1][c];
            const double(&unm)[N] = last ? st.U[K][prv] : st.U[s + 1][cm];

            double nl[N], nr[N], cl[N], cr[N];
            if constexpr (s == 0) {
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
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    nl[j] = pnl[j];
                    nr[j] = pnr[j];
                    if constexpr (COUPLED) {
                        cl[j] = pcl[j];
                        cr[j] = pcr[j];
                    }
                }
            }
            if (p == 0) {  // replicate the row above the image
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) um[j] = uc[j];
            }
            double acc[N];
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
            F rsv[N];
            if constexpr (s == 0 && PLAIN) {
#pragma unroll
                for (int j = 0; j < N; ++j) rsv[j] = (F)st.q0l[cur][j];
            } else if constexpr (s == 0) {
                // d_j = float32(shrink(q_j)), b_j = float32(q_j - d_j), t_j = ell (b_j - d_j) in float32   (:84-89,143-146)
                F(&t0m)[N] = st.t0[prv];
                F(&t0c)[N] = st.t0[cur];
                F b0c[N], b1c[N], t1c[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    F d0f, b0f, d1f, b1f;
                    split(st.q0l[cur][j], st.kql[cur][j], d0f, b0f);
                    split(st.q1l[cur][j], st.kql[cur][j], d1f, b1f);
                    t0c[j] = ell_f * (b0f - d0f);
                    t1c[j] = ell_f * (b1f - d1f);
                    b0c[j] = b0f;
                    b1c[j] = b1f;
                }
                if constexpr (KQV) {
                    // a negative or NaN threshold somewhere in the wave's row: np.maximum(|x| - k, 0) * np.sign(x) as written
                    if (__any(!(st.kql[cur][0] >= 0.0) || !(st.kql[cur][1] >= 0.0))) {
                        DXH_UNIFORM_BRANCH();
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            float d0f, b0f, d1f, b1f;
                            np2_split<false>(st.q0l[cur][j], st.kql[cur][j], d0f, b0f);
                            np2_split<false>(st.q1l[cur][j], st.kql[cur][j], d1f, b1f);
                            t0c[j] = ell_f * (b0f - d0f);
                            t1c[j] = ell_f * (b1f - d1f);
                            b0c[j] = b0f;
                            b1c[j] = b1f;
                        }
                    }
                }
                if (p == 0) {
                    DXH_UNIFORM_BRANCH();
#pragma unroll
                    for (int j = 0; j < N; ++j) t0m[j] = t0c[j];
                }
                F rhs[N];
                if constexpr (EDGE) {
                    F t1l[N];
                    shifted2f<-C>(t1c, t1l);
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const F rf = st.of0[j] + (t0c[j] - t0m[j]);
                        rhs[j] = rf + (t1c[j] - sel_bits(edgeL[j], t1c[j], t1l[j]));
                    }
                } else {
                    static_for<0, N>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const F rf = st.of0[j] + (t0c[j] - t0m[j]);
                        rhs[j] = rf + add_shifted_elem<-C, j, true>(t1c[j], t1c);
                    });
                }
                if (F32 && np2_first) {  // u is the float32 image: the reference's accumulator is float32 too (wave-uniform)
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
                div_rhs(rhs, rsv);
                if constexpr (L::FIFO) {
                    float2 fb0, fb1;
                    fb0.x = b0c[0];
                    fb0.y = b0c[1];
                    fb1.x = b1c[0];
                    fb1.y = b1c[1];
                    *reinterpret_cast<float2*>(lfifo + (PH & (L::NF - 1)) * L::FSLOT) = fb0;
                    *reinterpret_cast<float2*>(lfifo + (PH & (L::NF - 1)) * L::FSLOT + L::FHALF) = fb1;
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) rsv[j] = st.rs[s][cur][j];
            }
            double qa[N];
            div_acc(acc, qa);
#pragma unroll
            for (int j = 0; j < N; ++j) un[j] = (double)rsv[j] + qa[j];
            if constexpr (s + 1 < K) {
#pragma unroll
                for (int j = 0; j < N; ++j) st.rs[s + 1][prv][j] = rsv[j];
            }
            if constexpr (s + 1 < ACT) fetch(IC<s + 1>{});
            if (p32 >= nrow32) {  // below the image: replicate the new iterate
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) un[j] = unm[j];
            }
            if constexpr (last && PLAIN) {
                if (store_lane && p32 >= R0_32 && p32 < R1_32) {
                    if constexpr (NTS) store_vec_nt(at(u_out, p), un);
                    else store_vec(at(u_out, p), un);
                }
            } else if constexpr (last) {
                if (L::HAS_NORM && norm) {
                    DXH_UNIFORM_BRANCH();
                    const bool in_chunk = p32 >= R0_32 && p32 < R1_32;   // rows of this chunk only (each element once)
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const double dlt = un[j] - tu0[j];
                        const double sq = (in_chunk && inL[j]) ? dlt * dlt : 0.0;
                        t = j == 0 ? sq : t + sq;
                    }
                    lds_add_f64(linc, t);
                }
                if constexpr (!L::FIFO) {   // the head's shrinkage of these rows once more: b = q - d (np2_split_fast, float64)
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        F dd;
                        split(tq0[j], 0.0, dd, tb0m[j]);
                        split(tq1[j], 0.0, dd, tb1c[j]);
                    }
                }
                double unr[N], q0o[N], q1o[N];
                shifted<+C, true>(un, unr);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    q0o[j] = (un[j] - unm[j]) + (double)tb0m[j];
                    if constexpr (EDGE)
                        q1o[j] = (sel_bits(edgeR[j], un[j], unr[j]) - un[j]) + (double)tb1c[j];
                    else
                        q1o[j] = (unr[j] - un[j]) + (double)tb1c[j];
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
                u_put(s, un);   // the new row for stage s + 1, which meets it at the next step
            }
        });
        if constexpr (!PLAIN) {
#pragma unroll
            for (int j = 0; j < N; ++j) st.of0[j] = frn[j];
        }
    }
};

template <int C, bool COUPLED, int K, bool NTS, bool EDGE, bool KQV, typename F, bool PLAIN>
__device__ __forceinline__ void np2_sweeps_body(const FusedArgs<double>& a, const int strip, const int chunk, const int lane,
                                                char* lds_wave) {
    constexpr int N = 2;
    using SG = StripGeomK<double, C, K>;
    using PipeT = PipeNp2Sweeps<C, COUPLED, K, NTS, EDGE, KQV, F, PLAIN>;
    using L = Np2SweepLds<K, F>;
    constexpr bool F32 = std::is_same<F, float>::value;
    PipeT pp;
    chunk_rows(a, chunk, pp.R0, pp.R1);
    if (pp.R0 >= a.n0) return;
    const int64_t x0 = (int64_t)strip * SG::STRIDE + (int64_t)(lane - SG::LH) * N;
    int64_t xl = x0;
    if (xl < 0) xl = 0;
    if (xl > a.pitch - N) xl = a.pitch - N;
    pp.store_lane = lane >= SG::LH && lane < SG::LH + SG::VALID && x0 < a.L;
    pp.lfifo = lds_wave + L::FIFO_OFF + lane * 8;
    pp.linc = reinterpret_cast<double*>(lds_wave + L::INC_OFF) + lane;
    pp.norm = !PLAIN && L::HAS_NORM && a.inc_partial != nullptr;
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
    pp.kq_p = a.cp.kq;
    pp.u_out = a.u_out;
    pp.q0_out = a.q0_out;
    pp.q1_out = a.q1_out;
    pp.voff = (uint32_t)(xl * (int64_t)sizeof(double));
    pp.P = a.pitch;
    pp.nrow = a.n0;
    pp.nrow32 = (int)a.n0;
    pp.R0_32 = (int)pp.R0;
    pp.R1_32 = (int)pp.R1;
    pp.dsc = a.c.dsc;
    pp.rdsc = a.c.rdsc;
    pp.kq = a.c.kq;
    pp.negkq = -a.c.kq;
    pp.ell_f = (F)a.c.ell;
    pp.diag_f = (F)a.c.diag;
    pp.rdiag_f = F(1) / pp.diag_f;
    pp.div3f = F32 && a.np2_div3 != 0;
    pp.np2_first = F32 && a.np2_first != 0;

    Np2SweepState<K, F> st;
#pragma unroll
    for (int l = 0; l <= K; ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) st.U[l][i][j] = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            st.t0[i][j] = F(0);
            st.q0l[i][j] = st.q1l[i][j] = st.kql[i][j] = 0.0;
        }
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) st.rs[l][i][j] = F(0);
#pragma unroll
    for (int j = 0; j < N; ++j) st.of0[j] = F(0);
    {
        static_assert(L::BYTES % 16 == 0, "whole 16-byte words");
        for (int off = lane * 16; off < L::BYTES; off += kWave * 16)
            *reinterpret_cast<float4*>(lds_wave + off) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // prologue: phase 0 at row rs; rows rs - 1, rs, rs + 1 of u in ring slots 3, 0, 1, t_0 of row rs - 1 recomputed from the
    // stored q_0
    const int64_t rs = pp.R0 - (K - 1);
    {
        load_vec(pp.at(pp.u_in, pp.clampr(rs - 1)), st.U[0][3]);
        load_vec(pp.at(pp.u_in, pp.clampr(rs)), st.U[0][0]);
        load_vec(pp.at(pp.u_in, pp.clampr(rs + 1)), st.U[0][1]);
        load_vec(pp.at(pp.q0_in, pp.clampr(rs)), st.q0l[0]);   // (PLAIN: rs of row rs)
    }
    if constexpr (!PLAIN) {
        pp.load_of(pp.clampr(rs), st.of0);
        load_vec(pp.at(pp.q1_in, pp.clampr(rs)), st.q1l[0]);
        double q0v[N], kqm[N] = {0.0, 0.0};
        load_vec(pp.at(pp.q0_in, pp.clampr(rs - 1)), q0v);
        if constexpr (KQV) {
            load_vec(pp.at(pp.kq_p, pp.clampr(rs - 1)), kqm);
            load_vec(pp.at(pp.kq_p, pp.clampr(rs)), st.kql[0]);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            F df, bf;
            if constexpr (KQV) {
                if constexpr (F32) np2_split<false>(q0v[j], kqm[j], df, bf);   // (once per chunk: the general expression)
            } else {
                pp.split(q0v[j], 0.0, df, bf);
            }
            st.t0[1][j] = pp.ell_f * (bf - df);   // parity "prv" of phase 0
        }
    }
    const int64_t r_end = pp.R1 + (K - 1);
    int64_t r = rs;
    {
        // start-up: stage s first matters at step 2 s - 1 of the chunk (fused_pipe_body states the argument); eight steps
        // cover every K <= 5, sixteen K = 6
        constexpr int NG = (2 * K - 3 + 7) / 8;
        static_for<0, NG>([&](auto gc) {
            constexpr int t0 = 8 * decltype(gc)::value;
            constexpr auto act = [](int t) { return (t + 1) / 2 + 1 < K ? (t + 1) / 2 + 1 : K; };
            pp.template step<0, act(t0)>(st, r);
            pp.template step<1, act(t0 + 1)>(st, r + 1);
            pp.template step<2, act(t0 + 2)>(st, r + 2);
            pp.template step<3, act(t0 + 3)>(st, r + 3);
            pp.template step<4, act(t0 + 4)>(st, r + 4);
            pp.template step<5, act(t0 + 5)>(st, r + 5);
            pp.template step<6, act(t0 + 6)>(st, r + 6);
            pp.template step<7, act(t0 + 7)>(st, r + 7);
            r += 8;
        });
    }
    for (; r + 4 <= r_end; r += 8) {
        pp.template step<0>(st, r);
        pp.template step<1>(st, r + 1);
        pp.template step<2>(st, r + 2);
        pp.template step<3>(st, r + 3);
        pp.template step<4>(st, r + 4);
        pp.template step<5>(st, r + 5);
        pp.template step<6>(st, r + 6);
        pp.template step<7>(st, r + 7);
    }
    if (r <= r_end) {   // at most four steps left
        pp.template step<0>(st, r);
        pp.template step<1>(st, r + 1);
        pp.template step<2>(st, r + 2);
        pp.template step<3>(st, r + 3);
    }
    if constexpr (!PLAIN && L::HAS_NORM) if (pp.norm) {
        // a wave's LDS operations complete in order: the sum is final here.  Lanes outside the strip's stored part drop out
        double inc[1][N];
        const double v = *(volatile double*)pp.linc;
        inc[0][0] = pp.store_lane ? v : 0.0;
        inc[0][1] = 0.0;
        pipe_write_partials<1, 0>(a, inc, chunk * a.nstrips + strip, lane);
    }
}

template <int C, bool COUPLED, int K, bool KQV, typename F = float, bool PLAIN = false, bool NTS = true>
__global__ void __launch_bounds__(256, (std::is_same<F, float>::value ? np2_sweep_waves(K) : (PLAIN ? np2_plain_waves(K) : 2))) k_sweeps_np2(FusedArgs<double> a) {
    constexpr int N = 2;
    using SG = StripGeomK<double, C, K>;
    using L = Np2SweepLds<K, F>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int WPB = 4;
    const int nbx = (a.nstrips + WPB - 1) / WPB;
    int bx, chunk;
    if (!block_to_tile_sub((int)blockIdx.x, nbx, a, bx, chunk)) return;  // block-uniform
    const int strip = bx * WPB + wave;
    if (strip >= a.nstrips || chunk >= a.nchunks) return;  // wave-uniform
    const int64_t w0 = (int64_t)strip * SG::STRIDE - SG::LH * N;
    const bool edge = (w0 < C) || (w0 + kWave * N > a.L - C);     // wave-uniform
    __shared__ __attribute__((aligned(16))) char lds_all[WPB * L::BYTES];
    char* lds_wave = lds_all + (size_t)wave * L::BYTES;
    if (edge)
        np2_sweeps_body<C, COUPLED, K, NTS, true, KQV, F, PLAIN>(a, strip, chunk, lane, lds_wave);
    else
        np2_sweeps_body<C, COUPLED, K, NTS, false, KQV, F, PLAIN>(a, strip, chunk, lane, lds_wave);
    signal_chunk_done(a, chunk, lane);
}

}  // namespace dxh
