// K Jacobi sweeps of a VOLUME in one pass (dim = 3; utils/linear_solvers/jacobi.py:107-114 -- the inner solver of the
// reference's own volume workloads: examples/regularization.py:8-31 and examples/paper/pet_simulations_comparison_block_b.py:
// 700-713 run split Bregman on DICOM volumes with Jacobi(maxiter=20)):
//     x_{s+1} = rs + N(x_s) / dsc        rs = rhs / diag the same for every sweep, N = the six-neighbour sum with replicate ghosts
// One sweep per launch (k_jacobi_vol_x) streams x in, rs in, x out: 24 B per element and sweep in float64, and a 256^3 float64
// volume (3 x 134 MB) does not fit the Infinity Cache.  Here a block of 512 threads owns a TY x TX tile of the (axis 1, axis 2)
// plane and marches down axis 0 with a K-stage pipeline (3.5-D blocking): stage s applies sweep s to plane z - s.
//   * along axis 0 a thread keeps the two planes above of every level in registers (4 cells per thread: rows ty, ty + 8, ...
//     of one column) and takes the plane below from the stage before it, in the same step;
//   * in-plane neighbours come out of ONE LDS tile per level (the level's current plane, written at the end of the step
//     before): every stage reads its four neighbours, a barrier, then every level's tile is overwritten with its next plane,
//     a barrier -- two barriers per step whatever K;
//   * the halo shrinks by one cell per sweep: a tile stores (TY - 2K) x (TX - 2K) of its TY x TX cells (32 x 64, K = 3: 74 %);
//     cells outside the volume load clamped addresses and are never stored; the replicate rule is a per-thread neighbour
//     offset of zero.
// Bytes per element and PASS of K sweeps: (x in + rs in + x out) / 0.74 = 33 B in float64 instead of 72 B for three sweeps.
// Same expressions in the same order as k_jacobi_vol_x (the quotient through the host-rounded reciprocal with two residual
// corrections, IEEE division for numerators outside its proven range): same bits.  No tolerance; diag_scaled a scalar or (HET)
// a plane read once per stage (array-valued omega / ell: the IEEE division per element, as k_jacobi_vol_x does it).
#pragma once

#include "dxh_kernels.h"

namespace dxh {

template <typename T>
struct VolMultiArgs {
    const T* rs;
    const T* x_in;
    T* x_out;
    int64_t n0, n1, n2, pitch;   // pitch: elements per plane
    int tiles_y, tiles_x, nchunks;
    int64_t planes_per_chunk;
    T dsc, rdsc;                 // rdsc = RN(1 / dsc), 0: divide the IEEE way
    const T* dsc_plane;          // HET: diag_scaled per element (array-valued omega / ell), IEEE division
};

// 512 threads x 4 cells per tile (16 waves per CU at <= 128 registers: 256 x 8 needs 251 in float64, K = 3 -- 8 waves per CU;
// measured 256^3 float64 1.32 -> 1.27 ms per Jacobi(20) iteration, float32 0.76 -> 0.68); the float64 divisor-plane variant
// needs 130 registers at 4 cells and keeps 256 x 8 (1.93 against 2.07 ms)
template <typename T, int K, bool HET>
struct VolMultiGeom {
    static constexpr bool WIDE = !(HET && sizeof(T) == 8);
    static constexpr int TY = 32, TX = 64, E = WIDE ? 4 : 8, NT = WIDE ? 512 : 256;
    static constexpr int VY = TY - 2 * K, VX = TX - 2 * K;  // cells a tile stores
    static constexpr int TILE_BYTES = TY * TX * (int)sizeof(T);
    static constexpr int RS = NT / TX;                       // rows between a thread's cells
    static_assert(TY * TX == NT * E, "one cell per (thread, slot)");
};

template <typename T, int K, bool HET>
__global__ void __launch_bounds__((VolMultiGeom<T, K, HET>::NT), 2) k_jacobi_vol_multi(VolMultiArgs<T> a) {
    using G = VolMultiGeom<T, K, HET>;
    constexpr int TY = G::TY, TX = G::TX, E = G::E;
    __shared__ __attribute__((aligned(16))) char tiles[K * G::TILE_BYTES];
    // XCD x (= blockIdx % 8) works a contiguous run of (chunk, tile row, tile) blocks: neighbouring tiles share their halo in L2
    const int total = a.tiles_y * a.tiles_x * a.nchunks;
    const int per_xcd = (total + 7) >> 3;
    const int bid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || bid >= total) return;  // block-uniform
    const int txi = bid % a.tiles_x, tyi = (bid / a.tiles_x) % a.tiles_y, chunk = bid / (a.tiles_x * a.tiles_y);
    const int64_t Z0 = (int64_t)chunk * a.planes_per_chunk;
    if (Z0 >= a.n0) return;  // block-uniform
    const int64_t Z1 = (Z0 + a.planes_per_chunk < a.n0) ? Z0 + a.planes_per_chunk : a.n0;
    const int tx = (int)threadIdx.x & (TX - 1), ty0 = (int)threadIdx.x / TX;   // rows ty0, ty0 + RS, ...
    const int64_t gx = (int64_t)txi * G::VX - K + tx;
    const int64_t cx = gx < 0 ? 0 : (gx > a.n2 - 1 ? a.n2 - 1 : gx);
    const bool x_in_vol = gx >= 0 && gx < a.n2, x_stored = tx >= K && tx < TX - K && x_in_vol;
    // in-plane neighbour offsets within a tile (bytes); zero where the reference replicates or the tile ends
    const int xl_off = (tx == 0 || gx <= 0) ? 0 : -(int)sizeof(T);
    const int xr_off = (tx == TX - 1 || gx >= a.n2 - 1) ? 0 : (int)sizeof(T);
    int own[E], yl_off[E], yr_off[E];
    int goff[E];           // element offset within a plane (clamped into the volume; the host keeps n1 * n2 below 2^31)
    bool stored[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int y = ty0 + G::RS * i;
        const int64_t gy = (int64_t)tyi * G::VY - K + y;
        const int64_t cy = gy < 0 ? 0 : (gy > a.n1 - 1 ? a.n1 - 1 : gy);
        own[i] = (y * TX + tx) * (int)sizeof(T);
        yl_off[i] = (y == 0 || gy <= 0) ? 0 : -TX * (int)sizeof(T);
        yr_off[i] = (y == TY - 1 || gy >= a.n1 - 1) ? 0 : TX * (int)sizeof(T);
        goff[i] = (int)(cy * a.n2 + cx);
        stored[i] = x_stored && y >= K && y < TY - K && gy >= 0 && gy < a.n1;
    }
    auto clampz = [&](int64_t p) -> int64_t { return p < 0 ? 0 : (p > a.n0 - 1 ? a.n0 - 1 : p); };
    auto lds = [&](int s, int off) -> T { return *reinterpret_cast<const T*>(tiles + s * G::TILE_BYTES + off); };

    // level s: um[s], uc[s] = planes p - 1, p of x_s (p = z - s); the plane below is the stage before's output of this step
    T um[K][E], uc[K][E];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int i = 0; i < E; ++i) um[s][i] = uc[s][i] = T(0);
    const int64_t zs = Z0 - (K - 1);   // the last stage reaches Z0 with valid planes above it
    T below0[E];                        // level 0, plane z + 1 (loaded one step ahead)
#pragma unroll
    for (int i = 0; i < E; ++i) {
        um[0][i] = a.x_in[clampz(zs - 1) * a.pitch + goff[i]];
        uc[0][i] = a.x_in[clampz(zs) * a.pitch + goff[i]];
        below0[i] = a.x_in[clampz(zs + 1) * a.pitch + goff[i]];
    }
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int i = 0; i < E; ++i) *reinterpret_cast<T*>(tiles + s * G::TILE_BYTES + own[i]) = uc[s][i];
    __syncthreads();
    const bool recip = a.rdsc != T(0);
    for (int64_t z = zs; z <= Z1 - 1 + (K - 1); ++z) {
        T nxt0[E];   // level 0, plane z + 2
#pragma unroll
        for (int i = 0; i < E; ++i) nxt0[i] = a.x_in[clampz(z + 2) * a.pitch + goff[i]];
        T below[E];
#pragma unroll
        for (int i = 0; i < E; ++i) below[i] = below0[i];
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const int64_t p = z - s;
            const bool top = p <= 0, bottom = p >= a.n0 - 1;   // block-uniform: replicate ghosts (jacobi.py:43,52)
            const int64_t pc = clampz(p);
            T acc[E], r[E], xn[E];
#pragma unroll
            for (int i = 0; i < E; ++i) r[i] = a.rs[pc * a.pitch + goff[i]];
#pragma unroll
            for (int i = 0; i < E; ++i) {
                T v = T(0);
                v = v + (top ? uc[s][i] : um[s][i]);
                v = v + (bottom ? uc[s][i] : below[i]);
                v = v + lds(s, own[i] + yl_off[i]);
                v = v + lds(s, own[i] + yr_off[i]);
                v = v + lds(s, own[i] + xl_off);
                v = v + lds(s, own[i] + xr_off);
                acc[i] = v;
            }
            bool ieee = !recip;
            if constexpr (HET) {
                T dv[E];
#pragma unroll
                for (int i = 0; i < E; ++i) dv[i] = a.dsc_plane[pc * a.pitch + goff[i]];
#pragma unroll
                for (int i = 0; i < E; ++i) xn[i] = r[i] + acc[i] / dv[i];
                ieee = false;
            } else if (!ieee) {
                T lo = t_abs(acc[0]);
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    xn[i] = r[i] + div_by_const<T>(acc[i], a.dsc, a.rdsc);
                    lo = t_min(lo, t_abs(acc[i]));
                }
                if (__builtin_expect(__any(lo < DivSafe<T>::tiny), 0)) {
                    bool unsafe = false;
#pragma unroll
                    for (int i = 0; i < E; ++i) unsafe = unsafe || div_unsafe(acc[i]);
                    ieee = __any(unsafe);
                }
            }
            if (ieee) {
#pragma unroll
                for (int i = 0; i < E; ++i) xn[i] = r[i] + acc[i] / a.dsc;
            }
            if (s == K - 1) {
                if (p >= Z0 && p < Z1) {
#pragma unroll
                    for (int i = 0; i < E; ++i)
                        if (stored[i]) a.x_out[p * a.pitch + goff[i]] = xn[i];
                }
            }
            // rotate this level's window; its output is the plane below for the next level (which sits one plane up)
#pragma unroll
            for (int i = 0; i < E; ++i) {
                um[s][i] = uc[s][i];
                uc[s][i] = below[i];
                below[i] = xn[i];
            }
        }
#pragma unroll
        for (int i = 0; i < E; ++i) below0[i] = nxt0[i];
        __syncthreads();   // every stage has read its tile
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
            for (int i = 0; i < E; ++i) *reinterpret_cast<T*>(tiles + s * G::TILE_BYTES + own[i]) = uc[s][i];
        __syncthreads();   // the next step's tiles are in place
    }
}

// specialised in dxh_vol_multi.hip
template <typename T>
void launch_jacobi_vol_multi(int K, const VolMultiArgs<T>& a, unsigned grid, hipStream_t stream);

}  // namespace dxh
