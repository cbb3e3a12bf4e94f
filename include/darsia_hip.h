/*
 * darsia_hip.h -- C ABI of libdarsia_hip.so: MI355X (gfx950) kernels for DarSIA's
 * dense image hot path (split-Bregman TV denoising, Jacobi/H1 smoothing,
 * derivatives, affine colour apply).
 *
 * The reference (pmgbergen/DarSIA) is pure Python and has no FFI layer; its
 * boundary for this path is Python duck typing.  Every entry point below names
 * the reference function whose arithmetic it replaces (file:line relative to
 * /root/reference/src/darsia).  INTEGRATION.md shows the ctypes stub a DarSIA
 * maintainer would add to bind them.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DXH_E* code otherwise;
 *     dxh_last_error() gives the message for the calling thread.  Never aborts.
 *   - host arrays are caller-owned, C-contiguous, never retained.
 *   - a context owns its device buffers and (unless given one) its stream; one
 *     context per host thread at a time, not re-entrant.
 *   - dtype: 0 = float32, 1 = float64.
 *   - an image/volume is described as an array of up to 3 axes (n0, n1, n2)
 *     stored row-major; a 2-D image has ndim = 2 and n2 = 1.  `dim` is the
 *     number of leading SPATIAL axes (reference `dim`), so an (H, W, 3) colour
 *     image is ndim = 3, dim = 2.
 */
#ifndef DARSIA_HIP_H
#define DARSIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXH_OK 0
#define DXH_EINVAL -1   /* bad argument (maps to Python ValueError/TypeError) */
#define DXH_EHIP -2     /* HIP runtime error */
#define DXH_ENOMEM -3   /* device allocation failed */
#define DXH_ESTATE -4   /* call order violated (e.g. run before upload) */

#define DXH_F32 0
#define DXH_F64 1
#define DXH_U8 2  /* integer frames: dxh_map_coordinates, dxh_tvd_upload_signal_typed */
#define DXH_U16 3

/* which Jacobi neighbour sum to use */
#define DXH_NSUM_ALL_AXES 0 /* reference behaviour inside split_bregman_tvd: every array axis
                               (utils/linear_solvers/jacobi.py:41), i.e. colour channels are coupled */
#define DXH_NSUM_SPATIAL 1  /* only the `dim` spatial axes (what H1_regularization's per-channel loop
                               computes, restoration/h1_regularization.py:54-67) */

/* run flags */
#define DXH_FLAG_ISOTROPIC 1u /* split_bregman_tvd(isotropic=True), split_bregman_tvd.py:136-140 */
#define DXH_FLAG_NO_FUSED 2u  /* force the three-kernel path (rhs / Jacobi sweep / shrink) */
#define DXH_FLAG_NO_DPP 4u    /* fused kernel: use ds_bpermute lane exchange instead of DPP */
#define DXH_FLAG_EXPLICIT 16u /* keep the split variables d, b as explicit planes (needed by dxh_tvd_adapt_ell);
                                 the context stays in this mode */
#define DXH_FLAG_ASYNC 8u     /* enqueue only: no event timing, no stream synchronisation (loop_ms = NaN);
                                 ignored when eps or jacobi_tol need host-side tests */

const char* dxh_last_error(void);
int dxh_version(void);

/* number of visible HIP devices (0 when there is none; never an error) */
int dxh_device_count(int* count);

/*
 * Device-resident image arrays.  Wherever an entry point below takes a caller-owned IMAGE array (img, out, src, dst,
 * base, canvas, x0, rhs ...: the arrays whose size is that of an image), the pointer may be a host array or a dense
 * device array obtained here: every copy of caller data resolves its direction from the pointers.  Operators can then be
 * chained without leaving HBM, the way the reference chains them in host memory (image/image.py:188-199,
 * signals/models/combinedmodel.py:29-52).  Small parameter tables (matrices, patch records, tap tables) are host memory.
 */
int dxh_mem_alloc(int device, size_t bytes, void** ptr);
int dxh_mem_free(int device, void* ptr);
/* synchronous copy between any two of {host array, device array} */
int dxh_mem_copy(int device, void* dst, const void* src, size_t bytes);
/* dst = factor * src on dense DEVICE arrays of n elements, the factor rounded to the array's type first -- what NumPy does
 * for ``python_float * array`` (H1_regularization's right-hand side omega * img, restoration/h1_regularization.py:73, for
 * images that are in HBM already). */
int dxh_scale(int device, int dtype, size_t n, const void* src, double factor, void* dst);
/*
 * The TV contexts and dxh_mem_free keep the planes / arrays they release in a cache (exact-size re-use; at most 1/16 of the device memory,
 * DXH_POOL_MB overrides, 0 turns it off): hipFree costs 0.1 - 0.3 ms per plane and waits for the device, a whole call
 * releases 10 - 30 planes.  This hands everything cached back to the HIP runtime; *bytes (optional) = what was released.
 */
int dxh_mem_trim(size_t* bytes);

typedef struct dxh_tvd_ctx* dxh_tvd_t;

typedef struct dxh_tvd_desc {
    int32_t struct_size; /* = sizeof(dxh_tvd_desc) */
    int32_t device;      /* HIP device ordinal */
    int32_t dtype;       /* DXH_F32 / DXH_F64 */
    int32_t ndim;        /* 2 or 3 array axes */
    int32_t dim;         /* 2 or 3 spatial axes, dim <= ndim */
    int32_t nsum;        /* DXH_NSUM_* */
    int64_t n0, n1, n2;  /* array shape; n2 = 1 when ndim == 2.  n0 = rows of THIS slab, halo rows included */
    int32_t top_is_boundary;    /* 1: row 0 is the image boundary (replicate ghosts); 0: halo, rows decay */
    int32_t bottom_is_boundary; /* 1: row n0-1 is the image boundary */
    void* stream;        /* optional hipStream_t to run on (e.g. the current torch stream); NULL = own */
} dxh_tvd_desc;

/*
 * Device-resident state of one split-Bregman problem (or of one row slab of it).
 * Replaces the arrays set up in restoration/split_bregman_tvd.py:93-101.
 */
int dxh_tvd_create(dxh_tvd_t* ctx, const dxh_tvd_desc* desc);
int dxh_tvd_destroy(dxh_tvd_t ctx);

/*
 * Coefficients mu (TV weight), omega (mass weight), ell (Bregman regularisation);
 * split_bregman_tvd.py:18-20,62-68 and jacobi.py:57-72.  Each may be a scalar or a
 * host array with the image's shape and dtype (pointer non-NULL).  has_ell = 0
 * means ell = 2 * mu (split_bregman_tvd.py:62-63).  Scalars are doubles; all
 * derived constants (diag, diag/ell, mu/ell) are formed in double and rounded to
 * the image dtype once, like NumPy's weak Python scalars.
 */
int dxh_tvd_set_coeffs(dxh_tvd_t ctx, double mu, const void* mu_arr, double omega, const void* omega_arr,
                       int has_ell, double ell, const void* ell_arr);
/*
 * The same with coefficient arrays of either float type (DXH_F32 / DXH_F64 per array): an array whose type is not the
 * image's is converted on the device (round to nearest, like ndarray.astype) -- a float64 porosity map as the weight of a
 * float32 image (presets/workflows/restoration.py:118-140) costs a 2x larger upload instead of a pass over it on the host.
 * Arrays may be host or device pointers.  In a numpy2 context (dxh_tvd_set_promotion) a FLOAT32 mu array next to scalar omega
 * and ell becomes the threshold NumPy forms for it -- the float32 quotient mu / float32(ell) (split_bregman_tvd.py:144 with a
 * float32 array and a Python float), widened -- on the device; a float64 mu array is divided in float64.
 */
int dxh_tvd_set_coeffs_typed(dxh_tvd_t ctx, double mu, const void* mu_arr, int mu_dtype, double omega, const void* omega_arr,
                             int omega_dtype, int has_ell, double ell, const void* ell_arr, int ell_dtype);

/*
 * Upload the image (split_bregman_tvd.py:99) and, optionally, a warm start
 * x0 = (u0, d0, b0) (split_bregman_tvd.py:93-97).  d0/b0 use the reference's
 * layout: shape (*img.shape, dim), split index last.  u0 = NULL starts from img,
 * d0 = b0 = NULL from zeros.  Coefficients must have been set before.
 */
int dxh_tvd_upload(dxh_tvd_t ctx, const void* img, const void* u0, const void* d0, const void* b0);

typedef struct dxh_tvd_result {
    int32_t iters_done;   /* outer iterations executed */
    int32_t converged;    /* 1 if the eps criterion stopped the loop */
    double rel_increment; /* last ||u_new - u|| / ||img||  (NaN when eps was not requested) */
    double loop_ms;       /* device time of the iteration loop (hipEvents on the context stream) */
    int64_t launches;     /* kernels launched inside the timed loop */
} dxh_tvd_result;

/*
 * The iteration loop, split_bregman_tvd.py:129-175 with Jacobi.__call__
 * (jacobi.py:74-133) as inner solver.  eps < 0 or NaN: no convergence test.
 * jacobi_tol < 0 or NaN: fixed number of sweeps.  norm_scale: the caller may
 * pass ||img||_2 of the WHOLE image when this context holds only a slab
 * (<= 0: computed from this context's image).
 */
int dxh_tvd_run(dxh_tvd_t ctx, int n_iter, int jacobi_maxiter, double jacobi_tol, double eps, double norm_scale,
                uint32_t flags, dxh_tvd_result* result);

/* download u and (optionally) the split variables d, b in the reference layout (*img.shape, dim). */
int dxh_tvd_download(dxh_tvd_t ctx, void* u, void* d, void* b);

/*
 * The whole of split_bregman_tvd (split_bregman_tvd.py:14-177) for a HOST image with a fixed iteration count (eps and
 * the inner tolerance off, Jacobi inner solver, no warm start): host array in, host array out, PCIe overlapped with the
 * iterations.  The image is cut into row bands with n_iter * jacobi_maxiter extra rows above and one more below on
 * their open sides -- the dependence cone of n_iter iterations (jacobi.py:43-54, split_bregman_tvd.py:84-89,144) -- so
 * that every band reproduces the whole-image result on its own rows bit for bit; band b + 1 goes up and band b - 1
 * comes down while band b iterates (three streams, an uploader thread).  desc describes the WHOLE image (dim == 2,
 * both boundary flags set; desc->stream is ignored); coefficient arrays, if given, have the image's shape.
 * nbands: 0 = chosen from a transfer / compute / halo-recompute model, 1 = no banding; *nbands_used (optional) reports it.
 */
int dxh_tvd_solve_banded(const dxh_tvd_desc* desc, const void* img, void* out, double mu, const void* mu_arr, double omega,
                         const void* omega_arr, int has_ell, double ell, const void* ell_arr, int n_iter,
                         int jacobi_maxiter, uint32_t flags, int nbands, int* nbands_used);

/*
 * The same for a host image of another float type than the band contexts' (converted on the device, band by band), with
 * scalar coefficients and one Jacobi sweep.  promotion = DXH_PROMOTION_NUMPY2: host_dtype DXH_F32, desc->dtype DXH_F64 --
 * the reference's NumPy >= 2 dtype flow for a float32 image (dxh_tvd_set_promotion), bit-identical to the one-piece run.
 */
int dxh_tvd_solve_banded_typed(const dxh_tvd_desc* desc, const void* img, void* out, int host_dtype, int promotion, double mu,
                               double omega, int has_ell, double ell, int n_iter, uint32_t flags, int nbands,
                               int* nbands_used);

/*
 * ConcentrationAnalysis fusion (multi_image_analysis/concentrationanalysis.py:230-310): the elementwise stages
 * either side of the restoration run on the device, so that a frame goes up once (colour) and the
 * concentration comes down once (scalar).  The context is the scalar (H, W) TVD problem.
 *   set_base            baseline image (H, W, channels) in the context dtype, or NULL (concentrationanalysis.py:328-364)
 *   set_cleaning_filter threshold filter (H, W) or NULL (:381-395)
 *   upload_signal       signal = model_pre(clean(reduce(diff(probe, base)))) becomes the context's image
 *                       (what dxh_tvd_upload does with a host image); coefficients must be set before
 *   download_model      out = model_post(u)
 * diff_option: 0 plain, 1 positive, 2 negative, 3 absolute.  reduction (signals/reduction/monochromatic.py:66-95):
 * 0 none (channels == 1), 1 red, 2 green, 3 blue, 4 red+green, 5 gray (0.299, 0.587, 0.114 in float32, like
 * cv2.COLOR_RGB2GRAY), 6 negative-key.  Model ops are applied in order: kind 0: a*x + b (LinearModel,
 * signals/models/linearmodel.py:140-149), kind 1: a*x (ScalingModel :63), kind 2: clip to [a, b] with
 * flags bit0 = has lower, bit1 = has upper bound (ClipModel, clipmodel.py:87-106).
 */
typedef struct dxh_model_op {
    int32_t kind;
    int32_t flags;
    double a, b;
} dxh_model_op;

int dxh_tvd_set_base(dxh_tvd_t ctx, const void* base, int channels);
int dxh_tvd_set_cleaning_filter(dxh_tvd_t ctx, const void* filter);
int dxh_tvd_upload_signal(dxh_tvd_t ctx, const void* probe, int channels, int diff_option, int reduction,
                          const dxh_model_op* pre_ops, int n_pre);
int dxh_tvd_download_model(dxh_tvd_t ctx, void* out, const dxh_model_op* post_ops, int n_post);
/*
 * The same with frames in the type they are stored in.  The reference subtracts the baseline in NumPy's promotion of the two
 * arrays and reduces in that type; only the cleaning filter (float64 by construction, concentrationanalysis.py:154-197) widens
 * the signal.  So a float32 frame against a float32 baseline goes to the device as float32 even when the context is float64,
 * and integer frames (which the reference converts with img_as(float): x * (1 / 255) resp. (1 / 65535) in float64, with a
 * warning, concentrationanalysis.py:241-246) go as the bytes they are.  base_dtype: DXH_F32 / DXH_F64, not wider than the
 * context; probe_dtype: DXH_F32 / DXH_F64 / DXH_U8 / DXH_U16, DXH_EINVAL if probe - baseline would be wider than the context.
 */
int dxh_tvd_set_base_typed(dxh_tvd_t ctx, const void* base, int channels, int base_dtype);
int dxh_tvd_upload_signal_typed(dxh_tvd_t ctx, const void* probe, int probe_dtype, int channels, int diff_option,
                                int reduction, const dxh_model_op* pre_ops, int n_pre);

/*
 * Row-slab halo plumbing for multi-GPU runs (no reference counterpart: the
 * reference is single-process).  Rows [row0, row0 + nrows) of the three evolving
 * fields u, dub_0.. are copied to / from a caller-provided DEVICE buffer of
 * dxh_tvd_halo_bytes() bytes on the context stream.
 */
int dxh_tvd_halo_bytes(dxh_tvd_t ctx, int64_t nrows, size_t* bytes);
int dxh_tvd_pack_rows(dxh_tvd_t ctx, int64_t row0, int64_t nrows, void* dst_device);
int dxh_tvd_unpack_rows(dxh_tvd_t ctx, int64_t row0, int64_t nrows, const void* src_device);

/*
 * Multi-GPU TV sweep: one process per GPU, the image cut into row slabs (axis 0), halo rows exchanged between slab
 * neighbours only -- point to point, as xGMI is; no collective on the data path.  No reference counterpart (the
 * reference is single-process); what an exchange must deliver follows from the data dependences of one outer
 * iteration: the Jacobi neighbour sum reads u one row above and below per sweep (utils/linear_solvers/jacobi.py:43-54),
 * the right-hand side differences ell (b0 - d0) across the row above (restoration/split_bregman_tvd.py:84-89) and the
 * split update takes D+ of the new iterate across the row below (:144).  With halo_top_rows = k * jacobi_maxiter rows
 * above and halo_bottom_rows = k * jacobi_maxiter + 1 below the owned rows, k iterations run between two exchanges.
 *
 * A communicator is either RCCL (grouped ncclSend / ncclRecv straight from / into the state planes' row blocks on a
 * dedicated comm stream; librccl is dlopen'ed on first use) or host-staged through caller-provided callbacks (tests
 * and rehearsals with several ranks on one GPU, which RCCL refuses).  Bootstrap: rank 0 calls dxh_comm_unique_id and
 * hands the 128 bytes to the other ranks by any means (darsia_amd/parallel.py uses a file next to MASTER_PORT).
 */
typedef struct dxh_comm_s* dxh_comm_t;
#define DXH_COMM_ID_BYTES 128

int dxh_comm_unique_id(void* id /* DXH_COMM_ID_BYTES */);
int dxh_comm_create_rccl(dxh_comm_t* comm, int device, const void* id, int rank, int world);
/* host-staged transport: all n_ops transfers of one call progress together (is_send[i] ? send bufs[i] to peers[i] :
 * receive from peers[i] into bufs[i]); messages between a pair of ranks match in call order.  Return 0 on success. */
typedef int (*dxh_comm_exchange_fn)(void* user, int n_ops, const int* peers, const int* is_send, void* const* bufs,
                                    const size_t* bytes);
/* in-place all-reduce of n doubles over all ranks; op 0 = sum, 1 = max */
typedef int (*dxh_comm_allreduce_fn)(void* user, double* vals, int n, int op);
int dxh_comm_create_host(dxh_comm_t* comm, int device, int rank, int world, dxh_comm_exchange_fn exchange,
                         dxh_comm_allreduce_fn allreduce, void* user);
int dxh_comm_destroy(dxh_comm_t comm);
/* blocking all-reduce of n <= 16 host doubles (op 0 = sum, 1 = max); dxh_comm_barrier = all-reduce of one value */
int dxh_comm_allreduce(dxh_comm_t comm, double* vals, int n, int op);
int dxh_comm_barrier(dxh_comm_t comm);
/* any pointer may be NULL; kind 0 = RCCL, 1 = host-staged */
int dxh_comm_info(dxh_comm_t comm, int* rank, int* world, int* kind, int64_t* exchanges, int64_t* bytes_sent);

/*
 * Make `ctx` one slab of a distributed problem.  The context's n0 rows are [top halo | owned rows | bottom halo];
 * a side with desc.top_is_boundary / bottom_is_boundary set has no halo.  All ranks pass the same halo row counts.
 * A communicator destroyed while contexts are still attached stays alive until the last of them is destroyed (no exchange
 * may be started after dxh_comm_destroy).  A one-rank communicator with halos on BOTH sides makes the slab its own
 * neighbour: the problem is periodic along axis 0 (exercises exchange and overlap on a single GPU).
 */
int dxh_tvd_attach_comm(dxh_tvd_t ctx, dxh_comm_t comm, int64_t halo_top_rows, int64_t halo_bottom_rows);
/* refresh the halo rows from the neighbours now (enqueued behind everything the context has in flight) */
int dxh_tvd_exchange(dxh_tvd_t ctx);
#define DXH_FLAG_NO_OVERLAP 32u /* dxh_tvd_run_slab: ordinary launches, exchange in stream order behind them (A/B, tests) */
#define DXH_FLAG_CONTINUE 64u   /* dxh_tvd_run_slab: more iterations will follow -- if the halos are used up at the end
                                   of this call, refresh them now, overlapped with the last launch */
/*
 * dxh_tvd_run for a slab: runs n_iter outer iterations, exchanging halos whenever they are used up.  The launch before
 * an exchange puts the row chunks next to the slab interfaces first; each of their waves counts itself on a device counter
 * when its rows are in memory, and the exchange (stream wait on that counter, gather, one grouped ncclSend / ncclRecv per
 * direction, scatter) runs on the context's second stream while the interior chunks of that launch are still computing.
 * The launches after an exchange follow it on that second stream: a slab context alternates between its stream and one
 * of its own, so work a caller enqueues on desc.stream is NOT ordered against it -- dxh_tvd_sync / dxh_tvd_download are the
 * synchronisation points.  eps (>= 0): global stopping test of split_bregman_tvd.py:161-175 -- one all-reduce of the
 * owned rows' squared increment per iteration (and of ||img||^2 once); all ranks stop at the same iteration.
 * result->loop_ms is NaN with DXH_FLAG_ASYNC.
 */
int dxh_tvd_run_slab(dxh_tvd_t ctx, int n_iter, int jacobi_maxiter, double eps, uint32_t flags, dxh_tvd_result* result);

/*
 * The reference's own dtype flow for a FLOAT32 image under NumPy >= 2 ("numpy2" promotion).  With scalar omega / ell the
 * reference iterates u in float64 (utils/linear_solvers/jacobi.py:101-114: const_diag_scaled is an np.float64) while d,
 * b and the right-hand side stay float32 (restoration/split_bregman_tvd.py:84-89,99-101,129-146); a float64 weight map
 * next to a float32 image -- what presets/workflows/restoration.py:117-148 builds -- also makes the shrinkage threshold a
 * float64 array.  The native float32 flow of this library stays within the north-star tolerance of that (rtol 1e-4) for
 * ell = 2 mu but drifts past it for ell >= 0.5 after ~100 iterations; this mode reproduces the reference bit for bit.
 * Use: create the context with dtype DXH_F64, call dxh_tvd_set_promotion(ctx, DXH_PROMOTION_NUMPY2) BEFORE
 * dxh_tvd_set_coeffs, upload the float32 image with dxh_tvd_upload_typed, run with dxh_tvd_run (dim = 2,
 * jacobi_maxiter = 1, no tolerance, anisotropic, scalar omega and ell; mu scalar or a float64 array), fetch the float32
 * result with dxh_tvd_download_typed (convert_dtype(img_iter, float32), split_bregman_tvd.py:177).
 * DXH_PROMOTION_MU_IS_THRESHOLD: the mu array passed to dxh_tvd_set_coeffs already holds mu / ell (a float32 mu array:
 * NumPy forms the quotient in float32 before it meets the float64 iterate; the caller does that and widens).
 */
#define DXH_PROMOTION_NATIVE 0
#define DXH_PROMOTION_NUMPY2 1
#define DXH_PROMOTION_MU_IS_THRESHOLD 2
int dxh_tvd_set_promotion(dxh_tvd_t ctx, int mode);
int dxh_tvd_upload_typed(dxh_tvd_t ctx, const void* img, int img_dtype);
int dxh_tvd_download_typed(dxh_tvd_t ctx, void* u, int out_dtype);

/*
 * Warm start x0 = (img0, d0, b0) of a numpy2 context (split_bregman_tvd.py:92-96 with float32 arrays): img float32; u0
 * float32 or float64 (u0_dtype), or NULL for img itself; d0, b0 float32 arrays of shape (*img.shape, dim).  Host or device
 * pointers (a chain of warm-started calls can keep its state in device memory).
 * dxh_tvd_download_split_np2 hands d and b back as the reference leaves them in the caller's arrays (float32; either
 * pointer may be NULL).  Scalar omega and ell.
 */
int dxh_tvd_upload_warm_np2(dxh_tvd_t ctx, const void* img, const void* u0, int u0_dtype, const void* d0, const void* b0);
int dxh_tvd_download_split_np2(dxh_tvd_t ctx, void* d, void* b);

/*
 * Where the time of a slab context went (diagnostics; no reference counterpart).  Every exchange is bracketed by three
 * events on the stream it runs on: wait_signal_ms sums the time that stream idled until the interface waves of the
 * signalling launch had released their rows (0 for exchanges in plain stream order), exchange_ms the time from there to
 * the end of the scatter of the received rows (gather + grouped send / receive -- which includes waiting for the
 * neighbour -- + scatter).  Synchronises the context.  wait_timeouts > 0: the polling wait (devices without
 * hipStreamWaitValue32) gave up after DXH_SLAB_WAIT_TIMEOUT_MS (default 10 s); dxh_tvd_sync / dxh_tvd_run_slab /
 * dxh_tvd_download return DXH_EHIP from then on until the next dxh_tvd_upload.
 */
typedef struct dxh_slab_stats {
    int32_t struct_size;          /* sizeof(dxh_slab_stats), set by the caller */
    int32_t wait_mode;            /* 1 hipStreamWaitValue32, 2 polling kernel, 0 no communicator attached */
    int64_t launches;             /* sweep launches enqueued by dxh_tvd_run_slab */
    int64_t signalling_launches;  /* of these: interface chunks first, followed by an overlapped exchange */
    int64_t exchanges;            /* halo exchanges enqueued */
    int64_t exchanges_timed;      /* exchanges that contributed to the two sums below */
    int64_t wait_timeouts;
    double wait_signal_ms;
    double exchange_ms;
} dxh_slab_stats;
int dxh_tvd_slab_stats(dxh_tvd_t ctx, dxh_slab_stats* out);
/* sum over rows [row0, row0+nrows) of (u - u_prev)^2 of the last iteration and of img^2 (for a global eps test) */
int dxh_tvd_partial_norms(dxh_tvd_t ctx, int64_t row0, int64_t nrows, double* sq_increment, double* sq_img);
/* adaptive regularisation, split_bregman_tvd.py:149-158: ell <- 1 / max(|grad u|_1, 1e-12) per element for the rhs
 * and the shrinkage threshold (the Jacobi diagonal keeps its value, as the reference's cached one does).
 * Requires the context to be in explicit mode (run with DXH_FLAG_EXPLICIT). */
int dxh_tvd_adapt_ell(dxh_tvd_t ctx);
int dxh_tvd_sync(dxh_tvd_t ctx);
/* energy functional of split_bregman_tvd.py:71-77 (verbose output) of the current (0) or previous (1) iterate */
int dxh_tvd_energy(dxh_tvd_t ctx, int which, double* value);

/*
 * Diagnostic.  The scalar-coefficient float pipelines divide by the two wave-invariant Jacobi divisors
 * (jacobi.py:104-106) through reciprocals; whether ONE residual correction already gives the correctly rounded
 * quotient for every numerator is decided per coefficient set by trying all 2^23 float mantissas on the device.
 * This entry runs that test for divisors d0, d1 with reciprocals r0, r1 (0 = RN(1/d)); *bad gets bit 0 / bit 1
 * set when some x is NOT rounded correctly for d0 / d1.
 */
int dxh_div3_probe(int device, float d0, float r0, float d1, float r1, unsigned* bad);
/*
 * Diagnostic.  The float kernels take the square root of the isotropic shrinkage (split_bregman_tvd.py:107-126) and of
 * Chambolle's gradient norm through a five-instruction reciprocal-square-root chain where that equals the correctly
 * rounded square root, and through the IEEE sequence elsewhere.  This entry compares the function the kernels use with
 * sqrtf on ALL 2^32 float bit patterns: *mismatches must come back 0; *chain_inputs = how many inputs took the chain.
 */
int dxh_sqrt_probe(int device, unsigned long long* mismatches, unsigned long long* chain_inputs);
/*
 * Diagnostic.  Chambolle's update divides the components of p by one per-element norm (skimage _denoise.py:379-384, as
 * called from restoration/tvd.py:77-83).  The float kernels form ONE reciprocal, v_rcp_f32 refined by a Newton step, and
 * refine each quotient with exact residuals; that is the correctly rounded quotient provided the reciprocal is RN(1/d).
 * This entry checks it for every float d in [2^-100, 2^100] (*tested = how many d) and then compares the whole
 * three-quotient routine, screening included, with the IEEE division on 2.6e10 pseudo-random operand sets (all classes of
 * floats, image-like magnitudes, values around the screen's thresholds): *mismatches must come back 0.
 */
int dxh_rcp_probe(int device, unsigned long long* mismatches, unsigned long long* tested);
/* The isotropic shrink factor max(s - k, 0) / (s + 1e-18) of the float kernels (dxh_common.h: iso_s_fac_fast): the
 * rsq-seeded reciprocal against RN(1 / s) for EVERY float x = s^2 in [2^-60, 2^60) (mismatches_rcp), and the whole routine
 * against the IEEE expressions on ~8.6e9 pseudo-random (x, k) pairs (mismatches_fac). */
int dxh_isofac_probe(int device, unsigned long long* mismatches_rcp, unsigned long long* mismatches_fac, unsigned long long* tested);

/*
 * One-shot Jacobi solve  mass * x - diffusion * laplace(x) = rhs  (jacobi.py:74-133),
 * used by Jacobi.__call__ and H1_regularization (h1_regularization.py:54-70).
 * x0, rhs, out: arrays of shape (n0, n1, n2); rhs = NULL stands for rhs = mass * x0 computed on the device (scalar
 * mass; H1 regularisation's right-hand side, h1_regularization.py:73).  Returns the iterate count in
 * *sweeps_done (with tol the reference returns the iterate BEFORE the one that
 * met the tolerance; kept).
 */
int dxh_jacobi_solve(int device, int dtype, int ndim, int dim, int nsum, int64_t n0, int64_t n1, int64_t n2,
                     const void* x0, const void* rhs, void* out, double mass, const void* mass_arr,
                     double diffusion, const void* diffusion_arr, double h, int maxiter, double tol,
                     int* sweeps_done);

/*
 * The same solve in the reference's dtype flow for FLOAT32 operands under NumPy >= 2 (scalar coefficients): rhs / diag stays
 * float32, the first sweep sums the neighbours of the float32 x0 in float32 and divides by the np.float64 const_diag_scaled
 * (jacobi.py:101-114), every later sweep runs in float64.  x0, rhs (or NULL = mass * x0): float32 arrays, host or device;
 * out: out_dtype DXH_F64 (what Jacobi.__call__ returns in the reference) or DXH_F32 (convert_dtype(out, float32),
 * h1_regularization.py:76).
 */
int dxh_jacobi_solve_np2(int device, int ndim, int dim, int nsum, int64_t n0, int64_t n1, int64_t n2, const void* x0,
                         const void* rhs, void* out, int out_dtype, double mass, double diffusion, double h, int maxiter,
                         double tol, int* sweeps_done);

/*
 * Multigrid and conjugate-gradient solvers for  mass * x - diffusion * laplace(x) = rhs
 * (utils/linear_solvers/mg.py:13-270, cg.py:16-59; plug-in contract utils/linear_solvers/solver.py:12-72).
 * Arrays have exactly `dim` axes, shape (n0, n1[, n2]), dense host memory.  The object is stateful like the
 * reference's: MG keeps its smoother's cached diagonal (scalar coefficients) or its coefficient arrays, which
 * every V-cycle restricts and re-prolongates in place (heterogeneous; mg.py:124-181).
 *   MG: depth, smoother_iterations, maxiter, tol (< 0: none), heterogeneous (constructor-time flag, mg.py:72-75)
 *   CG: maxiter, tol (required), h (<= 0: none); stop rule |r| <= tol |b| of scipy.sparse.linalg.cg (SciPy < 1.14)
 */
typedef struct dxh_solver_s* dxh_solver_t;

#define DXH_SOLVER_JACOBI 0
#define DXH_SOLVER_MG 1
#define DXH_SOLVER_CG 2

typedef struct dxh_solver_desc {
    int32_t struct_size; /* sizeof(dxh_solver_desc) */
    int32_t device;
    int32_t dtype; /* DXH_F32 / DXH_F64 */
    int32_t dim;   /* 2 or 3 */
    int64_t n0, n1, n2;
    int32_t kind; /* DXH_SOLVER_MG / DXH_SOLVER_CG */
    int32_t depth;
    int32_t smoother_iterations;
    int32_t maxiter;
    int32_t heterogeneous;
    int32_t reserved;
    double tol;
    double h;
    void* stream; /* hipStream_t or NULL */
} dxh_solver_desc;

int dxh_solver_create(dxh_solver_t* out, const dxh_solver_desc* desc);
int dxh_solver_destroy(dxh_solver_t s);
/* Solver.update_params (solver.py:40-58): scalars or dense host arrays (an array wins over the scalar) */
int dxh_solver_set_params(dxh_solver_t s, double mass, const void* mass_arr, double diffusion, const void* diffusion_arr);
/* Solver.__call__(x0, rhs): *iters = V-cycles / CG iterations performed */
int dxh_solver_solve(dxh_solver_t s, const void* x0, const void* rhs, void* out, int32_t* iters);
/* current coefficient arrays of a heterogeneous MG (the reference's mutated solver.mass_coeff / diffusion_coeff) */
int dxh_solver_get_params(dxh_solver_t s, void* mass_out, void* diffusion_out);

/*
 * Use MG / CG instead of the Jacobi sweeps inside dxh_tvd_run (split_bregman_tvd.py:25,63-67,133: the `solver`
 * argument).  kind = DXH_SOLVER_JACOBI restores the default.  Scalar omega / ell only (as with the reference,
 * whose MG() default object is not heterogeneous).
 */
int dxh_tvd_set_solver(dxh_tvd_t ctx, int kind, int depth, int smoother_iterations, int maxiter, double tol);

/*
 * skimage.restoration.denoise_tv_bregman as DarSIA calls it (restoration/tvd.py:85-101, utils/segmentation.py:117-120;
 * third-party arithmetic, restated in oracle/bregman_gs_numpy.py): split Bregman with an in-place Gauss-Seidel
 * u-update over a (rows, cols, dims) image whose channels are independent 2-D problems with a joint stop test.
 * image, out: dense host arrays of the given dtype (the caller applies img_as_float).  *iters = sweeps done.
 */
int dxh_tv_bregman_gs(int device, int dtype, int64_t rows, int64_t cols, int64_t dims, const void* image, void* out,
                      double weight, int max_iter, double eps, int isotropic, int* iters, double* rmse);

/*
 * Finite differences (utils/derivatives.py:16-59): mode 0 = backward_diff
 * (x[i+1]-x[i], last 0), 1 = forward_diff (x[i]-x[i-1], first 0); h <= 0: no scaling.
 */
int dxh_diff(int device, int dtype, int64_t n0, int64_t n1, int64_t n2, int axis, int mode, double h,
             const void* in, void* out);

/*
 * Chambolle's TV projection algorithm: the arithmetic of DarSIA's default TVD() method, which the
 * reference obtains from skimage.restoration.denoise_tv_chambolle(img, weight, max_num_iter, eps)
 * (restoration/tvd.py:77-83; no channel axis, so an (H, W, 3) image is a 3-D volume, ndim = 3).
 * Stops when |E_prev - E| < eps * E_0; *iters_done = iterations completed before the stop.
 */
int dxh_chambolle(int device, int dtype, int ndim, int64_t n0, int64_t n1, int64_t n2, const void* img, void* out,
                  double weight, double eps, int max_num_iter, int* iters_done);

/*
 * Piecewise perspective warp: the per-patch  cv2.warpPerspective(patch.astype(float32), P, (w_eff, h_eff),
 * INTER_LINEAR)  +  canvas[roi] += tmp[...]  loop of PiecewisePerspectiveTransform.find_and_warp
 * (corrections/shape/piecewiseperspective.py:71-133) in one launch.  The caller prepares one record per
 * patch (homography already inverted, in double); patches are applied in table order.
 */
typedef struct dxh_warp_patch {
    double minv[9];       /* inverse of P: tile pixel (x, y, 1) -> source patch pixel, row-major */
    int32_t sr0, sc0;     /* source patch origin (row, col) in the base image */
    int32_t sh, sw;       /* source patch size */
    int32_t ox, oy;       /* tile origin in the canvas (col, row) = origin_eff */
    int32_t w_eff, h_eff; /* tile size passed to warpPerspective */
    int32_t nx, ny;       /* tile part added: canvas[oy:oy+ny, ox:ox+nx] += tile[:ny, :nx] */
    int32_t bw0;          /* OpenCV's internal block width for (w_eff, h_eff) */
    int32_t pad;
} dxh_warp_patch;

int dxh_warp_patches(int device, int H, int W, int C, const float* base, const dxh_warp_patch* patches, int npatch,
                     float* canvas);
/*
 * The same on images in the type they are stored in (dtype: DXH_F32 / DXH_F64 / DXH_U8 / DXH_U16, for base and canvas alike):
 * the reference reads every patch as patch.astype(float32), sums the warped patches on a float32 canvas and returns
 * canvas.astype(dtype) (piecewiseperspective.py:121-133) -- the kernel converts on load and on store, so a uint8 frame is
 * 1 byte per value over PCIe each way instead of 4.  Integer stores truncate towards zero through int32 and wrap, like
 * NumPy's cast on x86-64.
 */
int dxh_warp_patches_typed(int device, int H, int W, int C, int dtype, const void* base, const dxh_warp_patch* patches,
                           int npatch, void* canvas);

/*
 * Diagnostic: the projective scale 32 / W of dxh_warp_patches (cv2.warpPerspective's double-precision division,
 * piecewiseperspective.py:121-131) is computed without the scaling helpers of the compiler's division when the exponent of W
 * is far from both ends; this runs that code and the plain division side by side on the device over 3 x 2^29 pseudo-random
 * denominators (all exponents; the range a homography produces; the boundaries of the screen) and counts results that
 * differ in any bit.  Expected: 0.
 */
int dxh_warp_div_probe(int device, unsigned long long seed, unsigned long long* mismatches, unsigned long long* tested);

/*
 * Translation of every patch of img_dst relative to the same patch of img_src by phase correlation -- the patch estimator
 * darsia_amd's registration ships where the reference calls OpenCV (ORB features + RANSAC, translationestimator.py, out of reach:
 * SURVEY.md section 2; the registration takes any estimator with that signature).  Algorithm: oracle/phase_correlation_numpy.py.
 * Images (H, W) or (H, W, 3) of one dtype (DXH_F32 / DXH_F64 / DXH_U8 / DXH_U16), host or device; rois: npatch x
 * (r0, r1, c0, c1), the pixel rectangle of a patch in both images; shifts: npatch x (dx, dy) in pixels; ok: 1 where the
 * estimate is accepted (patch at least 8 x 8, |shift| <= max_shift_fraction of the patch size).  Patches longer than 8192 pixels
 * are read in cells of 2 x 2 (4 x 4 ...) pixels.
 */
int dxh_phase_correlation(int device, int H, int W, int C, int dtype, const void* img_src, const void* img_dst, int npatch,
                          const int32_t* rois, double max_shift_fraction, double* shifts, int32_t* ok);
/*
 * The same with a peak-quality test: an estimate is accepted only if its shift is finite and the correlation peak reaches
 * min_peak_ratio * sqrt(ln N / N), N = cells of the padded transform -- sqrt(ln N / N) is the level the maximum of the
 * correlation surface of two UNRELATED patterns reaches (unit-modulus cross-power spectrum), a flat patch gives 0, two equal
 * patterns 1.  min_peak_ratio = 0 switches the test off (NaN shifts are rejected in any case).  peaks (npatch doubles, or
 * NULL): the peak value per patch.  (The reference's ORB + RANSAC estimator reports such patches as not intact and lets the
 * spline interpolate over them: translationanalysis.py:225-275.)
 */
int dxh_phase_correlation_q(int device, int H, int W, int C, int dtype, const void* img_src, const void* img_dst, int npatch,
                            const int32_t* rois, double max_shift_fraction, double min_peak_ratio, double* shifts, int32_t* ok,
                            double* peaks);
/*
 * Thin-plate-spline evaluation of a fitted scipy.interpolate.RBFInterpolator (kernel 'thin_plate_spline',
 * degree 1, epsilon 1) at n points -- TranslationAnalysis.translation / DiffeomorphicImageRegistration.
 * displacement (translationanalysis.py:285-301, imageregistration.py:265-306).  centers (p, 2), coeffs
 * (p + 3, dout), powers (3, 2).  points (n, 2) or, if NULL, the grid  (gx0 + gsx*(i % grid_w), gy0 + gsy*(i / grid_w)).
 */
int dxh_tps_eval(int device, int p, int dout, const double* centers, const double* coeffs, const double* shift,
                 const double* scale, const int* powers, int64_t n, const double* points, int grid_w, double gx0,
                 double gy0, double gsx, double gsy, double* out);
/*
 * Two such splines with scalar values on the pixel grid at once -- the x and y translation maps of a
 * TranslationAnalysis (translationanalysis.py:285-294) share their patch centres and differ in the boundary points -- as
 * one pass over the union of their centres.  Point (col, row) = (gx0 + gsx * col, gy0 + gsy * row); coeffs0 / coeffs1 have
 * p0 + 3 / p1 + 3 entries; out is (grid_h * grid_w, 2) float64 (host or device) and receives
 * (out_scale0 * s0, out_scale1 * s1): with the voxel sizes (vx, -vy) that IS DiffeomorphicImageRegistration.displacement()
 * (imageregistration.py:265-306) without any pass over the field on the host.
 */
int dxh_tps_eval_pair(int device, int p0, const double* centers0, const double* coeffs0, const double* shift0,
                      const double* scale0, const int* powers0, int p1, const double* centers1, const double* coeffs1,
                      const double* shift1, const double* scale1, const int* powers1, int64_t grid_h, int64_t grid_w,
                      double gx0, double gy0, double gsx, double gsy, double out_scale0, double out_scale1, double* out);

/*
 * out = img_as_float(img) @ A + b, optional clip to [0, 1], float32 -- AffineBalance.apply_balance and the
 * tail of ColorCorrection.correct_array (corrections/color/colorbalance.py:186-196, colorcorrection.py:388-394).
 * in_dtype: 0 float32, 1 float64, 2 uint8 (/255), 3 uint16 (/65535); img (npix, 3); A row-major (3, 3).
 */
int dxh_affine_color(int device, int in_dtype, int64_t npix, const void* img, const double* A, const double* b,
                     int clip, float* out);

/*
 * scipy.ndimage.map_coordinates(channel, grid, order) with mode "constant" for every channel of an (H, W, C) image --
 * CurvatureCorrection._transform_image (corrections/shape/curvature.py:921-957).  dtype: 0 float32, 1 float64,
 * 2 uint8, 3 uint16 (output has the input's dtype, integers rounded like SciPy); rows / cols: the two rows of the
 * reference's `grid` (n coordinates each, float32 or float64); order 0 or 1; out (n, C).
 */
int dxh_map_coordinates(int device, int dtype, int64_t H, int64_t W, int C, const void* img, int coord_dtype,
                        const void* rows, const void* cols, int64_t n, int order, double cval, void* out);

/*
 * cv2.resize of every channel of an (H, W, C) image -- Resize.__call__ (restoration/resize.py:150-176).  The host
 * builds OpenCV's tap tables (darsia_amd/resizing.py); the device applies them with OpenCV's order of operations.
 * mode 0: nearest (xi, yi = source indices); 1: bilinear (xi, yi = first tap, xw, yw = float weight of the second
 * tap); 2: INTER_AREA with integer scales isx, isy; 3: INTER_AREA with weight tables in CSR form (xptr has DW + 1
 * entries, xsrc / xalpha one per table entry; likewise y).  dtype: 0 float32, 1 float64, 2 uint8.
 */
typedef struct dxh_resize_plan {
    int32_t struct_size; /* sizeof(dxh_resize_plan) */
    int32_t mode;
    int32_t isx, isy;
    const int32_t* xi;
    const float* xw;
    const int32_t* yi;
    const float* yw;
    const int32_t* xptr;
    const int32_t* xsrc;
    const float* xalpha;
    const int32_t* yptr;
    const int32_t* ysrc;
    const float* yalpha;
} dxh_resize_plan;

int dxh_resize(int device, int dtype, int64_t H, int64_t W, int C, const void* src, int64_t DH, int64_t DW,
               const dxh_resize_plan* plan, void* dst);

#ifdef __cplusplus
}
#endif
#endif /* DARSIA_HIP_H */
