/* vinterp.h - C-ABI of libvinterp.so, the MI355X (gfx950) implementation of the per-timestep
 * fit + evaluate hot path of amisr/volumetricinterp.
 *
 * The reference is pure Python and has NO C/FFI boundary: its only plug-in seam is
 * the Python `Model` class contract (volumetricinterp/models/sphharmlag.py:11-15).  This header is
 * therefore the boundary *underneath* the reference's Python surface; each entry point cites the
 * reference function whose arithmetic it replaces.  The binding a maintainer adds on the reference
 * side is a ctypes stub - see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (VI_OK) or a negative vi_status; nothing throws across the ABI;
 *     vi_last_error() returns a thread-local message for the last failure.
 *   - one vi_ctx per GPU, used from one host thread at a time.  All work is enqueued on the
 *     context's own HIP stream; entry points taking device pointers are ASYNCHRONOUS on that
 *     stream (call vi_ctx_sync), entry points taking host pointers stage H2D/D2H and return when
 *     the result is in the caller's buffer.
 *   - "d_" parameters are device pointers obtained from vi_dmalloc on the same context;
 *     "h_" parameters are caller-owned, C-contiguous host buffers.  All reals are IEEE fp64.
 *   - a failed timestep is reported through its outputs (NaN rows), never through the status.
 */
#ifndef VINTERP_H
#define VINTERP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): vi_warm_solve_f64 / vi_basis_solve_f64 / vi_warm_rebase_f64 take a trailing d_sweeps pointer,
 * vi_warm_chi2_one_f64 writes THREE doubles to h_chi2, vi_brent_warm_supported added, h_rebase of vi_brent_warm_f64 /
 * vi_brent_host_one_f64 is 10 doubles (jump rule).  _lib.py refuses any other version. */
#define VI_ABI_VERSION 2

typedef enum vi_status {
    VI_OK = 0,
    VI_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unsupported order) */
    VI_ERR_HIP = -2,         /* HIP runtime failure */
    VI_ERR_ROCBLAS = -3,
    VI_ERR_ROCSOLVER = -4,
    VI_ERR_NOMEM = -5,
    VI_ERR_UNSUPPORTED = -6,
    VI_ERR_RCCL = -7
} vi_status;

typedef struct vi_ctx vi_ctx;       /* one per GPU: device id, stream, rocBLAS handle, workspaces */
typedef struct vi_model vi_model;   /* device-resident tables of one basis model */

enum { VI_MODEL_SPHHARMLAG = 1, VI_MODEL_RADBASFUN = 2 };

/* One group of spherical-cap-harmonic degrees nu_l = nv_l + v0 that share the fractional part v0
 * (sphharmlag.py:101-115 `nu`).  The Legendre functions P_nu^m(cos theta) of scipy.special.lpmv
 * (called at sphharmlag.py:141) are produced by an upward degree recurrence seeded at degrees
 * v0+m and v0+m+1, normalised so that one step is  p_j = x p_{j-1} - c[j][m] p_{j-2}. */
typedef struct vi_sph_group {
    double v0;               /* fractional part of the degrees of this group (0 -> closed-form seeds) */
    int32_t nvmax;           /* largest integer part in the group */
    int32_t nterms;          /* length of the 2F1 series tables (0 when v0 == 0) */
    const int32_t* pick;     /* [nvmax+2]   l whose degree has integer part j, or -1 */
    const double* c;         /* [(nvmax+2) x maxl]  normalised recurrence coefficient c[j][m] (one degree beyond
                                nvmax: grad_basis evaluates lpmv(m, nu+1, x), sphharmlag.py:177) */
    const double* seed_pref; /* [2 x maxl]  prefactor of the two seeds of chain m (v0 != 0) */
    const double* seed_q;    /* [2 x maxl x nterms]  term ratios of the 2F1 series (v0 != 0) */
} vi_sph_group;

/* Model description (host side).  Built by volumetricinterp_amd.models.<NAME>.Model from the same
 * INI keys the reference parses (sphharmlag.py:65-75, radbasfun.py:65-78). */
typedef struct vi_model_desc {
    int32_t kind;            /* VI_MODEL_* */
    int32_t nbasis;          /* N = maxk*maxl^2 (sphharmlag.py:59) or NUMGRIDPNT^3 (radbasfun.py:60) */
    /* --- sphharmlag --- */
    int32_t maxk, maxl;
    double rot_cos, rot_sin; /* cos/sin(theta0) of the Rodrigues rotation, sphharmlag.py:346,353 */
    double rot_kx, rot_ky;   /* rotation axis k = (kx, ky, 0), sphharmlag.py:349 */
    double earth_radius;     /* RE, sphharmlag.py:9 */
    int32_t ngroups;
    const vi_sph_group* groups;
    const double* coef_scale;/* [maxl^2] per (l, signed m): Kvm(nu_l,|m|) (sphharmlag.py:305-321) x the
                                lpmv negative-order factor x the chain normalisation */
    const double* coef_scale1;/* [maxl^2] the same constant for degree nu_l + 1 (gradient basis only) */
    const double* nu;        /* [maxl] the degrees nu_l (gradient basis only) */
    /* --- radbasfun --- */
    const double* centers;   /* [N x 3] ECEF metres (radbasfun.py:59) */
    double eps;              /* radbasfun.py:72 */
} vi_model_desc;

/* ---- context / memory --------------------------------------------------------------------- */
int  vi_abi_version(void);
int  vi_device_count(int* count);
int  vi_ctx_create(int device, vi_ctx** out);
void vi_ctx_destroy(vi_ctx* ctx);
int  vi_ctx_sync(vi_ctx* ctx);
const char* vi_last_error(void);

int  vi_dmalloc(vi_ctx* ctx, size_t bytes, void** d_ptr);
int  vi_dfree(vi_ctx* ctx, void* d_ptr);
int  vi_h2d(vi_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);   /* synchronous */
int  vi_d2h(vi_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* synchronous */
int  vi_dmemset(vi_ctx* ctx, void* d_ptr, int value, size_t bytes);
/* A download that does not hold the context's stream (the reference has no counterpart: its covariances, interpolate.py:466,
 * are host arrays already).  vi_d2h_side_mark: after the launches that produce the data, from the launching thread.
 * vi_d2h_side: blocking, from any host thread - copies on a second stream once the marked work has finished, while the
 * context's stream runs on.  The source must stay untouched until vi_d2h_side returns. */
int  vi_d2h_side_mark(vi_ctx* ctx);
int  vi_d2h_side(vi_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* HIP-event timing on the context's stream (bench.py measures kernels on the stream they run on) */
/* free / total memory of the context's device (hipMemGetInfo) */
int  vi_mem_info(vi_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int  vi_timer_start(vi_ctx* ctx);
int  vi_timer_stop_ms(vi_ctx* ctx, double* ms);   /* synchronises on the stop event */

/* ---- model -------------------------------------------------------------------------------- */
int  vi_model_create(vi_ctx* ctx, const vi_model_desc* desc, vi_model** out);
void vi_model_destroy(vi_model* model);

/* ---- basis assembly: replaces Model.basis (sphharmlag.py:118-145, radbasfun.py:83-112) ------
 * A[p*ld_p + n*ld_n] for p < P, n < N.  (ld_p, ld_n) = (N, 1) gives the reference's row-major
 * (P, N) layout; (1, P) gives the N x P layout the fit kernels consume (coalesced stores). */
int  vi_basis_f64(vi_model* model, int64_t P, const double* d_lat, const double* d_lon,
                  const double* d_alt, double* d_A, int64_t ld_p, int64_t ld_n);
/* gradient basis: replaces Model.grad_basis (sphharmlag.py:148-184; next row N1 - the reference never calls it).
 * G[p*ld_p + c*ld_c + n*ld_n], c = 0,1,2 = components along z, theta, phi.  sphharmlag only. */
int  vi_grad_basis_f64(vi_model* model, int64_t P, const double* d_lat, const double* d_lon,
                       const double* d_alt, double* d_G, int64_t ld_p, int64_t ld_c, int64_t ld_n);
/* gradient of the fitted parameter, out[q*3 + c] = sum_n grad_basis[q][c][n] * C[n] (c = z, theta, phi components as
 * sphharmlag.py:148-184 defines them): the contraction the reference's Estimate.__call__ advertises as `calcgrad`
 * (estimate.py:125-147, dead code there); the (Q, 3, N) array is never formed. */
int  vi_eval_grad_f64(vi_model* model, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                      const double* d_C, double* d_out);
/* standard error of the fitted parameter, err[q] = sqrt(a_q^T dC a_q) with a_q the basis row of point q and dC the
 * (N x N) coefficient covariance of the record: the `calcerr` output of the same dead branch (estimate.py:139-145) */
int  vi_eval_err_f64(vi_model* model, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                     const double* d_dC, double* d_out);
/* model coordinates (z, theta, phi) of sphharmlag.py:324-359 `transform_coord`; ECEF x,y,z for RBF */
int  vi_transform_f64(vi_model* model, int64_t P, const double* d_lat, const double* d_lon,
                      const double* d_alt, double* d_c0, double* d_c1, double* d_c2);

/* ---- fused evaluation: replaces Estimate.__call__ (estimate.py:110-123) ---------------------
 * out[t*Q + q] = sum_n basis_n(q) * C[t*N + n]; NaN where the point fails the convex-hull test
 * (estimate.py:119-121, :153-178) when hull_eq != NULL: a point is inside iff
 * max_f (hull_eq[f][0..2] . ecef(q) + hull_eq[f][3]) <= hull_tol.  The basis matrix is never
 * materialised.  hull_eq is what scipy.spatial.ConvexHull(...).equations holds (unit normals, metres); the test is exact
 * in fp64 for any list - a reduced-precision pass only sorts out the points further from the surface than its own error
 * bound, which assumes nothing but scales with the longest normal (lists of much shorter normals run slower, not wrong). */
int  vi_eval_f64(vi_model* model, int64_t Q, const double* d_lat, const double* d_lon,
                 const double* d_alt, int64_t T, const double* d_C,
                 const double* d_hull_eq, int32_t F, double hull_tol, double* d_out);
/* Many timesteps on ONE grid (BASELINE configs[3]: a GPU's share of 10 000 timesteps, all on the same 256^3 grid - Estimate.__call__
 * (estimate.py:110-123) once per timestep in the reference, which rebuilds the basis of the grid every time): the basis matrix
 * of the grid is assembled once and kept in HBM, and every batch of timesteps is one matrix product.
 *   vi_eval_basis_f64     d_Y[n*Q + q] = basis_n(q), N x Q doubles (19 GB at N = 144 on 256^3); with hull_eq != NULL the
 *                         entries of a point that fails the hull test (as in vi_eval_f64) are NaN
 *   vi_eval_resident_f64  out[t*Q + q] = sum_n d_Y[n*Q + q] * C[t*N + n]  - NaN outside the hull through the NaN of d_Y,
 *                         NaN for a timestep whose coefficients are NaN (a failed fit), as vi_eval_f64 gives them.
 * Agrees with vi_eval_f64 to rounding (the sum over n is taken in the library's order; tests/test_gpu_eval_resident.py). */
int  vi_eval_basis_f64(vi_model* model, int64_t Q, const double* d_lat, const double* d_lon, const double* d_alt,
                       const double* d_hull_eq, int32_t F, double hull_tol, double* d_Y);
int  vi_eval_resident_f64(vi_model* model, int64_t Q, int64_t T, const double* d_Y, const double* d_C, double* d_out);
/* Arithmetic of the Legendre degree recurrences inside vi_eval_f64 for this model: 0 = fp64 (default; the reference
 * computes in float64 throughout, sphharmlag.py:118-145), 1 = fp32 chains with everything else in fp64 - the variant
 * BASELINE configs[4] sweeps against the 1e-6 tolerance.  Orders with an fp32 kernel: (MAXL, MAXK) = (6,4), (2,8), (12,8);
 * others return VI_ERR_UNSUPPORTED from vi_eval_f64 while the flag is set. */
int  vi_model_set_eval_precision(vi_model* model, int32_t chain_f32);
/* device time (ms) of the evaluation kernel launches of the last vi_eval_f64 / vi_eval_resident_f64 call on this context, from
 * HIP events recorded on the context's stream around them (the preparation kernels are excluded).  The events are recorded
 * only while vi_ctx_set_eval_timing(ctx, 1) is in force (default: off - the pair costs a 0.2 ms call about 7 us);
 * vi_eval_kernel_ms fails with VI_ERR_ARG while it is off or before a call has been timed. */
int  vi_ctx_set_eval_timing(vi_ctx* ctx, int32_t on);
int  vi_eval_kernel_ms(vi_ctx* ctx, double* ms);
/* device time of the eigen-solve kernel launches (the kernel the fit spends its time in), from one HIP event pair
 * per launch on the context's stream.  enable = 1 starts / resets recording, 0 stops it, -1 only reads; the
 * outputs (each may be NULL) describe the period since the last reset: launches, systems solved, number of
 * launches whose duration is included (the 2048 most recent at most), their summed and maximal duration (ms). */
int  vi_solve_timing(vi_ctx* ctx, int enable, int64_t* launches, int64_t* systems, int64_t* timed,
                     double* total_ms, double* max_ms);
/* Jacobi rounds (one round = one pass of the LDS-resident matrix through the registers) summed over the systems of
 * all launches since vi_solve_timing(enable = 1); read it BEFORE the next vi_solve_timing call with enable >= 0,
 * which resets it.  The unit the LDS roofline of the eigen-solve kernel is priced in (bench.py). */
int  vi_solve_rounds(vi_ctx* ctx, int64_t* rounds);
/* host-pointer form of the same call (what Estimate.__call__ uses): chunked, coordinates up on one stream while densities
 * come down on another; the staging buffers live on the model.  h_out rows have length Q. */
int  vi_eval_f64_host(vi_model* model, int64_t Q, const double* h_lat, const double* h_lon,
                      const double* h_alt, int64_t T, const double* h_C,
                      const double* h_hull_eq, int32_t F, double hull_tol, double* h_out);

/* page-locked host memory: arrays handed to vi_eval_f64_host from it move at the full rate of the link in both directions
 * at once (pageable arrays go through the runtime's own staging) */
int  vi_host_alloc(size_t bytes, void** out);
int  vi_host_free(void* p);

/* ---- fit: replaces Interpolate.eval_C (interpolate.py:432-469) -----------------------------
 * Normal equations for T records sharing one basis matrix (records differ only in W and b;
 * dropped points carry W = 0, b = 0 - algebraically the row removal of interpolate.py:516-520):
 *   AWA[t] = A^T diag(W[t]) A  (interpolate.py:456),  y[t] = A^T (W[t] .* b[t])  (interpolate.py:458)
 * d_At is the N x P basis (ld_n = P layout of vi_basis_f64). */
int  vi_normal_eq_f64(vi_ctx* ctx, int64_t T, int64_t P, int32_t N, const double* d_At,
                      const double* d_W, const double* d_b, double* d_AWA, double* d_y);

/* X[i] = AWA[rec[i]] + alpha[i] * R   for a batch of B (record, alpha) pairs (interpolate.py:460-461);
 * d_rec == NULL means rec[i] = i; d_AWA == NULL accumulates a further penalty term, X[i] += alpha[i] * R. */
int  vi_form_system_f64(vi_ctx* ctx, int64_t B, int32_t N, const double* d_AWA, const int32_t* d_rec,
                        const double* d_alpha, const double* d_R, double* d_X);

/* Minimum-norm solve of the symmetric systems X[i] c = y[rec[i]] with singular values below
 * rcond * sigma_max treated as zero - scipy.linalg.lstsq / LAPACK gelsd at interpolate.py:462
 * (rcond = eps), via a batched symmetric eigendecomposition (sigma_i = |lambda_i|).
 * d_X is destroyed.  Optionally also H = pinv(X) with its own cutoff (interpolate.py:465). */
int  vi_solve_trunc_f64(vi_ctx* ctx, int64_t B, int32_t N, double* d_X, const double* d_y,
                        const int32_t* d_rec, double rcond, double* d_C, int32_t* d_rank,
                        double pinv_rcond, double* d_H /* may be NULL */);

/* chi2[i] = sum_p W[rec[i]][p] (A[p,:] . C[i] - b[rec[i]][p])^2   (interpolate.py:258-259, :569) */
int  vi_chi2_f64(vi_ctx* ctx, int64_t B, int64_t P, int32_t N, const double* d_At, const double* d_C,
                 const int32_t* d_rec, const double* d_W, const double* d_b, double* d_chi2);

/* dC[t] = H[t] AWA[t] H[t]   (interpolate.py:466) */
int  vi_cov_f64(vi_ctx* ctx, int64_t T, int32_t N, const double* d_H, const double* d_AWA, double* d_dC);

/* ---- warm-started regularisation-parameter search ----------------------------------------------
 * Brent's iterates (interpolate.py:214) solve nearly identical systems X(alpha) = AWA[rec] + alpha R.
 * vi_warm_prepare_f64 decomposes X(alpha0[i]) of B records with eigenvectors (returning the solution C
 * at alpha0 like vi_solve_trunc_f64) and stores V, D1 = V^T AWA V, D2 = V^T R V, yt = V^T y per record
 * (slot i); vi_warm_solve_f64 then solves (D1 + alpha D2) c' = yt for B (slot, alpha) pairs with the same
 * truncation rule and returns C = V c'.  Used for the chi^2 search only; the final coefficients of a
 * record always come from vi_solve_trunc_f64 on the untransformed system. */
int  vi_warm_prepare_f64(vi_ctx* ctx, int64_t B, int32_t N, const double* d_AWA, const int32_t* d_rec,
                         const double* d_alpha0, const double* d_R, const double* d_y, double rcond,
                         double* d_C, int32_t* d_rank, double* d_V, double* d_D1, double* d_D2, double* d_yt);
int  vi_warm_solve_f64(vi_ctx* ctx, int64_t B, int32_t N, const double* d_D1, const double* d_D2,
                       const double* d_yt, const double* d_V, const int32_t* d_slot, const double* d_alpha,
                       double rcond, double* d_C, int32_t* d_rank, int32_t* d_sweeps /* may be NULL */);
/* d_sweeps (here and in vi_basis_solve_f64 / vi_warm_rebase_f64): Jacobi sweeps each system took, or the sweep cap + 1
 * (vi_max_sweeps() + 1) when the cap ended the iteration before it converged - such a solution must not decide a sign
 * of chi^2 - nu (interpolate.py:193-203); the caller solves that system again from X(alpha) itself. */
int  vi_max_sweeps(void);

/* The two phases of vi_warm_prepare_f64 as separate calls: vi_decompose_f64 forms and decomposes X(alpha0[i]) of B records
 * (solution to d_C as vi_solve_trunc_f64 gives it) and leaves the rotation logs - vi_rotation_log_bytes(N) per system - and
 * the rounds they hold in buffers of the caller; vi_warm_finish_f64 turns B consecutive logs into V, D1, D2, yt.  A record
 * fitted alone decomposes the middle of every candidate bracket in the launch of its bracket walk and finishes only the one
 * the walk points at. */
size_t vi_rotation_log_bytes(int32_t N);
int  vi_decompose_f64(vi_ctx* ctx, int64_t B, int32_t N, const double* d_AWA, const int32_t* d_rec,
                      const double* d_alpha0, const double* d_R, const double* d_y, double rcond, double* d_C,
                      int32_t* d_rank, void* d_log, int32_t* d_nround);
int  vi_warm_finish_f64(vi_ctx* ctx, int64_t B, int32_t N, const void* d_log, const int32_t* d_nround,
                        const double* d_AWA, const int32_t* d_rec, const double* d_R, const double* d_y,
                        double* d_V, double* d_D1, double* d_D2, double* d_yt);

/* vi_warm_solve_f64 for B (slot, record, alpha) triples that also MOVES each slot's rotated system to alpha: the
 * eigenvectors of the rotated system come out of the rotation log, V <- V Vw, and D1, D2, yt are formed again from the
 * untransformed AWA[rec], R, y[rec].  Brent's late iterates (interpolate.py:214) sit within 1e-3 decades of each other;
 * from a basis that close a warm solve takes 1-3 sweeps instead of 6-13.  A slot may appear once per call.  The first
 * nplain triples are plain warm solves (no re-basing) that share the eigen-solve launch of the others. */
int  vi_warm_rebase_f64(vi_ctx* ctx, int64_t B, int64_t nplain, int32_t N, const double* d_AWA, const double* d_R,
                        const double* d_y, const int32_t* d_rec, const int32_t* d_slot, const double* d_alpha, double rcond,
                        double* d_V, double* d_D1, double* d_D2, double* d_yt, double* d_C, int32_t* d_rank,
                        int32_t* d_sweeps /* may be NULL */);

/* out[t] = the alpha below which alpha R vanishes from AWA[t] + alpha R in floating point (alpha |R_ij| under a quarter
 * of eps |AWA_ij| in every element): the systems of the bracket walk (interpolate.py:186-203) below it are one and the
 * same matrix and are solved once. */
int  vi_reg_floor_f64(vi_ctx* ctx, int64_t T, int32_t N, const double* d_AWA, const double* d_R, double* d_out);

/* The bracket walk (interpolate.py:186-203) in SHARED bases: the walk systems of the records of one geometry resemble each
 * other decade by decade, so they are solved in the eigenbasis of a reference system of the same decade - d_V / d_D2 as
 * vi_warm_prepare_f64 leaves them for a reference record (mean weights of the batch), one slot per decade.  For B
 * (record, basis slot, alpha) triples: (V^T AWA[rec] V + alpha D2) c' = V^T y[rec] with the truncation rule of
 * vi_solve_trunc_f64, C = V c'.  1-4 Jacobi sweeps per system instead of 8-24.  Its chi^2 only decides the signs of the
 * walk (the search asks for the bracket ends, and for values near the target, again from cold solves), so the iteration ends
 * at |a_pq| <= 1e-6 sqrt|a_pp a_qq| (VINTERP_WALK_TOL; csrc/vi_fit.hip walk_tolerance). */
int  vi_basis_solve_f64(vi_ctx* ctx, int64_t B, int32_t N, const double* d_AWA, const double* d_y,
                        const int32_t* d_rec, const int32_t* d_basis, const double* d_alpha, const double* d_V,
                        const double* d_D2, double rcond, double* d_C, int32_t* d_rank, int32_t* d_sweeps /* may be NULL */);

/* One root-finder iterate of ONE record in a single call (single-record latency path): vi_warm_solve_f64 for
 * (slot, alpha) followed by vi_chi2_f64 against record `rec`; the scalars travel as kernel arguments and the only
 * synchronisation is the read-back into h_chi2 (host, THREE doubles: chi^2, an internal word, and in the low 32 bits of
 * the third the sweep count of the solve - cap + 1 when it did not converge).  d_scratch: N + 8 doubles of device memory. */
int  vi_warm_chi2_one_f64(vi_ctx* ctx, int32_t N, int64_t P, const double* d_D1, const double* d_D2,
                          const double* d_yt, const double* d_V, int32_t slot, double alpha, double rcond,
                          const double* d_At, int32_t rec, const double* d_W, const double* d_b,
                          double* d_scratch, double* h_chi2);

/* The whole root-finder phase of the search (scipy.optimize.brentq at interpolate.py:214 on chi^2(10^x) - nu) for ntask
 * records in ONE launch: a workgroup owns a record from its unit bracket [xa, xb] (values fa, fb) to its root, every
 * function value computed in place in the rotated system of the record's slot (vi_warm_prepare_f64) by the code of
 * vi_warm_solve_f64 + vi_chi2_f64, alpha = vi_exp10 (below).  Device arrays of length ntask in and out: root and other end
 * of the final bracket (log10 alpha), iterations / function calls as brentq counts them, status (low byte; bits 8 and up
 * count how often the record's rotated system was re-based) 0 = converged, 2 = a solve
 * was ended by the sweep cap (run that record's iteration on the host), 3 = more than 100 iterations.  Bit for bit the
 * result of the host-driven iteration from the same rotated system.
 * The rotated systems (d_D1, d_D2, d_yt, d_V, by slot) are MOVED next to the root as the host path does
 * (vi_warm_rebase_f64) by the rule in h_rebase (host, 10 doubles: number of thresholds, up to four thresholds in decades
 * between consecutive abscissae, then the late move: after so many requests, within so many decades, enabled; then the early
 * end on a JUMP of chi^2(alpha) - a sign change without a root, where an eigenvalue of X(alpha) crosses the truncation cut,
 * which brentq bisects to 2e-12 in 40-60 values: [8] width in decades of log10 alpha, [9] fraction of nu - the iteration ends
 * once the bracket is narrower than the width while both ends miss nu by more than that fraction; zeros: brentq's own end)
 * - from d_AWA, d_y (by record) and d_R. */
int  vi_brent_warm_f64(vi_ctx* ctx, int64_t ntask, int32_t N, int64_t P, double* d_D1, double* d_D2, double* d_yt,
                       double* d_V, const double* d_AWA, const double* d_R, const double* d_y, const double* h_rebase,
                       const double* d_At, const double* d_W, const double* d_b,
                       const int32_t* d_rec, const int32_t* d_slot, const double* d_xa, const double* d_xb,
                       const double* d_fa, const double* d_fb, const double* d_nu, double rcond, double* d_root,
                       double* d_other, int32_t* d_iters, int32_t* d_funcalls, int32_t* d_status);
/* 1 when vi_brent_warm_f64 serves order N with P data points per record (in-LDS Jacobi range, and the chi^2 partial sums of
 * a record - one per 256 points - fit beside its system in a CU's LDS: P <= ~2 million at N = 144), else 0: drive Brent's
 * iteration from the host then (vi_warm_solve_f64 + vi_chi2_f64), which has no such limit - the reference accepts any P
 * (interpolate.py:214). */
int  vi_brent_warm_supported(int32_t N, int64_t P);
/* The same root-finder phase for ONE record, driven from the host in C (the loop of a record fitted alone: one dependent
 * launch chain per function value, vi_warm_chi2_one_f64, and vi_warm_rebase_f64 + vi_chi2_f64 for the value that moves the
 * rotated system) - brentq's state machine and the re-basing rule are the ones vi_brent_warm_f64 runs on the device, compiled
 * for the host: same requests, same values, same root, without an interpreter between a value and the next request.
 * h_out (host, 6 doubles): root, other end of the final bracket, iterations, function calls, status (0 / 2 / 3 as above),
 * re-basings.  d_scratch: N + 8 doubles of device memory. */
int  vi_brent_host_one_f64(vi_ctx* ctx, int32_t N, int64_t P, double* d_D1, double* d_D2, double* d_yt, double* d_V,
                           const double* d_AWA, const double* d_R, const double* d_y, const double* h_rebase,
                           const double* d_At, const double* d_W, const double* d_b, int32_t rec, int32_t slot,
                           double xa, double xb, double fa, double fb, double nu, double rcond, double* d_scratch,
                           double* h_out);
/* out[i] = 10^x[i] (host arrays) in plain IEEE operations - the routine the device-side iteration uses, so that host and
 * device form alpha = 10^(log10 alpha) identically (interpolate.py:216, :246) */
int  vi_exp10_f64(const double* x, double* out, int64_t n);

/* ---- generalised cross validation: replaces the loop of Interpolate.gcvobjfunct (interpolate.py:332-351) ----
 * For ONE record (d_AWA N x N, d_y N, d_W / d_b P) and one alpha: res[i] = W_p (a_p . C_(-p) - b_p)^2 for the np
 * data points p = pidx[i], where C_(-p) is the regularised truncated solution of the fit that leaves point p
 * out - a rank-one down-date of the normal equations, np eigen-solves in one batch.  Synchronous. */
int  vi_gcv_terms_f64(vi_ctx* ctx, int64_t np, int64_t P, int32_t N, const double* d_At, const int32_t* d_pidx,
                      const double* d_AWA, const double* d_y, const double* d_W, const double* d_b,
                      double alpha, const double* d_R, double rcond, double* d_res);

/* ---- multi-GPU: one broadcast of shared parameters over RCCL (xGMI) -----------------------------
 * Records are independent (interpolate.py:511), so the fit/evaluate path has no collective; the caller shards
 * records across one process per GPU.  The only exchange is this broadcast of the parameters every rank
 * needs (beam geometry, regularisation matrices, hull facets).  Rank 0 obtains a 128-byte id, ships it over
 * its own control channel, then every rank calls vi_rccl_init.  RCCL is dlopen'ed on first use. */
int  vi_rccl_unique_id(char* out128);
int  vi_rccl_init(vi_ctx* ctx, int nranks, int rank, const char* id128);
int  vi_rccl_bcast_f64(vi_ctx* ctx, double* d_buf, int64_t count, int root);
int  vi_rccl_destroy(vi_ctx* ctx);

/* Diagnostic: eigenvalues (unsorted) of B symmetric N x N systems by the in-LDS Jacobi kernel that
 * vi_solve_trunc_f64 uses, and the sweeps each system needed.  d_X is rescaled in place. */
int  vi_eigvals_f64(vi_ctx* ctx, int64_t B, int32_t N, double* d_X, double* d_lam, int32_t* d_sweeps);

/* Stage-test entry of the pre-conditioner of the cold solves (csrc/vi_qr.hip): one column-pivoted Householder QR step
 * X P = Q R of each of B symmetric systems, returned as the similar matrix X1 = Q^T X Q (same eigenvalues; the matrix
 * the Jacobi kernel then iterates on), y1 = Q^T y and the explicit Q (Q[:, j] contiguous).  The solve it serves is
 * scipy.linalg.lstsq at interpolate.py:462; d_X is only read. */
int  vi_qr_similarity_f64(vi_ctx* ctx, int64_t B, int32_t N, const double* d_X, const double* d_y, double* d_X1,
                          double* d_y1, double* d_Q);

#ifdef __cplusplus
}
#endif
#endif /* VINTERP_H */
