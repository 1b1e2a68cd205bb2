// Fit entry points: batched normal equations, regularised minimum-norm solves, chi^2, covariance.
// Reference: Interpolate.eval_C (volumetricinterp/interpolate.py:432-469) and the chi^2 objective
// (interpolate.py:255-259).  The basis matrix is shared by all records (geometry is per file, not per
// record), so T records become one strided-batched contraction instead of T x (#alpha) einsums.
#include "vi_common.h"
#include "vi_gemm_device.h"

#include <rocsolver/rocsolver.h>

#include <cstdlib>

#define VI_ROCSOLVER(call)                                                                      \
    do {                                                                                        \
        rocblas_status s_ = (call);                                                             \
        if (s_ != rocblas_status_success) {                                                     \
            vi_set_error("%s:%d: %s -> rocsolver status %d", __FILE__, __LINE__, #call, (int)s_); \
            return VI_ERR_ROCSOLVER;                                                            \
        }                                                                                       \
    } while (0)

size_t vi_jacobi_lds_bytes(int N);
size_t vi_jacobi_log_bytes(int N, int max_sweeps);
bool vi_jacobi_supported(int N);
int vi_jacobi_solve(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_scl, const double* d_y,
                    const int* d_rec, double rcond, double* d_C, int* d_rank, void* d_log, int max_sweeps,
                    int* d_sweeps, double* d_lam, int lam_raw, int* d_nround, double abs_floor, int64_t log_stride = 0,
                    double conv_tol = 0.0);
bool vi_jacobi_vectors_supported(int N);
int vi_jacobi_vectors(vi_ctx* c, int64_t B, int N, const void* d_log, int max_sweeps, const int* d_nround, double* d_V,
                      int64_t log_stride = 0);
// K3p (vi_qr.hip): X1 = Q^T X Q, y1 = Q^T y by one column-pivoted Householder QR step; back-transformations c <- Q c
bool vi_qr_supported(int N);
size_t vi_qr_hh_bytes(int N);
int vi_qr_precond(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_y, const int* d_rec, double* d_X1,
                  double* d_y1, double* d_hh, double* d_scr, int64_t hh_stride = 0);
int vi_qr_back_vec(vi_ctx* c, int64_t B, int N, const double* d_hh, double* d_C, int64_t hh_stride = 0);
int vi_qr_back_mat(vi_ctx* c, int64_t B, int N, const double* d_hh, double* d_V, int64_t hh_stride = 0);

namespace {

inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// B[t][n][p] = At[n][p] * W[min(t, tvalid - 1)][p]   (entries beyond tvalid pad the last GEMM group)
__global__ void k_scale_rows(int64_t P, int N, const double* __restrict__ At, const double* __restrict__ W,
                             double* __restrict__ B, int tvalid)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y, t = blockIdx.z;
    if (p >= P) return;
    const int ts = t < tvalid ? t : tvalid - 1;
    B[((int64_t)t * N + n) * P + p] = At[(int64_t)n * P + p] * W[(int64_t)ts * P + p];
}

__global__ void k_mul(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] * b[i];
}

// X[i] = AWA[rec[i]] + alpha[i] * R   (AWA == nullptr: X[i] += alpha[i] * R, for further penalty terms)
__global__ void k_form_system(int NN, const double* __restrict__ AWA, const int* __restrict__ rec,
                              const double* __restrict__ alpha, const double* __restrict__ R, double* __restrict__ X)
{
    const int64_t i = blockIdx.x;
    const int64_t r = rec ? rec[i] : i;
    const double a = R ? alpha[i] : 0.0;
    for (int e = threadIdx.x; e < NN; e += blockDim.x) {
        double v = AWA ? AWA[r * NN + e] : X[i * NN + e];
        if (R) v = fma(a, R[e], v);
        X[i * NN + e] = v;
    }
}

// X[i] = D1[slot[i]] + alpha[i] * D2[slot[i]]   (warm-started search: both terms are per-record)
// slot1 null: D1[i] (X may then be D1 itself); slot2: where D2 of system i lives
__global__ void k_form_pair(int NN, const double* D1, const double* __restrict__ D2, const int* __restrict__ slot1,
                            const int* __restrict__ slot2, const double* __restrict__ alpha, double* X)
{
    const int64_t i = blockIdx.x;
    const int64_t w1 = slot1 ? slot1[i] : i;
    const int64_t w = slot2[i];
    const double a = alpha[i];
    for (int e = threadIdx.x; e < NN; e += blockDim.x) X[i * NN + e] = fma(a, D2[w * NN + e], D1[w1 * NN + e]);
}

// out[i][k] = sum_r V[w][k*N + r] * v[src][r]      (V^T v; V in LAPACK eigenvector layout)
__global__ void k_vt_vec(int N, const double* __restrict__ V, const int* __restrict__ vslot, const double* __restrict__ v,
                         const int* __restrict__ vidx, double* __restrict__ out)
{
    const int64_t i = blockIdx.x;
    const double* Vi = V + (int64_t)(vslot ? vslot[i] : i) * N * N;
    const double* vi = v + (int64_t)(vidx ? vidx[i] : i) * N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int k = wave; k < N; k += nw) {
        double acc = 0.0;
        for (int r = lane; r < N; r += 64) acc = fma(Vi[(int64_t)k * N + r], vi[r], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) out[i * N + k] = acc;
    }
}

// out[i][r] = sum_k V[w][k*N + r] * c[i][k]        (V c)
__global__ void k_v_vec(int N, const double* __restrict__ V, const int* __restrict__ vslot, const double* __restrict__ cin,
                        double* __restrict__ out)
{
    extern __shared__ double shc[];
    const int64_t i = blockIdx.x;
    const double* Vi = V + (int64_t)(vslot ? vslot[i] : i) * N * N;
    for (int k = threadIdx.x; k < N; k += blockDim.x) shc[k] = cin[i * N + k];
    __syncthreads();
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < N; ++k) acc = fma(Vi[(int64_t)k * N + r], shc[k], acc);
        out[i * N + r] = acc;
    }
}

// dst[slot[i]] = src[i]   (NN elements each)
__global__ void k_scatter_mat(int NN, const double* __restrict__ src, const int* __restrict__ slot, double* __restrict__ dst)
{
    const int64_t i = blockIdx.x;
    const double* s = src + i * NN;
    double* d = dst + (int64_t)slot[i] * NN;
    for (int e = threadIdx.x; e < NN; e += blockDim.x) d[e] = s[e];
}

// out[slot[i]][k] = sum_r V[i][k*N + r] * v[rec[i]][r]     (k_vt_vec with the result scattered to the slots)
__global__ void k_vt_vec_slot(int N, const double* __restrict__ V, const double* __restrict__ v, const int* __restrict__ rec,
                              const int* __restrict__ slot, double* __restrict__ out)
{
    const int64_t i = blockIdx.x;
    const double* Vi = V + i * N * N;
    const double* vi = v + (int64_t)rec[i] * N;
    double* o = out + (int64_t)slot[i] * N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int k = wave; k < N; k += nw) {
        double acc = 0.0;
        for (int r = lane; r < N; r += 64) acc = fma(Vi[(int64_t)k * N + r], vi[r], acc);
        for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_down(acc, o2);
        if (lane == 0) o[k] = acc;
    }
}

// Leave-one-out systems of generalised cross validation (interpolate.py:333-349): deleting data point p from a
// record is a rank-one down-date of its normal equations,
//   X_p = AWA - W_p a_p a_p^T + alpha R,   y_p = y - W_p b_p a_p,      a_p = column p of the N x P basis.
__global__ void k_form_loo(int N, int64_t P, const double* __restrict__ At, const int* __restrict__ pidx,
                           const double* __restrict__ AWA, const double* __restrict__ yv, const double* __restrict__ W,
                           const double* __restrict__ b, double alpha, const double* __restrict__ R,
                           double* __restrict__ X, double* __restrict__ yl)
{
    extern __shared__ double sha[];            // a_p [N]
    const int64_t i = blockIdx.x;
    const int64_t p = pidx[i];
    for (int n = threadIdx.x; n < N; n += blockDim.x) sha[n] = At[(int64_t)n * P + p];
    __syncthreads();
    const double w = W[p], bw = W[p] * b[p];
    const int NN = N * N;
    for (int e = threadIdx.x; e < NN; e += blockDim.x) {
        const int r = e / N, c = e - r * N;
        double v = fma(-w * sha[r], sha[c], AWA[e]);
        if (R) v = fma(alpha, R[e], v);
        X[i * NN + e] = v;
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) yl[i * N + n] = fma(-bw, sha[n], yv[n]);
}

// res[i] = (a_p . C_i - b_p)^2 W_p
__global__ void k_loo_resid(int64_t n, int N, int64_t P, const double* __restrict__ At, const int* __restrict__ pidx,
                            const double* __restrict__ C, const double* __restrict__ W, const double* __restrict__ b,
                            double* __restrict__ res)
{
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const int64_t p = pidx[i];
    double acc = 0.0;
    for (int n = lane; n < N; n += 64) acc = fma(At[(int64_t)n * P + p], C[i * N + n], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) {
        const double d = acc - b[p];
        res[i] = d * d * W[p];
    }
}

// rocSOLVER's syevd loses accuracy on matrices of tiny magnitude (A^T W A entries are ~1e-19 because
// W = sigma^-2 ~ 1e-22): measured rel(C) 0.6 on the raw system vs 2e-8 once it is scaled.  Scale every
// system by an exact power of two to max|X| in [1, 2); eigenvalues are scaled back in k_trunc_apply.
template <int BS>
__global__ __launch_bounds__(BS) void k_scale_system(int NN, double* __restrict__ X, double* __restrict__ scl)
{
    __shared__ double red[BS];
    const int64_t i = blockIdx.x;
    const int tid = threadIdx.x;
    double* Xi = X + i * NN;
    double mx = 0.0;
    for (int e = tid; e < NN; e += BS) mx = fmax(mx, fabs(Xi[e]));
    red[tid] = mx;
    __syncthreads();
    for (int s = BS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    int ex = 0;
    double f = 1.0;
    if (mx > 0.0 && mx < 1.7e308) {
        (void)frexp(mx, &ex);            // mx = m * 2^ex, m in [0.5, 1)
        f = ldexp(1.0, 1 - ex);          // mx * f in [1, 2)
    }
    for (int e = tid; e < NN; e += BS) Xi[e] *= f;
    if (tid == 0) scl[i] = 1.0 / f;      // exact (power of two)
}

// k_form_pair followed by k_scale_system in one pass: X[i] = f * (D1[slot] + alpha D2[slot]), scl[i] = 1 / f with f the
// power of two that brings max|X| into [1, 2).  The elements stay in registers between the max reduction and the
// store (NN <= EPT * BS), so D1 and D2 are read once and X is written once (a single-record root-finder iterate spent
// 45 us in the two separate kernels against ~1 ms in the solve).
// slot1 / slot2: where D1 / D2 of system i live (slot1 null: D1[i], then X may be D1 itself - every element is read and
// written by the same thread).
template <int BS, int EPT>
__global__ __launch_bounds__(BS) void k_form_pair_scaled(int NN, const double* D1, const double* __restrict__ D2,
                                                         const int* __restrict__ slot1, const int* __restrict__ slot2,
                                                         const double* __restrict__ alpha, double* X, double* __restrict__ scl)
{
    __shared__ double red[BS / 64];
    const int64_t i = blockIdx.x;
    const int64_t w1 = slot1 ? slot1[i] : i;
    const int64_t w = slot2[i];
    const double a = alpha[i];
    const int tid = threadIdx.x;
    double v[EPT];
    double mx = 0.0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int e = tid + u * BS;
        v[u] = e < NN ? fma(a, D2[w * NN + e], D1[w1 * NN + e]) : 0.0;
        mx = fmax(mx, fabs(v[u]));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = 0.0;
#pragma unroll
    for (int q = 0; q < BS / 64; ++q) mx = fmax(mx, red[q]);
    int ex = 0;
    double f = 1.0;
    if (mx > 0.0 && mx < 1.7e308) {
        (void)frexp(mx, &ex);            // mx = m * 2^ex, m in [0.5, 1)
        f = ldexp(1.0, 1 - ex);          // mx * f in [1, 2)
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int e = tid + u * BS;
        if (e < NN) X[i * NN + e] = v[u] * f;
    }
    if (tid == 0) scl[i] = 1.0 / f;      // exact (power of two)
}

// X[i] = f (D1[slot1[i]] + alpha[i] D2[slot2[i]]), scl[i] = 1 / f with f the power of two that brings max|X| into [1, 2):
// one fused pass while a system fits the registers of a 1024-thread block (N <= 156), two kernels beyond
static void form_pair_scaled(vi_ctx* c, int64_t bc, int NN, const double* D1, const double* D2, const int* slot1,
                             const int* slot2, const double* alpha, double* X, double* scl)
{
    if (NN <= 24 * 1024) {
        hipLaunchKernelGGL((k_form_pair_scaled<1024, 24>), dim3((unsigned)bc), dim3(1024), 0, c->stream, NN, D1, D2, slot1, slot2,
                           alpha, X, scl);
    } else {
        hipLaunchKernelGGL(k_form_pair, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, D1, D2, slot1, slot2, alpha, X);
        hipLaunchKernelGGL(k_scale_system<256>, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, X, scl);
    }
}

// One workgroup per system: given eigenpairs (V column-major, lam) form the truncated minimum-norm
// solution C = V diag(1/lam | |lam| > rcond*max|lam|) V^T y   (gelsd semantics for symmetric X)
// and optionally the scaled eigenvectors Vs = V diag(w_pinv) for H = Vs V^T.
template <int BS>
__global__ __launch_bounds__(BS) void k_trunc_apply(int N, const double* __restrict__ V, const double* __restrict__ lam,
                                                    const double* __restrict__ scl, const double* __restrict__ y,
                                                    const int* __restrict__ rec, double rcond,
                                                    double* __restrict__ C, int* __restrict__ rank,
                                                    double pinv_rcond, double* __restrict__ Vs)
{
    extern __shared__ double sh[];
    double* g = sh;            // [N] V^T y scaled
    double* red = sh + N;      // [BS]
    const int i = blockIdx.x, tid = threadIdx.x;
    const double* Vi = V + (int64_t)i * N * N;
    const double* li = lam + (int64_t)i * N;      // eigenvalues of the scaled system
    const double sc = scl[i];                     // true eigenvalue = li[j] * sc
    const double* yi = y + (int64_t)(rec ? rec[i] : i) * N;
    // max |lambda|
    double mx = 0.0;
    for (int j = tid; j < N; j += BS) mx = fmax(mx, fabs(li[j]));
    red[tid] = mx;
    __syncthreads();
    for (int s = BS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    const double thr = rcond * mx, pthr = pinv_rcond * mx;
    // g_j = (v_j . y) / lam_j : one wave per eigenvector, coalesced column reads
    const int wave = tid >> 6, lane = tid & 63, nw = BS >> 6;
    int rk = 0;
    for (int j = wave; j < N; j += nw) {
        double acc = 0.0;
        for (int r = lane; r < N; r += 64) acc = fma(Vi[(int64_t)j * N + r], yi[r], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) {
            const bool keep = fabs(li[j]) > thr;
            g[j] = keep ? acc / (li[j] * sc) : 0.0;
            rk += keep ? 1 : 0;
        }
    }
    red[tid] = (double)rk;
    __syncthreads();
    for (int s = BS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0 && rank) rank[i] = (int)red[0];
    __syncthreads();
    // C_r = sum_j V[r,j] g_j
    for (int r = tid; r < N; r += BS) {
        double acc = 0.0;
        for (int j = 0; j < N; ++j) acc = fma(Vi[(int64_t)j * N + r], g[j], acc);
        C[(int64_t)i * N + r] = acc;
    }
    if (Vs) {
        double* Vo = Vs + (int64_t)i * N * N;
        for (int j = 0; j < N; ++j) {
            const double w = fabs(li[j]) > pthr ? 1.0 / (li[j] * sc) : 0.0;
            for (int r = tid; r < N; r += BS) Vo[(int64_t)j * N + r] = Vi[(int64_t)j * N + r] * w;
        }
    }
}

// chi^2 of S systems per workgroup, one data point per thread:
//   part[i][blockIdx.y] = sum over the block's points p of W[rec_i][p] (sum_n At[n][p] C[i][n] - b[rec_i][p])^2.
// The arithmetic of a system does not depend on the batch it is evaluated in: the model value of a point is ONE chain of
// fused multiply-adds over n = 0 .. N-1, the points of a block are summed by a fixed tree, the blocks by k_chi2_sum in
// order - whatever S (which only decides how many systems share a block's loads of At), whatever the batch size.  (The
// library GEMM this replaces chose its kernel, hence its summation order, by the number of systems: a record's chi^2
// changed in the 10th digit with the batch it was fitted in, and Brent's path with it.)
template <int BS, int S>
__global__ __launch_bounds__(BS) void k_chi2_part(int64_t P, int N, int64_t B, const double* __restrict__ At,
                                                  const double* __restrict__ C, const int* __restrict__ rec,
                                                  int64_t rec_base, const double* __restrict__ W,
                                                  const double* __restrict__ b, double* __restrict__ part)
{
    extern __shared__ double shC[];                 // S x N coefficients, then BS partial sums
    double* red = shC + (size_t)S * N;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * S;
    const int ns = (int)((B - i0) < S ? (B - i0) : S);
    for (int e = tid; e < S * N; e += BS) shC[e] = (e / N) < ns ? C[i0 * N + e] : 0.0;
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.y * BS + tid;
    double acc[S];
#pragma unroll
    for (int q = 0; q < S; ++q) acc[q] = 0.0;
    if (p < P) {
#pragma unroll 8
        for (int n = 0; n < N; ++n) {
            const double a = At[(int64_t)n * P + p];
#pragma unroll
            for (int q = 0; q < S; ++q) acc[q] = fma(a, shC[q * N + n], acc[q]);
        }
    }
    for (int q = 0; q < ns; ++q) {
        double v = 0.0;
        if (p < P) {
            const int64_t r = rec ? rec[i0 + q] : rec_base + i0 + q;
            const double d = acc[q] - b[r * P + p];
            v = d * d * W[r * P + p];
        }
        red[tid] = v;
        __syncthreads();
        for (int h = BS / 2; h > 0; h >>= 1) {
            if (tid < h) red[tid] += red[tid + h];
            __syncthreads();
        }
        if (tid == 0) part[(i0 + q) * gridDim.y + blockIdx.y] = red[0];
        __syncthreads();
    }
}

__global__ void k_chi2_sum(int64_t B, int nb, const double* __restrict__ part, double* __restrict__ chi2)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double acc = 0.0;
    for (int j = 0; j < nb; ++j) acc += part[i * nb + j];
    chi2[i] = acc;
}

// y[t][n] = sum_p At[n][p] W[t][p] b[t][p]: one workgroup per (n, group of 8 records), fixed summation order per record
template <int BS>
__global__ __launch_bounds__(BS) void k_atwb(int64_t P, int N, int64_t T, const double* __restrict__ At,
                                             const double* __restrict__ W, const double* __restrict__ b,
                                             double* __restrict__ y)
{
    __shared__ double red[BS];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.y * 8;
    const int nt = (int)((T - t0) < 8 ? (T - t0) : 8);
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (int64_t p = tid; p < P; p += BS) {
        const double a = At[(int64_t)n * P + p];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < nt) acc[q] = fma(a, W[(t0 + q) * P + p] * b[(t0 + q) * P + p], acc[q]);
    }
    for (int q = 0; q < nt; ++q) {
        red[tid] = acc[q];
        __syncthreads();
        for (int h = BS / 2; h > 0; h >>= 1) {
            if (tid < h) red[tid] += red[tid + h];
            __syncthreads();
        }
        if (tid == 0) y[(t0 + q) * N + n] = red[0];
        __syncthreads();
    }
}

// 2: in-LDS parallel Jacobi (vi_jacobi.hip, default where it fits), 1: rocSOLVER syevj, 0: rocSOLVER syevd.
// syevd (divide & conquer) measured 1.4e-6 on the coefficients of one screened fixture against the 1e-6
// gate; both Jacobi variants give <= 1e-8.
int eig_method()
{
    static int m = -1;
    if (m < 0) {
        const char* e = getenv("VINTERP_EIG");
        m = 2;
        if (e && !strcmp(e, "syevj")) m = 1;
        if (e && !strcmp(e, "syevd")) m = 0;
        if (e && !strcmp(e, "jacobi")) m = 2;
    }
    return m;
}

// sweep cap of the in-LDS Jacobi solver (the rotation-log workspace is sized by it); VINTERP_MAX_SWEEPS overrides it
int jacobi_max_sweeps()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VINTERP_MAX_SWEEPS");
        v = e ? atoi(e) : 24;
        if (v < 4) v = 4;
        if (v > 200) v = 200;
    }
    return v;
}
#define JACOBI_MAX_SWEEPS jacobi_max_sweeps()
constexpr double JACOBI_FLOOR_COLD = 1e-22;
// Absolute rotation floor of the warm (rotated-system) solves, relative to the scaled matrix (VINTERP_WARM_FLOOR overrides).
// It was 1e-16 - about half the truncation cut eps * max|lambda| - so couplings of that size between eigenvalues next to
// the cut were never rotated away, the warm and the cold solve disagreed about which of them survive, and at the default
// order half of the records came back with a root of the warm chi^2 that the cold final solve missed by > 1e-4 nu
// (measured, 1000 records 26 x 100: 464 searches redone with the floor at 1e-16, none at 1e-18, for 15 % more rounds)
double jacobi_floor_warm()
{
    static double v = -1.0;
    if (v < 0.0) {
        const char* e = getenv("VINTERP_WARM_FLOOR");
        v = e ? atof(e) : 1e-18;
        if (!(v > 0.0)) v = 1e-18;
    }
    return v;
}
#define JACOBI_FLOOR_WARM jacobi_floor_warm()

// K3p: the cold solves (walk ends, bracket bases, final solves - everything that starts from X(alpha) itself) are
// pre-conditioned by one pivoted-QR similarity step (vi_qr.hip): 5-8 Jacobi sweeps instead of 20-24 at the default order.
// VINTERP_QRPRE=0 switches it off.  The rotated systems of the warm / shared-basis solves are nearly diagonal already.
bool qr_enabled(int N)
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VINTERP_QRPRE");
        v = (e && !strcmp(e, "0")) ? 0 : 1;
    }
    return v == 1 && vi_qr_supported(N);
}
inline size_t up16(size_t b) { return (b + 15) & ~(size_t)15; }

}  // namespace

namespace {
// The library picks its GEMM kernel - hence the summation order - by the problem size AND the batch count, so a record's
// normal equations and rotated systems used to change in the 16th digit with the number of records computed alongside,
// and (chi^2(alpha) being what it is at the default order) Brent's path with them.  Every batched product that feeds a
// record's numbers is therefore issued in groups of exactly GEMM_GROUP matrices through the pointer-array interface;
// the last group is padded with repeats of its last entry whose results go to a scratch area.
constexpr int GEMM_GROUP = 32;

// out[i] = base + min(i, count - 1) * stride   (i < countp; inputs)          or, with scratch != nullptr (outputs),
// out[i] = i < count ? base + i * stride : scratch + (i - count) * sstride
// idx != nullptr: entry i lives at base + idx[i] * stride (records / slots picked out of a larger array)
__global__ void k_group_ptrs(int64_t count, int64_t countp, const double* base, int64_t stride, double* scratch,
                             int64_t sstride, const double** out, const int* __restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= countp) return;
    if (i < count || !scratch) {
        const int64_t j = i < count ? i : count - 1;
        out[i] = base + (idx ? (int64_t)idx[j] : j) * stride;
    } else {
        out[i] = scratch + (i - count) * sstride;
    }
}

inline int64_t group_pad(int64_t n) { return (n + GEMM_GROUP - 1) / GEMM_GROUP * GEMM_GROUP; }

int group_ptrs(vi_ctx* c, int64_t count, const double* base, int64_t stride, double* scratch, int64_t sstride,
               const double** out, const int* idx = nullptr)
{
    const int64_t cp = group_pad(count);
    hipLaunchKernelGGL(k_group_ptrs, dim3(nblk(cp, 256)), dim3(256), 0, c->stream, count, cp, base, stride, scratch, sstride,
                       out, idx);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

int gemm_groups(vi_ctx* c, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double** pA, int lda,
                const double** pB, int ldb, const double** pC, int ldc, int64_t count)
{
    const double one = 1.0, zero = 0.0;
    for (int64_t g0 = 0; g0 < group_pad(count); g0 += GEMM_GROUP)
        VI_ROCBLAS(rocblas_dgemm_batched(c->blas, ta, tb, m, n, k, &one, pA + g0, lda, pB + g0, ldb, &zero,
                                         (double* const*)(pC + g0), ldc, GEMM_GROUP));
    return VI_OK;
}

// C[i] = op(A[i]) B[i] for `count` N x N column-major matrices by wg_gemm (vi_gemm_device.h): the products of the re-basing,
// whose bits the device-side search (k_brent_warm) must reproduce.  Few products (a record fitted alone) are cut into
// blocks of 8 x 8 thread tiles of 3 x 3 elements, one workgroup of 64 threads each: the same elements (a sum does not
// know its tile), 36 times the CUs at N = 144.
template <bool TA>
__global__ __launch_bounds__(640) void k_wg_gemm(int N, const double* const* __restrict__ pA, const double* const* __restrict__ pB,
                                                 const double* const* __restrict__ pC, int bt)
{
    extern __shared__ __align__(16) double shg[];
    if (bt <= 0) {
        wg_gemm<TA>(N, pA[blockIdx.x], pB[blockIdx.x], const_cast<double*>(pC[blockIdx.x]), shg);
    } else {
        const int nt = (N + 2) / 3, nb = (nt + bt - 1) / bt;
        const int bi = blockIdx.y % nb, bj = blockIdx.y / nb;
        wg_gemm<TA, 3>(N, pA[blockIdx.x], pB[blockIdx.x], const_cast<double*>(pC[blockIdx.x]), shg, bi * bt, bt, bj * bt, bt);
    }
}

int wg_gemm_batched(vi_ctx* c, bool ta, int N, const double** pA, const double** pB, const double** pC, int64_t count)
{
    const size_t shm = wg_gemm_lds_doubles(N) * sizeof(double);
    const int nt = (N + 5) / 6;
    int threads = ((nt * nt + 63) / 64) * 64;
    if (threads > 640) threads = 640;
    int bt = 0;
    dim3 grid((unsigned)count);
    if (count * 4 <= c->n_cu && N > 24) {           // the chip is far from full: blocks of 8 x 8 thread tiles of 3 x 3
        bt = 8;
        const int nb = ((N + 2) / 3 + bt - 1) / bt;
        grid = dim3((unsigned)count, (unsigned)(nb * nb));
        threads = 64;
    }
    if (ta) {
        VI_HIP(hipFuncSetAttribute((const void*)k_wg_gemm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(k_wg_gemm<true>, grid, dim3(threads), shm, c->stream, N, pA, pB, pC, bt);
    } else {
        VI_HIP(hipFuncSetAttribute((const void*)k_wg_gemm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(k_wg_gemm<false>, grid, dim3(threads), shm, c->stream, N, pA, pB, pC, bt);
    }
    VI_HIP(hipGetLastError());
    return VI_OK;
}
}  // namespace

extern "C" int vi_normal_eq_f64(vi_ctx* c, int64_t T, int64_t P, int32_t N, const double* d_At, const double* d_W,
                                const double* d_b, double* d_AWA, double* d_y)
{
    VI_REQUIRE(c && d_At && d_W && d_b && d_AWA && d_y, "null argument");
    VI_REQUIRE(T >= 0 && P > 0 && N > 0, "bad size");
    if (T == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    const size_t per_t = (size_t)N * P * sizeof(double);
    const int NN = N * N;
    int64_t Tc = (int64_t)((size_t)1 << 30) / (int64_t)per_t;      // <= 1 GiB of scaled copies at a time
    if (Tc < 1) Tc = 1;
    if (Tc > T) Tc = T;
    const int64_t Tcp = group_pad(Tc);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)Tc * per_t + (size_t)GEMM_GROUP * NN * sizeof(double) + (size_t)3 * Tcp * sizeof(void*),
                              &ws);
    if (rc != VI_OK) return rc;
    double* Bs = (double*)ws;
    double* scratch = Bs + (size_t)Tc * N * P;
    const double** pA = (const double**)(scratch + (size_t)GEMM_GROUP * NN);
    const double** pB = pA + Tcp;
    const double** pC = pB + Tcp;
    for (int64_t t0 = 0; t0 < T; t0 += Tc) {
        const int64_t tc = (T - t0) < Tc ? (T - t0) : Tc;
        hipLaunchKernelGGL(k_scale_rows, dim3(nblk(P, 256), N, (unsigned)tc), dim3(256), 0, c->stream, P, N, d_At,
                           d_W + t0 * P, Bs, (int)tc);
        VI_HIP(hipGetLastError());
        // AWA_t (N x N) = A^T (N x P) * B_t (P x N), column-major views of the row-major N x P arrays
        if ((rc = group_ptrs(c, tc, d_At, 0, nullptr, 0, pA)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, tc, Bs, (int64_t)N * P, nullptr, 0, pB)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, tc, d_AWA + t0 * NN, NN, scratch, NN, pC)) != VI_OK) return rc;
        if ((rc = gemm_groups(c, rocblas_operation_transpose, rocblas_operation_none, N, N, (int)P, pA, (int)P, pB, (int)P,
                              pC, N, tc)) != VI_OK)
            return rc;
    }
    hipLaunchKernelGGL(k_atwb<256>, dim3(N, nblk(T, 8)), dim3(256), 0, c->stream, P, N, T, d_At, d_W, d_b, d_y);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

extern "C" int vi_form_system_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_AWA, const int32_t* d_rec,
                                  const double* d_alpha, const double* d_R, double* d_X)
{
    VI_REQUIRE(c && d_X, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    VI_REQUIRE(d_AWA || d_R, "nothing to do");
    VI_REQUIRE(!d_R || d_alpha, "regularisation matrix given without parameters");
    if (B == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    hipLaunchKernelGGL(k_form_system, dim3((unsigned)B), dim3(256), 0, c->stream, NN, d_AWA, d_rec, d_alpha, d_R, d_X);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

extern "C" int vi_solve_trunc_f64(vi_ctx* c, int64_t B, int32_t N, double* d_X, const double* d_y,
                                  const int32_t* d_rec, double rcond, double* d_C, int32_t* d_rank,
                                  double pinv_rcond, double* d_H)
{
    VI_REQUIRE(c && d_X && d_y && d_C, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    if (!d_H && eig_method() == 2 && vi_jacobi_supported(N)) {
        // in-LDS Jacobi: chunk the batch so that the rotation logs stay within 4 GiB of workspace
        const bool qr = qr_enabled(N);
        const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS);
        const size_t hhb = qr ? up16(vi_qr_hh_bytes(N)) : 0;
        const size_t per = logb + sizeof(double) + (qr ? hhb + (size_t)(N * N + N) * sizeof(double) : 0);
        int64_t Bc = (int64_t)(((size_t)4 << 30) / per);
        if (Bc < 1) Bc = 1;
        if (Bc > B) Bc = B;
        void* ws = nullptr;
        int rc = vi_ctx_workspace(c, (size_t)Bc * per + 256, &ws);
        if (rc != VI_OK) return rc;
        double* scl = (double*)((char*)ws + (size_t)Bc * logb);
        double* hh = scl + Bc + (Bc & 1);
        double* scr = (double*)((char*)hh + (size_t)Bc * hhb);
        double* y1 = scr + (size_t)Bc * N * N;
        for (int64_t i0 = 0; i0 < B; i0 += Bc) {
            const int64_t bc = (B - i0) < Bc ? (B - i0) : Bc;
            double* Xc = d_X + i0 * N * N;
            const double* yc = d_rec ? d_y : d_y + i0 * N;
            const int* rcc = d_rec ? d_rec + i0 : nullptr;
            hipLaunchKernelGGL(k_scale_system<256>, dim3((unsigned)bc), dim3(256), 0, c->stream, N * N, Xc, scl);
            VI_HIP(hipGetLastError());
            if (qr) {
                if ((rc = vi_qr_precond(c, bc, N, Xc, yc, rcc, Xc, y1, hh, scr, (int64_t)(hhb / 8))) != VI_OK) return rc;
                yc = y1;
                rcc = nullptr;
            }
            rc = vi_jacobi_solve(c, bc, N, Xc, scl, yc, rcc, rcond, d_C + i0 * N, d_rank ? d_rank + i0 : nullptr, ws,
                                 JACOBI_MAX_SWEEPS, nullptr, nullptr, 0, nullptr, JACOBI_FLOOR_COLD);
            if (rc != VI_OK) return rc;
            if (qr && (rc = vi_qr_back_vec(c, bc, N, hh, d_C + i0 * N, (int64_t)(hhb / 8))) != VI_OK) return rc;
        }
        return VI_OK;
    }
    if (d_H && eig_method() == 2 && vi_jacobi_vectors_supported(N)) {
        // final solves with H = pinv(X): in-LDS Jacobi (C, eigenvalues, rotation log) -> eigenvectors from the
        // log -> H = V diag(1/lam | kept) V^T
        const bool qr = qr_enabled(N);
        const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS);
        const size_t hhb = qr ? up16(vi_qr_hh_bytes(N)) : 0;
        int64_t Bc = (int64_t)(((size_t)4 << 30) / logb);
        if (Bc < 1) Bc = 1;
        if (Bc > B) Bc = B;
        void* ws = nullptr;
        const size_t per = logb + (size_t)(N + 1) * sizeof(double) + sizeof(int) + (size_t)2 * N * N * sizeof(double) +
                           (qr ? hhb + (size_t)N * sizeof(double) : 0);
        int rc = vi_ctx_workspace(c, (size_t)Bc * per + 1024, &ws);
        if (rc != VI_OK) return rc;
        char* wp = (char*)ws + (size_t)Bc * logb;
        double* scl = (double*)wp;
        double* lam = scl + Bc;
        double* V = lam + (size_t)Bc * N;
        double* Vs = V + (size_t)Bc * N * N;
        double* y1 = Vs + (size_t)Bc * N * N;
        double* hh = y1 + (qr ? (size_t)Bc * N : 0);
        int* nrd = (int*)((char*)hh + (size_t)Bc * hhb);
        const double one = 1.0, zero = 0.0;
        for (int64_t i0 = 0; i0 < B; i0 += Bc) {
            const int64_t bc = (B - i0) < Bc ? (B - i0) : Bc;
            double* Xc = d_X + i0 * N * N;
            const double* yc = d_rec ? d_y : d_y + i0 * N;
            const int* rc_ = d_rec ? d_rec + i0 : nullptr;
            hipLaunchKernelGGL(k_scale_system<256>, dim3((unsigned)bc), dim3(256), 0, c->stream, N * N, Xc, scl);
            VI_HIP(hipGetLastError());
            // pre-conditioned: the Jacobi kernel works on X1 = Q^T X Q (Vs is free until k_trunc_apply: QR scratch);
            // its eigenvectors are turned back, V = Q V1, and k_trunc_apply then sees X's own eigenpairs and y
            if (qr && (rc = vi_qr_precond(c, bc, N, Xc, yc, rc_, Xc, y1, hh, Vs, (int64_t)(hhb / 8))) != VI_OK) return rc;
            rc = vi_jacobi_solve(c, bc, N, Xc, scl, qr ? y1 : yc, qr ? nullptr : rc_, rcond, d_C + i0 * N,
                                 d_rank ? d_rank + i0 : nullptr, ws, JACOBI_MAX_SWEEPS, nullptr, lam, 1, nrd, JACOBI_FLOOR_COLD);
            if (rc != VI_OK) return rc;
            rc = vi_jacobi_vectors(c, bc, N, ws, JACOBI_MAX_SWEEPS, nrd, V);
            if (rc != VI_OK) return rc;
            if (qr && (rc = vi_qr_back_mat(c, bc, N, hh, V, (int64_t)(hhb / 8))) != VI_OK) return rc;
            constexpr int BS = 256;
            hipLaunchKernelGGL(k_trunc_apply<BS>, dim3((unsigned)bc), dim3(BS), (size_t)(N + BS) * sizeof(double), c->stream,
                               N, V, lam, scl, yc, rc_, rcond, d_C + i0 * N, d_rank ? d_rank + i0 : nullptr, pinv_rcond, Vs);
            VI_HIP(hipGetLastError());
            VI_ROCBLAS(rocblas_dgemm_strided_batched(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N,
                                                     &one, Vs, N, (rocblas_stride)N * N, V, N, (rocblas_stride)N * N, &zero,
                                                     d_H + i0 * N * N, N, (rocblas_stride)N * N, (rocblas_int)bc));
        }
        return VI_OK;
    }
    // The library path in launches of at most 4 GiB of systems (404 at N = 1152): one batched syevd call on 1618 systems of
    // that order - 17 GB, 2.147e9 elements, a hair under 2^31 - ended in a GPU memory access fault inside the library
    // (round 4); up to 392 systems per call is what every earlier round ran.  The cost per system is flat from 64 systems on.
    {
        int64_t Bmax = (int64_t)(((size_t)4 << 30) / ((size_t)N * N * sizeof(double)));
        if (Bmax < 1) Bmax = 1;
        if (B > Bmax) {
            for (int64_t i0 = 0; i0 < B; i0 += Bmax) {
                const int64_t bc = (B - i0) < Bmax ? (B - i0) : Bmax;
                const int rc2 = vi_solve_trunc_f64(c, bc, N, d_X + i0 * N * N, d_rec ? d_y : d_y + i0 * N, d_rec ? d_rec + i0 : nullptr,
                                                   rcond, d_C + i0 * N, d_rank ? d_rank + i0 : nullptr, pinv_rcond,
                                                   d_H ? d_H + i0 * N * N : nullptr);
                if (rc2 != VI_OK) return rc2;
            }
            return VI_OK;
        }
    }
    // workspace: eigenvalues [B][N], E [B][N], info [B], (Vs [B][N][N] when H is wanted)
    const size_t nD = (size_t)B * N;
    size_t bytes = 2 * nD * sizeof(double) + (size_t)B * sizeof(double) + (size_t)B * sizeof(int) * 4 + 256;
    if (d_H) bytes += (size_t)B * N * N * sizeof(double);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, bytes, &ws);
    if (rc != VI_OK) return rc;
    double* D = (double*)ws;
    double* E = D + nD;
    double* scl = E + nD;
    double* Vs = d_H ? scl + B : nullptr;
    int* info = (int*)((char*)ws + (2 * nD + (size_t)B) * sizeof(double) + (d_H ? (size_t)B * N * N * sizeof(double) : 0));
    hipLaunchKernelGGL(k_scale_system<256>, dim3((unsigned)B), dim3(256), 0, c->stream, N * N, d_X, scl);
    VI_HIP(hipGetLastError());
    // orders beyond the in-LDS solver (N > ~200): divide and conquer.  Measured at N = 1152 (configs[4]): syevd 10.6 ms
    // per system against 295 ms for rocSOLVER's Jacobi (syevj), which is only used when asked for (VINTERP_EIG=syevj).
    if (eig_method() == 1) {
        int* nsweeps = info + B;
        double* resid = E;
        VI_ROCSOLVER(rocsolver_dsyevj_strided_batched(c->blas, rocblas_esort_ascending, rocblas_evect_original,
                                                      rocblas_fill_upper, N, d_X, N, (rocblas_stride)N * N, 0.0, resid,
                                                      100, nsweeps, D, N, info, (rocblas_int)B));
    } else {
        VI_ROCSOLVER(rocsolver_dsyevd_strided_batched(c->blas, rocblas_evect_original, rocblas_fill_upper, N, d_X, N,
                                                      (rocblas_stride)N * N, D, N, E, N, info, (rocblas_int)B));
    }
    constexpr int BS = 256;
    const size_t shm = (size_t)(N + BS) * sizeof(double);
    hipLaunchKernelGGL(k_trunc_apply<BS>, dim3((unsigned)B), dim3(BS), shm, c->stream, N, d_X, D, scl, d_y, d_rec, rcond,
                       d_C, d_rank, pinv_rcond, Vs);
    VI_HIP(hipGetLastError());
    if (d_H) {
        const double one = 1.0, zero = 0.0;
        // H = Vs V^T (column-major N x N; symmetric)
        VI_ROCBLAS(rocblas_dgemm_strided_batched(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N,
                                                 &one, Vs, N, (rocblas_stride)N * N, d_X, N, (rocblas_stride)N * N,
                                                 &zero, d_H, N, (rocblas_stride)N * N, (rocblas_int)B));
    }
    return VI_OK;
}

extern "C" int vi_chi2_f64(vi_ctx* c, int64_t B, int64_t P, int32_t N, const double* d_At, const double* d_C,
                           const int32_t* d_rec, const double* d_W, const double* d_b, double* d_chi2)
{
    VI_REQUIRE(c && d_At && d_C && d_W && d_b && d_chi2, "null argument");
    VI_REQUIRE(B >= 0 && P > 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    const int nb = (int)nblk(P, 256);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)B * nb * sizeof(double), &ws);
    if (rc != VI_OK) return rc;
    double* part = (double*)ws;
    // S systems share a block's loads of the basis; a system's arithmetic is the same for every S (see k_chi2_part)
    if (B >= 2048 && (size_t)(8 * N + 256) * sizeof(double) <= 48 * 1024) {          // (48 KB: the default LDS limit of a launch)
        hipLaunchKernelGGL((k_chi2_part<256, 8>), dim3(nblk(B, 8), nb), dim3(256), (size_t)(8 * N + 256) * sizeof(double),
                           c->stream, P, N, B, d_At, d_C, d_rec, (int64_t)0, d_W, d_b, part);
    } else if (B >= 256) {
        hipLaunchKernelGGL((k_chi2_part<256, 2>), dim3(nblk(B, 2), nb), dim3(256), (size_t)(2 * N + 256) * sizeof(double),
                           c->stream, P, N, B, d_At, d_C, d_rec, (int64_t)0, d_W, d_b, part);
    } else {
        hipLaunchKernelGGL((k_chi2_part<256, 1>), dim3((unsigned)B, nb), dim3(256), (size_t)(N + 256) * sizeof(double),
                           c->stream, P, N, B, d_At, d_C, d_rec, (int64_t)0, d_W, d_b, part);
    }
    VI_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_chi2_sum, dim3(nblk(B, 256)), dim3(256), 0, c->stream, B, nb, part, d_chi2);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

extern "C" int vi_cov_f64(vi_ctx* c, int64_t T, int32_t N, const double* d_H, const double* d_AWA, double* d_dC)
{
    VI_REQUIRE(c && d_H && d_AWA && d_dC, "null argument");
    VI_REQUIRE(T >= 0 && N > 0, "bad size");
    if (T == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)T * N * N * sizeof(double), &ws);
    if (rc != VI_OK) return rc;
    double* tmp = (double*)ws;
    const double one = 1.0, zero = 0.0;
    const rocblas_stride s = (rocblas_stride)N * N;
    // all three factors are symmetric, so row-/column-major views coincide
    VI_ROCBLAS(rocblas_dgemm_strided_batched(c->blas, rocblas_operation_none, rocblas_operation_none, N, N, N, &one,
                                             d_H, N, s, d_AWA, N, s, &zero, tmp, N, s, (rocblas_int)T));
    VI_ROCBLAS(rocblas_dgemm_strided_batched(c->blas, rocblas_operation_none, rocblas_operation_none, N, N, N, &one,
                                             tmp, N, s, d_H, N, s, &zero, d_dC, N, s, (rocblas_int)T));
    return VI_OK;
}

// Diagnostic / test entry: eigenvalues of B symmetric systems by the in-LDS Jacobi kernel, with the number
// of sweeps each one took.  d_X is scaled in place (power of two) and otherwise only read.
extern "C" int vi_eigvals_f64(vi_ctx* c, int64_t B, int32_t N, double* d_X, double* d_lam, int32_t* d_sweeps)
{
    VI_REQUIRE(c && d_X && d_lam, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_jacobi_supported(N)) {
        vi_set_error("vi_eigvals_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const bool qr = qr_enabled(N);
    const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS);
    const size_t hhb = qr ? up16(vi_qr_hh_bytes(N)) : 0;
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)B * logb + (size_t)B * (N + 1) * sizeof(double) * 3 + 256 +
                                     (qr ? (size_t)B * (hhb + (size_t)N * N * sizeof(double)) : 0), &ws);
    if (rc != VI_OK) return rc;
    double* scl = (double*)((char*)ws + (size_t)B * logb);
    double* yz = scl + B + (B & 1);  // zero right-hand sides
    double* Cz = yz + (size_t)B * N;
    double* y1 = Cz + (size_t)B * N;
    double* hh = y1 + (size_t)B * N + ((B * N) & 1);
    double* scr = (double*)((char*)hh + (size_t)B * hhb);
    VI_HIP(hipMemsetAsync(yz, 0, (size_t)B * N * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_scale_system<256>, dim3((unsigned)B), dim3(256), 0, c->stream, N * N, d_X, scl);
    VI_HIP(hipGetLastError());
    // the same pre-conditioning as vi_solve_trunc_f64 (X1 = Q^T X Q has X's eigenvalues)
    if (qr && (rc = vi_qr_precond(c, B, N, d_X, yz, nullptr, d_X, y1, hh, scr, (int64_t)(hhb / 8))) != VI_OK) return rc;
    return vi_jacobi_solve(c, B, N, d_X, scl, yz, nullptr, 2.220446049250313e-16, Cz, nullptr, ws,
                           JACOBI_MAX_SWEEPS, d_sweeps, d_lam, 0, nullptr, JACOBI_FLOOR_COLD);
}

// ---- warm-started search -------------------------------------------------------------------------------
// Brent's iterates inside one unit bracket of log10(alpha) solve nearly identical systems.  Once per record
// the system at the first iterate alpha0 is decomposed with eigenvectors, X(alpha0) = V L V^T, and
// D1 = V^T AWA V, D2 = V^T R V, yt = V^T y are formed; every later iterate then solves the rotated system
// D1 + alpha D2 (3-10 Jacobi sweeps instead of 17-24 in the rank-deficient regime) and maps back C = V c'.
namespace {
// Rotation logs of systems that are decomposed now and turned into eigenvectors later lie in records of
// log_record_bytes(N): the log itself, then (pre-conditioned solves) the Householder reflectors of the system's QR step.
size_t log_record_bytes(int N)
{
    return vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS) + (qr_enabled(N) ? up16(vi_qr_hh_bytes(N)) : 0);
}

// phase 1 of setting up rotated systems: X(alpha0) of bc records formed, scaled, pre-conditioned (y1, scr: N and N x N
// doubles of scratch per system) and decomposed (cold), the truncated solution to Cc, the rotation logs (+ reflectors) to
// the records at `log`, the rounds they hold to nrd
int prep_decompose(vi_ctx* c, int64_t bc, int N, const double* d_AWA, const int32_t* recc, const double* alpha0c,
                   const double* d_R, const double* d_y, double rcond, double* Cc, int32_t* rankc, void* log, double* scl,
                   double* lam, double* T0, int* nrd, double* y1, double* scr)
{
    const int NN = N * N;
    const bool qr = qr_enabled(N);
    const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS), recb = log_record_bytes(N);
    double* hh = (double*)((char*)log + logb);
    hipLaunchKernelGGL(k_form_system, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, d_AWA, recc, alpha0c, d_R, T0);
    hipLaunchKernelGGL(k_scale_system<256>, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, T0, scl);
    VI_HIP(hipGetLastError());
    int rc;
    if (qr && (rc = vi_qr_precond(c, bc, N, T0, d_y, recc, T0, y1, hh, scr, (int64_t)(recb / 8))) != VI_OK) return rc;
    rc = vi_jacobi_solve(c, bc, N, T0, scl, qr ? y1 : d_y, qr ? nullptr : recc, rcond, Cc, rankc, log, JACOBI_MAX_SWEEPS, nullptr,
                         lam, 1, nrd, JACOBI_FLOOR_COLD, (int64_t)(recb / 16));
    if (rc != VI_OK) return rc;
    if (qr) return vi_qr_back_vec(c, bc, N, hh, Cc, (int64_t)(recb / 8));
    return VI_OK;
}

// phase 2: eigenvectors from the logs, D1 = V^T (AWA V), D2 = V^T (R V) - four batched products in groups of fixed size
// (see GEMM_GROUP) - and yt = V^T y.  parr: room for 6 pointer arrays of Bcp entries.
int prep_finish(vi_ctx* c, int64_t bc, int N, const void* log, const int* nrd, const double* d_AWA, const int32_t* recc,
                const double* d_R, const double* d_y, double* Vc, double* D1c, double* D2c, double* ytc, double* T0, double* T1,
                double* scrT, double* scrD, const double** parr, int64_t Bcp)
{
    const int NN = N * N;
    const double** pT0 = parr;
    const double** pV = pT0 + Bcp;
    const double** pT1 = pV + Bcp;
    const double** pR = pT1 + Bcp;
    const double** pD1 = pR + Bcp;
    const double** pD2 = pD1 + Bcp;
    const size_t recb = log_record_bytes(N);
    int rc = vi_jacobi_vectors(c, bc, N, log, JACOBI_MAX_SWEEPS, nrd, Vc, (int64_t)(recb / 16));
    if (rc != VI_OK) return rc;
    if (qr_enabled(N)) {          // eigenvectors of X1 = Q^T X Q -> eigenvectors of X
        const double* hh = (const double*)((const char*)log + vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS));
        if ((rc = vi_qr_back_mat(c, bc, N, hh, Vc, (int64_t)(recb / 8))) != VI_OK) return rc;
    }
    hipLaunchKernelGGL(k_form_system, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, d_AWA, recc, nullptr, nullptr, T0);
    VI_HIP(hipGetLastError());
    if ((rc = group_ptrs(c, bc, T0, NN, nullptr, 0, pT0)) != VI_OK) return rc;
    if ((rc = group_ptrs(c, bc, Vc, NN, nullptr, 0, pV)) != VI_OK) return rc;
    if ((rc = group_ptrs(c, bc, T1, NN, scrT, NN, pT1)) != VI_OK) return rc;
    if ((rc = group_ptrs(c, bc, d_R, 0, nullptr, 0, pR)) != VI_OK) return rc;
    if ((rc = group_ptrs(c, bc, D1c, NN, scrD, NN, pD1)) != VI_OK) return rc;
    if ((rc = group_ptrs(c, bc, D2c, NN, scrD, NN, pD2)) != VI_OK) return rc;
    if ((rc = gemm_groups(c, rocblas_operation_none, rocblas_operation_none, N, N, N, pT0, N, pV, N, pT1, N, bc)) != VI_OK) return rc;
    if ((rc = gemm_groups(c, rocblas_operation_transpose, rocblas_operation_none, N, N, N, pV, N, pT1, N, pD1, N, bc)) != VI_OK)
        return rc;
    if ((rc = gemm_groups(c, rocblas_operation_none, rocblas_operation_none, N, N, N, pR, N, pV, N, pT1, N, bc)) != VI_OK) return rc;
    if ((rc = gemm_groups(c, rocblas_operation_transpose, rocblas_operation_none, N, N, N, pV, N, pT1, N, pD2, N, bc)) != VI_OK)
        return rc;
    hipLaunchKernelGGL(k_vt_vec, dim3((unsigned)bc), dim3(256), 0, c->stream, N, Vc, nullptr, d_y, recc, ytc);
    VI_HIP(hipGetLastError());
    return VI_OK;
}
}  // namespace

extern "C" int vi_warm_prepare_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_AWA, const int32_t* d_rec,
                                   const double* d_alpha0, const double* d_R, const double* d_y, double rcond,
                                   double* d_C, int32_t* d_rank, double* d_V, double* d_D1, double* d_D2, double* d_yt)
{
    VI_REQUIRE(c && d_AWA && d_rec && d_alpha0 && d_R && d_y && d_C && d_V && d_D1 && d_D2 && d_yt, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_jacobi_vectors_supported(N)) {
        vi_set_error("vi_warm_prepare_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    const size_t logb = log_record_bytes(N);
    const size_t per = logb + (size_t)(2 * N + 1) * sizeof(double) + sizeof(int) + (size_t)2 * NN * sizeof(double) + 6 * sizeof(void*);
    // chunks of records, so that the rotation logs and temporaries stay within 4 GiB of workspace (4.3 MB per record at
    // N = 144: 10 000 records at once would ask for 43 GB)
    int64_t Bc = (int64_t)(((size_t)4 << 30) / per);
    if (Bc < 1) Bc = 1;
    if (Bc > B) Bc = B;
    const int64_t Bcp = group_pad(Bc);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)Bc * per + (size_t)2 * GEMM_GROUP * NN * sizeof(double) + 6 * GEMM_GROUP * sizeof(void*) + 1024,
                              &ws);
    if (rc != VI_OK) return rc;
    char* wp = (char*)ws + (size_t)Bc * logb;
    double* scl = (double*)wp;
    double* lam = scl + Bc;
    double* T0 = lam + (size_t)Bc * N;     // X0, later AWA[rec]
    double* T1 = T0 + (size_t)Bc * NN;
    double* scrT = T1 + (size_t)Bc * NN;   // results of the padding entries of the last product group
    double* scrD = scrT + (size_t)GEMM_GROUP * NN;
    const double** parr = (const double**)(scrD + (size_t)GEMM_GROUP * NN);
    double* y1 = (double*)(parr + 6 * Bcp);
    int* nrd = (int*)(y1 + (size_t)Bc * N);
    for (int64_t i0 = 0; i0 < B; i0 += Bc) {
        const int64_t bc = (B - i0) < Bc ? (B - i0) : Bc;
        rc = prep_decompose(c, bc, N, d_AWA, d_rec + i0, d_alpha0 + i0, d_R, d_y, rcond, d_C + i0 * N,
                            d_rank ? d_rank + i0 : nullptr, ws, scl, lam, T0, nrd, y1, T1);
        if (rc != VI_OK) return rc;
        rc = prep_finish(c, bc, N, ws, nrd, d_AWA, d_rec + i0, d_R, d_y, d_V + i0 * NN, d_D1 + i0 * NN, d_D2 + i0 * NN,
                         d_yt + i0 * N, T0, T1, scrT, scrD, parr, Bcp);
        if (rc != VI_OK) return rc;
    }
    return VI_OK;
}

// The two phases of vi_warm_prepare_f64 as separate calls, with the rotation logs in a buffer of the caller
// (vi_rotation_log_bytes(N) per system): a record fitted alone decomposes the systems at the middle of ALL its candidate
// brackets in the launch of its bracket walk (room for 256 systems, as long as the slowest one) and finishes - eigenvectors
// and the three products - only the one the walk then points at.
extern "C" size_t vi_rotation_log_bytes(int32_t N) { return log_record_bytes(N); }
extern "C" int vi_max_sweeps(void) { return JACOBI_MAX_SWEEPS; }
double vi_floor_warm() { return JACOBI_FLOOR_WARM; }

// End of the Jacobi iteration in the solves of the bracket walk (vi_basis_solve_f64), whose chi^2 only decides signs - values
// within 5e-3 of the target are asked for again from cold solves (alpha_search.WALK_SIGN_MARGIN): every |a_pq| <= 1e-6 x
// sqrt|a_pp a_qq| instead of eps x.  Measured on 1000 records x 49 decades (tools/exp_walk_floor.py, tools/exp_walk_tol.sh):
// rotating sweeps per system 2.21 -> 1.59, the walk 141 -> 124 ms, and the deviation of the walk's chi^2 from the cold one
// unchanged to three digits (max 1.13e-3, 99.9 % 2.94e-4 - it comes from eigenvalues next to the truncation cut, not from
// the end of the iteration; at 1e-4 the maximum grows to 2.4e-3, at 1e-3 to 1.2e-2).  An off-diagonal element of relative
// size tol moves an eigenvalue by tol^2 of itself and a coefficient's contribution to the fit by tol of the data norm.
// VINTERP_WALK_TOL overrides (0 = the criterion of all other solves).
namespace {
double walk_tolerance()
{
    static const double v = [] {
        const char* e = getenv("VINTERP_WALK_TOL");
        return e ? atof(e) : 1e-6;
    }();
    return v;
}
}  // namespace

extern "C" int vi_decompose_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_AWA, const int32_t* d_rec,
                                const double* d_alpha0, const double* d_R, const double* d_y, double rcond, double* d_C,
                                int32_t* d_rank, void* d_log, int32_t* d_nround)
{
    VI_REQUIRE(c && d_AWA && d_rec && d_alpha0 && d_R && d_y && d_C && d_log && d_nround, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_jacobi_vectors_supported(N)) {
        vi_set_error("vi_decompose_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)B * ((size_t)(2 * N + 1 + 2 * NN) * sizeof(double)) + 1024, &ws);
    if (rc != VI_OK) return rc;
    double* scl = (double*)ws;
    double* lam = scl + B;
    double* T0 = lam + (size_t)B * N;
    double* scr = T0 + (size_t)B * NN;
    double* y1 = scr + (size_t)B * NN;
    return prep_decompose(c, B, N, d_AWA, d_rec, d_alpha0, d_R, d_y, rcond, d_C, d_rank, d_log, scl, lam, T0, d_nround, y1, scr);
}

extern "C" int vi_warm_finish_f64(vi_ctx* c, int64_t B, int32_t N, const void* d_log, const int32_t* d_nround,
                                  const double* d_AWA, const int32_t* d_rec, const double* d_R, const double* d_y,
                                  double* d_V, double* d_D1, double* d_D2, double* d_yt)
{
    VI_REQUIRE(c && d_log && d_nround && d_AWA && d_rec && d_R && d_y && d_V && d_D1 && d_D2 && d_yt, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_jacobi_vectors_supported(N)) {
        vi_set_error("vi_warm_finish_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    const int64_t Bp = group_pad(B);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)(2 * B + 2 * GEMM_GROUP) * NN * sizeof(double) + (size_t)6 * Bp * sizeof(void*) + 1024, &ws);
    if (rc != VI_OK) return rc;
    double* T0 = (double*)ws;
    double* T1 = T0 + (size_t)B * NN;
    double* scrT = T1 + (size_t)B * NN;
    double* scrD = scrT + (size_t)GEMM_GROUP * NN;
    const double** parr = (const double**)(scrD + (size_t)GEMM_GROUP * NN);
    return prep_finish(c, B, N, d_log, d_nround, d_AWA, d_rec, d_R, d_y, d_V, d_D1, d_D2, d_yt, T0, T1, scrT, scrD, parr, Bp);
}

extern "C" int vi_warm_solve_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_D1, const double* d_D2,
                                 const double* d_yt, const double* d_V, const int32_t* d_slot, const double* d_alpha,
                                 double rcond, double* d_C, int32_t* d_rank, int32_t* d_sweeps)
{
    VI_REQUIRE(c && d_D1 && d_D2 && d_yt && d_V && d_slot && d_alpha && d_C, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_jacobi_supported(N)) {
        vi_set_error("vi_warm_solve_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS);
    int64_t Bc = (int64_t)(((size_t)4 << 30) / logb);
    if (Bc < 1) Bc = 1;
    if (Bc > B) Bc = B;
    void* ws = nullptr;
    const size_t per = logb + sizeof(double) + (size_t)NN * sizeof(double) + (size_t)N * sizeof(double);
    int rc = vi_ctx_workspace(c, (size_t)Bc * per + 1024, &ws);
    if (rc != VI_OK) return rc;
    char* wp = (char*)ws + (size_t)Bc * logb;
    double* scl = (double*)wp;
    double* X = scl + Bc;
    double* cp = X + (size_t)Bc * NN;
    for (int64_t i0 = 0; i0 < B; i0 += Bc) {
        const int64_t bc = (B - i0) < Bc ? (B - i0) : Bc;
        form_pair_scaled(c, bc, NN, d_D1, d_D2, d_slot + i0, d_slot + i0, d_alpha + i0, X, scl);
        VI_HIP(hipGetLastError());
        rc = vi_jacobi_solve(c, bc, N, X, scl, d_yt, d_slot + i0, rcond, cp, d_rank ? d_rank + i0 : nullptr, ws,
                             JACOBI_MAX_SWEEPS, d_sweeps ? d_sweeps + i0 : nullptr, nullptr, 0, nullptr, JACOBI_FLOOR_WARM);
        if (rc != VI_OK) return rc;
        hipLaunchKernelGGL(k_v_vec, dim3((unsigned)bc), dim3(256), (size_t)N * sizeof(double), c->stream, N, d_V,
                           d_slot + i0, cp, d_C + i0 * N);
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}

namespace {
// out[t] = min over the elements with R != 0 of (eps / 4) |AWA[t]| / |R|
template <int BS>
__global__ __launch_bounds__(BS) void k_reg_floor(int NN, const double* __restrict__ AWA, const double* __restrict__ R,
                                                  double* __restrict__ out)
{
    __shared__ double red[BS / 64];
    const double* Xt = AWA + (int64_t)blockIdx.x * NN;
    double mn = 1.7976931348623157e308;
    for (int e = threadIdx.x; e < NN; e += BS) {
        const double r = fabs(R[e]);
        if (r > 0.0) mn = fmin(mn, (0.25 * 2.220446049250313e-16) * fabs(Xt[e]) / r);
    }
    for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, __shfl_xor(mn, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < BS / 64; ++q) mn = fmin(mn, red[q]);
        out[blockIdx.x] = mn;
    }
}
}  // namespace

// Below which alpha does the regularisation term vanish from X(alpha) = AWA + alpha R in floating point?  For
// alpha |R_ij| < (eps / 4) |AWA_ij| in every element, fl(AWA_ij + alpha R_ij) = AWA_ij (the term is under half an ulp), so
// all the systems of the bracket walk below that alpha are the SAME matrix bit for bit - in the reference as well, whose
// LAPACK calls then return the same numbers again and again - and need one solve, not one per decade.
extern "C" int vi_reg_floor_f64(vi_ctx* c, int64_t T, int32_t N, const double* d_AWA, const double* d_R, double* d_out)
{
    VI_REQUIRE(c && d_AWA && d_R && d_out, "null argument");
    VI_REQUIRE(T >= 0 && N > 0, "bad size");
    if (T == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_reg_floor<256>, dim3((unsigned)T), dim3(256), 0, c->stream, N * N, d_AWA, d_R, d_out);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

namespace {
// pointer arrays of the two batched products of vi_basis_solve_f64
__global__ void k_basis_ptrs(int64_t B, int NN, const double* AWA, const int* __restrict__ rec, const double* V,
                             const int* __restrict__ basis, double* T1, double* D1, const double** pA, const double** pV,
                             double** pT, double** pD)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    pA[i] = AWA + (int64_t)rec[i] * NN;
    pV[i] = V + (int64_t)basis[i] * NN;
    pT[i] = T1 + i * NN;
    pD[i] = D1 + i * NN;
}
}  // namespace

// The bracket walk in SHARED bases.  The records of one geometry differ in their weights, not in their basis functions
// or their regularisation matrix, so their walk systems X_j(alpha_k) = AWA_j + alpha_k R resemble each other decade by
// decade: in the eigenbasis V_k of ONE reference system of the decade (mean weights of the batch; set up with
// vi_warm_prepare_f64, which also gives D2_k = V_k^T R V_k) the system of every other record is nearly diagonal and
// the Jacobi iteration needs 1-4 sweeps instead of 8 (alpha >= 1e-22) or 19-24 (below) - measured on the BASELINE
// configs[2] geometry, tools/exp_shared_basis.py: 1349 -> 197 sweeps over 28 decades x 3 records.  For B (record, basis,
// alpha) triples: D1 = V^T AWA V (two batched rocBLAS products, 8 N^3 flop), (D1 + alpha D2) c' = V^T y, C = V c'.
// Any orthonormal V gives the same solution in exact arithmetic - a poor reference costs sweeps, not correctness.
extern "C" int vi_basis_solve_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_AWA, const double* d_y,
                                  const int32_t* d_rec, const int32_t* d_basis, const double* d_alpha, const double* d_V,
                                  const double* d_D2, double rcond, double* d_C, int32_t* d_rank, int32_t* d_sweeps)
{
    VI_REQUIRE(c && d_AWA && d_y && d_rec && d_basis && d_alpha && d_V && d_D2 && d_C, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_jacobi_supported(N)) {
        vi_set_error("vi_basis_solve_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS);
    const size_t per = logb + sizeof(double) + (size_t)2 * NN * sizeof(double) + (size_t)2 * N * sizeof(double) + 4 * sizeof(void*);
    int64_t Bc = (int64_t)(((size_t)8 << 30) / per);
    if (Bc > 256) Bc &= ~(int64_t)255;            // whole rounds of the 256 CUs
    if (Bc < 1) Bc = 1;
    if (Bc > B) Bc = B;
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)Bc * per + 1024, &ws);
    if (rc != VI_OK) return rc;
    char* wp = (char*)ws + (size_t)Bc * logb;
    double* scl = (double*)wp;
    double* T1 = scl + Bc;
    double* D1 = T1 + (size_t)Bc * NN;            // D1, then the scaled system in place
    double* yt = D1 + (size_t)Bc * NN;
    double* cp = yt + (size_t)Bc * N;
    const double** pA = (const double**)(cp + (size_t)Bc * N);
    const double** pV = pA + Bc;
    double** pT = (double**)(pV + Bc);
    double** pD = pT + Bc;
    const double one = 1.0, zero = 0.0;
    for (int64_t i0 = 0; i0 < B; i0 += Bc) {
        const int64_t bc = (B - i0) < Bc ? (B - i0) : Bc;
        hipLaunchKernelGGL(k_basis_ptrs, dim3((unsigned)((bc + 255) / 256)), dim3(256), 0, c->stream, bc, NN, d_AWA,
                           d_rec + i0, d_V, d_basis + i0, T1, D1, pA, pV, pT, pD);
        VI_HIP(hipGetLastError());
        VI_ROCBLAS(rocblas_dgemm_batched(c->blas, rocblas_operation_none, rocblas_operation_none, N, N, N, &one, pA, N,
                                         pV, N, &zero, pT, N, (rocblas_int)bc));
        VI_ROCBLAS(rocblas_dgemm_batched(c->blas, rocblas_operation_transpose, rocblas_operation_none, N, N, N, &one, pV, N,
                                         (const double* const*)pT, N, &zero, pD, N, (rocblas_int)bc));
        hipLaunchKernelGGL(k_vt_vec, dim3((unsigned)bc), dim3(256), 0, c->stream, N, d_V, d_basis + i0, d_y, d_rec + i0, yt);
        form_pair_scaled(c, bc, NN, D1, d_D2, nullptr, d_basis + i0, d_alpha + i0, D1, scl);
        VI_HIP(hipGetLastError());
        rc = vi_jacobi_solve(c, bc, N, D1, scl, yt, nullptr, rcond, cp, d_rank ? d_rank + i0 : nullptr, ws,
                             JACOBI_MAX_SWEEPS, d_sweeps ? d_sweeps + i0 : nullptr, nullptr, 0, nullptr, JACOBI_FLOOR_WARM, 0,
                             walk_tolerance());
        if (rc != VI_OK) return rc;
        hipLaunchKernelGGL(k_v_vec, dim3((unsigned)bc), dim3(256), (size_t)N * sizeof(double), c->stream, N, d_V,
                           d_basis + i0, cp, d_C + i0 * N);
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}

// Re-basing of rotated systems.  Brent's iterates close in on the root within a few steps, and the cost of a warm solve
// grows with the distance between alpha and the alpha0 the rotated system was set up at (measured, N = 144: 13 sweeps at
// 0.5 decades, 5-9 at 0.1, 3 at 1e-3, 2 at 1e-6, 1 at 1e-9; tools/exp_rebase_sweeps.py) - the middle of the unit bracket,
// typically 0.1-0.5 decades from the root.  For B (slot, record, alpha) triples this call solves (D1 + alpha D2) c' = yt
// like vi_warm_solve_f64 (C = V c' is returned) AND moves the slot's basis to alpha: the eigenvectors Vw of the rotated
// system come out of the rotation log, V <- V Vw, and D1 = V^T AWA V, D2 = V^T R V, yt = V^T y are formed again from the
// untransformed matrices (no compounding of the transform's rounding).  One warm solve + the eigenvector replay + six
// small products instead of a cold decomposition.
// nplain > 0: the first nplain triples are plain warm solves (vi_warm_solve_f64) that ride in the same eigen-solve launch -
// a round of Brent's iteration of a batch has both kinds, and a launch lasts as long as its slowest system whether it holds
// 30 systems or 250.
extern "C" int vi_warm_rebase_f64(vi_ctx* c, int64_t B, int64_t nplain, int32_t N, const double* d_AWA, const double* d_R,
                                  const double* d_y, const int32_t* d_rec, const int32_t* d_slot, const double* d_alpha,
                                  double rcond, double* d_V, double* d_D1, double* d_D2, double* d_yt, double* d_C,
                                  int32_t* d_rank, int32_t* d_sweeps)
{
    VI_REQUIRE(c && d_AWA && d_R && d_y && d_rec && d_slot && d_alpha && d_V && d_D1 && d_D2 && d_yt && d_C, "null argument");
    VI_REQUIRE(B >= 0 && N > 0 && nplain >= 0 && nplain <= B, "bad size");
    if (B == 0) return VI_OK;
    if (nplain == B) return vi_warm_solve_f64(c, B, N, d_D1, d_D2, d_yt, d_V, d_slot, d_alpha, rcond, d_C, d_rank, d_sweeps);
    if (!vi_jacobi_vectors_supported(N)) {
        vi_set_error("vi_warm_rebase_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    const size_t logb = vi_jacobi_log_bytes(N, JACOBI_MAX_SWEEPS);
    const size_t per = logb + sizeof(double) + sizeof(int) + (size_t)3 * NN * sizeof(double) + (size_t)N * sizeof(double) +
                       8 * sizeof(void*);
    int64_t Bc = (int64_t)(((size_t)4 << 30) / per);
    if (Bc < 1) Bc = 1;
    if (nplain > 0 && B > Bc) {
        // more than one workspace chunk: the plain solves on their own, then the re-basing ones
        int rc0 = vi_warm_solve_f64(c, nplain, N, d_D1, d_D2, d_yt, d_V, d_slot, d_alpha, rcond, d_C, d_rank, d_sweeps);
        if (rc0 != VI_OK) return rc0;
        return vi_warm_rebase_f64(c, B - nplain, 0, N, d_AWA, d_R, d_y, d_rec + nplain, d_slot + nplain, d_alpha + nplain, rcond,
                                  d_V, d_D1, d_D2, d_yt, d_C + nplain * N, d_rank ? d_rank + nplain : nullptr,
                                  d_sweeps ? d_sweeps + nplain : nullptr);
    }
    if (Bc > B) Bc = B;
    const int64_t Bcp = group_pad(Bc);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)Bc * per + (size_t)2 * GEMM_GROUP * NN * sizeof(double) + 8 * GEMM_GROUP * sizeof(void*) + 1024,
                              &ws);
    if (rc != VI_OK) return rc;
    char* wp = (char*)ws + (size_t)Bc * logb;
    double* scl = (double*)wp;
    double* X = scl + Bc;                       // the scaled rotated system, later AWA[rec]
    double* Vw = X + (size_t)Bc * NN;           // eigenvectors of the rotated system, later a product
    double* Vn = Vw + (size_t)Bc * NN;          // V Vw
    double* cp = Vn + (size_t)Bc * NN;
    double* scrA = cp + (size_t)Bc * N;
    double* scrB = scrA + (size_t)GEMM_GROUP * NN;
    const double** pX = (const double**)(scrB + (size_t)GEMM_GROUP * NN);
    const double** pVold = pX + Bcp;
    const double** pVw = pVold + Bcp;
    const double** pVn = pVw + Bcp;
    const double** pR = pVn + Bcp;
    const double** pD1 = pR + Bcp;
    const double** pD2 = pD1 + Bcp;
    const double** pVslot = pD2 + Bcp;
    int* nrd = (int*)(pVslot + Bcp);
    for (int64_t i0 = 0; i0 < B; i0 += Bc) {
        int64_t bc = (B - i0) < Bc ? (B - i0) : Bc;
        const int32_t* slotc = d_slot + i0;
        const int32_t* recc = d_rec + i0;
        form_pair_scaled(c, bc, NN, d_D1, d_D2, slotc, slotc, d_alpha + i0, X, scl);
        VI_HIP(hipGetLastError());
        rc = vi_jacobi_solve(c, bc, N, X, scl, d_yt, slotc, rcond, cp, d_rank ? d_rank + i0 : nullptr, ws, JACOBI_MAX_SWEEPS,
                             d_sweeps ? d_sweeps + i0 : nullptr, nullptr, 0, nrd, JACOBI_FLOOR_WARM);
        if (rc != VI_OK) return rc;
        hipLaunchKernelGGL(k_v_vec, dim3((unsigned)bc), dim3(256), (size_t)N * sizeof(double), c->stream, N, d_V, slotc, cp,
                           d_C + i0 * N);
        VI_HIP(hipGetLastError());
        // from here on only the systems that re-base (nplain > 0 implies a single chunk, i0 = 0)
        const int64_t skip = i0 == 0 ? nplain : 0;
        rc = vi_jacobi_vectors(c, bc - skip, N, (const char*)ws + (size_t)skip * logb, JACOBI_MAX_SWEEPS, nrd + skip, Vw);
        if (rc != VI_OK) return rc;
        slotc += skip;
        recc += skip;
        bc -= skip;
        // V_new = V_old Vw (into Vn, then back into the slots), then the rotated system from the untransformed matrices
        if ((rc = group_ptrs(c, bc, d_V, NN, nullptr, 0, pVold, slotc)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, bc, Vw, NN, scrA, NN, pVw)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, bc, Vn, NN, scrB, NN, pVn)) != VI_OK) return rc;
        // (own products with a fixed summation order, not the library's: k_brent_warm re-bases with the same code)
        if ((rc = wg_gemm_batched(c, false, N, pVold, pVw, pVn, bc)) != VI_OK) return rc;
        hipLaunchKernelGGL(k_scatter_mat, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, Vn, slotc, d_V);
        hipLaunchKernelGGL(k_form_system, dim3((unsigned)bc), dim3(256), 0, c->stream, NN, d_AWA, recc, nullptr, nullptr, X);
        VI_HIP(hipGetLastError());
        if ((rc = group_ptrs(c, bc, X, NN, nullptr, 0, pX)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, bc, d_R, 0, nullptr, 0, pR)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, bc, d_D1, NN, scrB, NN, pD1, slotc)) != VI_OK) return rc;
        if ((rc = group_ptrs(c, bc, d_D2, NN, scrB, NN, pD2, slotc)) != VI_OK) return rc;
        // (pVn reads Vn, which still holds V_new; pVw now names the product buffer)
        if ((rc = wg_gemm_batched(c, false, N, pX, pVn, pVw, bc)) != VI_OK) return rc;
        if ((rc = wg_gemm_batched(c, true, N, pVn, pVw, pD1, bc)) != VI_OK) return rc;
        if ((rc = wg_gemm_batched(c, false, N, pR, pVn, pVw, bc)) != VI_OK) return rc;
        if ((rc = wg_gemm_batched(c, true, N, pVn, pVw, pD2, bc)) != VI_OK) return rc;
        hipLaunchKernelGGL(k_vt_vec_slot, dim3((unsigned)bc), dim3(256), 0, c->stream, N, Vn, d_y, recc, slotc, d_yt);
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}

// One root-finder iterate of ONE record in a single call (the latency path of a single-record fit: ~14 dependent
// iterates per record): the scalars travel as kernel arguments instead of three host-to-device copies, and the only
// synchronisation is the 8-byte read-back of chi^2.  d_scratch: N + 8 doubles owned by the caller.
namespace {
__global__ void k_set_one(double* scratch, double alpha, int slot, int rec)
{
    scratch[0] = alpha;
    int* iv = reinterpret_cast<int*>(scratch + 2);
    iv[0] = slot;
    iv[1] = rec;
}
}  // namespace

extern "C" int vi_warm_chi2_one_f64(vi_ctx* c, int32_t N, int64_t P, const double* d_D1, const double* d_D2,
                                    const double* d_yt, const double* d_V, int32_t slot, double alpha, double rcond,
                                    const double* d_At, int32_t rec, const double* d_W, const double* d_b,
                                    double* d_scratch, double* h_chi2)
{
    VI_REQUIRE(c && d_D1 && d_D2 && d_yt && d_V && d_At && d_W && d_b && d_scratch && h_chi2, "null argument");
    VI_REQUIRE(N > 0 && P > 0 && slot >= 0 && rec >= 0, "bad size");
    VI_HIP(hipSetDevice(c->device));
    double* d_alpha = d_scratch;
    double* d_chi = d_scratch + 1;
    int32_t* d_slot = reinterpret_cast<int32_t*>(d_scratch + 2);
    int32_t* d_rec = d_slot + 1;
    int32_t* d_sw = reinterpret_cast<int32_t*>(d_scratch + 3);
    double* d_C = d_scratch + 8;
    hipLaunchKernelGGL(k_set_one, dim3(1), dim3(1), 0, c->stream, d_scratch, alpha, (int)slot, (int)rec);
    VI_HIP(hipGetLastError());
    int rc = vi_warm_solve_f64(c, 1, N, d_D1, d_D2, d_yt, d_V, d_slot, d_alpha, rcond, d_C, nullptr, d_sw);
    if (rc != VI_OK) return rc;
    rc = vi_chi2_f64(c, 1, P, N, d_At, d_C, d_rec, d_W, d_b, d_chi);
    if (rc != VI_OK) return rc;
    // chi^2, alpha's slot word and the sweep count travel back together: h_chi2[0] = chi^2, and the low 32 bits of the
    // third double are the sweeps of the solve (cap + 1: it did not converge and the value must not be used)
    VI_HIP(hipMemcpyAsync(h_chi2, d_chi, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    VI_HIP(hipStreamSynchronize(c->stream));
    return VI_OK;
}

// ---- generalised cross validation objective (interpolate.py:299-351) -----------------------------------
// For one record (normal equations d_AWA (N x N), d_y (N), weights / data d_W, d_b (P)) and one alpha: the sum
// over the np listed data points of the squared, weighted residual of each point against the fit that leaves it
// out.  np truncated eigen-solves in one batch.  d_res receives the np individual terms.
extern "C" int vi_gcv_terms_f64(vi_ctx* c, int64_t np, int64_t P, int32_t N, const double* d_At, const int32_t* d_pidx,
                                const double* d_AWA, const double* d_y, const double* d_W, const double* d_b,
                                double alpha, const double* d_R, double rcond, double* d_res)
{
    VI_REQUIRE(c && d_At && d_pidx && d_AWA && d_y && d_W && d_b && d_res, "null argument");
    VI_REQUIRE(np >= 0 && P > 0 && N > 0, "bad size");
    if (np == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    const int NN = N * N;
    // systems are formed and solved in chunks; X / y / C live in a private allocation because
    // vi_solve_trunc_f64 uses the context workspace itself
    int64_t Bc = (int64_t)(((size_t)1 << 30) / ((size_t)NN * sizeof(double)));
    if (Bc < 1) Bc = 1;
    if (Bc > np) Bc = np;
    double* buf = nullptr;
    VI_HIP(hipMalloc((void**)&buf, (size_t)Bc * (NN + 2 * (size_t)N) * sizeof(double)));
    double* X = buf;
    double* yl = X + (size_t)Bc * NN;
    double* Cl = yl + (size_t)Bc * N;
    int rc = VI_OK;
    for (int64_t i0 = 0; i0 < np && rc == VI_OK; i0 += Bc) {
        const int64_t bc = (np - i0) < Bc ? (np - i0) : Bc;
        hipLaunchKernelGGL(k_form_loo, dim3((unsigned)bc), dim3(256), (size_t)N * sizeof(double), c->stream, N, P, d_At,
                           d_pidx + i0, d_AWA, d_y, d_W, d_b, alpha, d_R, X, yl);
        if (hipGetLastError() != hipSuccess) { vi_set_error("k_form_loo launch failed"); rc = VI_ERR_HIP; break; }
        rc = vi_solve_trunc_f64(c, bc, N, X, yl, nullptr, rcond, Cl, nullptr, 0.0, nullptr);
        if (rc != VI_OK) break;
        hipLaunchKernelGGL(k_loo_resid, dim3((unsigned)((bc + 3) / 4)), dim3(256), 0, c->stream, bc, N, P, d_At, d_pidx + i0,
                           Cl, d_W, d_b, d_res + i0);
        if (hipGetLastError() != hipSuccess) { vi_set_error("k_loo_resid launch failed"); rc = VI_ERR_HIP; }
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(buf);
    return rc;
}
