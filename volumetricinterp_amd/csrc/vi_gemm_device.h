// C = A B or A^T B for small square matrices by ONE workgroup, with a summation order that is a property of the code:
// every element is one fma chain over k = 0 .. N-1.  Used where a product feeds a record's numbers on two paths that must
// agree bit for bit: the re-basing of a rotated system by the host-driven search (vi_warm_rebase_f64, one workgroup per
// matrix) and by the device-side search (k_brent_warm, inside the record's workgroup).  (The library GEMM picks its kernel -
// and with it the order of its sums - by the problem size and the batch count.)
// All matrices N x N, column-major (element (r, c) at [c * N + r] - the LAPACK layout of the eigenvector matrices).
#pragma once
#include "vi_common.h"

namespace {

constexpr int WG_GEMM_KP = 16;          // depth of a k panel staged in LDS

inline __host__ __device__ size_t wg_gemm_lds_doubles(int N) { return (size_t)2 * WG_GEMM_KP * ((N + 7) & ~7); }

// lds: wg_gemm_lds_doubles(N) doubles.  All threads of the workgroup call it; A, B are only read, C must not alias them.
// Thread tiles of TS x TS outputs (6: 24 x 24 tiles = 576 threads at N = 144); per k a thread reads TS + TS operands from LDS.
// A workgroup computes the thread tiles [ti0, ti0 + nti) x [tj0, tj0 + ntj) of the product (all of them by default): an
// element's value does not depend on how the product is cut into workgroups, so the host path spreads a product over
// several of them (k_wg_gemm) and still gets the bits of the one workgroup of k_brent_warm.
template <bool TA, int TS = 6>
__device__ __forceinline__ void wg_gemm(int N, const double* __restrict__ A, const double* __restrict__ B,
                                        double* __restrict__ C, double* __restrict__ lds, int ti0 = 0, int nti = -1, int tj0 = 0,
                                        int ntj = -1)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    const int NP = (N + 7) & ~7;                 // padded leading dimension of the LDS panels (zeros beyond N)
    double* As = lds;                            // [KP][NP]  As[kk][r] = op(A)(r, kp + kk)
    double* Bs = lds + WG_GEMM_KP * NP;          // [KP][NP]  Bs[kk][c] = B(kp + kk, c)
    const int nt = (N + TS - 1) / TS;            // tiles per dimension
    if (nti < 0) nti = nt - ti0;
    if (ntj < 0) ntj = nt - tj0;
    if (nti > nt - ti0) nti = nt - ti0;
    if (ntj > nt - tj0) ntj = nt - tj0;
    const int ntile = nti > 0 && ntj > 0 ? nti * ntj : 0;
    const int r_lo = TS * ti0, r_hi = TS * (ti0 + nti) < NP ? TS * (ti0 + nti) : NP;        // rows / columns of the panels this
    const int c_lo = TS * tj0, c_hi = TS * (tj0 + ntj) < NP ? TS * (tj0 + ntj) : NP;        // workgroup reads
    const int nr = r_hi - r_lo, nc = c_hi - c_lo;
    for (int t0 = 0; t0 < ntile; t0 += NT) {
        const int t = t0 + tid;
        const bool live = t < ntile;
        const int ti = ti0 + (live ? t % nti : 0), tj = tj0 + (live ? t / nti : 0);
        double acc[TS][TS];
#pragma unroll
        for (int i = 0; i < TS; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) acc[i][j] = 0.0;
        for (int kp = 0; kp < N; kp += WG_GEMM_KP) {
            __syncthreads();
            for (int e = tid; e < WG_GEMM_KP * nr; e += NT) {
                if (TA) {
                    const int r = r_lo + e / WG_GEMM_KP, kk = e % WG_GEMM_KP;           // A^T(r, k) = A(k, r): k contiguous
                    As[kk * NP + r] = (r < N && kp + kk < N) ? A[(int64_t)r * N + kp + kk] : 0.0;
                } else {
                    const int kk = e / nr, r = r_lo + e % nr;                           // A(r, k): r contiguous
                    As[kk * NP + r] = (r < N && kp + kk < N) ? A[(int64_t)(kp + kk) * N + r] : 0.0;
                }
            }
            for (int e = tid; e < WG_GEMM_KP * nc; e += NT) {
                const int c = c_lo + e / WG_GEMM_KP, kb = e % WG_GEMM_KP;               // B(k, c): k contiguous
                Bs[kb * NP + c] = (c < N && kp + kb < N) ? B[(int64_t)c * N + kp + kb] : 0.0;
            }
            __syncthreads();
            if (live) {
                const int kmax = (N - kp) < WG_GEMM_KP ? (N - kp) : WG_GEMM_KP;
                for (int kk = 0; kk < kmax; ++kk) {
                    double a[TS], b[TS];
#pragma unroll
                    for (int i = 0; i < TS; ++i) a[i] = (TS * ti + i < NP) ? As[kk * NP + TS * ti + i] : 0.0;
#pragma unroll
                    for (int j = 0; j < TS; ++j) b[j] = (TS * tj + j < NP) ? Bs[kk * NP + TS * tj + j] : 0.0;
#pragma unroll
                    for (int i = 0; i < TS; ++i)
#pragma unroll
                        for (int j = 0; j < TS; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < TS; ++j)
#pragma unroll
                for (int i = 0; i < TS; ++i) {
                    const int r = TS * ti + i, c = TS * tj + j;
                    if (r < N && c < N) C[(int64_t)c * N + r] = acc[i][j];
                }
        }
    }
    __syncthreads();
}

}  // namespace
