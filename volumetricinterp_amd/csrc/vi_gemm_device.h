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
// Thread tiles of 6 x 6 outputs (24 x 24 tiles = 576 threads at N = 144); per k a thread reads 6 + 6 operands from LDS.
template <bool TA>
__device__ __forceinline__ void wg_gemm(int N, const double* __restrict__ A, const double* __restrict__ B,
                                        double* __restrict__ C, double* __restrict__ lds)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    const int NP = (N + 7) & ~7;                 // padded leading dimension of the LDS panels (zeros beyond N)
    double* As = lds;                            // [KP][NP]  As[kk][r] = op(A)(r, kp + kk)
    double* Bs = lds + WG_GEMM_KP * NP;          // [KP][NP]  Bs[kk][c] = B(kp + kk, c)
    const int nt = (N + 5) / 6;                  // tiles per dimension
    const int ntile = nt * nt;
    for (int t0 = 0; t0 < ntile; t0 += NT) {
        const int t = t0 + tid;
        const bool live = t < ntile;
        const int ti = live ? t % nt : 0, tj = live ? t / nt : 0;
        double acc[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = 0.0;
        for (int kp = 0; kp < N; kp += WG_GEMM_KP) {
            __syncthreads();
            for (int e = tid; e < WG_GEMM_KP * NP; e += NT) {
                if (TA) {
                    const int r = e / WG_GEMM_KP, kk = e - r * WG_GEMM_KP;              // A^T(r, k) = A(k, r): k contiguous
                    As[kk * NP + r] = (r < N && kp + kk < N) ? A[(int64_t)r * N + kp + kk] : 0.0;
                } else {
                    const int kk = e / NP, r = e - kk * NP;                             // A(r, k): r contiguous
                    As[kk * NP + r] = (r < N && kp + kk < N) ? A[(int64_t)(kp + kk) * N + r] : 0.0;
                }
                const int c = e / WG_GEMM_KP, kb = e - c * WG_GEMM_KP;                  // B(k, c): k contiguous
                Bs[kb * NP + c] = (c < N && kp + kb < N) ? B[(int64_t)c * N + kp + kb] : 0.0;
            }
            __syncthreads();
            if (live) {
                const int kmax = (N - kp) < WG_GEMM_KP ? (N - kp) : WG_GEMM_KP;
                for (int kk = 0; kk < kmax; ++kk) {
                    double a[6], b[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) a[i] = (6 * ti + i < NP) ? As[kk * NP + 6 * ti + i] : 0.0;
#pragma unroll
                    for (int j = 0; j < 6; ++j) b[j] = (6 * tj + j < NP) ? Bs[kk * NP + 6 * tj + j] : 0.0;
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int r = 6 * ti + i, c = 6 * tj + j;
                    if (r < N && c < N) C[(int64_t)c * N + r] = acc[i][j];
                }
        }
    }
    __syncthreads();
}

}  // namespace
