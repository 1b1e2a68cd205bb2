// K3: batched truncated minimum-norm solve of small symmetric (indefinite, possibly rank-deficient)
// systems by a parallel cyclic Jacobi eigensolver that lives entirely in one CU's LDS.
//
// Replaces, for the hundreds of trial systems per record of the regularisation-parameter search,
// scipy.linalg.lstsq (LAPACK gelsd, rcond = eps) at volumetricinterp/interpolate.py:462:
// for symmetric X = V diag(lam) V^T the SVD truncation |sigma| <= rcond*sigma_max is the spectral
// truncation |lam| <= rcond*max|lam|, and the minimum-norm solution is V diag(1/lam | kept) V^T y.
//
// One 256-thread workgroup per system.  The symmetric half of X (N(N+1)/2 doubles: 83.5 KB at
// N = 144, inside the 160 KB LDS of a gfx950 CU) stays in LDS for the whole decomposition.  A round
// of the round-robin ordering holds N/2 disjoint index pairs; the two-sided update A <- J^T A J then
// factorises into independent 2x2 blocks B_PQ <- R_P^T B_PQ R_Q over pairs-of-pairs, which is what the
// threads iterate over - no atomics, two barriers per round.  The eigenvector matrix is never formed:
// the right-hand side is rotated along (y <- J^T y), the rotations (c, s) are logged to HBM (16 B each,
// ~1.3 MB per system at N = 144) and replayed in reverse on the truncated, scaled vector to give
// C = V g.  Convergence is the relative criterion |a_pq| <= eps*sqrt(|a_pp a_qq|) for every pair of a
// whole sweep, which keeps small eigenvalues accurate to high relative accuracy.
#include "vi_common.h"

namespace {

constexpr int JBS = 256;

__device__ __forceinline__ int tri(int i, int j)      // index into the packed lower triangle
{
    const int a = i > j ? i : j, b = i > j ? j : i;
    return ((a * (a + 1)) >> 1) + b;
}

struct JacobiLds {
    double* A;            // [Np(Np+1)/2]
    double* yv;           // [Np]
    double2* cs;          // [m]
    double* nd;           // [2m] new diagonal values of the pair
    unsigned short* top;  // [2][m]
    unsigned short* bot;  // [2][m]
    unsigned short* blk;  // [m(m-1)/2][2]
};

// music-chairs step of the round-robin tournament (Brent & Luk): position 0 of the top row is fixed
__device__ __forceinline__ void next_pairing(const unsigned short* top, const unsigned short* bot,
                                             unsigned short* ntop, unsigned short* nbot, int m, int P)
{
    if (m == 1) { ntop[0] = top[0]; nbot[0] = bot[0]; return; }
    ntop[P] = (P == 0) ? top[0] : (P == 1 ? bot[0] : top[P - 1]);
    nbot[P] = (P == m - 1) ? top[m - 1] : bot[P + 1];
}

__device__ __forceinline__ void prev_pairing(const unsigned short* top, const unsigned short* bot,
                                             unsigned short* ptop, unsigned short* pbot, int m, int P)
{
    if (m == 1) { ptop[0] = top[0]; pbot[0] = bot[0]; return; }
    // inverse of next_pairing
    ptop[P] = (P == 0) ? top[0] : (P == m - 1 ? bot[m - 1] : top[P + 1]);
    pbot[P] = (P == 0) ? top[1] : bot[P - 1];
}

__global__ __launch_bounds__(JBS) void k_jacobi_solve(int N, const double* __restrict__ X,
                                                      const double* __restrict__ scl, const double* __restrict__ y,
                                                      const int* __restrict__ rec, double rcond,
                                                      double* __restrict__ C, int* __restrict__ rank,
                                                      double2* __restrict__ rotlog, int64_t log_stride,
                                                      int max_sweeps, int* __restrict__ sweeps_out,
                                                      double* __restrict__ lam_out)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int Np = (N + 1) & ~1;          // padded to even with an inert index
    const int m = Np >> 1;
    const int ntri = (Np * (Np + 1)) >> 1;
    const int nblk = (m * (m - 1)) >> 1;
    JacobiLds L;
    L.A = reinterpret_cast<double*>(lds_raw);
    L.yv = L.A + ntri;
    L.cs = reinterpret_cast<double2*>(L.yv + Np + ((ntri + Np) & 1));     // keep the double2 array 16-B aligned
    L.nd = reinterpret_cast<double*>(L.cs + m);
    L.top = reinterpret_cast<unsigned short*>(L.nd + 2 * m);
    L.bot = L.top + 2 * m;
    L.blk = L.bot + 2 * m;

    const int tid = threadIdx.x;
    const int64_t sys = blockIdx.x;
    const double* Xs = X + sys * (int64_t)N * N;
    const double* ys = y + (int64_t)(rec ? rec[sys] : sys) * N;
    double2* logp = rotlog + sys * log_stride;

    // ---- load ---------------------------------------------------------------------------------
    for (int e = tid; e < ntri; e += JBS) L.A[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < N * N; e += JBS) {
        const int i = e / N, j = e - i * N;
        if (j <= i) L.A[((i * (i + 1)) >> 1) + j] = Xs[e];
    }
    for (int i = tid; i < Np; i += JBS) L.yv[i] = i < N ? ys[i] : 0.0;
    for (int P = tid; P < m; P += JBS) { L.top[P] = (unsigned short)P; L.bot[P] = (unsigned short)(m + P); }
    for (int k = tid; k < nblk; k += JBS) {
        // k -> (P, Q) with P < Q: Q is the largest integer with Q(Q-1)/2 <= k
        int Q = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)k)) * 0.5f);
        while ((Q * (Q - 1)) / 2 > k) --Q;
        while (((Q + 1) * Q) / 2 <= k) ++Q;
        L.blk[2 * k] = (unsigned short)(k - (Q * (Q - 1)) / 2);
        L.blk[2 * k + 1] = (unsigned short)Q;
    }
    __syncthreads();

    const double eps = 2.220446049250313e-16;
    int cur = 0, sweep = 0;
    int64_t nround = 0;
    const int rounds_per_sweep = Np - 1;
    for (; sweep < max_sweeps; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < rounds_per_sweep; ++r, ++nround) {
            const unsigned short* top = L.top + cur * m;
            const unsigned short* bot = L.bot + cur * m;
            // ---- phase 1: rotation of every pair -----------------------------------------------
            if (tid < m) {
                const int p = top[tid], q = bot[tid];
                const double app = L.A[tri(p, p)], aqq = L.A[tri(q, q)], apq = L.A[tri(p, q)];
                double c = 1.0, s = 0.0, npp = app, nqq = aqq;
                const double aa = fabs(apq);
                if (aa > eps * sqrt(fabs(app) * fabs(aqq)) && aa > 1e-300) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    double t;
                    if (fabs(tau) > 1e150) t = 0.5 / tau;
                    else t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    npp = app - t * apq;
                    nqq = aqq + t * apq;
                    rotated = 1;
                    const double yp = L.yv[p], yq = L.yv[q];
                    L.yv[p] = c * yp - s * yq;       // y <- J^T y
                    L.yv[q] = s * yp + c * yq;
                }
                L.cs[tid] = make_double2(c, s);
                L.nd[2 * tid] = npp;
                L.nd[2 * tid + 1] = nqq;
                logp[nround * m + tid] = make_double2(c, s);
            }
            __syncthreads();
            // ---- phase 2: 2x2 blocks B_PQ <- R_P^T B_PQ R_Q ---------------------------------------
            for (int k = tid; k < nblk; k += JBS) {
                const int P = L.blk[2 * k], Q = L.blk[2 * k + 1];
                const double2 rp = L.cs[P], rq = L.cs[Q];
                if (rp.y == 0.0 && rq.y == 0.0) continue;
                const int p1 = top[P], p2 = bot[P], q1 = top[Q], q2 = bot[Q];
                const int i11 = tri(p1, q1), i12 = tri(p1, q2), i21 = tri(p2, q1), i22 = tri(p2, q2);
                const double b11 = L.A[i11], b12 = L.A[i12], b21 = L.A[i21], b22 = L.A[i22];
                // T = R_P^T B
                const double t11 = rp.x * b11 - rp.y * b21, t12 = rp.x * b12 - rp.y * b22;
                const double t21 = rp.y * b11 + rp.x * b21, t22 = rp.y * b12 + rp.x * b22;
                // B' = T R_Q
                L.A[i11] = rq.x * t11 - rq.y * t12;
                L.A[i12] = rq.y * t11 + rq.x * t12;
                L.A[i21] = rq.x * t21 - rq.y * t22;
                L.A[i22] = rq.y * t21 + rq.x * t22;
            }
            if (tid < m) {
                const int p = top[tid], q = bot[tid];
                if (L.cs[tid].y != 0.0) {
                    L.A[tri(p, p)] = L.nd[2 * tid];
                    L.A[tri(q, q)] = L.nd[2 * tid + 1];
                    L.A[tri(p, q)] = 0.0;
                }
                next_pairing(top, bot, L.top + (cur ^ 1) * m, L.bot + (cur ^ 1) * m, m, tid);
            }
            cur ^= 1;
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) { ++sweep; break; }
    }
    // ---- truncated solve in the eigenbasis ---------------------------------------------------------
    // (reuse nd[] as scratch for the reduction of max |lambda|; Np <= 2m so it has room for partials)
    double mx = 0.0;
    for (int i = tid; i < N; i += JBS) mx = fmax(mx, fabs(L.A[tri(i, i)]));
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    __syncthreads();
    if ((tid & 63) == 0) L.nd[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(L.nd[0], L.nd[1]), fmax(L.nd[2], L.nd[3]));
    const double thr = rcond * mx;
    const double sc = scl ? scl[sys] : 1.0;
    int rk = 0;
    for (int i = tid; i < Np; i += JBS) {
        double g = 0.0;
        if (i < N) {
            const double lam = L.A[tri(i, i)];
            const bool keep = fabs(lam) > thr;
            g = keep ? L.yv[i] / (lam * sc) : 0.0;
            rk += keep ? 1 : 0;
            if (lam_out) lam_out[sys * N + i] = lam * sc;
        }
        L.yv[i] = g;
    }
    __syncthreads();
    {
        for (int o = 32; o > 0; o >>= 1) rk += __shfl_xor(rk, o);
        __syncthreads();
        if ((tid & 63) == 0) L.nd[4 + (tid >> 6)] = (double)rk;
        __syncthreads();
        if (tid == 0) {
            if (rank) rank[sys] = (int)(L.nd[4] + L.nd[5] + L.nd[6] + L.nd[7]);
            if (sweeps_out) sweeps_out[sys] = sweep;
        }
    }
    // ---- C = V g : replay the rotations in reverse -----------------------------------------------------
    // `cur` holds the pairing that the next (never executed) round would use; step back once per round.
    constexpr int PF = 8;                               // rounds of (c, s) prefetched per batch
    for (int64_t r1 = nround; r1 > 0; r1 -= PF) {
        const int nb = r1 >= PF ? PF : (int)r1;
        double2 pf[PF];
        if (tid < m) {
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (u < nb) pf[u] = logp[(r1 - 1 - u) * m + tid];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (u < nb) {
                if (tid < m) prev_pairing(L.top + cur * m, L.bot + cur * m, L.top + (cur ^ 1) * m, L.bot + (cur ^ 1) * m, m, tid);
                cur ^= 1;
                __syncthreads();
                if (tid < m && pf[u].y != 0.0) {
                    const int p = L.top[cur * m + tid], q = L.bot[cur * m + tid];
                    const double gp = L.yv[p], gq = L.yv[q];
                    L.yv[p] = pf[u].x * gp + pf[u].y * gq;      // g <- J g
                    L.yv[q] = -pf[u].y * gp + pf[u].x * gq;
                }
                __syncthreads();
            }
        }
    }
    for (int i = tid; i < N; i += JBS) C[sys * N + i] = L.yv[i];
}

}  // namespace

size_t vi_jacobi_lds_bytes(int N)
{
    const int Np = (N + 1) & ~1, m = Np / 2;
    const int nd = Np * (Np + 1) / 2 + Np;
    size_t b = (size_t)(nd + (nd & 1)) * 8 + (size_t)m * 16 + (size_t)2 * m * 8;
    b += (size_t)(4 * m) * 2 + (size_t)(m * (m - 1)) * 2;
    return (b + 15) & ~(size_t)15;
}

bool vi_jacobi_supported(int N) { return N >= 8 && vi_jacobi_lds_bytes(N) <= 159 * 1024 && ((N + 1) / 2) <= JBS; }

// d_X: scaled systems (B x N x N), destroyed: no.  Returns VI_OK or an error.
int vi_jacobi_solve(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_scl, const double* d_y,
                    const int* d_rec, double rcond, double* d_C, int* d_rank, void* d_log, int max_sweeps,
                    int* d_sweeps, double* d_lam)
{
    const size_t shm = vi_jacobi_lds_bytes(N);
    static size_t attr_max = 0;
    if (shm > attr_max) {
        VI_HIP(hipFuncSetAttribute((const void*)k_jacobi_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_max = shm;
    }
    const int Np = (N + 1) & ~1, m = Np / 2;
    const int64_t log_stride = (int64_t)max_sweeps * (Np - 1) * m;
    hipLaunchKernelGGL(k_jacobi_solve, dim3((unsigned)B), dim3(JBS), shm, c->stream, N, d_X, d_scl, d_y, d_rec, rcond,
                       d_C, d_rank, (double2*)d_log, log_stride, max_sweeps, d_sweeps, d_lam);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

size_t vi_jacobi_log_bytes(int N, int max_sweeps)
{
    const int Np = (N + 1) & ~1, m = Np / 2;
    return (size_t)max_sweeps * (Np - 1) * m * sizeof(double2);
}
