// K3p: orthogonal PRE-CONDITIONING of the truncated solve by one column-pivoted Householder QR step.
//
// The systems X(alpha) = A^T W A + alpha R of the regularisation-parameter search (volumetricinterp/interpolate.py:460-462)
// are graded over ~50 decades and rank deficient; a cyclic Jacobi iteration started from X itself spends 12-16 of its
// 20-24 sweeps in a linear phase in which every pair rotates and the grading is sorted out one scale at a time.  With
// X P = Q R (Householder QR, columns pivoted by remaining norm) the similar matrix X1 = Q^T X Q = R P^T Q has its scales
// concentrated row by row, and the same Jacobi kernel converges on it in 5-8 sweeps - and, the rotations being far fewer
// and better scaled, reproduces 50-digit arithmetic to 1e-10 in chi^2 and 1e-7 in the coefficients where the plain
// iteration reaches 1e-5 / 4e-3 (tools/exp_qr_precond.py, tests/golden/exact_default_c2.npz).  The solve becomes
//     X1 = Q^T X Q,  y1 = Q^T y   (k_qr_sim, this file)  ->  X1 c1 = y1  (k_jacobi_solve)  ->  c = Q c1  (k_qr_back_*).
//
// k_qr_sim: one workgroup per system, the matrix in REGISTERS - a 144 x 144 fp64 matrix (166 KB) does not fit the 160 KB
// of LDS, but a column split over four lanes (rows 4i + q in lane q) is 36 doubles per thread.  LDS holds the current
// Householder vector (with the mask of live rows) and the packed store of all vectors.
//   phase 1 (N - 1 steps, two barriers each): pivot = the live column of largest remaining norm (norms are recomputed
//     in the update pass of the previous step, not down-dated: the matrices are graded); its owner forms
//     v = x - alpha e_k, tau = 2 / v^T v; every column (and y, carried along as column N) takes c -= tau (v^T c) v.
//     Row k of the updated matrix - row k of R P^T - leaves the registers for global memory when step k is done.
//   phase 2 (no barriers): X1 = Q^T X Q = (R P^T) Q, and X1 is symmetric, so its column j is Q^T applied to row j of
//     R P^T: every quad loads its row and takes all reflectors; y1 = Q^T y the same way.
// The reflectors leave in packed form (v_k: rows k .. N-1, then the N taus) for the back-transformation kernels.
#include "vi_common.h"

#ifdef VI_STAMPS
// diagnostic build only (-DVI_STAMPS): cycle sums of thread 0 of workgroup 0 per part of k_qr_sim; vi_debug_qr_stamps
__device__ unsigned long long g_qr_stamps[8];
#define QR_STAMP(k)                                                        \
    do {                                                                   \
        const unsigned long long t_ = __builtin_readcyclecounter();        \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_qr_stamps[k] += t_ - stamp_t; \
        stamp_t = __builtin_readcyclecounter();                            \
    } while (0)
#else
#define QR_STAMP(k)
#endif

namespace {

// sum over the four lanes of a column group (quad): DPP quad_perm [1,0,3,2] and [2,3,0,1]
__device__ __forceinline__ double quad_xor(double v, const int ctrl)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl == 1) {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v)
{
    v += quad_xor(v, 1);
    v += quad_xor(v, 2);
    return v;
}

// Packed store of the reflectors.  v_k lives in rows k .. N-1; it is stored from row 16 (k >> 4) on (zeros above k) up to
// row NP - 1 (NP = N rounded up to 4, zeros below N), so that a quad can take it in whole groups of four local rows
// (16 matrix rows) without a predicate per element.
__host__ __device__ __forceinline__ int hh_off(int k, int NP)
{
    const int B = k >> 4;
    return 16 * NP * B - 128 * B * (B - 1) + (k & 15) * (NP - 16 * B);
}
__host__ __device__ __forceinline__ size_t hh_doubles(int N)
{
    const int NP = (N + 3) & ~3;
    return (size_t)hh_off(N, NP) + (size_t)N;              // N vectors (the last one unused), then the N taus
}

// one reflector on one column held by a quad (rows 4 i + q in lane q): c -= tau (v^T c) v with vk[r] valid (zero-padded)
// for 16 (k >> 4) <= r < 4 RPT' (RPT' = NP / 4 local rows).  The vector is read from LDS twice (dot product, update)
// instead of being kept: 36 more doubles do not fit the 168 registers of three waves per SIMD.  Groups of four local rows
// wholly above the reflector, or wholly below the matrix, are skipped (uniform branches).
template <int RPT>
__device__ __forceinline__ void reflect(double (&a)[RPT], int nloc, int k, int q, const double* __restrict__ vk, double tau)
{
    const int g0 = k >> 4;                 // first group of four local rows the reflector touches
    const double* vq = vk + q;
    double dot = 0.0;
#pragma unroll
    for (int g = 0; g < (RPT + 3) / 4; ++g) {
        if (g >= g0 && 4 * g < nloc) {
#pragma unroll
            for (int i = 4 * g; i < 4 * g + 4 && i < RPT; ++i)
                if (i < nloc) dot = fma(vq[4 * i], a[i], dot);
            asm volatile("" ::: "memory");           // one group of loads in flight at a time (registers)
        }
    }
    const double w = tau * quad_sum(dot);
    asm volatile("" ::: "memory");               // keeps the compiler from holding the first pass's loads for the second
#pragma unroll
    for (int g = 0; g < (RPT + 3) / 4; ++g) {
        if (g >= g0 && 4 * g < nloc) {
#pragma unroll
            for (int i = 4 * g; i < 4 * g + 4 && i < RPT; ++i)
                if (i < nloc) a[i] = fma(-w, vq[4 * i], a[i]);
            asm volatile("" ::: "memory");
        }
    }
}

// c <- (I - tau_k v_k v_k^T) c for k = 0 .. K-1 in that order (c <- Q^T c); reflectors from the packed LDS store.  No
// synchronisation: a column only needs itself.
template <int RPT>
__device__ __forceinline__ void apply_all_fwd(double (&a)[RPT], int NP, int K, int q, const double* __restrict__ Vst,
                                              const double* __restrict__ taus)
{
    for (int k = 0; k < K; ++k) reflect<RPT>(a, NP >> 2, k, q, Vst + hh_off(k, NP) - 16 * (k >> 4), taus[k]);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the global stores in flight (rows of
// R P^T leave for global memory in every step of the factorisation; nobody reads them before the phase is over), a
// round trip of ~1 us, twice per step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// max over the wave of a value that is uniform within each quad (result valid in every lane): row_shr:4, row_shr:8 inside
// the rows of 16 lanes, row_bcast:15 / row_bcast:31 across them, then a read of lane 63 - ~12 DPP moves instead of the
// 12 dependent ds_bpermute round trips of a shuffle butterfly (the reduction sits on the critical path of every step).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_max(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return fmax(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double wave_max_quads(double v)
{
    v = dpp_max<0x114, 0xf>(v);
    v = dpp_max<0x118, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    v = dpp_max<0x143, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// 1 / sqrt(x), 1 / x for x well inside the normal range: hardware estimate + one third-order / two Newton steps
// (see fast_rsqrt in vi_jacobi.hip)
__device__ __forceinline__ double qr_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}
__device__ __forceinline__ double qr_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}

template <int RPT>
__global__ __launch_bounds__(RPT <= 16 ? 320 : 640) void k_qr_sim(
    int N, const double* __restrict__ X, const double* __restrict__ y, const int* __restrict__ rec, double* X1,
    double* __restrict__ y1, double* __restrict__ hh, int64_t hh_stride, double* __restrict__ Rscr)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int NP = (N + 3) & ~3, nloc = NP >> 2;
    constexpr int NR = 4 * RPT;                                          // rows a quad can hold
    constexpr int NG = (RPT + 3) / 4;                                    // groups of four local rows
    double* vcur = reinterpret_cast<double*>(lds_raw);                   // [2][NR] current reflector (double-buffered)
    double* Vst = vcur + 2 * NR;                                         // packed reflectors
    const int nvst = hh_off(N, NP);
    double* taus = Vst + nvst;                                           // [N]
    double* redv = taus + N;                                             // [16]
    int* redi = reinterpret_cast<int*>(redv + 16);                       // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int q = tid & 3, cj = tid >> 2;
    const int64_t sys = blockIdx.x;
    const double* Xs = X + sys * (int64_t)N * N;
    const double* ys = y + (int64_t)(rec ? rec[sys] : sys) * N;
    double* Rs = Rscr + sys * (int64_t)N * N;
    const bool is_mat = cj < N, is_y = cj == N;

#ifdef VI_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter();
#endif
    double a[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = 4 * i + q;
        // the lower triangle mirrored: exactly symmetric, and the elements the plain Jacobi kernel reads
        a[i] = (r < N) ? (is_mat ? Xs[(int64_t)(r > cj ? r : cj) * N + (r > cj ? cj : r)] : (is_y ? ys[r] : 0.0)) : 0.0;
    }
    for (int r = tid; r < 2 * NR; r += blockDim.x) vcur[r] = 0.0;
    for (int k = tid; k < N; k += blockDim.x) taus[k] = 0.0;
    for (int e = tid; e < nvst; e += blockDim.x) Vst[e] = 0.0;
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) nrm = fma(a[i], a[i], nrm);
    nrm = quad_sum(nrm);
    bool done = !is_mat;
    int K = 0;
    __syncthreads();
    QR_STAMP(0);
    // ---- phase 1: pivoted Householder QR.  Invariant at the top of step k: rows < k of every column are zero in the
    //      registers (row k' went to Rs[k'][.] = (R P^T)[k'][.] when step k' finished), nrm = the column's remaining norm^2.
    for (int k = 0; k < N - 1; ++k) {
        const double cv = done ? -1.0 : nrm;
        const double wm = wave_max_quads(cv);
        const unsigned long long hit = __ballot(!done && cv == wm);
        if (lane == 0) {
            redv[wave] = hit ? wm : -1.0;
            redi[wave] = hit ? (wave * 16 + ((__ffsll((long long)hit) - 1) >> 2)) : 0;
        }
        lds_barrier();
        QR_STAMP(1);
        double pmax = redv[0];
        int p = redi[0];
        for (int w = 1; w < nw; ++w) {
            const double ov = redv[w];
            const int oi = redi[w];
            const bool take = ov > pmax;                 // ties: the first wave, i.e. the smallest column index
            pmax = take ? ov : pmax;
            p = take ? oi : p;
        }
        if (!(pmax > 0.0)) break;                    // what is left is exactly zero: the remaining reflectors are identities
        const int ik = k >> 2, qk = k & 3;           // row k = local row ik of lane qk
        double* vb = vcur + (k & 1) * NR + q;
        if (cj == p) {
            double xl = 0.0;
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                if (i == ik) xl = a[i];
            const double xk = __shfl(xl, (lane & ~3) | qk);
            const double ri = qr_rsqrt(pmax);
            const double nx = pmax * ri;                                // ||x||
            const double alpha = -copysign(nx, xk);                     // H x = alpha e_k
            const double vk = xk - alpha;                               // no cancellation: |vk| = |xk| + ||x||
            const double tau = ri * qr_rcp(nx + fabs(xk));              // 2 / v^T v = 1 / (||x|| (||x|| + |xk|))
            double* vs = Vst + hh_off(k, NP) - 16 * (k >> 4) + q;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                if (i >= 4 * (k >> 4) && i < nloc) {
                    const double v = (i == ik && q == qk) ? vk : a[i];  // rows < k are zero
                    vb[4 * i] = v;
                    vs[4 * i] = v;
                }
            }
            if (q == 0) taus[k] = tau;
            done = true;
        }
        lds_barrier();
        QR_STAMP(2);
        K = k + 1;
        const double tau = taus[k];
        const int g0 = k >> 4;
        double dot = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= g0 && 4 * g < nloc) {
#pragma unroll
                for (int i = 4 * g; i < 4 * g + 4 && i < RPT; ++i) dot = fma(vb[4 * i], a[i], dot);
                asm volatile("" ::: "memory");
            }
        }
        const double w = tau * quad_sum(dot);
        nrm = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= g0 && 4 * g < nloc) {
#pragma unroll
                for (int i = 4 * g; i < 4 * g + 4 && i < RPT; ++i) {
                    a[i] = fma(-w, vb[4 * i], a[i]);
                    if (i == ik) {                    // row k is finished: it leaves for (R P^T)[k][cj]
                        if (q == qk) {
                            if (is_mat) Rs[(int64_t)k * N + cj] = a[i];
                            a[i] = 0.0;
                        }
                    }
                    nrm = fma(a[i], a[i], nrm);
                }
                asm volatile("" ::: "memory");
            }
        }
        nrm = quad_sum(nrm);
        QR_STAMP(3);
    }
    __syncthreads();
    // rows K .. N-1 of R P^T (what is left in the registers when the loop ends: the last row, or zeros after a break)
    if (is_mat) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = 4 * i + q;
            if (r >= K && r < N) Rs[(int64_t)r * N + cj] = a[i];
        }
    }
    {
        double* hs = hh + sys * hh_stride;
        for (int e = tid; e < nvst + N; e += blockDim.x) hs[e] = Vst[e];         // the taus follow the vectors in LDS as well
    }
    __syncthreads();
    QR_STAMP(4);
    // ---- phase 2: X1 = Q^T X Q = (R P^T) Q = (Q^T (R P^T)^T)^T, and X1 is symmetric: column cj of (R P^T)^T is row cj
    //      of Rs, contiguous; it takes all reflectors.  y1 = Q^T y the same way from y itself.
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = 4 * i + q;
        a[i] = (r < N) ? (is_mat ? Rs[(int64_t)cj * N + r] : (is_y ? ys[r] : 0.0)) : 0.0;
    }
    QR_STAMP(5);
    apply_all_fwd<RPT>(a, NP, K, q, Vst, taus);
    QR_STAMP(6);
    if (is_mat) {
        double* Xo = X1 + sys * (int64_t)N * N;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = 4 * i + q;
            if (r < N) Xo[(int64_t)cj * N + r] = a[i];
        }
    } else if (is_y) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = 4 * i + q;
            if (r < N) y1[sys * N + r] = a[i];
        }
    }
}

// c <- Q c = H_0 H_1 ... H_{K-1} c for one vector per system: one wave, the vector in its registers (element lane + 64 j),
// the reflectors streamed from global memory one step ahead of their use.
template <int EPL>
__global__ __launch_bounds__(64) void k_qr_back_vec(int N, const double* __restrict__ hh, int64_t hh_stride, double* __restrict__ C)
{
    const int lane = threadIdx.x;
    const int NP = (N + 3) & ~3;
    const int64_t sys = blockIdx.x;
    const double* hs = hh + sys * hh_stride;
    const double* taus = hs + hh_off(N, NP);
    double* c = C + sys * N;
    double x[EPL], v[EPL], vn[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int r = lane + 64 * j;
        x[j] = r < N ? c[r] : 0.0;
    }
    int k = N - 2;
    if (k < 0) return;
    {
        const double* vk = hs + hh_off(k, NP) - 16 * (k >> 4);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int r = lane + 64 * j;
            vn[j] = (r >= k && r < N) ? vk[r] : 0.0;
        }
    }
    double taun = taus[k];
    for (; k >= 0; --k) {
        const double tau = taun;
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[j] = vn[j];
        if (k > 0) {
            const double* vk = hs + hh_off(k - 1, NP) - 16 * ((k - 1) >> 4);
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int r = lane + 64 * j;
                vn[j] = (r >= k - 1 && r < N) ? vk[r] : 0.0;
            }
            taun = taus[k - 1];
        }
        if (tau == 0.0) continue;                    // no reflector was formed at this step (the rest of the matrix was zero)
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < EPL; ++j) dot = fma(v[j], x[j], dot);
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        const double w = tau * dot;
#pragma unroll
        for (int j = 0; j < EPL; ++j) x[j] = fma(-w, v[j], x[j]);
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int r = lane + 64 * j;
        if (r < N) c[r] = x[j];
    }
}

// V[:, j] <- Q V[:, j] for the N vectors of a system (eigenvectors in the LAPACK layout: vector j contiguous): one
// workgroup per system, reflectors staged in LDS, one vector per quad.
template <int RPT>
__global__ __launch_bounds__(RPT <= 16 ? 320 : 640) void k_qr_back_mat(int N, const double* __restrict__ hh, int64_t hh_stride,
                                                                       double* __restrict__ V)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* Vst = reinterpret_cast<double*>(lds_raw);
    const int NP = (N + 3) & ~3;
    const int nvst = hh_off(N, NP);
    double* taus = Vst + nvst;
    const int tid = threadIdx.x, q = tid & 3, cj = tid >> 2;
    const int64_t sys = blockIdx.x;
    const double* hs = hh + sys * hh_stride;
    for (int e = tid; e < nvst + N; e += blockDim.x) Vst[e] = hs[e];
    __syncthreads();
    if (cj >= N) return;
    double* vj = V + sys * (int64_t)N * N + (int64_t)cj * N;
    double a[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = 4 * i + q;
        a[i] = r < N ? vj[r] : 0.0;
    }
    for (int k = N - 2; k >= 0; --k) {
        const double tau = taus[k];
        if (tau != 0.0) reflect<RPT>(a, NP >> 2, k, q, Vst + hh_off(k, NP) - 16 * (k >> 4), tau);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = 4 * i + q;
        if (r < N) vj[r] = a[i];
    }
}

}  // namespace
#ifdef VI_STAMPS
extern "C" int vi_debug_qr_stamps(double* out, int reset)
{
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qr_stamps), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qr_stamps), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
namespace {

int qr_rpt(int N)
{
    const int need = (N + 3) / 4;
    const int opts[] = {8, 16, 24, 36};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}

size_t qr_lds_bytes(int N)
{
    return (size_t)2 * 4 * qr_rpt(N) * 8 + (hh_doubles(N) + 16) * 8 + 16 * 4;
}

}  // namespace

bool vi_qr_supported(int N) { return N >= 8 && qr_rpt(N) > 0 && qr_lds_bytes(N) <= 159 * 1024; }
size_t vi_qr_hh_bytes(int N) { return hh_doubles(N) * sizeof(double); }

// d_X: B scaled symmetric systems (read twice, not written unless d_X1 == d_X); d_y (+ d_rec) as in vi_jacobi_solve.
// Outputs: d_X1 (B x N x N; may alias d_X), d_y1 (B x N), d_hh (B x vi_qr_hh_bytes(N)); d_scr: B x N x N scratch.
// hh_stride: distance in doubles between the reflector sets of consecutive systems (0: back to back).
int vi_qr_precond(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_y, const int* d_rec, double* d_X1,
                  double* d_y1, double* d_hh, double* d_scr, int64_t hh_stride)
{
    if (hh_stride <= 0) hh_stride = (int64_t)hh_doubles(N);
    const int rpt = qr_rpt(N);
    const int threads = ((4 * (N + 1) + 63) / 64) * 64;
    const size_t shm = qr_lds_bytes(N);
#define VI_QR(R)                                                                                                            \
    do {                                                                                                                    \
        VI_HIP(hipFuncSetAttribute((const void*)k_qr_sim<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));        \
        hipLaunchKernelGGL(k_qr_sim<R>, dim3((unsigned)B), dim3(threads), shm, c->stream, N, d_X, d_y, d_rec, d_X1, d_y1,   \
                           d_hh, hh_stride, d_scr);                                                                                   \
    } while (0)
    switch (rpt) {
    case 8: VI_QR(8); break;
    case 16: VI_QR(16); break;
    case 24: VI_QR(24); break;
    case 36: VI_QR(36); break;
    default: vi_set_error("vi_qr_precond: N=%d unsupported", N); return VI_ERR_UNSUPPORTED;
    }
#undef VI_QR
    VI_HIP(hipGetLastError());
    return VI_OK;
}

int vi_qr_back_vec(vi_ctx* c, int64_t B, int N, const double* d_hh, double* d_C, int64_t hh_stride)
{
    if (hh_stride <= 0) hh_stride = (int64_t)hh_doubles(N);
    if (N <= 64) hipLaunchKernelGGL(k_qr_back_vec<1>, dim3((unsigned)B), dim3(64), 0, c->stream, N, d_hh, hh_stride, d_C);
    else if (N <= 128) hipLaunchKernelGGL(k_qr_back_vec<2>, dim3((unsigned)B), dim3(64), 0, c->stream, N, d_hh, hh_stride, d_C);
    else hipLaunchKernelGGL(k_qr_back_vec<3>, dim3((unsigned)B), dim3(64), 0, c->stream, N, d_hh, hh_stride, d_C);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

int vi_qr_back_mat(vi_ctx* c, int64_t B, int N, const double* d_hh, double* d_V, int64_t hh_stride)
{
    if (hh_stride <= 0) hh_stride = (int64_t)hh_doubles(N);
    const int rpt = qr_rpt(N);
    const int threads = ((4 * N + 63) / 64) * 64;
    const size_t shm = hh_doubles(N) * sizeof(double);
#define VI_QB(R)                                                                                                            \
    do {                                                                                                                    \
        VI_HIP(hipFuncSetAttribute((const void*)k_qr_back_mat<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));   \
        hipLaunchKernelGGL(k_qr_back_mat<R>, dim3((unsigned)B), dim3(threads), shm, c->stream, N, d_hh, hh_stride, d_V);    \
    } while (0)
    switch (rpt) {
    case 8: VI_QB(8); break;
    case 16: VI_QB(16); break;
    case 24: VI_QB(24); break;
    case 36: VI_QB(36); break;
    default: vi_set_error("vi_qr_back_mat: N=%d unsupported", N); return VI_ERR_UNSUPPORTED;
    }
#undef VI_QB
    VI_HIP(hipGetLastError());
    return VI_OK;
}

// Stage-test entry (tests/test_gpu_search_stages.py): X1 = Q^T X Q and the explicit Q of B symmetric systems.
// d_X is only read; d_X1, d_Q: B x N x N (Q in the LAPACK layout of eigenvector matrices: Q[:, j] contiguous).
extern "C" int vi_qr_similarity_f64(vi_ctx* c, int64_t B, int32_t N, const double* d_X, const double* d_y, double* d_X1,
                                    double* d_y1, double* d_Q)
{
    VI_REQUIRE(c && d_X && d_y && d_X1 && d_y1 && d_Q, "null argument");
    VI_REQUIRE(B >= 0 && N > 0, "bad size");
    if (B == 0) return VI_OK;
    if (!vi_qr_supported(N)) {
        vi_set_error("vi_qr_similarity_f64: N=%d outside the range of the register-resident QR", N);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)B * (vi_qr_hh_bytes(N) + (size_t)N * N * sizeof(double)) + 256, &ws);
    if (rc != VI_OK) return rc;
    double* hhp = (double*)ws;
    double* scr = hhp + (size_t)B * hh_doubles(N);
    if ((rc = vi_qr_precond(c, B, N, d_X, d_y, nullptr, d_X1, d_y1, hhp, scr, 0)) != VI_OK) return rc;
    // explicit Q: the back-transformation of the identity
    std::vector<double> eye((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i) eye[(size_t)i * N + i] = 1.0;
    for (int64_t b = 0; b < B; ++b)
        VI_HIP(hipMemcpyAsync(d_Q + b * (int64_t)N * N, eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    VI_HIP(hipStreamSynchronize(c->stream));
    return vi_qr_back_mat(c, B, N, hhp, d_Q, 0);
}
