// K3: batched truncated minimum-norm solve of small symmetric (indefinite, possibly rank-deficient)
// systems by a parallel cyclic Jacobi eigensolver that lives entirely in one CU's LDS.
//
// Replaces, for the hundreds of trial systems per record of the regularisation-parameter search,
// scipy.linalg.lstsq (LAPACK gelsd, rcond = eps) at volumetricinterp/interpolate.py:462:
// for symmetric X = V diag(lam) V^T the SVD truncation |sigma| <= rcond*sigma_max is the spectral
// truncation |lam| <= rcond*max|lam|, and the minimum-norm solution is V diag(1/lam | kept) V^T y.
//
// One workgroup per system.  The symmetric half of X (N(N+1)/2 doubles: 83.5 KB at N = 144, inside the
// 160 KB LDS of a gfx950 CU) stays in LDS for the whole decomposition.
//  * Ordering: "matches" of four indices.  The indices sit in slots; unit u = slots (2u, 2u+1); match a = units
//    (2a, 2a+1) = slots 4a .. 4a+3 = (U0, U1, V0, V1).  One ROUND rotates, in every match, the four cross pairs of
//    its two units - inner round 1: (U0,V0), (U1,V1); inner round 2: (U0,V1), (U1,V0) - and then moves the UNITS
//    one step along the Brent-Luk "music chairs" ring (unit 0 fixed), so that after m - 1 rounds (m units) every
//    pair of units has met once.  The intra-unit pairs (U0,U1), (V0,V1) get one extra round per sweep (round 0).
//    Every pair of indices is rotated exactly once per sweep, as in the classical one-pair-per-round round robin,
//    but the matrix passes through the registers only m times per sweep instead of 2m - 1: the kernel is bound by
//    LDS traffic (a round reads and writes every element once), so this is what sets its speed.  Sweep counts are
//    the same or lower (tools/proto_jacobi4.py: 10 vs 12 on the screened N = 32 fixture, 16 vs 21 at N = 144).
//  * DATA movement instead of index movement: the matrix is stored by slot as 4x4 "super-blocks" S_ab (rows of
//    match a, columns of match b, a < b), one per thread, and after every round the unit permutation is applied
//    to the slots.  Because a round reads every super-block into registers before the barrier and writes it after,
//    the permutation is free: the elements are simply stored at the permuted positions.  All LDS addresses of a
//    thread are round-invariant and live in registers.
//  * The two-sided update A <- J^T A J with J the direct sum of the rotations of a round factorises into
//    independent updates S_ab <- R_a^T S_ab R_b (R_a = the four rotations of match a): no atomics, two barriers per
//    round.  The 4x4 diagonal block of a match (10 unique elements) is handled by one thread of wave 0, which
//    computes the rotations of both inner rounds in registers.
//  * The eigenvector matrix is never formed: the right-hand side is rotated along (y <- J^T y), the rotations
//    (c, s) are logged to HBM (16 B each, ~1.3 MB per system at N = 144 and 8 sweeps) and replayed in reverse on
//    the truncated, scaled vector to give C = V g.
//  * Convergence: a pair is rotated while |a_pq| > eps*sqrt(|a_pp a_qq|) (relative criterion: small kept
//    eigenvalues stay accurate), except below an absolute floor and inside the subspace that the truncation is
//    going to drop anyway; after every sweep all pairs are tested in place and the iteration ends when none would
//    rotate.
#include "vi_common.h"

#include <cstdlib>
#include <cstring>

#ifdef VI_STAMPS
// diagnostic build only (make STAMPS=1): per-phase cycle sums of workgroup 0, for wave 0 (slots 0-7) and the last wave
// (slots 8-15); read and reset by vi_debug_jacobi_stamps.  No stamp executes in the product build.
__device__ unsigned long long g_jacobi_stamps[16];
#define VI_STAMP(k)                                                                          \
    do {                                                                                     \
        const unsigned long long t_ = __builtin_readcyclecounter();                          \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64))         \
            g_jacobi_stamps[(threadIdx.x == 0 ? 0 : 8) + (k)] += t_ - stamp_t;               \
        stamp_t = __builtin_readcyclecounter();                                              \
    } while (0)
#else
#define VI_STAMP(k)
#endif

#ifdef VI_ROTHIST
// diagnostic build only (-DVI_ROTHIST): rotations and solves by sweep index over all k_jacobi_solve launches; vi_debug_rot_hist
__device__ unsigned long long g_rot_hist[32][2];
#define VI_ROT_COUNT(sweep, r0, r1, r2, r3)                                                                              \
    atomicAdd(&g_rot_hist[(sweep) < 31 ? (sweep) : 31][0],                                                               \
              (unsigned long long)(((r0).y != 0.0) + ((r1).y != 0.0) + ((r2).y != 0.0) + ((r3).y != 0.0)))
#define VI_ROT_SWEEP(sweep)                                                                                               \
    do { if (threadIdx.x == 0) atomicAdd(&g_rot_hist[(sweep) < 31 ? (sweep) : 31][1], 1ull); } while (0)
#endif

#include "vi_jacobi_device.h"
#include "vi_jacobi_v2_device.h"
#include <cstdlib>
#include <cstring>

namespace {

template <int IT>
__global__ __launch_bounds__(IT == 1 ? 768 : 512) void k_jacobi_solve(
    int N, const double* __restrict__ X, const double* __restrict__ scl, const double* __restrict__ y,
    const int* __restrict__ rec, double rcond, double abs_floor, double* __restrict__ C, int* __restrict__ rank,
    double2* __restrict__ rotlog, int64_t log_stride, int max_sweeps, int* __restrict__ sweeps_out,
    double* __restrict__ lam_out, int lam_raw, int* __restrict__ nround_out, unsigned long long* __restrict__ round_acc,
    double conv_tol)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int64_t sys = blockIdx.x;
    jacobi_system<IT>(lds_raw, N, X + sys * (int64_t)N * N, scl ? scl[sys] : 1.0, y + (int64_t)(rec ? rec[sys] : sys) * N, rcond,
                      abs_floor, C + sys * N, rank ? rank + sys : nullptr, rotlog + sys * log_stride, max_sweeps,
                      sweeps_out ? sweeps_out + sys : nullptr, lam_out ? lam_out + sys * N : nullptr, lam_raw,
                      nround_out ? nround_out + sys : nullptr, round_acc, conv_tol);
}

// The role-separated kernel (vi_jacobi_v2_device.h): blockDim = 64 (set-up wave) + one thread per super-block.
__global__ __launch_bounds__(768) void k_jacobi_solve_v2(
    int N, const double* __restrict__ X, const double* __restrict__ scl, const double* __restrict__ y,
    const int* __restrict__ rec, double rcond, double abs_floor, double* __restrict__ C, int* __restrict__ rank,
    double2* __restrict__ rotlog, int64_t log_stride, int max_sweeps, int* __restrict__ sweeps_out,
    double* __restrict__ lam_out, int lam_raw, int* __restrict__ nround_out, unsigned long long* __restrict__ round_acc,
    double conv_tol)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int64_t sys = blockIdx.x;
    jacobi_system_v2(lds_raw, N, X + sys * (int64_t)N * N, scl ? scl[sys] : 1.0, y + (int64_t)(rec ? rec[sys] : sys) * N, rcond,
                     abs_floor, C + sys * N, rank ? rank + sys : nullptr, rotlog + sys * log_stride, max_sweeps,
                     sweeps_out ? sweeps_out + sys : nullptr, lam_out ? lam_out + sys * N : nullptr, lam_raw,
                     nround_out ? nround_out + sys : nullptr, round_acc, conv_tol);
}

// Eigenvectors from the rotation log: V = J_1 J_2 ... J_K, so column k of V is the reverse replay applied to the unit
// vector of (final) slot k.  One WAVE per strip of CPW columns, the columns in the wave's registers (WaveReplay) - no
// LDS, no barrier.  CPW = 1 for a single system (N independent waves spread over as many CUs: latency); CPW = 8 for
// batches, so that a system's rotation log is read N / 8 times instead of N times.  Output is the LAPACK / rocSOLVER
// layout Vout[k*N + r] (eigenvector k contiguous).  N <= 256.
template <int CPW>
__global__ __launch_bounds__(64) void k_jacobi_vectors_wave(int N, const double2* __restrict__ rotlog, int64_t log_stride,
                                                            const int* __restrict__ nround_in, double* __restrict__ Vout)
{
    const int64_t sys = blockIdx.y;
    jacobi_vector_strip<CPW>(N, rotlog + sys * log_stride, nround_in[sys], blockIdx.x * CPW, threadIdx.x,
                             Vout + sys * (int64_t)N * N);
}

}  // namespace

#ifdef VI_STAMPS
extern "C" int vi_debug_jacobi_stamps(double* out, int reset)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_jacobi_stamps), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out[i] = (double)h[i];
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_jacobi_stamps), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef VI_ROTHIST
extern "C" int vi_debug_rot_hist(double* out, int reset)
{
    unsigned long long h[32][2];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rot_hist), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 32; ++i) { out[2 * i] = (double)h[i][0]; out[2 * i + 1] = (double)h[i][1]; }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rot_hist), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace {

int jacobi_nsb(int N)
{
    const int M = ((N + 3) & ~3) / 4;
    return M * (M - 1) / 2;
}
// launch geometry: one super-block per thread up to 768 threads (three waves per SIMD, <= 168 VGPRs), then two or three
// per thread in 512-thread workgroups
void jacobi_geometry(int N, int& threads, int& it)
{
    const int nsb = jacobi_nsb(N);
    if (nsb <= 768) {
        it = 1;
        threads = ((nsb + 63) / 64) * 64;
        if (threads < 64) threads = 64;
    } else {
        threads = 512;
        it = (nsb + 511) / 512;
    }
}

}  // namespace

// The role-separated kernel serves the orders with one super-block per thread that leave room for the set-up wave in a
// 768-thread workgroup, from 24 matches on (N = 93 ... 148: the benchmarked order 144 with 704 threads; below, the fixed cost of
// its counters outweighs what it overlaps);
// VINTERP_K3=v1 keeps the two-barrier kernel everywhere.  Same bits either way (tests/test_gpu_search_stages.py).
bool vi_jacobi_use_v2(int N)
{
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("VINTERP_K3");
        forced = (e && !strcmp(e, "v1")) ? 1 : 0;
    }
    if (forced) return false;
    int threads, it;
    jacobi_geometry(N, threads, it);
    const int M = ((N + 3) & ~3) / 4;
    static int minm = -1;
    if (minm < 0) { const char* e2 = getenv("VINTERP_K3_MINM"); minm = e2 ? atoi(e2) : 24; if (minm < 3) minm = 3; }
    return it == 1 && M >= minm && M <= 64 && threads + 64 <= 768;     // (from N = 93: measured -2 % at N = 100, -9 % at 144, +2 % at 32)
}

// LDS of a workgroup: the image, y x 2, the rotations, scratch - and for the role-separated kernel the rotations twice, the
// mailbox, the counters and the destination table of the diagonal blocks
size_t vi_jacobi_lds_bytes(int N)
{
    const int Np = (N + 3) & ~3, M = Np / 4;
    const int nt = Np * (Np + 1) / 2;
    size_t b = (size_t)(nt + 2 * Np) * 8 + (size_t)4 * M * 16 + 16 * 8;
    if (vi_jacobi_use_v2(N)) b += (size_t)4 * M * 16 + (size_t)4 * M * 8 + 16 + (size_t)14 * M * 4;
    return (b + 15) & ~(size_t)15;
}

bool vi_jacobi_supported(int N)
{
    int threads, it;
    jacobi_geometry(N, threads, it);
    return N >= 8 && vi_jacobi_lds_bytes(N) <= 159 * 1024 && it <= 3 && ((N + 3) / 4) <= 64;
}

// rounds per sweep: m = Np / 2 (one intra-unit round + m - 1 rounds of cross pairs); 4 rotations per match and round
static int64_t jacobi_log_stride(int N, int max_sweeps)
{
    const int Np = (N + 3) & ~3, m = Np / 2, M = Np / 4;
    return (int64_t)max_sweeps * m * 4 * M;
}
size_t vi_jacobi_log_bytes(int N, int max_sweeps) { return (size_t)jacobi_log_stride(N, max_sweeps) * sizeof(double2); }

template <int IT>
static int launch_jacobi(vi_ctx* c, int threads, int64_t B, int N, const double* d_X, const double* d_scl, const double* d_y,
                         const int* d_rec, double rcond, double* d_C, int* d_rank, void* d_log, int max_sweeps,
                         int* d_sweeps, double* d_lam, int lam_raw, int* d_nround, double abs_floor, int64_t log_stride_in,
                         double conv_tol)
{
    const size_t shm = vi_jacobi_lds_bytes(N);
    // per launch, not cached: a context per device may exist in one process and the attribute is per device
    VI_HIP(hipFuncSetAttribute((const void*)k_jacobi_solve<IT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const int64_t log_stride = log_stride_in > 0 ? log_stride_in : jacobi_log_stride(N, max_sweeps);
    // abs_floor: the systems are scaled to max|X| in [1, 2).  Cold solves pass 1e-22: off-diagonal elements
    // below it cannot move any kept eigenvalue (|lambda| > eps * max|lambda|) by more than 1e-6 of itself.
    // Warm solves pass 1e-16: the rotated system D1 + alpha D2 carries formation errors of N*eps anyway.
    const int slot = (int)(c->solve_launches % vi_ctx::NSOLVE_EV);
    if (c->solve_timing) VI_HIP(hipEventRecord(c->evs[slot][0], c->stream));
    hipLaunchKernelGGL(k_jacobi_solve<IT>, dim3((unsigned)B), dim3(threads), shm, c->stream, N, d_X, d_scl, d_y, d_rec,
                       rcond, abs_floor, d_C, d_rank, (double2*)d_log, log_stride, max_sweeps, d_sweeps, d_lam, lam_raw, d_nround,
                       c->solve_timing ? c->d_rounds : nullptr, conv_tol);
    VI_HIP(hipGetLastError());
    if (c->solve_timing) {
        VI_HIP(hipEventRecord(c->evs[slot][1], c->stream));
        c->solve_launches += 1;
        c->solve_systems += B;
    }
    return VI_OK;
}

// d_X: systems scaled by k_scale_system (B x N x N, only read).  log_stride (in 16-byte rotation entries, 0 = the logs
// lie back to back): distance between the rotation logs of consecutive systems in d_log.  conv_tol > 0: a looser end of the
// iteration (jacobi_system).
int vi_jacobi_solve(vi_ctx* c, int64_t B, int N, const double* d_X, const double* d_scl, const double* d_y,
                    const int* d_rec, double rcond, double* d_C, int* d_rank, void* d_log, int max_sweeps,
                    int* d_sweeps, double* d_lam, int lam_raw, int* d_nround, double abs_floor, int64_t log_stride, double conv_tol)
{
    int threads, it;
    jacobi_geometry(N, threads, it);
    if (vi_jacobi_use_v2(N)) {
        const size_t shm = vi_jacobi_lds_bytes(N);
        VI_HIP(hipFuncSetAttribute((const void*)k_jacobi_solve_v2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        const int64_t ls = log_stride > 0 ? log_stride : jacobi_log_stride(N, max_sweeps);
        const int slot = (int)(c->solve_launches % vi_ctx::NSOLVE_EV);
        if (c->solve_timing) VI_HIP(hipEventRecord(c->evs[slot][0], c->stream));
        hipLaunchKernelGGL(k_jacobi_solve_v2, dim3((unsigned)B), dim3(threads + 64), shm, c->stream, N, d_X, d_scl, d_y, d_rec, rcond,
                           abs_floor, d_C, d_rank, (double2*)d_log, ls, max_sweeps, d_sweeps, d_lam, lam_raw, d_nround,
                           c->solve_timing ? c->d_rounds : nullptr, conv_tol);
        VI_HIP(hipGetLastError());
        if (c->solve_timing) {
            VI_HIP(hipEventRecord(c->evs[slot][1], c->stream));
            c->solve_launches += 1;
            c->solve_systems += B;
        }
        return VI_OK;
    }
#define VI_J(IT) return launch_jacobi<IT>(c, threads, B, N, d_X, d_scl, d_y, d_rec, rcond, d_C, d_rank, d_log, max_sweeps, d_sweeps, d_lam, lam_raw, d_nround, abs_floor, log_stride, conv_tol)
    if (it <= 1) VI_J(1);
    if (it <= 2) VI_J(2);
    VI_J(3);
#undef VI_J
}

// Eigenvectors (column k = eigenvector of slot k, LAPACK layout) from the rotation logs of vi_jacobi_solve.
bool vi_jacobi_vectors_supported(int N) { return vi_jacobi_supported(N); }

int vi_jacobi_vectors(vi_ctx* c, int64_t B, int N, const void* d_log, int max_sweeps, const int* d_nround, double* d_V,
                      int64_t log_stride_in)
{
    // one wave per column strip with the columns in registers; very few systems (one record's prepare / final solve):
    // one column per wave, spread over as many CUs as there are columns (latency); many systems: eight columns per wave,
    // so that a system's rotation log is read N / 8 times (throughput).  The strips of one system are neighbours in the
    // grid (x fastest), so they read its log together through L2.
    const int64_t log_stride = log_stride_in > 0 ? log_stride_in : jacobi_log_stride(N, max_sweeps);
    // columns per wave: as few as keeps ~8 waves per CU busy (a wave with 8 columns takes 8 times as long as one with 1;
    // the handful of systems of a re-basing round waited 1.2-2.9 ms for 18 waves per system while the chip stood empty).
    // All instances do the same arithmetic per column (WaveReplay::inv), so the choice does not show in the result.
    const int64_t cols = B * N, target = (int64_t)8 * c->n_cu;
#define VI_VEC(CPW)                                                                                                           \
    hipLaunchKernelGGL(k_jacobi_vectors_wave<CPW>, dim3((unsigned)((N + CPW - 1) / CPW), (unsigned)B), dim3(64), 0, c->stream, \
                       N, (const double2*)d_log, log_stride, d_nround, d_V)
    if (cols <= target) VI_VEC(1);
    else if (cols <= 2 * target) VI_VEC(2);
    else if (cols <= 4 * target) VI_VEC(4);
    else VI_VEC(8);
#undef VI_VEC
    VI_HIP(hipGetLastError());
    return VI_OK;
}
