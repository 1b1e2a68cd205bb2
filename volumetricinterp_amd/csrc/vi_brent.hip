// The root-finder phase of the regularisation-parameter search, one launch for a whole batch.
//
// Reference: scipy.optimize.brentq at volumetricinterp/interpolate.py:214 on f(x) = chi^2(10^x) - nu
// (interpolate.py:220-261), x = log10(alpha) inside the unit bracket the walk found.  Brent's iteration is a chain of
// dependent function values - 12-15 per record, 40-60 on the 15 % of the records whose bracket holds a jump of chi^2 -
// and driven from the host (alpha_search.BrentBatch) every link of the chain is a round of launches for the whole batch:
// form the rotated systems, solve, map back, chi^2, read back, NumPy.  A round lasts as long as its slowest system, the
// last 50 rounds of a 1000-record batch carry a few dozen systems each, and the Python between the rounds is what four
// concurrent pipelines end up waiting for.
// Here a WORKGROUP owns a record from its bracket to its root: k_brent_warm takes records off a queue (atomic counter) and
// runs brentq for each - the state machine in one lane, statement for statement alpha_search.brentq_gen with
// multiply-add contraction off - with every function value computed in place by the code of the host-driven path:
//   X = D1 + alpha D2 scaled to max|X| in [1, 2)            (k_form_pair_scaled, vi_fit.hip)
//   (D1 + alpha D2) c' = yt, truncated                        (jacobi_system, vi_jacobi_device.h - the K3 kernel's body)
//   C = V c'                                                  (k_v_vec)
//   chi^2 = sum_p W_p (A_p . C - b_p)^2                       (k_chi2_part<256, 1> + k_chi2_sum: same chains, same tree)
// and alpha = vi_exp10(x) (vi_exp10.h, the same function on the host).  The result is bit for bit what the host-driven
// iteration gives from the same rotated system (tests/test_gpu_search_stages.py::test_device_brent_equals_host_brent); no
// record waits for another, and the host sees one call.  The rotated system stays at the middle of the bracket (the host
// path's re-basing next to the root needs eigenvectors and six matrix products per record).
// A solve that the sweep cap ends before it converges makes the record leave with status 2: the host runs that record's
// iteration itself (FitEngine: rotated-system solves that did not converge are solved again from X(alpha)).
#include "vi_jacobi_device.h"
#include "vi_jacobi_v2_device.h"
#include "vi_gemm_device.h"
#include "vi_exp10.h"

#include <cstdlib>
#include <cstring>

size_t vi_jacobi_lds_bytes(int N);
size_t vi_jacobi_log_bytes(int N, int max_sweeps);
bool vi_jacobi_supported(int N);
bool vi_jacobi_use_v2(int N);           // vi_jacobi.hip: the role-separated K3 serves this order
double vi_floor_warm();                  // vi_fit.hip: absolute rotation floor of the rotated-system solves
extern "C" int vi_max_sweeps(void);

#ifdef VI_STAMPS
// diagnostic build only (-DVI_STAMPS): cycle sums of thread 0 of every workgroup per part of k_brent_warm; vi_debug_brent_stamps
__device__ unsigned long long g_brent_stamps[16];
__device__ unsigned long long g_brent_hist[64][2];     // by iteration index: sweeps, solves
#define BR_STAMP(k)                                                                   \
    do {                                                                              \
        const unsigned long long t_ = __builtin_readcyclecounter();                   \
        if (threadIdx.x == 0) atomicAdd(&g_brent_stamps[k], t_ - stamp_t);            \
        stamp_t = __builtin_readcyclecounter();                                       \
    } while (0)
#else
#define BR_STAMP(k)
#endif

namespace {

struct BrentState {
    double xpre, xcur, xblk, fpre, fcur, fblk, spre, scur;
    double last_x;                    // the record's previous abscissa (NaN: none yet) - FitEngine._last_x
    int it, funcalls, done, status;
    int nreq, rebased, rebase_now;    // FitEngine._nreq, ._rebased; this iterate moves the rotated system to its alpha
};

// The re-basing rule of the host path (FitEngine._wants_rebase), looking at the record's own abscissae only.
struct RebaseRule {
    double sched[4];                  // k-th move: the first time two consecutive abscissae lie closer than sched[k]
    int nsched;
    int again_after;                  // one more move after this many requests ...
    double again_within;              // ... when two consecutive abscissae lie closer than this (records bisecting a jump)
    int again_on;
    // the early end on a jump of chi^2 (alpha_search.jump_rule): the iteration ends once the sign change is confined to
    // jump_width decades while both ends miss nu by more than jump_frac nu; 0 = brentq's own end
    double jump_width, jump_frac;
};

// h_rebase of the C-ABI: 10 doubles
__host__ inline int rule_from_host(const double* h, RebaseRule& rr)
{
    rr.nsched = (int)h[0];
    if (rr.nsched < 0 || rr.nsched > 4) return -1;
    for (int i = 0; i < 4; ++i) rr.sched[i] = h[1 + i];
    rr.again_after = (int)h[5];
    rr.again_within = h[6];
    rr.again_on = (int)h[7];
    rr.jump_width = h[8];
    rr.jump_frac = h[9];
    return 0;
}

__host__ __device__ __forceinline__ void rebase_decide(BrentState& s, const RebaseRule& rr)
{
    const double x = s.xcur;
    bool rb = false;
    if (s.last_x == s.last_x) {                                   // not NaN
        const double d = fabs(x - s.last_x);
        if (s.rebased < rr.nsched)
            rb = d < rr.sched[s.rebased];
        else
            rb = rr.again_on && s.rebased == rr.nsched && s.nreq >= rr.again_after && d < rr.again_within;
    }
    if (rb) s.rebased += 1;
    s.rebase_now = rb ? 1 : 0;
    s.last_x = x;
    s.nreq += 1;
}

// brentq_gen from the top of its loop to the next request (alpha_search.py: BrentBatch._top for one record); returns true
// when the iteration has ended (root in xcur, other end in xblk).
__host__ __device__ __forceinline__ bool brent_top(BrentState& s, double xtol, double rtol, double jump_width, double jump_fmin)
{
#pragma clang fp contract(off)
    if (s.fpre != 0.0 && s.fcur != 0.0 && (signbit(s.fpre) != signbit(s.fcur))) {
        s.xblk = s.xpre;
        s.fblk = s.fpre;
        s.spre = s.scur = s.xcur - s.xpre;
    }
    if (fabs(s.fblk) < fabs(s.fcur)) {
        const double xp = s.xcur, xc = s.xblk, xb = s.xcur;
        const double fp = s.fcur, fc = s.fblk, fb = s.fcur;
        s.xpre = xp; s.xcur = xc; s.xblk = xb;
        s.fpre = fp; s.fcur = fc; s.fblk = fb;
    }
    const double delta = (xtol + rtol * fabs(s.xcur)) / 2;
    const double sbis = (s.xblk - s.xcur) / 2;
    if (s.fcur == 0.0 || fabs(sbis) < delta) return true;
    if (jump_width > 0.0 && fabs(s.xblk - s.xcur) <= jump_width && fabs(s.fcur) > jump_fmin) return true;   // |fblk| >= |fcur| here
    if (fabs(s.spre) > delta && fabs(s.fcur) < fabs(s.fpre)) {
        double stry;
        if (s.xpre == s.xblk) {
            stry = -s.fcur * (s.xcur - s.xpre) / (s.fcur - s.fpre);                          // secant
        } else {
            const double dpre = (s.fpre - s.fcur) / (s.xpre - s.xcur);                       // inverse quadratic
            const double dblk = (s.fblk - s.fcur) / (s.xblk - s.xcur);
            stry = -s.fcur * (s.fblk * dblk - s.fpre * dpre) / (dblk * dpre * (s.fblk - s.fpre));
        }
        const double a = fabs(s.spre), b = 3 * fabs(sbis) - delta;
        if (2 * fabs(stry) < (b < a ? b : a)) {
            s.spre = s.scur;
            s.scur = stry;
        } else {
            s.spre = sbis;
            s.scur = sbis;
        }
    } else {
        s.spre = sbis;
        s.scur = sbis;
    }
    s.xpre = s.xcur;
    s.fpre = s.fcur;
    if (fabs(s.scur) > delta)
        s.xcur += s.scur;
    else
        s.xcur += (sbis > 0 ? delta : -delta);
    return false;
}

// Out of line: the solver keeps the register budget it has as a kernel of its own (inlined into the loop below it shares the
// 168 registers of 12 waves per CU with thirty live pointers and spills in its rounds).
template <int IT>
__device__ __noinline__ void jacobi_system_call(unsigned char* lds_raw, int N, const double* Xs, double sc, const double* ys,
                                                double rcond, double abs_floor, double* Cs, double2* logp, int max_sweeps,
                                                int* sweeps_s, int* nround_s, unsigned long long* round_acc)
{
    // IT == 0: the role-separated kernel body (vi_jacobi_v2_device.h; blockDim = 64 + one thread per super-block)
    if constexpr (IT == 0)
        jacobi_system_v2(lds_raw, N, Xs, sc, ys, rcond, abs_floor, Cs, nullptr, logp, max_sweeps, sweeps_s, nullptr, 0, nround_s,
                         round_acc);
    else
        jacobi_system<IT>(lds_raw, N, Xs, sc, ys, rcond, abs_floor, Cs, nullptr, logp, max_sweeps, sweeps_s, nullptr, 0, nround_s,
                          round_acc);
}

// The move of a record's rotated system (vi_warm_rebase_f64's kernels in its order), out of line for the same reason.
__device__ __noinline__ void rebase_call(unsigned char* lds_raw, int N, const double2* logp, int64_t nround, double* Vs,
                                         double* VwS, double* VnS, const double* AWAr, const double* Rm, double* D1s, double* D2s,
                                         const double* yr, double* yts
#ifdef VI_STAMPS
                                         , unsigned long long& stamp_t
#endif
                                         )
{
    const int tid = threadIdx.x, NT = blockDim.x, nw = NT >> 6, lane = tid & 63, wave = tid >> 6;
    const int NN = N * N;
    for (int col0 = wave * 4; col0 < N; col0 += nw * 4) jacobi_vector_strip<4>(N, logp, nround, col0, lane, VwS);
    __syncthreads();
    BR_STAMP(6);
    double* ldsg = reinterpret_cast<double*>(lds_raw);
    wg_gemm<false>(N, Vs, VwS, VnS, ldsg);
    for (int e = tid; e < NN; e += NT) Vs[e] = VnS[e];
    __syncthreads();
    wg_gemm<false>(N, AWAr, VnS, VwS, ldsg);
    wg_gemm<true>(N, VnS, VwS, D1s, ldsg);
    wg_gemm<false>(N, Rm, VnS, VwS, ldsg);
    wg_gemm<true>(N, VnS, VwS, D2s, ldsg);
    for (int k = wave; k < N; k += nw) {                 // k_vt_vec_slot
        double acc = 0.0;
        for (int r = lane; r < N; r += 64) acc = fma(VnS[(int64_t)k * N + r], yr[r], acc);
        for (int o2 = 32; o2 > 0; o2 >>= 1) acc += __shfl_down(acc, o2);
        if (lane == 0) yts[k] = acc;
    }
    __syncthreads();
}

template <int IT>
__global__ __launch_bounds__(IT <= 1 ? 768 : 512) void k_brent_warm(
    int N, int64_t P, int ntask, double* D1, double* D2, double* yt, double* V, const double* __restrict__ AWA,
    const double* __restrict__ Rm, const double* __restrict__ ysrc, RebaseRule rr, double* __restrict__ VwW, double* __restrict__ VnW,
    const double* __restrict__ At, const double* __restrict__ W, const double* __restrict__ b,
    const int* __restrict__ t_rec, const int* __restrict__ t_slot, const double* __restrict__ t_xa,
    const double* __restrict__ t_xb, const double* __restrict__ t_fa, const double* __restrict__ t_fb,
    const double* __restrict__ t_nu, double rcond, double abs_floor, int max_sweeps, double xtol, double rtol, int maxiter,
    int* __restrict__ queue, double* __restrict__ Xw, double2* __restrict__ logw, int64_t log_stride, double* __restrict__ cw,
    double* __restrict__ o_root, double* __restrict__ o_other, int* __restrict__ o_iters, int* __restrict__ o_funcalls,
    int* __restrict__ o_status, size_t lds_jacobi, int npart, unsigned long long* __restrict__ round_acc)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // behind the Jacobi kernel's LDS image: coefficients, chi^2 reduction, state
    double* shC = reinterpret_cast<double*>(lds_raw + lds_jacobi);        // [N]
    double* red = shC + ((N + 1) & ~1);                                    // [768]
    double* part = red + 768;                                              // [npart] partial sums of chi^2, one per block of 256 points
    double* shd = part + npart;                                            // [16] max reduction
    BrentState* st = reinterpret_cast<BrentState*>(shd + 16);
    int* shi = reinterpret_cast<int*>(st + 1);                             // [0] task, [1] sweeps, [2] rounds of the last solve

    const int tid = threadIdx.x, NT = blockDim.x, nw = NT >> 6;
    const int NN = N * N;
    double* Xs = Xw + (int64_t)blockIdx.x * NN;
    double2* logp = logw + (int64_t)blockIdx.x * log_stride;
    double* cp = cw + (int64_t)blockIdx.x * N;
    double* VwS = VwW + (int64_t)blockIdx.x * NN;                          // eigenvectors of a rotated system / product scratch
    double* VnS = VnW + (int64_t)blockIdx.x * NN;                          // the new basis V Vw
    const int nb = (int)((P + 255) / 256);

#ifdef VI_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter();
    const unsigned long long wall0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&g_brent_stamps[12], wall0);
#endif
    for (;;) {
        if (tid == 0) shi[0] = atomicAdd(queue, 1);
        __syncthreads();
        const int task = shi[0];
        if (task >= ntask) {
#ifdef VI_STAMPS
            if (threadIdx.x == 0) { const unsigned long long w1 = wall_clock64(); atomicAdd(&g_brent_stamps[11], w1 - wall0); atomicMax(&g_brent_stamps[13], w1); }
#endif
            break;
        }
        const int64_t slot = t_slot[task], rec = t_rec[task];
        const double nu = t_nu[task];
        double* D1s = D1 + slot * NN;
        double* D2s = D2 + slot * NN;
        double* Vs = V + slot * NN;
        const double* Wr = W + rec * P;
        const double* br = b + rec * P;
        if (tid == 0) {
            BrentState s;
            s.xpre = t_xa[task]; s.xcur = t_xb[task]; s.fpre = t_fa[task]; s.fcur = t_fb[task];
            s.xblk = s.fblk = s.spre = s.scur = 0.0;
            s.it = 1; s.funcalls = 0; s.status = 0;
            s.last_x = __builtin_nan("");
            s.nreq = 0; s.rebased = 0; s.rebase_now = 0;
            s.done = brent_top(s, xtol, rtol, rr.jump_width, rr.jump_frac * nu) ? 1 : 0;
            if (!s.done) rebase_decide(s, rr);
            *st = s;
        }
        __syncthreads();
        BR_STAMP(0);
        while (!st->done) {
            const double alpha = vi_exp10(st->xcur);
            // ---- X = f (D1 + alpha D2), f the power of two that brings max|X| into [1, 2)   (k_form_pair_scaled)
            double mx = 0.0;
            for (int e = tid; e < NN; e += NT) mx = fmax(mx, fabs(fma(alpha, D2s[e], D1s[e])));
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
            if ((tid & 63) == 0) shd[tid >> 6] = mx;
            __syncthreads();
            mx = 0.0;
            for (int q = 0; q < nw; ++q) mx = fmax(mx, shd[q]);
            int ex = 0;
            double f = 1.0;
            if (mx > 0.0 && mx < 1.7e308) {
                (void)frexp(mx, &ex);
                f = ldexp(1.0, 1 - ex);
            }
            for (int e = tid; e < NN; e += NT) Xs[e] = fma(alpha, D2s[e], D1s[e]) * f;
            __syncthreads();
            BR_STAMP(1);
            // ---- the truncated solve in the rotated system (the K3 kernel's body)
            jacobi_system_call<IT>(lds_raw, N, Xs, 1.0 / f, yt + slot * N, rcond, abs_floor, cp, logp, max_sweeps, shi + 1, shi + 2,
                                   round_acc);
            __syncthreads();
            BR_STAMP(2);
#ifdef VI_STAMPS
            if (tid == 0) { const int ii_ = st->it < 63 ? st->it : 63; atomicAdd(&g_brent_hist[ii_][0], (unsigned long long)shi[1]); atomicAdd(&g_brent_hist[ii_][1], 1ull); atomicAdd(&g_brent_stamps[8], (unsigned long long)shi[1]); atomicAdd(&g_brent_stamps[9], 1ull); atomicAdd(&g_brent_stamps[10], (unsigned long long)shi[2]); }
#endif
            // ---- C = V c'   (k_v_vec)
            for (int k = tid; k < N; k += NT) red[k] = cp[k];
            __syncthreads();
            for (int r = tid; r < N; r += NT) {
                // one fma chain over k, as k_v_vec; the loads of sixteen links are issued before their fmas (left to the
                // compiler every link waited for its own load: 100 us per function value for 144 x 144 elements)
                double acc = 0.0;
                for (int k0 = 0; k0 < N; k0 += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = (k0 + u < N) ? Vs[(int64_t)(k0 + u) * N + r] : 0.0;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (k0 + u < N) acc = fma(v[u], red[k0 + u], acc);
                }
                shC[r] = acc;
            }
            __syncthreads();
            BR_STAMP(3);
            // ---- chi^2   (k_chi2_part<256, 1>: one fma chain over n per data point, a fixed tree over the 256 points of a
            //      block; k_chi2_sum: the blocks in order).  Two blocks of points at a time, one per half of 512 threads.
            // Three blocks of points at a time (one per 256 threads of the 768; two with 512 threads).
            const int nbp = NT >> 8;
            for (int b0 = 0; b0 < nb; b0 += nbp) {
                const int sub = tid >> 8, t = tid & 255;
                const int blk = b0 + sub;
                double v = 0.0;
                if (sub < nbp && blk < nb) {
                    const int64_t p = (int64_t)blk * 256 + t;
                    if (p < P) {
                        double acc = 0.0;
#pragma unroll 8
                        for (int n = 0; n < N; ++n) acc = fma(At[(int64_t)n * P + p], shC[n], acc);
                        const double d = acc - br[p];
                        v = d * d * Wr[p];
                    }
                }
                if (sub < nbp) red[tid] = v;
                __syncthreads();
                for (int h = 128; h > 0; h >>= 1) {
                    if (sub < nbp && t < h) red[tid] += red[tid + h];
                    __syncthreads();
                }
                if (sub < nbp && t == 0 && blk < nb) part[blk] = red[tid];
                __syncthreads();
            }
            BR_STAMP(4);
            if (tid == 0) {
                double chi = 0.0;
                for (int j = 0; j < nb; ++j) chi += part[j];
                BrentState s = *st;
                if (shi[1] > max_sweeps) {
                    s.status = 2;                    // the solve did not converge: this record goes back to the host
                    s.done = 1;
                } else {
                    s.fcur = chi - nu;
                    s.funcalls += 1;
                    s.it += 1;
                    if (s.it > maxiter) {
                        s.status = 3;
                        s.done = 1;
                    } else {
                        s.done = brent_top(s, xtol, rtol, rr.jump_width, rr.jump_frac * nu) ? 1 : 0;
                    }
                }
                *st = s;
            }
            __syncthreads();
            BR_STAMP(5);
            // every wave takes its copy of the decision BEFORE lane 0 may write the next iterate's flags (rebase_decide below):
            // a wave that read st->rebase_now late could otherwise see the next iterate's 1 and enter rebase_call - which holds
            // barriers - alone
            const bool do_rebase = st->rebase_now && st->status == 0 && !st->done;
            const bool done_now = st->done != 0;
            __syncthreads();
            if (do_rebase) {
                // ---- move the rotated system to this iterate's alpha (vi_warm_rebase_f64): eigenvectors Vw of the rotated
                //      system out of the rotation log of the solve just done, V <- V Vw, and D1 = V^T AWA V, D2 = V^T R V,
                //      yt = V^T y from the untransformed matrices - the host path's kernels, the host path's order
                rebase_call(lds_raw, N, logp, shi[2], Vs, VwS, VnS, AWA + rec * NN, Rm, D1s, D2s, ysrc + rec * N, yt + slot * N
#ifdef VI_STAMPS
                            , stamp_t
#endif
                            );
                BR_STAMP(7);
            }
            if (tid == 0 && !done_now) {
                BrentState s = *st;
                rebase_decide(s, rr);
                *st = s;
            }
            __syncthreads();
        }
        if (tid == 0) {
            o_root[task] = st->xcur;
            o_other[task] = st->xblk;
            o_iters[task] = st->it;
            o_funcalls[task] = st->funcalls;
            o_status[task] = st->status | (st->rebased << 8);      // low byte: the status; above it: how often the system moved
        }
        __syncthreads();
    }
}

void brent_geometry(int N, int& threads, int& it)
{
    if (vi_jacobi_use_v2(N)) {                       // set-up wave + one thread per super-block (at least 512: the chi^2 pass)
        const int Mv = ((N + 3) & ~3) / 4, nsbv = Mv * (Mv - 1) / 2;
        it = 0;
        threads = 64 + ((nsbv + 63) / 64) * 64;
        if (threads < 512) threads = 512;
        return;
    }
    const int M = ((N + 3) & ~3) / 4;
    const int nsb = M * (M - 1) / 2;
    if (nsb <= 768) {
        it = 1;
        threads = ((nsb + 63) / 64) * 64;
        if (threads < 512) threads = 512;          // the chi^2 pass wants two blocks of 256 points
    } else {
        threads = 512;
        it = (nsb + 511) / 512;
    }
}

// LDS of a k_brent_warm workgroup: the Jacobi image (or the product scratch of a re-basing, whichever is larger), the
// coefficients, the chi^2 reduction tree, one partial sum per block of 256 data points, the state.
constexpr size_t VI_LDS_BYTES_PER_CU = 160 * 1024;
size_t brent_lds_bytes(int N, int64_t P, size_t* ldsj_eff, int* npart)
{
    const size_t ldsj = (vi_jacobi_lds_bytes(N) + 15) & ~(size_t)15;
    const size_t ldsj2 = wg_gemm_lds_doubles(N) * sizeof(double);
    const size_t eff = ldsj > ldsj2 ? ldsj : ((ldsj2 + 15) & ~(size_t)15);
    const int64_t nb = (P + 255) / 256;
    const int64_t np = nb < 64 ? 64 : ((nb + 1) & ~(int64_t)1);
    if (ldsj_eff) *ldsj_eff = eff;
    if (npart) *npart = (int)(np > (1 << 20) ? (1 << 20) : np);
    return eff + ((size_t)((N + 1) & ~1) + 768 + (size_t)np + 16) * sizeof(double) + sizeof(BrentState) + 64;
}

}  // namespace

// 1 when vi_brent_warm_f64 serves this order and record size (in-LDS Jacobi range; the chi^2 partial sums of P points fit
// beside the system in one CU's LDS), 0 otherwise: the caller then drives Brent's iteration from the host, which has no limit.
extern "C" int vi_brent_warm_supported(int32_t N, int64_t P)
{
    if (N <= 0 || P <= 0 || !vi_jacobi_supported(N)) return 0;
    return brent_lds_bytes(N, P, nullptr, nullptr) <= VI_LDS_BYTES_PER_CU ? 1 : 0;
}

#ifdef VI_STAMPS
extern "C" int vi_debug_brent_hist(double* out, int reset)
{
    unsigned long long h[64][2];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_brent_hist), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 64; ++i) { out[2 * i] = (double)h[i][0]; out[2 * i + 1] = (double)h[i][1]; }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_brent_hist), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int vi_debug_brent_stamps(double* out, int reset)
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_brent_stamps), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out[i] = (double)h[i];
    if (reset) {
        memset(h, 0, sizeof(h));
        h[12] = ~0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_brent_stamps), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int vi_exp10_f64(const double* x, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = vi_exp10(x[i]);
    return VI_OK;
}

// Brent's iteration (interpolate.py:214) of ONE record driven from the host, in C: the loop of FitEngine's host-driven path for a
// record fitted alone - the same state machine (brent_top / rebase_decide, compiled for the host), the same calls per function
// value (vi_warm_chi2_one_f64; vi_warm_rebase_f64 + vi_chi2_f64 for the value that moves the rotated system) - without the
// interpreter between a value and the next request (~35 us of each ~135 us a dependent iterate spends outside its solve).
// h_out: root, other end, iterations, function calls, status (0 converged, 2 a solve hit the sweep cap - the caller runs the
// record's iteration itself -, 3 maxiter), re-basings.  d_scratch: N + 8 doubles.
extern "C" int vi_warm_chi2_one_f64(vi_ctx* c, int32_t N, int64_t P, const double* d_D1, const double* d_D2, const double* d_yt,
                                    const double* d_V, int32_t slot, double alpha, double rcond, const double* d_At, int32_t rec,
                                    const double* d_W, const double* d_b, double* d_scratch, double* h_chi2);
extern "C" int vi_warm_rebase_f64(vi_ctx* c, int64_t B, int64_t nplain, int32_t N, const double* d_AWA, const double* d_R,
                                  const double* d_y, const int32_t* d_rec, const int32_t* d_slot, const double* d_alpha,
                                  double rcond, double* d_V, double* d_D1, double* d_D2, double* d_yt, double* d_C,
                                  int32_t* d_rank, int32_t* d_sweeps);
extern "C" int vi_chi2_f64(vi_ctx* c, int64_t B, int64_t P, int32_t N, const double* d_At, const double* d_C,
                           const int32_t* d_rec, const double* d_W, const double* d_b, double* d_chi2);
namespace {
__global__ void k_set_one_b(double* scratch, double alpha, int slot, int rec)
{
    scratch[0] = alpha;
    int* iv = reinterpret_cast<int*>(scratch + 2);
    iv[0] = slot;
    iv[1] = rec;
}
}  // namespace

extern "C" int vi_brent_host_one_f64(vi_ctx* c, int32_t N, int64_t P, double* d_D1, double* d_D2, double* d_yt, double* d_V,
                                     const double* d_AWA, const double* d_R, const double* d_y, const double* h_rebase,
                                     const double* d_At, const double* d_W, const double* d_b, int32_t rec, int32_t slot,
                                     double xa, double xb, double fa, double fb, double nu, double rcond, double* d_scratch,
                                     double* h_out)
{
    VI_REQUIRE(c && d_D1 && d_D2 && d_yt && d_V && d_AWA && d_R && d_y && h_rebase && d_At && d_W && d_b && d_scratch && h_out,
               "null argument");
    VI_REQUIRE(N > 0 && P > 0 && rec >= 0 && slot >= 0, "bad size");
    VI_HIP(hipSetDevice(c->device));
    RebaseRule rr;
    if (rule_from_host(h_rebase, rr)) { vi_set_error("vi_brent_host_one_f64: at most four re-basing thresholds"); return VI_ERR_INVALID; }
    const double xtol = 2e-12, rtol = 4 * 2.220446049250313e-16;
    const double jump_fmin = rr.jump_frac * nu;
    const int maxiter = 100, max_sweeps = vi_max_sweeps();
    BrentState s;
    s.xpre = xa; s.xcur = xb; s.fpre = fa; s.fcur = fb;
    s.xblk = s.fblk = s.spre = s.scur = 0.0;
    s.it = 1; s.funcalls = 0; s.status = 0;
    s.last_x = __builtin_nan("");
    s.nreq = 0; s.rebased = 0; s.rebase_now = 0;
    s.done = brent_top(s, xtol, rtol, rr.jump_width, jump_fmin) ? 1 : 0;
    if (!s.done) rebase_decide(s, rr);
    double* d_alpha = d_scratch;
    double* d_chi = d_scratch + 1;
    int32_t* d_slot = reinterpret_cast<int32_t*>(d_scratch + 2);
    int32_t* d_rec = d_slot + 1;
    int32_t* d_sw = reinterpret_cast<int32_t*>(d_scratch + 3);
    double* d_C = d_scratch + 8;
    while (!s.done) {
        const double alpha = vi_exp10(s.xcur);
        double back[3] = {0.0, 0.0, 0.0};
        int rc;
        if (s.rebase_now) {
            hipLaunchKernelGGL(k_set_one_b, dim3(1), dim3(1), 0, c->stream, d_scratch, alpha, (int)slot, (int)rec);
            VI_HIP(hipGetLastError());
            rc = vi_warm_rebase_f64(c, 1, 0, N, d_AWA, d_R, d_y, d_rec, d_slot, d_alpha, rcond, d_V, d_D1, d_D2, d_yt, d_C, nullptr,
                                    d_sw);
            if (rc != VI_OK) return rc;
            rc = vi_chi2_f64(c, 1, P, N, d_At, d_C, d_rec, d_W, d_b, d_chi);
            if (rc != VI_OK) return rc;
            VI_HIP(hipMemcpyAsync(back, d_chi, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            VI_HIP(hipStreamSynchronize(c->stream));
        } else {
            rc = vi_warm_chi2_one_f64(c, N, P, d_D1, d_D2, d_yt, d_V, slot, alpha, rcond, d_At, rec, d_W, d_b, d_scratch, back);
            if (rc != VI_OK) return rc;
        }
        int sweeps;
        memcpy(&sweeps, &back[2], sizeof(int));
        if (sweeps > max_sweeps) {
            s.status = 2;
            break;
        }
        s.fcur = back[0] - nu;
        s.funcalls += 1;
        s.it += 1;
        if (s.it > maxiter) {
            s.status = 3;
            break;
        }
        s.done = brent_top(s, xtol, rtol, rr.jump_width, jump_fmin) ? 1 : 0;
        if (!s.done) rebase_decide(s, rr);
    }
    h_out[0] = s.xcur;
    h_out[1] = s.xblk;
    h_out[2] = (double)s.it;
    h_out[3] = (double)s.funcalls;
    h_out[4] = (double)s.status;
    h_out[5] = (double)s.rebased;
    return VI_OK;
}

// Brent's iteration (interpolate.py:214) of ntask records in one launch, each in the rotated system of its slot
// (vi_warm_prepare_f64).  Task arrays are device arrays of length ntask; outputs likewise: root and the other end of the
// final bracket (log10 alpha), iterations and function calls as brentq counts them, status (low byte; the bits above count the
// record's re-basings) 0 = converged, 2 = a solve hit the
// sweep cap (the caller runs that record's iteration on the host), 3 = maxiter exceeded.
extern "C" int vi_brent_warm_f64(vi_ctx* c, int64_t ntask, int32_t N, int64_t P, double* d_D1, double* d_D2, double* d_yt,
                                 double* d_V, const double* d_AWA, const double* d_R, const double* d_y,
                                 const double* h_rebase /* 10 doubles: nsched, sched[4], again_after, again_within, again_on, jump_width, jump_frac */,
                                 const double* d_At, const double* d_W,
                                 const double* d_b, const int32_t* d_rec, const int32_t* d_slot, const double* d_xa,
                                 const double* d_xb, const double* d_fa, const double* d_fb, const double* d_nu, double rcond,
                                 double* d_root, double* d_other, int32_t* d_iters, int32_t* d_funcalls, int32_t* d_status)
{
    const double abs_floor = vi_floor_warm();
    const int max_sweeps = vi_max_sweeps();
    size_t ldsj_eff;
    VI_REQUIRE(c && d_D1 && d_D2 && d_yt && d_V && d_AWA && d_R && d_y && h_rebase && d_At && d_W && d_b && d_rec && d_slot && d_xa && d_xb && d_fa && d_fb && d_nu &&
                   d_root && d_other && d_iters && d_funcalls && d_status, "null argument");
    VI_REQUIRE(ntask >= 0 && N > 0 && P > 0, "bad size");
    if (ntask == 0) return VI_OK;
    if (!vi_jacobi_supported(N)) {
        vi_set_error("vi_brent_warm_f64: N=%d outside the in-LDS Jacobi range", N);
        return VI_ERR_UNSUPPORTED;
    }
    int npart;
    const size_t shm = brent_lds_bytes(N, P, &ldsj_eff, &npart);
    if (shm > VI_LDS_BYTES_PER_CU) {
        vi_set_error("vi_brent_warm_f64: %lld data points per record need %zu bytes of LDS (the chi^2 partial sums of a record "
                     "sit beside its rotated system); the host-driven iteration has no such limit", (long long)P, shm);
        return VI_ERR_UNSUPPORTED;
    }
    VI_HIP(hipSetDevice(c->device));
    int threads, it;
    brent_geometry(N, threads, it);
    const int64_t nwg = ntask < c->n_cu ? ntask : c->n_cu;
    const size_t logb = vi_jacobi_log_bytes(N, max_sweeps);
    void* ws = nullptr;
    int rc = vi_ctx_workspace(c, (size_t)nwg * (logb + (size_t)(3 * N * N + N) * sizeof(double)) + 256, &ws);
    if (rc != VI_OK) return rc;
    double2* logw = (double2*)ws;
    double* Xw = (double*)((char*)ws + (size_t)nwg * logb);
    double* VwW = Xw + (size_t)nwg * N * N;
    double* VnW = VwW + (size_t)nwg * N * N;
    double* cw = VnW + (size_t)nwg * N * N;
    RebaseRule rr;
    if (rule_from_host(h_rebase, rr)) { vi_set_error("vi_brent_warm_f64: at most four re-basing thresholds"); return VI_ERR_INVALID; }
    int* queue = (int*)(cw + (size_t)nwg * N + 1);
    VI_HIP(hipMemsetAsync(queue, 0, sizeof(int), c->stream));
#define VI_B(ITV)                                                                                                             \
    do {                                                                                                                      \
        VI_HIP(hipFuncSetAttribute((const void*)k_brent_warm<ITV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));    \
        hipLaunchKernelGGL(k_brent_warm<ITV>, dim3((unsigned)nwg), dim3(threads), shm, c->stream, N, P, (int)ntask, d_D1, d_D2, \
                           d_yt, d_V, d_AWA, d_R, d_y, rr, VwW, VnW, d_At, d_W, d_b, d_rec, d_slot, d_xa, d_xb, d_fa, d_fb, d_nu, rcond, abs_floor,          \
                           (int)max_sweeps, 2e-12, 4 * 2.220446049250313e-16, 100, queue, Xw, logw,                           \
                           (int64_t)(logb / sizeof(double2)), cw, d_root, d_other, d_iters, d_funcalls, d_status, ldsj_eff, npart, \
                           c->solve_timing ? c->d_rounds : nullptr);                                                        \
    } while (0)
    // (vi_solve_timing: the launch counts among the eigen-solve launches - it is one, record after record)
    const int tslot = (int)(c->solve_launches % vi_ctx::NSOLVE_EV);
    if (c->solve_timing) VI_HIP(hipEventRecord(c->evs[tslot][0], c->stream));
    if (it == 0) VI_B(0);
    else if (it <= 1) VI_B(1);
    else if (it <= 2) VI_B(2);
    else VI_B(3);
#undef VI_B
    VI_HIP(hipGetLastError());
    if (c->solve_timing) {
        VI_HIP(hipEventRecord(c->evs[tslot][1], c->stream));
        c->solve_launches += 1;
        c->solve_systems += ntask;
    }
    return VI_OK;
}
