// K2s: fused evaluation for HIGH orders (MAXL = 12; BASELINE configs[4]: MAXK 8 x MAXL 12, N = 1152).
//
// k_eval_sph_fast keeps all L Legendre chains of a point live at once: at L = 12 that is cur/prev/cos/sin for twelve
// orders plus the Laguerre factors - the compiler ends at 256 VGPRs + 64 AGPRs of spill moves, one wave per SIMD, and
// 13.7 k VALU instructions per 64 points where ~6.5 k are arithmetic (rocprofv3 SQ_INSTS_VALU).
// Here the orders are processed in NH groups of L / NH consecutive orders, one group after the other: every group runs
// its own chains over all degrees and contracts its own share of every picked degree into the same accumulators.  The
// recurrence work is the same (each chain still runs once); what is repeated per group is only the walk over the
// degree table.  Live state per group: 2 x L/NH chain values + 2 x L/NH trig values.  Three groups of four orders measured
// best at MAXK 8 x MAXL 12: 0.73 ms per 128^3 points against 1.93 ms (2.9e9 points/s, 34 % of the 8.6e9 VALU ceiling).
// Same staging as the fast kernel: recurrence table and coefficient tile in LDS, wave-uniform reads.
#include "vi_common.h"
#include "vi_sph_device.h"

#include <cstdlib>
#include <cstring>

namespace {

template <int L, int K, int TT, int M0, int MW>
struct SplitEval {
    const double* shC;      // LDS [TT][L*L*K] coefficient tile, [t][r = l(l+1)+m][k]
    const double* Lk;       // [K] Laguerre factors of the point
    double* acc;            // [TT] accumulators shared by all groups
    // orders M0 .. M0+MW-1 of degree l (cur / cm / sm are indexed by m - M0)
    template <int l, typename CT>
    __device__ __forceinline__ void consume(const CT* cur, const double* cm, const double* sm)
    {
        constexpr int r0 = l * (l + 1);
        constexpr int NB = L * L * K;
#pragma unroll
        for (int mm = 0; mm < MW; ++mm) {
            const int m = M0 + mm;
            if (m <= l) {
                const double pc = (double)cur[mm] * cm[mm];
                const double ps = (double)cur[mm] * sm[mm];
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const double* cp = shC + t * NB + (r0 + m) * K;
                    const double* cn = shC + t * NB + (r0 - m) * K;
                    double Sp = 0.0, Sm = 0.0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        Sp = fma(cp[k], Lk[k], Sp);
                        if (m > 0) Sm = fma(cn[k], Lk[k], Sm);
                    }
                    acc[t] = fma(pc, Sp, acc[t]);
                    if (m > 0) acc[t] = fma(ps, Sm, acc[t]);
                }
            }
        }
    }
};

template <int L, class Ev, int l, int M0>
struct SplitPickAt {
    template <typename CT>
    __device__ static __forceinline__ void run(Ev& E, const int* nvl, int j, const CT* cur, const double* cm,
                                               const double* sm)
    {
        if constexpr (l >= M0) {
            if (nvl[l] == j) E.template consume<l>(cur, cm, sm);
        }
        if constexpr (l + 1 < L) SplitPickAt<L, Ev, l + 1, M0>::run(E, nvl, j, cur, cm, sm);
    }
};

template <int L, class Ev, int l, int M0, int MW>
struct SplitSegments {
    template <typename CT>
    __device__ static __forceinline__ void run(Ev& E, const CT* shc, const int* nvl, int& j, CT x, CT* cur,
                                               CT* prev, const double* cm, const double* sm)
    {
        const int jend = nvl[l];
        if (jend > L) {
#pragma unroll 2
            for (; j <= jend; ++j) {
                const CT* cj = shc + j * L + M0;
#pragma unroll
                for (int mm = 0; mm < MW; ++mm) {
                    const CT nw = fma(x, cur[mm], -(cj[mm] * prev[mm]));
                    prev[mm] = cur[mm];
                    cur[mm] = nw;
                }
            }
            if constexpr (l >= M0) E.template consume<l>(cur, cm, sm);
        }
        if constexpr (l + 1 < L) SplitSegments<L, Ev, l + 1, M0, MW>::run(E, shc, nvl, j, x, cur, prev, cm, sm);
    }
};

// one group of orders M0 .. M0+MW-1; CT = arithmetic type of the Legendre chains (double, or float for the fp32 variant:
// seeds, trigonometric factors, Laguerre factors and the contraction stay fp64)
template <int L, int K, int TT, int M0, int MW, typename CT>
__device__ __forceinline__ void run_group(const SphGroupDev& G, const Geom& g, const CT* shc, const double* shC,
                                          const int* nvl, int nj, const double* Lk, double* acc)
{
    using Ev = SplitEval<L, K, TT, M0, MW>;
    Ev E;
    E.shC = shC;
    E.Lk = Lk;
    E.acc = acc;
    // cos(m phi), sin(m phi) of the group by angle addition (same recurrence as everywhere else)
    double c0 = 1.0, s0 = 0.0;
#pragma unroll
    for (int i = 0; i < M0; ++i) {
        const double c2 = c0 * g.cphi - s0 * g.sphi;
        s0 = s0 * g.cphi + c0 * g.sphi;
        c0 = c2;
    }
    double cm[MW], sm[MW];
    cm[0] = c0;
    sm[0] = s0;
#pragma unroll
    for (int mm = 1; mm < MW; ++mm) {
        cm[mm] = cm[mm - 1] * g.cphi - sm[mm - 1] * g.sphi;
        sm[mm] = sm[mm - 1] * g.cphi + cm[mm - 1] * g.sphi;
    }
    // (-1)^m (2m-1)!! s^m and s^m for m = M0 - 1 (the start-up below advances them order by order)
    double pmm = 1.0, spow = 1.0;
#pragma unroll
    for (int i = 1; i < M0; ++i) { pmm *= -(2.0 * i - 1.0) * g.s; spow *= g.s; }
    const CT x = (CT)g.x;
    const double zz = 0.5 * (1.0 - g.x);
    const bool intseed = (G.nterms == 0);
    CT cur[MW], prev[MW];
#pragma unroll
    for (int mm = 0; mm < MW; ++mm) { cur[mm] = (CT)0.0; prev[mm] = (CT)0.0; }
    // ---- start-up: degrees j = 0 .. L, compile-time triangular structure ----------------------------------
#pragma unroll
    for (int j = 0; j <= L; ++j) {
#pragma unroll
        for (int mm = 0; mm < MW; ++mm) {
            const int m = M0 + mm;
            if (j > m + 1) {
                if (j < nj) {
                    const CT nw = fma(x, cur[mm], -(shc[j * L + m] * prev[mm]));
                    prev[mm] = cur[mm];
                    cur[mm] = nw;
                }
            } else if (j == m) {
                if (m > 0) { pmm *= -(2.0 * m - 1.0) * g.s; spow *= g.s; }
                if (intseed) cur[mm] = (CT)pmm;
                else cur[mm] = (CT)(G.pref[m] * spow * hyp_series(G.q + (size_t)m * G.nterms, G.nterms, zz));
            } else if (j == m + 1) {
                prev[mm] = cur[mm];
                if (intseed) cur[mm] = (CT)(g.x * (2.0 * m + 1.0) * pmm);
                else cur[mm] = (CT)(G.pref[L + m] * spow * hyp_series(G.q + (size_t)(L + m) * G.nterms, G.nterms, zz));
            }
        }
        SplitPickAt<L, Ev, 0, M0>::run(E, nvl, j, cur, cm, sm);
    }
    // ---- main: the group's chains in recurrence mode; one segment per degree l ----------------------------
    int j = L + 1;
    SplitSegments<L, Ev, 0, M0, MW>::run(E, shc, nvl, j, x, cur, prev, cm, sm);
}

template <int L, int K, int TT, int NH, int G_>
struct GroupLoop {
    template <typename CT>
    __device__ static __forceinline__ void run(const SphGroupDev& G, const Geom& g, const CT* shc, const double* shC,
                                               const int* nvl, int nj, const double* Lk, double* acc)
    {
        constexpr int MW = L / NH;
        run_group<L, K, TT, G_ * MW, MW, CT>(G, g, shc, shC, nvl, nj, Lk, acc);
        if constexpr (G_ + 1 < NH) GroupLoop<L, K, TT, NH, G_ + 1>::run(G, g, shc, shC, nvl, nj, Lk, acc);
    }
};

template <int L, int K, int TT, int NH, typename CT>
__global__ __launch_bounds__(BLOCK) void k_eval_sph_split(SphDev M, int64_t Q, const double* __restrict__ lat,
                                                          const double* __restrict__ lon, const double* __restrict__ alt,
                                                          int tcount, const double* __restrict__ Cp,
                                                          const unsigned char* __restrict__ mask, int F,
                                                          double* __restrict__ out)
{
    static_assert(L % NH == 0, "orders must split evenly");
    extern __shared__ __align__(16) double sh[];
    constexpr int NB = L * L * K;
    const SphGroupDev G = M.groups[0];
    const int nj = G.nvmax + 1;
    CT* shc = reinterpret_cast<CT*>(sh);                // [nj][L] recurrence table in the chain's arithmetic type
    double* shC = sh + ((nj * L + 1) & ~1);             // [TT][NB]
    int* nvl = reinterpret_cast<int*>(shC + TT * NB);   // [L]
    for (int i = threadIdx.x; i < nj * L; i += BLOCK) shc[i] = (CT)G.c[i];
    for (int i = threadIdx.x; i < TT * NB; i += BLOCK) shC[i] = i < tcount * NB ? Cp[i] : 0.0;
    for (int j = threadIdx.x; j < nj; j += BLOCK) {
        const int l = G.pick[j];
        if (l >= 0) nvl[l] = j;
    }
    __syncthreads();

    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t qc = q < Q ? q : Q - 1;
    const Geom g = sph_geom(M, lat[qc], lon[qc], alt[qc]);
    bool in = true;
    if (F > 0) {
        in = mask[qc] != 0;
        if (!__any(in && q < Q)) {
            if (q < Q)
                for (int t = 0; t < tcount; ++t) out[(int64_t)t * Q + q] = __builtin_nan("");
            return;
        }
    }
    double Lk[K];
    laguerre<K>(K, g.z, Lk);
    double acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = 0.0;
    GroupLoop<L, K, TT, NH, 0>::template run<CT>(G, g, shc, shC, nvl, nj, Lk, acc);
    const double Ez = exp(-0.5 * g.z);
    if (q < Q) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < tcount) out[(int64_t)t * Q + q] = in ? Ez * acc[t] : __builtin_nan("");
    }
}

inline unsigned nblocks_s(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

template <int L, int K, int NH, typename CT>
int launch_split(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                 const double* Cp, const unsigned char* hull, int F, double* out)
{
    const int N = m->N;
    const int nj = m->nvmax0 + 1;
    auto shm = [&](int TT) { return (size_t)(((nj * L + 1) & ~1) + TT * N) * sizeof(double) + L * sizeof(int) + 16; };
    // per call, not cached: the attribute is per device and several device contexts may live in one process
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_sph_split<L, K, 4, NH, CT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               64 * 1024));
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_sph_split<L, K, 1, NH, CT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               64 * 1024));
    int64_t t = 0;
    while (t < T) {
        if (T - t >= 4 && shm(4) <= 60 * 1024) {
            hipLaunchKernelGGL((k_eval_sph_split<L, K, 4, NH, CT>), dim3(nblocks_s(Q, BLOCK)), dim3(BLOCK), shm(4), m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 4, Cp + t * N, hull, F, out + t * Q);
            t += 4;
        } else {
            hipLaunchKernelGGL((k_eval_sph_split<L, K, 1, NH, CT>), dim3(nblocks_s(Q, BLOCK)), dim3(BLOCK), shm(1), m->ctx->stream,
                               m->sph, Q, lat, lon, alt, 1, Cp + t * N, hull, F, out + t * Q);
            t += 1;
        }
        VI_HIP(hipGetLastError());
    }
    return VI_OK;
}

}  // namespace

// High-order evaluation (MAXL = 12, one degree group).  Returns VI_OK with *handled = 1 when it took the call;
// VINTERP_EVAL_SPLIT=0 hands everything back to k_eval_sph_fast.
int vi_eval_sph_split(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                      const double* Cp, const unsigned char* hull, int F, double* out, int* handled)
{
    *handled = 0;
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("VINTERP_EVAL_SPLIT");
        enabled = (e && !strcmp(e, "0")) ? 0 : 1;
    }
    if (!enabled || m->sph.ngroups != 1) return VI_OK;
    const int L = m->sph.maxl, K = m->sph.maxk;
    const int nj = m->nvmax0 + 1;
    if ((size_t)(((nj * L + 1) & ~1) + m->N) * sizeof(double) + 64 > 60 * 1024) return VI_OK;
    int rc = VI_OK;
    static int nh = -1;
    if (nh < 0) {
        const char* e = getenv("VINTERP_SPLIT_NH");
        nh = e ? atoi(e) : 3;
    }
    // measured at MAXK 8 x MAXL 12, 128^3 points: 1.93 ms with all twelve chains at once (k_eval_sph_fast), 1.13 ms in two
    // groups of six (190 VGPRs, two waves per SIMD), 0.73 ms in three groups of four (115 VGPRs, four waves), 0.77 ms in
    // four groups of three
    if (m->chain_f32) {          // fp32 Legendre chains (vi_model_set_eval_precision)
        if (L == 12 && K == 8) rc = launch_split<12, 8, 3, float>(m, Q, lat, lon, alt, T, Cp, hull, F, out);
        else return VI_OK;
    } else if (L == 12 && K == 8 && nh == 2) rc = launch_split<12, 8, 2, double>(m, Q, lat, lon, alt, T, Cp, hull, F, out);
    else if (L == 12 && K == 8 && nh == 4) rc = launch_split<12, 8, 4, double>(m, Q, lat, lon, alt, T, Cp, hull, F, out);
    else if (L == 12 && K == 8) rc = launch_split<12, 8, 3, double>(m, Q, lat, lon, alt, T, Cp, hull, F, out);
    else if (L == 12 && K == 2) rc = launch_split<12, 2, 3, double>(m, Q, lat, lon, alt, T, Cp, hull, F, out);
    // (the default order gains nothing from groups: 0.155 ms at once, 0.171 / 0.165 ms in two / three groups)
    else return VI_OK;
    if (rc == VI_OK) *handled = 1;
    return rc;
}
