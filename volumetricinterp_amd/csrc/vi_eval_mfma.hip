// K2m: evaluation of MANY timesteps on one query grid (Estimate.__call__, estimate.py:110-123,
// called once per timestep by the reference) with the contraction on the gfx950 matrix cores.
//
//   out[t][q] = exp(-z_q/2) * sum_r sum_k  Cp[t][r][k] * ( Y_r(q) * Lag_k(z_q) ),   r = l(l+1)+m,  k = Laguerre index
//
// is, per basis row r and quad kq of Laguerre indices, one v_mfma_f64_16x16x4_f64 per 16 x 16 (timestep, point) tile:
//   D[16 timesteps][16 points] += A[16 t][4 k] * B[4 k][16 points].
//
// Two lane roles in one wave:
//  * RECURRENCE role: lane = point (64 points per wave), every lane runs all L Legendre chains of its point exactly
//    as the VALU kernel k_eval_sph_fast does (wave-uniform table reads, compile-time triangular start-up) - no
//    redundancy, full VALU efficiency.
//  * OPERAND role: the 64 points are four sub-tiles of 16; for sub-tile i, lane (p = lane & 15, g = lane >> 4)
//    supplies B[g][p] = Y_r(point 16 i + p) * Lag_{4 kq + g}(point 16 i + p).  Y_r moves from the recurrence role
//    to the operand role by ONE wave shuffle per sub-tile; the Laguerre factors of the four operand points of a
//    lane are gathered once per 64 points.  A[t = p][k = g] is read from an LDS image of the coefficient tile
//    stored in operand order (ds_read_b64 at lane * 8: linear, conflict-free) and is reused by the four sub-tiles.
//    D comes back as col = lane & 15 (point), row = (lane >> 4) + 4 v (timestep): the 16 lanes of a lane group
//    store 128 contiguous bytes of one timestep's output row.
//
// Why MFMA although the fp64 matrix path has the SAME peak as the fp64 VALU on gfx950 (78.6 TF), and although the
// two do NOT overlap (rocprofv3 SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this kernel: v_mfma_f64 executes on the same
// double-precision units as v_fma_f64): one v_mfma_f64_16x16x4 retires 2048 flop with two register operands and runs
// at the full rate (measured 77.7 TF sustained, 64 cycles per instruction per SIMD; VALU FMA sustains 62.9 TF;
// tools/microbench/mfma_f64_rate.hip), whereas the VALU contraction of k_eval_sph_fast spends issue slots on LDS
// broadcast reads and reaches 24 % of peak at 16 timesteps per tile.  Measured at 128^3, N = 144 (r1): 1.2e11
// point-timesteps/s for T >= 32 (44 % of peak at 2N flop per pair) against 6.6e10 for the VALU tile kernel.
// Per 64 points x 32 timesteps a wave issues 288 MFMA (18.4 k cycles) and ~2.7 k VALU instructions (10.8 k cycles).
// A workgroup keeps its coefficient image for `ngrp` x 256 points, so the image costs ~1 B of L2 traffic per
// point-timestep.  Tiles of 64 timesteps were measured too: 128 accumulator registers leave one wave per SIMD and
// the rate is the same, so the widest tile is 32.
#include "vi_common.h"
#include "vi_sph_device.h"

#include <cstdlib>
#include <cstring>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// 64-bit wave shuffle with a precomputed ds_bpermute byte address (4 * source lane): two LDS-crossbar instructions,
// no address arithmetic (the generic __shfl recomputes lane id and width masks on every call)
__device__ __forceinline__ double bperm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int L, int KQ, int NT>
struct MfmaSink {
    static constexpr int R = L * L;
    const double* shA;        // LDS [NT][R][KQ][64], already offset by the lane
    double Lg[4][KQ];         // Lag_{4 kq + g} of this lane's operand point in sub-tile i
    int src[4];               // ds_bpermute address of the lane holding (recurrence role) the operand point of sub-tile i
    v4f64 D[4][NT];

    __device__ __forceinline__ void feed(int r, double y)
    {
        double a[NT][KQ];
#pragma unroll
        for (int tile = 0; tile < NT; ++tile)
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) a[tile][kq] = shA[((tile * R + r) * KQ + kq) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double yi = bperm(src[i], y);
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const double b = yi * Lg[i][kq];
#pragma unroll
                for (int tile = 0; tile < NT; ++tile)
                    D[i][tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tile][kq], b, D[i][tile], 0, 0, 0);
            }
        }
    }
    // degree l reached: cur[m] = chain value of order m at degree nu_l (this lane's own point)
    template <int l>
    __device__ __forceinline__ void consume(const double* cur, const double* cm, const double* sm)
    {
        constexpr int r0 = l * (l + 1);
#pragma unroll
        for (int m = 0; m <= l; ++m) {
            feed(r0 + m, cur[m] * cm[m]);
            if (m > 0) feed(r0 - m, cur[m] * sm[m]);
        }
    }
};

template <int L, class Sink, int l>
struct PickAt {
    // consume degree l if its integer part equals j (start-up phase only: j <= L)
    __device__ static __forceinline__ void run(Sink& E, const int* nvr, int j, const double* cur, const double* cm,
                                               const double* sm)
    {
        if (nvr[l] == j) E.template consume<l>(cur, cm, sm);
        if constexpr (l + 1 < L) PickAt<L, Sink, l + 1>::run(E, nvr, j, cur, cm, sm);
    }
};

template <int L, class Sink, int l>
struct MainSeg {
    __device__ static __forceinline__ void run(Sink& E, const double* shc, const int* nvr, int& j, double x, double* cur,
                                               double* prev, const double* cm, const double* sm)
    {
        const int jend = nvr[l];
        if (jend > L) {
#pragma unroll 2
            for (; j <= jend; ++j) {
                const double* cj = shc + j * L;
#pragma unroll
                for (int m = 0; m < L; ++m) {
                    const double nw = fma(x, cur[m], -(cj[m] * prev[m]));
                    prev[m] = cur[m];
                    cur[m] = nw;
                }
            }
            E.template consume<l>(cur, cm, sm);
        }
        if constexpr (l + 1 < L) MainSeg<L, Sink, l + 1>::run(E, shc, nvr, j, x, cur, prev, cm, sm);
    }
};

constexpr int MBLOCK = 256;      // 4 waves x 64 points

template <int L, int KQ, int NT>
__global__ __launch_bounds__(MBLOCK, (NT >= 2 ? 2 : 1)) void k_eval_sph_mfma(SphDev M, int64_t Q, const double* __restrict__ lat,
                                                          const double* __restrict__ lon, const double* __restrict__ alt,
                                                          int tcount, const double* __restrict__ Cp,
                                                          const unsigned char* __restrict__ mask, int F, int ngrp,
                                                          double* __restrict__ out)
{
    extern __shared__ __align__(16) double sh[];
    using Sink = MfmaSink<L, KQ, NT>;
    constexpr int R = L * L, TT = NT * 16;
    const SphGroupDev G = M.groups[0];
    const int K = M.maxk, NB = R * K;
    const int nj = G.nvmax + 1;
    double* shc = sh;                                   // [nj][L] recurrence table
    double* shA = sh + ((nj * L + 1) & ~1);             // [NT][R][KQ][4][16] coefficient image in operand order
    int* nvl = reinterpret_cast<int*>(shA + NT * R * KQ * 64);   // [L]
    const int tid = threadIdx.x;
    for (int i = tid; i < nj * L; i += MBLOCK) shc[i] = G.c[i];
    if (K < 4 * KQ || tcount < TT)
        for (int i = tid; i < NT * R * KQ * 64; i += MBLOCK) shA[i] = 0.0;
    for (int j = tid; j < nj; j += MBLOCK) {
        const int l = G.pick[j];
        if (l >= 0) nvl[l] = j;
    }
    __syncthreads();
    {
        const int ne = (tcount < TT ? tcount : TT) * NB;     // coalesced read of Cp[t][r*K + k], scattered LDS write
        for (int e = tid; e < ne; e += MBLOCK) {
            const int t = e / NB, n = e - t * NB;
            const int r = n / K, k = n - r * K;
            shA[(((t >> 4) * R + r) * KQ + (k >> 2)) * 64 + (k & 3) * 16 + (t & 15)] = Cp[e];
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int pi = lane & 15, g = lane >> 4;
    Sink E;
    E.shA = shA + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) E.src[i] = (16 * i + pi) << 2;
    const bool intseed = (G.nterms == 0);
    int nvr[L];                      // integer part of nu_l: wave-uniform, kept in scalar registers
#pragma unroll
    for (int l = 0; l < L; ++l) nvr[l] = __builtin_amdgcn_readfirstlane(nvl[l]);

    for (int it = 0; it < ngrp; ++it) {
        const int64_t q0 = ((int64_t)blockIdx.x * ngrp + it) * MBLOCK + wave * 64;
        if (q0 >= Q) break;                                  // wave-uniform
        // ---- recurrence role: lane = point --------------------------------------------------------------
        const int64_t q = q0 + lane;
        const int64_t qc = q < Q ? q : Q - 1;
        bool in = true;
        if (F > 0) in = mask[qc] != 0;
        if (F > 0 && !__any(in && q < Q)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tile = 0; tile < NT; ++tile)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int t = tile * 16 + g + 4 * v;
                        const int64_t qo = q0 + 16 * i + pi;
                        if (t < tcount && qo < Q) out[(int64_t)t * Q + qo] = __builtin_nan("");
                    }
            continue;
        }
        const Geom gm = sph_geom(M, lat[qc], lon[qc], alt[qc]);
        double cm[L], sm[L];
        cm[0] = 1.0;
        sm[0] = 0.0;
#pragma unroll
        for (int m = 1; m < L; ++m) {
            cm[m] = cm[m - 1] * gm.cphi - sm[m - 1] * gm.sphi;
            sm[m] = sm[m - 1] * gm.cphi + cm[m - 1] * gm.sphi;
        }
        // exp(-z/2) of the own point, then of the four operand points; NaN marks a point outside the hull
        const double Eown = in ? exp(-0.5 * gm.z) : __builtin_nan("");
        double Eop[4];
        {
            double Lall[4 * KQ];
#pragma unroll
            for (int k = 0; k < 4 * KQ; ++k) Lall[k] = 0.0;
            laguerre<4 * KQ>(K, gm.z, Lall);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Eop[i] = bperm(E.src[i], Eown);
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) {
                    const double a = bperm(E.src[i], Lall[4 * kq]), b = bperm(E.src[i], Lall[4 * kq + 1]);
                    const double c = bperm(E.src[i], Lall[4 * kq + 2]), d = bperm(E.src[i], Lall[4 * kq + 3]);
                    E.Lg[i][kq] = g == 0 ? a : g == 1 ? b : g == 2 ? c : d;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tile = 0; tile < NT; ++tile) E.D[i][tile] = v4f64{0.0, 0.0, 0.0, 0.0};
        const double x = gm.x;
        const double zz = 0.5 * (1.0 - x);
        double cur[L], prev[L];
#pragma unroll
        for (int m = 0; m < L; ++m) { cur[m] = 0.0; prev[m] = 0.0; }
        double pmm = 1.0, spow = 1.0;
        // ---- start-up: degrees j = 0 .. L, compile-time triangular structure (as k_eval_sph_fast) ------------
#pragma unroll
        for (int j = 0; j <= L; ++j) {
#pragma unroll
            for (int m = 0; m < L; ++m) {
                if (j > m + 1) {
                    if (j < nj) {
                        const double nw = fma(x, cur[m], -(shc[j * L + m] * prev[m]));
                        prev[m] = cur[m];
                        cur[m] = nw;
                    }
                } else if (j == m) {
                    if (m > 0) { pmm *= -(2.0 * m - 1.0) * gm.s; spow *= gm.s; }
                    if (intseed) cur[m] = pmm;
                    else cur[m] = G.pref[m] * spow * hyp_series(G.q + (size_t)m * G.nterms, G.nterms, zz);
                } else if (j == m + 1) {
                    prev[m] = cur[m];
                    if (intseed) cur[m] = x * (2.0 * m + 1.0) * cur[m];
                    else cur[m] = G.pref[L + m] * spow * hyp_series(G.q + (size_t)(L + m) * G.nterms, G.nterms, zz);
                }
            }
            PickAt<L, Sink, 0>::run(E, nvr, j, cur, cm, sm);
        }
        // ---- main: all chains in recurrence mode; one segment per degree l ----------------------------------
        int j = L + 1;
        MainSeg<L, Sink, 0>::run(E, shc, nvr, j, x, cur, prev, cm, sm);
        // ---- operand role: D[i][tile][v] is (timestep tile*16 + g + 4 v, point q0 + 16 i + pi) ---------------
        if (tcount == TT && q0 + 64 <= Q) {                  // whole tile in range (wave-uniform): plain stores
            double* o = out + (int64_t)g * Q + q0 + pi;
#pragma unroll
            for (int tile = 0; tile < NT; ++tile)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double* ot = o + (int64_t)(tile * 16 + 4 * v) * Q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ot[16 * i] = Eop[i] * E.D[i][tile][v];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t qo = q0 + 16 * i + pi;
#pragma unroll
                for (int tile = 0; tile < NT; ++tile)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int t = tile * 16 + g + 4 * v;
                        if (t < tcount && qo < Q) out[(int64_t)t * Q + qo] = Eop[i] * E.D[i][tile][v];
                    }
            }
        }
    }
}

inline unsigned nblocks64(int64_t n, int64_t b) { return (unsigned)((n + b - 1) / b); }

template <int L, int KQ, int NT>
int launch_tile(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int tcount,
                const double* Cp, const unsigned char* hull, int F, double* out)
{
    const int nj = m->nvmax0 + 1;
    const size_t shm = (size_t)(((nj * L + 1) & ~1) + NT * L * L * KQ * 64) * sizeof(double) + L * sizeof(int) + 16;
    // per call, not cached: the attribute is per device and several device contexts may live in one process
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_sph_mfma<L, KQ, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    // points per workgroup: enough groups of 256 to amortise the coefficient image, yet >= ~8 workgroups per CU
    int ngrp = 8;
    while (ngrp > 1 && (Q + MBLOCK * ngrp - 1) / (MBLOCK * ngrp) < (int64_t)8 * m->ctx->n_cu) ngrp >>= 1;
    hipLaunchKernelGGL((k_eval_sph_mfma<L, KQ, NT>), dim3(nblocks64(Q, (int64_t)MBLOCK * ngrp)), dim3(MBLOCK), shm,
                       m->ctx->stream, m->sph, Q, lat, lon, alt, tcount, Cp, hull, F, ngrp, out);
    VI_HIP(hipGetLastError());
    return VI_OK;
}

template <int L, int KQ>
int launch_mfma(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                const double* Cp, const unsigned char* hull, int F, double* out, int64_t* done)
{
    const int N = m->N;
    const int nj = m->nvmax0 + 1;
    auto fits = [&](int NT) {
        return (size_t)(((nj * L + 1) & ~1) + NT * L * L * KQ * 64) * sizeof(double) + L * sizeof(int) + 16 <= 80 * 1024;
    };
    int64_t t = 0;
    static int ntmax = -1;             // experiment switch: widest timestep tile (in units of 16)
    if (ntmax < 0) {
        const char* e = getenv("VINTERP_MFMA_NT");
        ntmax = e ? atoi(e) : 2;
    }
    while (T - t >= 16) {
        int rc;
        if (T - t >= 32 && fits(2) && ntmax >= 2) {
            rc = launch_tile<L, KQ, 2>(m, Q, lat, lon, alt, 32, Cp + t * N, hull, F, out + t * Q);
            t += 32;
        } else if (fits(1)) {
            rc = launch_tile<L, KQ, 1>(m, Q, lat, lon, alt, 16, Cp + t * N, hull, F, out + t * Q);
            t += 16;
        } else {
            break;
        }
        if (rc != VI_OK) return rc;
    }
    *done = t;
    return VI_OK;
}

}  // namespace

// Evaluate as many leading timesteps of the (prepared) coefficient block as the matrix-core kernel takes - whole
// tiles of 16 - and report how many were done; the caller finishes the rest with the VALU kernels.
// Orders without an instantiation, models with more than one degree group and VINTERP_EVAL_MFMA=0 report 0.
int vi_eval_sph_mfma(vi_model* m, int64_t Q, const double* lat, const double* lon, const double* alt, int64_t T,
                     const double* Cp, const unsigned char* hull, int F, double* out, int64_t* done)
{
    *done = 0;
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("VINTERP_EVAL_MFMA");
        enabled = (e && !strcmp(e, "0")) ? 0 : 1;
    }
    if (!enabled || m->sph.ngroups != 1 || T < 16) return VI_OK;
    const int L = m->sph.maxl, K = m->sph.maxk;
#define VI_MFMA(LL, KK) \
    if (L == LL && K == KK) return launch_mfma<LL, (KK + 3) / 4>(m, Q, lat, lon, alt, T, Cp, hull, F, out, done)
    VI_MFMA(6, 4);
    VI_MFMA(3, 4);
    VI_MFMA(2, 8);
    // MAXL = 12: the 144-row coefficient image exceeds the LDS share of a workgroup, and with a 96 KB image the kernel
    // measured no faster than the VALU tile kernel (5.48 vs 5.50 ms for 32 timesteps x 128^3 at MAXK = 2): not built
#undef VI_MFMA
    return VI_OK;
}
