// K2r: evaluation of many timesteps on one grid from the RESIDENT basis matrix of the grid (vi_eval_basis_f64):
//
//   out[t][q] = sum_n C[t][n] * Y[n][q]        (Estimate.__call__, estimate.py:110-123, once per timestep in the reference)
//
// a matrix product with a short inner dimension (N = 144 at the default order) and a very long one (Q = 2^24 points of a
// 256^3 grid), on the fp64 matrix cores: per k-step of four basis functions one v_mfma_f64_16x16x4 per 16 x 16
// (timestep, point) tile, D[16 t][16 q] += A[16 t][4 n] * B[4 n][16 q].
//
// Work of a wave: 64 timesteps x 64 points = 4 x 4 tiles, 128 accumulator registers, 16 independent MFMA per k-step (no
// accumulator is touched again before 15 others have been issued).
//  * B comes straight from HBM / L2, 32 bytes per lane and k-step: lane (p = lane & 15, g = lane >> 4) reads the FOUR
//    consecutive points 4p .. 4p + 3 of basis row 4 ks + g and uses them as its column p of the four column tiles - tile j
//    holds the points 4p + j.  A row of 16 lanes reads 512 contiguous bytes, and on the way out each lane stores its four
//    points of a timestep as one 32-byte piece of that timestep's row (the 16 lanes: 512 contiguous bytes).  The loads of a
//    stage of four k-steps are issued before the MFMAs of the previous stage (two register buffers): a k-step is
//    16 x 64 = 1024 matrix-core cycles.
//  * A is the coefficient tile of the workgroup's 64 timesteps, laid out in LDS in operand order once per workgroup
//    ([k-step][row tile][lane]: ds_read_b64 at lane * 8, linear and conflict-free) and reused for `groups` x 256 points:
//    73 KB at N = 144, two workgroups per CU (one computes while the other sets up its tile).
//  * HBM: Y is 0.15 B per flop at 64 timesteps per pass, so the tiles of the SAME points for different timesteps must meet
//    in a cache.  Workgroups go round-robin over the 8 XCDs (each with its own L2): block b runs on XCD b % 8, so the
//    timestep tiles of one group of points get the block indices 8 apart - same XCD, launched together - and Y leaves HBM
//    once per call, whatever the number of timesteps.
// Bound: the fp64 matrix peak (78.6 TF): 2 N flop per point-timestep against 8 (N / T_call + 1) bytes.
#include "vi_common.h"

#include <cstdlib>
#include <cstring>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int RT = 4;        // row tiles (16 timesteps each) per wave
constexpr int PF = 4;        // k-steps of B in flight

__global__ __launch_bounds__(256, 2) void k_eval_resident(int N, int KSp, int64_t Q, int64_t T, int ntt, int groups, int64_t npg,
                                                          const double* __restrict__ Y, const double* __restrict__ C,
                                                          double* __restrict__ out)
{
    extern __shared__ __align__(16) double shA[];            // [KSp][RT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    // XCD-aware decode: the timestep tiles of a group of points on one XCD, next to each other in launch order
    const int64_t bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const int64_t r = bid >> 3;
    const int tt = (int)(r % ntt);
    const int64_t pg = (r / ntt) * 8 + xcd;
    if (pg >= npg) return;
    const int64_t t0 = (int64_t)tt * (16 * RT);
    // ---- coefficient tile in operand order: shA[(ks * RT + i) * 64 + (g * 16 + p)] = C[t0 + 16 i + p][4 ks + g]
    // (read along the coefficient rows - contiguous -, written to where the operand order wants them)
    const int Np = 4 * KSp;
    for (int e = tid; e < 16 * RT * Np; e += 256) {
        const int tl = e / Np, n = e - tl * Np;
        const int64_t t = t0 + tl;
        shA[((n >> 2) * RT + (tl >> 4)) * 64 + ((n & 3) * 16 + (tl & 15))] = (t < T && n < N) ? C[t * N + n] : 0.0;
    }
    __syncthreads();
    const int KS = (N + 3) >> 2;                              // k-steps that carry basis functions (the rest is padding)
    for (int grp = 0; grp < groups; ++grp) {
        const int64_t qa = (pg * groups + grp) * 256 + wave * 64 + 4 * p;      // this lane's four points
        const bool valid = qa < Q;                                              // Q % 4 == 0: all four or none
        if (__ballot(valid) == 0) break;
        const double* yp = Y + (valid ? qa : 0);
        v4f64 D[RT][4];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) D[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
        // B in two named buffers of PF k-steps: while the MFMAs of one run, the loads of the next stage are in flight (written
        // as a ring of registers replaced one by one, the compiler moved every load next to its use: no distance at all)
        v4f64 yA[PF], yB[PF];
        auto load_stage = [&](v4f64 (&y)[PF], int ks0) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int n = 4 * (ks0 + u) + g;
                y[u] = *reinterpret_cast<const v4f64*>(yp + (int64_t)((ks0 + u < KS && n < N) ? n : 0) * Q);
            }
        };
        auto mfma_stage = [&](const v4f64 (&y)[PF], int ks0) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                double a[RT];
#pragma unroll
                for (int i = 0; i < RT; ++i) a[i] = shA[((ks0 + u) * RT + i) * 64 + lane];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    D[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], y[u].x, D[i][0], 0, 0, 0);
                    D[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], y[u].y, D[i][1], 0, 0, 0);
                    D[i][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], y[u].z, D[i][2], 0, 0, 0);
                    D[i][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], y[u].w, D[i][3], 0, 0, 0);
                }
            }
        };
        load_stage(yA, 0);
        int ks0 = 0;
        for (; ks0 + PF < KSp; ks0 += 2 * PF) {
            load_stage(yB, ks0 + PF);
            asm volatile("" ::: "memory");
            mfma_stage(yA, ks0);
            if (ks0 + 2 * PF < KSp) load_stage(yA, ks0 + 2 * PF);
            asm volatile("" ::: "memory");
            mfma_stage(yB, ks0 + PF);
        }
        if (ks0 < KSp) mfma_stage(yA, ks0);
        // D[i][j][v]: timestep t0 + 16 i + g + 4 v, point qa + j
        if (valid) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t t = t0 + 16 * i + g + 4 * v;
                    if (t < T)
                        *reinterpret_cast<v4f64*>(out + t * Q + qa) = (v4f64){D[i][0][v], D[i][1][v], D[i][2][v], D[i][3][v]};
                }
        }
    }
}

bool use_own_kernel()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VINTERP_EVAL_RESIDENT");
        v = (e && strcmp(e, "blas") == 0) ? 0 : 1;
    }
    return v != 0;
}

}  // namespace

// out[t*Q + q] = sum_n Y[n*Q + q] C[t*N + n] by K2r; *handled = 0 when the shape is not the kernel's (the caller then uses the
// library): Q a multiple of 4 and the matrices 32-byte aligned (the 32-byte pieces), the coefficient tile within the LDS.
int vi_eval_resident_mfma(vi_ctx* c, int N, int64_t Q, int64_t T, const double* d_Y, const double* d_C, double* d_out, int* handled)
{
    *handled = 0;
    if (!use_own_kernel()) return VI_OK;
    const int KSp = (((N + 3) / 4 + PF - 1) / PF) * PF;
    const size_t shm = (size_t)KSp * RT * 64 * sizeof(double);
    if ((Q & 3) != 0 || (((uintptr_t)d_Y | (uintptr_t)d_out) & 31) != 0 || shm > 150 * 1024 || Q < 256) return VI_OK;
    const int ntt = (int)((T + 16 * RT - 1) / (16 * RT));
    // points per workgroup: 256 x groups - the coefficient tile (73 KB through L2) is set up once per workgroup
    // (measured, T = 256: 128^3 points 52 TF with 4 groups, 57 with 32; 256^3 the same from 16 up: as many as leave every CU
    // a few workgroups)
    int groups = (int)((Q >> 16) < 1 ? 1 : ((Q >> 16) > 32 ? 32 : (Q >> 16)));
    if (const char* e = getenv("VINTERP_K2R_GROUPS")) { const int g = atoi(e); if (g >= 1 && g <= 256) groups = g; }      // experiments
    const int64_t npg = (Q + (int64_t)256 * groups - 1) / ((int64_t)256 * groups);
    const int64_t nblk = ((npg + 7) / 8) * 8 * ntt;
    if (nblk > 0x7fffffffLL) return VI_OK;
    VI_HIP(hipFuncSetAttribute((const void*)k_eval_resident, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k_eval_resident, dim3((unsigned)nblk), dim3(256), shm, c->stream, N, KSp, Q, T, ntt, groups, npg, d_Y, d_C,
                       d_out);
    VI_HIP(hipGetLastError());
    *handled = 1;
    return VI_OK;
}
