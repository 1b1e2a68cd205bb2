// Microbenchmark for DESIGN.md section 4 ("a matrix-core K3"): the two building blocks of a two-sided BLOCK Jacobi eigen-solver
// at N = 144 with 16-wide blocks (9 blocks, 36 block pairs per sweep = 9 steps of 4 disjoint pairs), measured instead of sized:
//   A  k_pivot32:    one wave diagonalises a 32 x 32 symmetric pivot block [A_II A_IJ; A_JI A_JJ] by cyclic Jacobi (round robin,
//                    31 rounds of 16 disjoint rotations per inner sweep; rotation parameters by the K3 formulas - v_rsq_f64 +
//                    one third-order step, no division), accumulating the 32 x 32 orthogonal factor U.  Matrix and U in LDS.
//   B  k_panel_mfma: the 32 x 144 row-panel update  [A_I,: ; A_J,:] <- U^T [A_I,: ; A_J,:]  on v_mfma_f64_16x16x4 (2 x 9 output
//                    tiles, 8 k-steps each: 144 MFMAs per panel), one wave per panel, four waves (panels) per workgroup.
// Built as a shared library and driven by tools/exp_block_jacobi.py (inputs from the reference's default-order systems,
// eigenvalues checked against LAPACK).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/microbench/block_jacobi_n144.hip -o tools/microbench/libblockjacobi.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ double fast_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}

constexpr int NB = 32, LD = 33;

// pair k (0..15) of round r (0..30) of the round robin of 32 indices (31 on a circle, index 31 fixed)
__device__ __forceinline__ void rr_pair(int r, int k, int& p, int& q)
{
    if (k == 0) { p = r; q = 31; }
    else { p = (r + k) % 31; q = (r + 31 - k) % 31; }
    if (p > q) { const int t = p; p = q; q = t; }
}

__global__ __launch_bounds__(64) void k_pivot32(const double* __restrict__ Ain, int max_sweeps, double drop, double abs_floor,
                                                double* __restrict__ lam,
                                                double* __restrict__ Uout, int* __restrict__ sweeps_out,
                                                unsigned long long* __restrict__ cycles_out)
{
    __shared__ double A[NB * LD], U[NB * LD];
    __shared__ double2 cs[16];
    __shared__ int pq[16][2];
    const int lane = threadIdx.x;
    const double* Ai = Ain + (int64_t)blockIdx.x * NB * NB;
    for (int e = lane; e < NB * NB; e += 64) {
        const int i = e / NB, j = e % NB;
        A[i * LD + j] = Ai[e];
        U[i * LD + j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < 31; ++r) {
            if (lane < 16) {
                int p, q;
                rr_pair(r, lane, p, q);
                const double app = A[p * LD + p], aqq = A[q * LD + q], apq = A[p * LD + q];
                const double aa = fabs(apq);
                // the K3 criterion (vi_jacobi_device.h rot_params): relative, with an absolute floor, and pairs wholly inside the
                // subspace the truncation is going to drop are left alone
                const bool tiny = fmax(fmax(fabs(app), fabs(aqq)), aa) < 0.0625 * drop;
                const bool rot = aa * aa > eps2 * fabs(app * aqq) && aa > abs_floor && !tiny;
                const double d = aqq - app;
                const double ri = fast_rsqrt(fma(d, d, 4.0 * apq * apq));
                const double w = fma(0.5 * fabs(d), ri, 0.5);
                const double rw = fast_rsqrt(w);
                const double qq = copysign(aa * ri, d * apq);
                cs[lane] = rot ? make_double2(w * rw, qq * rw) : make_double2(1.0, 0.0);
                pq[lane][0] = p;
                pq[lane][1] = q;
                rotated |= rot ? 1 : 0;
            }
            __syncthreads();
            // rows p, q <- R^T (rows p, q): 16 pairs x 32 columns = 512 items, 8 per lane
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int item = lane + 64 * t, k = item >> 5, j = item & 31;
                const int p = pq[k][0], q = pq[k][1];
                const double2 r_ = cs[k];
                const double x = A[p * LD + j], z = A[q * LD + j];
                A[p * LD + j] = r_.x * x - r_.y * z;
                A[q * LD + j] = r_.y * x + r_.x * z;
            }
            __syncthreads();
            // columns p, q <- (columns p, q) R, for A and for the accumulated U
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int item = lane + 64 * t, k = item >> 5, i = item & 31;
                const int p = pq[k][0], q = pq[k][1];
                const double2 r_ = cs[k];
                const double x = A[i * LD + p], z = A[i * LD + q];
                A[i * LD + p] = r_.x * x - r_.y * z;
                A[i * LD + q] = r_.y * x + r_.x * z;
                const double ux = U[i * LD + p], uz = U[i * LD + q];
                U[i * LD + p] = r_.x * ux - r_.y * uz;
                U[i * LD + q] = r_.y * ux + r_.x * uz;
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) { ++sweep; break; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane < NB) lam[(int64_t)blockIdx.x * NB + lane] = A[lane * LD + lane];
    for (int e = lane; e < NB * NB; e += 64) Uout[(int64_t)blockIdx.x * NB * NB + e] = U[(e / NB) * LD + e % NB];
    if (lane == 0) {
        sweeps_out[blockIdx.x] = sweep;
        cycles_out[blockIdx.x] = t1 - t0;
    }
}

// One wave per 32 x 144 panel: P <- U^T P, `iters` times (the panel stays in LDS, as it would inside a solver).
// v_mfma_f64_16x16x4: lane l supplies a = A[l & 15][l >> 4] (16 x 4), b = B[l >> 4][l & 15] (4 x 16); D[(l >> 4) + 4 v][l & 15].
constexpr int NC = 144, PLD = 146;
__global__ __launch_bounds__(256) void k_panel_mfma(const double* __restrict__ Uin, double* __restrict__ Pio, int iters,
                                                    unsigned long long* __restrict__ cycles_out)
{
    extern __shared__ double sh[];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    double* P = sh + wave * (NB * PLD);
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    const double* Ui = Uin + id * NB * NB;
    double* Pg = Pio + id * NB * NC;
    for (int e = l; e < NB * NC; e += 64) P[(e / NC) * PLD + e % NC] = Pg[e];
    // A operand: (U^T)[row][k] = U[k][row]; row tile rt, k-step ks: a = U[ks * 4 + (l >> 4)][rt * 16 + (l & 15)]
    double ua[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) ua[rt][ks] = Ui[(ks * 4 + (l >> 4)) * NB + rt * 16 + (l & 15)];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int ct = 0; ct < 9; ++ct) {
            v4f64 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            double b[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) b[ks] = P[(ks * 4 + (l >> 4)) * PLD + ct * 16 + (l & 15)];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[0][ks], b[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[1][ks], b[ks], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                P[((l >> 4) + 4 * v) * PLD + ct * 16 + (l & 15)] = acc0[v];
                P[(16 + (l >> 4) + 4 * v) * PLD + ct * 16 + (l & 15)] = acc1[v];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    for (int e = l; e < NB * NC; e += 64) Pg[e] = P[(e / NC) * PLD + e % NC];
    if (l == 0) cycles_out[id] = t1 - t0;
}

}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

// B pivot blocks (host, B x 32 x 32) -> eigenvalues (B x 32), U (B x 32 x 32), sweeps (B), cycles (B); returns kernel ms in *ms
extern "C" int mb_pivot32(const double* hA, int B, int max_sweeps, double drop, double abs_floor, double* hLam, double* hU, int* hSweeps, double* hCycles, double* ms)
{
    double *dA, *dL, *dU;
    int* dS;
    unsigned long long* dC;
    CK(hipMalloc(&dA, (size_t)B * 1024 * 8));
    CK(hipMalloc(&dL, (size_t)B * 32 * 8));
    CK(hipMalloc(&dU, (size_t)B * 1024 * 8));
    CK(hipMalloc(&dS, (size_t)B * 4));
    CK(hipMalloc(&dC, (size_t)B * 8));
    CK(hipMemcpy(dA, hA, (size_t)B * 1024 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_pivot32, dim3(B), dim3(64), 0, 0, dA, max_sweeps, drop, abs_floor, dL, dU, dS, dC);      // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pivot32, dim3(B), dim3(64), 0, 0, dA, max_sweeps, drop, abs_floor, dL, dU, dS, dC);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float f;
    CK(hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    unsigned long long* hc = new unsigned long long[B];
    CK(hipMemcpy(hLam, dL, (size_t)B * 32 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hU, dU, (size_t)B * 1024 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hSweeps, dS, (size_t)B * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc, dC, (size_t)B * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; ++i) hCycles[i] = (double)hc[i];
    delete[] hc;
    hipFree(dA); hipFree(dL); hipFree(dU); hipFree(dS); hipFree(dC);
    return 0;
}

// Bp panels (Bp a multiple of 4): P <- U^T P `iters` times; hP in/out (Bp x 32 x 144), hU (Bp x 32 x 32)
extern "C" int mb_panel_mfma(const double* hU, double* hP, int Bp, int iters, double* hCycles, double* ms)
{
    double *dU, *dP;
    unsigned long long* dC;
    CK(hipMalloc(&dU, (size_t)Bp * 1024 * 8));
    CK(hipMalloc(&dP, (size_t)Bp * 32 * 144 * 8));
    CK(hipMalloc(&dC, (size_t)Bp * 8));
    CK(hipMemcpy(dU, hU, (size_t)Bp * 1024 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, hP, (size_t)Bp * 32 * 144 * 8, hipMemcpyHostToDevice));
    const size_t shm = (size_t)4 * NB * PLD * sizeof(double);
    CK(hipFuncSetAttribute((const void*)k_panel_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_panel_mfma, dim3(Bp / 4), dim3(256), shm, 0, dU, dP, iters, dC);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float f;
    CK(hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    unsigned long long* hc = new unsigned long long[Bp];
    CK(hipMemcpy(hP, dP, (size_t)Bp * 32 * 144 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc, dC, (size_t)Bp * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < Bp; ++i) hCycles[i] = (double)hc[i];
    delete[] hc;
    hipFree(dU); hipFree(dP); hipFree(dC);
    return 0;
}
