// Microbenchmark: the LDS exchange of one K3 round in isolation (csrc/vi_jacobi_device.h, N = 144: 640 threads, one 4 x 4
// super-block of 16 doubles per thread, sixteen planes of 630 doubles).  Every round each thread reads its block (16 ds_read_b64,
// lane-linear: conflict-free), waits at a barrier, and stores the 16 elements at given addresses (16 ds_write_b64), barrier.
// Two address sets are timed: the kernel's real permuted destinations (29 % of its LDS cycles are bank conflicts by the
// profiler, 30 % by tools/sim/k3_bank_sim.py - all of them on these stores) and the identity (conflict-free), which is what a
// perfect re-layout could reach.  The difference is the whole prize of VERDICT round 3 item 3a.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/microbench/lds_exchange.hip -o tools/microbench/libldsexchange.so
#include <hip/hip_runtime.h>
#include <cstdio>

namespace {
__global__ __launch_bounds__(640) void k_exchange(const int* __restrict__ dst, int rounds, int nsb, double* __restrict__ out,
                                                  unsigned long long* __restrict__ cycles)
{
    extern __shared__ double A[];
    const int tid = threadIdx.x;
    const bool live = tid < nsb;
    int d[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = dst[tid * 16 + e];
    for (int e = tid; e < 16 * nsb; e += blockDim.x) A[e] = 1.0 + 1e-6 * e;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        double b[16];
        if (live) {
#pragma unroll
            for (int e = 0; e < 16; ++e) b[e] = A[tid + e * nsb];
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int e = 0; e < 16; ++e) A[d[e]] = b[e] * 1.0000001;
            acc += b[0];
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}
}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

// dst: 640 x 16 destination indices (doubles) inside the 16 x nsb image; one workgroup per CU; returns the median cycles per round
extern "C" int mb_lds_exchange(const int* h_dst, int nsb, int rounds, int nwg, double* cycles_per_round)
{
    int* d_dst;
    double* d_out;
    unsigned long long* d_cyc;
    CK(hipMalloc(&d_dst, 640 * 16 * sizeof(int)));
    CK(hipMalloc(&d_out, (size_t)nwg * 640 * sizeof(double)));
    CK(hipMalloc(&d_cyc, (size_t)nwg * 8));
    CK(hipMemcpy(d_dst, h_dst, 640 * 16 * sizeof(int), hipMemcpyHostToDevice));
    const size_t shm = (size_t)16 * nsb * sizeof(double);
    CK(hipFuncSetAttribute((const void*)k_exchange, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_exchange, dim3(nwg), dim3(640), shm, 0, d_dst, rounds, nsb, d_out, d_cyc);
        CK(hipDeviceSynchronize());
    }
    unsigned long long* h = new unsigned long long[nwg];
    CK(hipMemcpy(h, d_cyc, (size_t)nwg * 8, hipMemcpyDeviceToHost));
    // median
    for (int i = 0; i < nwg; ++i)
        for (int j = i + 1; j < nwg; ++j)
            if (h[j] < h[i]) { unsigned long long t = h[i]; h[i] = h[j]; h[j] = t; }
    *cycles_per_round = (double)h[nwg / 2] / rounds;
    delete[] h;
    hipFree(d_dst); hipFree(d_out); hipFree(d_cyc);
    return 0;
}
