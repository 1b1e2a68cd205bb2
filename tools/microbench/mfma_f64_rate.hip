// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 (NACC independent accumulators per wave) and of fp64
// VALU FMA on gfx950, to put the evaluation kernels' "fraction of peak" on a measured footing.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(int iters, double* out)
{
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k_fma(int iters, double* out)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
double timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double* out;
    hipMalloc(&out, 256 * 64 * 256 * sizeof(double));
    const int iters = 20000;
    for (int wpb : {4, 8, 16}) {          // waves per block = waves per CU (one block per CU)
        const int blocks = 256;
        double ms = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(64 * wpb), 0, 0, iters, out); });
        double fl = (double)blocks * wpb * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4  4 acc  %2d waves/CU: %.2f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wpb, ms,
               fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * wpb / 4));
        ms = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(64 * wpb), 0, 0, iters, out); });
        fl = (double)blocks * wpb * iters * 1 * 2048.0;
        printf("mfma_f64_16x16x4  1 acc  %2d waves/CU: %.2f ms  %.1f TFLOP/s\n", wpb, ms, fl / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(64 * wpb), 0, 0, iters, out); });
        fl = (double)blocks * wpb * 64 * iters * 8 * 2.0;
        printf("fp64 VALU fma     8 acc  %2d waves/CU: %.2f ms  %.1f TFLOP/s\n", wpb, ms, fl / ms / 1e9);
    }
    return 0;
}
