// Rate of the inner step of k_hull_mask_mx: one v_mfma_f32_32x32x16_f16 followed by NMAX v_max3_f32 over its sixteen results,
// at 1, 2, 4 waves per SIMD (max code 8: a tree of eight v_max3_f32; 80: a chain of eight; 16: sixteen v_max_f32; 4 / 1: four / one
// v_max3_f32; 'no mfma': the same maxima over registers that stay).  Prints cycles per step and SIMD (2.4 GHz assumed) - the matrix pipe alone is 32.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_max3_rate.hip -o mfma_max3_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NMAX, int NG, bool NOMFMA = false>
__global__ __launch_bounds__(256) void k(const h8* a, const h8* b, float* out, int iters)
{
    const h8 A = a[threadIdx.x & 63];
    h8 B[NG];
    float dmax[NG];
    for (int g = 0; g < NG; ++g) { B[g] = b[(threadIdx.x + g) & 63]; dmax[g] = -3e38f; }
    const f16v zero = {0};
    f16v keep = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B[0], zero, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f16v acc;
            if (NOMFMA) {
                acc = keep;
                asm volatile("" : "+v"(acc));
            } else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B[g], zero, 0, 0, 0);
            if (NMAX == 8) {
                const float t0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), t1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]),
                            t2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), t3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]),
                            t4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]), t5 = fmaxf(fmaxf(acc[15], dmax[g]), t0),
                            t6 = fmaxf(fmaxf(t1, t2), t3);
                dmax[g] = fmaxf(fmaxf(t4, t5), t6);
            } else if (NMAX == 4) {
                const float t0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), t1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]),
                            t2 = fmaxf(fmaxf(acc[6], acc[15]), dmax[g]);
                dmax[g] = fmaxf(fmaxf(t0, t1), t2);
            } else if (NMAX == 80) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) dmax[g] = fmaxf(fmaxf(dmax[g], acc[i]), acc[i + 1]);
            } else if (NMAX == 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(dmax[g]) : "v"(acc[i]));
            } else {
                dmax[g] = fmaxf(fmaxf(dmax[g], acc[0]), acc[15]);
            }
        }
        asm volatile("" ::: "memory");
    }
    float r = 0.f;
    for (int g = 0; g < NG; ++g) r += dmax[g];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NMAX, int NG, bool NOMFMA = false>
void run(const h8* a, const h8* b, float* out, int blocks_per_cu)
{
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NMAX, NG, NOMFMA>), dim3(blocks), dim3(256), 0, 0, a, b, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NMAX, NG, NOMFMA>), dim3(blocks), dim3(256), 0, 0, a, b, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)iters * NG * blocks_per_cu;      // one wave of each block on every SIMD
    printf("%smax code %d, groups %2d, waves per SIMD %d: %6.1f cycles per step and SIMD at 2.4 GHz\n", NOMFMA ? "no mfma, " : "", NMAX, NG, blocks_per_cu,
           ms * 1e-3 * 2.4e9 / steps_per_simd);
}
int main()
{
    h8 *a, *b; float* out;
    (void)hipMalloc(&a, 64 * 16); (void)hipMalloc(&b, 64 * 16); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    (void)hipMemset(a, 0, 64 * 16); (void)hipMemset(b, 0, 64 * 16);
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 16>(a, b, out, w);
        run<4, 16>(a, b, out, w);
        run<8, 16>(a, b, out, w);
    }
    run<80, 16>(a, b, out, 4);
    run<16, 16>(a, b, out, 4);
    run<8, 16, true>(a, b, out, 4);
    run<80, 16, true>(a, b, out, 4);
    run<16, 16, true>(a, b, out, 4);
    run<8, 16, true>(a, b, out, 1);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
