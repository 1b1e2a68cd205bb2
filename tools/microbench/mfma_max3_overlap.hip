// Does the VALU work after a v_mfma_f32_32x32x16_f16 overlap with the next one?  Four waves per SIMD, sixteen steps per
// iteration, cycles per step and SIMD (2.4 GHz assumed):
//   dep     : the eight v_max3_f32 of a step read the result of the step's own product (what the compiler emits for the loop
//             of k_hull_mask_mx: one accumulator, product -> wait -> maxima -> next product)
//   alien   : eight v_max3_f32 on registers no product writes
//   piped   : two accumulators by hand (inline asm): product g + 1 is issued, then the maxima of product g
// Build: hipcc --offload-arch=gfx950 -O3 mfma_max3_overlap.hip -o mfma_max3_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MAXTREE(acc, d)                                                                                               \
    {                                                                                                                 \
        const float t0 = fmaxf(fmaxf(acc[0], acc[1]), acc[2]), t1 = fmaxf(fmaxf(acc[3], acc[4]), acc[5]),            \
                    t2 = fmaxf(fmaxf(acc[6], acc[7]), acc[8]), t3 = fmaxf(fmaxf(acc[9], acc[10]), acc[11]),          \
                    t4 = fmaxf(fmaxf(acc[12], acc[13]), acc[14]), t5 = fmaxf(fmaxf(acc[15], d), t0),                 \
                    t6 = fmaxf(fmaxf(t1, t2), t3);                                                                    \
        d = fmaxf(fmaxf(t4, t5), t6);                                                                                 \
    }
template <int MODE>
__global__ __launch_bounds__(256) void k(const h8* a, const h8* b, float* out, int iters, long long* clk)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    constexpr int NG = 16;
    const h8 A = a[threadIdx.x & 63];
    h8 B[NG];
    float dmax[NG];
    for (int g = 0; g < NG; ++g) { B[g] = b[(threadIdx.x + g) & 63]; dmax[g] = -3e38f; }
    const f16v zero = {0};
    f16v alien = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B[1], zero, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f16v acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B[g], zero, 0, 0, 0);
                MAXTREE(acc, dmax[g]);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f16v acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B[g], zero, 0, 0, 0);
                asm volatile("" : "+v"(alien));
                MAXTREE(alien, dmax[g]);
                dmax[g] = fmaxf(dmax[g], acc[0]);
            }
        } else {
            f16v p, q;
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(p) : "v"(A), "v"(B[0]));
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(q) : "v"(A), "v"(B[g + 1]));
                asm volatile("s_nop 7\n s_nop 7" : "+v"(p));                       // the product of p is 16+ issue slots old
                MAXTREE(p, dmax[g]);
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(p) : "v"(A), "v"(B[(g + 2) % NG]), "v"(dmax[g]));
                asm volatile("s_nop 7\n s_nop 7" : "+v"(q));
                MAXTREE(q, dmax[g + 1]);
            }
            asm volatile("s_nop 7\n s_nop 7" : "+v"(p));
            dmax[0] = fmaxf(dmax[0], p[0]);
        }
    }
    float r = 0.f;
    for (int g = 0; g < NG; ++g) r += dmax[g];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}
template <int MODE>
void run(const h8* a, const h8* b, float* out, int blocks_per_cu, const char* name)
{
    static long long* clk = nullptr;
    if (!clk) (void)hipMalloc(&clk, 16);
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, a, b, out, 10, clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, a, b, out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long hc[2];
    (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-6s waves per SIMD %d: %6.1f cycles per step and SIMD at 2.4 GHz; shader clock of block 0: %.0f MHz\n", name, blocks_per_cu,
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * blocks_per_cu), 100.0 * hc[0] / hc[1]);
}
int main()
{
    h8 *a, *b; float* out;
    (void)hipMalloc(&a, 64 * 16); (void)hipMalloc(&b, 64 * 16); (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int data = 0; data < 2; ++data) {
    if (data == 0) { (void)hipMemset(a, 0, 64 * 16); (void)hipMemset(b, 0, 64 * 16); printf("operands: zeros\n"); }
    else {
        unsigned short h[512];
        for (int i = 0; i < 512; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x9fff));      // fp16 of mixed sign, 2^-3 ... 2^4
        (void)hipMemcpy(a, h, 1024, hipMemcpyHostToDevice);
        for (int i = 0; i < 512; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x9fff));
        (void)hipMemcpy(b, h, 1024, hipMemcpyHostToDevice);
        printf("operands: random\n");
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<0>(a, b, out, w, "dep");
        run<1>(a, b, out, w, "alien");
        run<2>(a, b, out, w, "piped");
    }
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
