// Operand / result lane maps of v_mfma_f32_32x32x16_f16 and the semantics of v_permlane32_swap, with exact integer data.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_map.hip -o mfma_f16_map ; run: ./mfma_f16_map out.bin
// out.bin: int32 a[64][8], b[64][8], float c[64][16], uint32 swap_in0[64], swap_in1[64], swap_out0[64], swap_out1[64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const int* a, const int* b, float* c, const unsigned* s0, const unsigned* s1, unsigned* o0, unsigned* o1)
{
    const int l = threadIdx.x;
    h8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)(float)a[l * 8 + j]; B[j] = (_Float16)(float)b[l * 8 + j]; }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
    const auto w = __builtin_amdgcn_permlane32_swap(s0[l], s1[l], false, false);
    o0[l] = w[0];
    o1[l] = w[1];
}
int main(int argc, char** argv)
{
    int ha[512], hb[512];
    unsigned hs0[64], hs1[64], ho0[64], ho1[64];
    float hc[1024];
    srand(1);
    for (int i = 0; i < 512; ++i) { ha[i] = rand() % 17 - 8; hb[i] = rand() % 17 - 8; }
    for (int i = 0; i < 64; ++i) { hs0[i] = 1000 + i; hs1[i] = 2000 + i; }
    int *da, *db; float* dc; unsigned *d0, *d1, *e0, *e1;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc);
    hipMalloc(&d0, 256); hipMalloc(&d1, 256); hipMalloc(&e0, 256); hipMalloc(&e1, 256);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(d0, hs0, 256, hipMemcpyHostToDevice); hipMemcpy(d1, hs1, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, d0, d1, e0, e1);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost); hipMemcpy(ho0, e0, 256, hipMemcpyDeviceToHost);
    hipMemcpy(ho1, e1, 256, hipMemcpyDeviceToHost);
    FILE* f = fopen(argc > 1 ? argv[1] : "mfma_f16_map.bin", "wb");
    fwrite(ha, 1, sizeof ha, f); fwrite(hb, 1, sizeof hb, f); fwrite(hc, 1, sizeof hc, f);
    fwrite(hs0, 1, 256, f); fwrite(hs1, 1, 256, f); fwrite(ho0, 1, 256, f); fwrite(ho1, 1, 256, f);
    fclose(f);
    // the map of the guide: C[row][col], col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); A / B: k = 8 (lane >> 5) + j
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            int sum = 0;
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 8; ++j) sum += ha[(row + 32 * h) * 8 + j] * hb[(col + 32 * h) * 8 + j];
            if ((float)sum != hc[l * 16 + r]) ++bad;
        }
    printf("mfma_f32_32x32x16_f16 against the bf16 map of the guide: %d of 1024 results differ\n", bad);
    printf("permlane32_swap(x = 1000 + lane, y = 2000 + lane): w[0] lanes 0, 31, 32, 63 = %u %u %u %u; w[1] = %u %u %u %u\n",
           ho0[0], ho0[31], ho0[32], ho0[63], ho1[0], ho1[31], ho1[32], ho1[63]);
    return 0;
}
