#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
}
int main() { int* d; hipMalloc(&d, 128*4); hipLaunchKernelGGL(k, 1, 64, 0, 0, d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("shl1:"); for (int i=0;i<66;i+=1) if(i<4||i>59) printf(" [%d]=%d", i, h[i]); printf("\nshr1:"); for (int i=64;i<128;i++) if(i<68||i>123) printf(" [%d]=%d", i-64, h[i]); printf("\n"); }
