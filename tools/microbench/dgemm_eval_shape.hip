// Rate of the library GEMM at the shape of a basis-resident evaluation: out(Q x T) = Y(Q x N) C(N x T), column-major,
// Q = 2^24 grid points, N = 144 basis functions, T timesteps per call.   hipcc -O2 --offload-arch=gfx950 dgemm_eval_shape.hip -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != 0) { printf("fail %s line %d\n", #x, __LINE__); return 1; } } while (0)
__global__ void fill(double* p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v * (double)((i * 2654435761u) & 1023) / 1024.0; }
int main(int argc, char** argv)
{
    const size_t Q = argc > 1 ? atol(argv[1]) : (size_t)1 << 24;
    const int N = argc > 2 ? atoi(argv[2]) : 144;
    rocblas_handle h; CK(rocblas_create_handle(&h));
    double *Y, *C, *O;
    const int Tmax = 256;
    CK(hipMalloc(&Y, Q * N * 8)); CK(hipMalloc(&C, (size_t)N * Tmax * 8)); CK(hipMalloc(&O, Q * Tmax * 8));
    fill<<<(Q * N + 255) / 256, 256>>>(Y, Q * N, 1.0); fill<<<(N * Tmax + 255) / 256, 256>>>(C, (size_t)N * Tmax, 0.5);
    CK(hipDeviceSynchronize());
    const double one = 1.0, zero = 0.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int T : {16, 32, 64, 128, 256}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            CK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, (rocblas_int)Q, T, N, &one, Y, (rocblas_int)Q, C, N, &zero, O, (rocblas_int)Q));
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("Q %zu N %d T %4d: %8.3f ms  %.1f TFLOP/s  %.3e point-timesteps/s  (Y read %.1f GB, out %.1f GB -> %.0f GB/s)\n", Q, N, T, ms,
                                 2.0 * Q * N * T / ms / 1e9, (double)Q * T / (ms * 1e-3), Q * N * 8 / 1e9, Q * T * 8.0 / 1e9, (Q * N * 8 + Q * T * 8.0) / 1e9 / (ms * 1e-3));
        }
    }
    return 0;
}
