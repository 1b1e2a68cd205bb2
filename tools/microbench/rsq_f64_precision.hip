// Precision of the hardware estimates v_rsq_f64 / v_rcp_f64 (how many Newton steps the rotation set-up needs).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* c0, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    r0[i] = y;
    double e = fma(-v * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    r1[i] = y;
    e = fma(-v * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    r2[i] = y;
    c0[i] = __builtin_amdgcn_rcp(v);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 140) - 100);
    double *x, *r0, *r1, *r2, *c0;
    hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8); hipMalloc(&c0, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(x, r0, r1, r2, c0, n);
    std::vector<double> a(n), b(n), c(n), d(n);
    hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), c0, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)h[i]);
        e0 = fmax(e0, fabs((double)((a[i] - t) / t))); e1 = fmax(e1, fabs((double)((b[i] - t) / t))); e2 = fmax(e2, fabs((double)((c[i] - t) / t)));
        ec = fmax(ec, fabs((double)((d[i] - 1.0L / h[i]) * h[i])));
    }
    printf("v_rsq_f64: max rel error %.3e (2^%.1f); after 1 Newton step %.3e; after 2 %.3e | v_rcp_f64 %.3e (2^%.1f)\n", e0, log2(e0), e1, e2, ec, log2(ec));
    return 0;
}
