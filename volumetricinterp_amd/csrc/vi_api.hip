// Context, memory, model-table and host-staging entry points of libvinterp.so.
#include "vi_common.h"

#include <cstdarg>

static thread_local char g_err[1024] = "";

void vi_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vi_last_error(void) { return g_err; }
extern "C" int vi_abi_version(void) { return VI_ABI_VERSION; }

extern "C" int vi_device_count(int* count)
{
    VI_REQUIRE(count, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return VI_OK;
}

extern "C" int vi_ctx_create(int device, vi_ctx** out)
{
    VI_REQUIRE(out, "null argument");
    *out = nullptr;
    int n = 0;
    VI_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        vi_set_error("vi_ctx_create: device %d out of range (%d visible)", device, n);
        return VI_ERR_INVALID;
    }
    VI_HIP(hipSetDevice(device));
    vi_ctx* c = new vi_ctx();
    c->device = device;
    VI_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    VI_HIP(hipEventCreate(&c->ev0));
    VI_HIP(hipEventCreate(&c->ev1));
    VI_HIP(hipEventCreate(&c->evk0));
    VI_HIP(hipEventCreate(&c->evk1));
    hipDeviceProp_t prop;
    VI_HIP(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    VI_ROCBLAS(rocblas_create_handle(&c->blas));
    VI_ROCBLAS(rocblas_set_stream(c->blas, c->stream));
    VI_ROCBLAS(rocblas_set_pointer_mode(c->blas, rocblas_pointer_mode_host));
    *out = c;
    return VI_OK;
}

extern "C" void vi_ctx_destroy(vi_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ws) (void)hipFree(c->ws);
    if (c->blas) rocblas_destroy_handle(c->blas);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evk0) (void)hipEventDestroy(c->evk0);
    if (c->evk1) (void)hipEventDestroy(c->evk1);
    for (int i = 0; i < vi_ctx::NSOLVE_EV; ++i)
        for (int j = 0; j < 2; ++j)
            if (c->evs[i][j]) (void)hipEventDestroy(c->evs[i][j]);
    if (c->d_rounds) (void)hipFree(c->d_rounds);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int vi_ctx_sync(vi_ctx* c)
{
    VI_REQUIRE(c, "null context");
    VI_HIP(hipStreamSynchronize(c->stream));
    return VI_OK;
}

int vi_ctx_workspace(vi_ctx* c, size_t bytes, void** out)
{
    if (bytes > c->ws_bytes) {
        VI_HIP(hipStreamSynchronize(c->stream));
        if (c->ws) VI_HIP(hipFree(c->ws));
        c->ws = nullptr;
        c->ws_bytes = 0;
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&c->ws, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            vi_set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
            return VI_ERR_NOMEM;
        }
        c->ws_bytes = want;
    }
    *out = c->ws;
    return VI_OK;
}

extern "C" int vi_dmalloc(vi_ctx* c, size_t bytes, void** d_ptr)
{
    VI_REQUIRE(c && d_ptr, "null argument");
    VI_HIP(hipSetDevice(c->device));
    *d_ptr = nullptr;
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 8);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        vi_set_error("vi_dmalloc: %zu bytes: %s", bytes, hipGetErrorString(e));
        return VI_ERR_NOMEM;
    }
    return VI_OK;
}

extern "C" int vi_dfree(vi_ctx* c, void* d_ptr)
{
    VI_REQUIRE(c, "null context");
    if (!d_ptr) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipStreamSynchronize(c->stream));
    VI_HIP(hipFree(d_ptr));
    return VI_OK;
}

extern "C" int vi_h2d(vi_ctx* c, void* d_dst, const void* h_src, size_t bytes)
{
    VI_REQUIRE(c && (bytes == 0 || (d_dst && h_src)), "null argument");
    if (!bytes) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    VI_HIP(hipStreamSynchronize(c->stream));
    return VI_OK;
}

extern "C" int vi_d2h(vi_ctx* c, void* h_dst, const void* d_src, size_t bytes)
{
    VI_REQUIRE(c && (bytes == 0 || (h_dst && d_src)), "null argument");
    if (!bytes) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    VI_HIP(hipStreamSynchronize(c->stream));
    return VI_OK;
}

extern "C" int vi_dmemset(vi_ctx* c, void* d_ptr, int value, size_t bytes)
{
    VI_REQUIRE(c && (bytes == 0 || d_ptr), "null argument");
    if (!bytes) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipMemsetAsync(d_ptr, value, bytes, c->stream));
    return VI_OK;
}

extern "C" int vi_timer_start(vi_ctx* c)
{
    VI_REQUIRE(c, "null context");
    VI_HIP(hipEventRecord(c->ev0, c->stream));
    return VI_OK;
}

extern "C" int vi_timer_stop_ms(vi_ctx* c, double* ms)
{
    VI_REQUIRE(c && ms, "null argument");
    VI_HIP(hipEventRecord(c->ev1, c->stream));
    VI_HIP(hipEventSynchronize(c->ev1));
    float f = 0.f;
    VI_HIP(hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = (double)f;
    return VI_OK;
}

// duration of the evaluation kernel launches of the last vi_eval_f64 call (HIP events recorded on the
// context's stream right around them, excluding the coefficient / hull preparation kernels)
extern "C" int vi_eval_kernel_ms(vi_ctx* c, double* ms)
{
    VI_REQUIRE(c && ms, "null argument");
    VI_REQUIRE(c->evk_valid, "no vi_eval_f64 call has been timed on this context");
    VI_HIP(hipEventSynchronize(c->evk1));
    float f = 0.f;
    VI_HIP(hipEventElapsedTime(&f, c->evk0, c->evk1));
    *ms = (double)f;
    return VI_OK;
}

// Timing of the eigen-solve kernel (k_jacobi_solve), the kernel the fit spends its time in: enable = 1 starts
// (and resets) recording of one HIP event pair per launch on the context's stream; a later call returns the
// number of launches and systems since then and the summed / maximal launch duration of the (at most 128 most
// recent) recorded launches.  enable = -1 only reads.  Any output pointer may be NULL.
extern "C" int vi_solve_timing(vi_ctx* c, int enable, int64_t* launches, int64_t* systems, int64_t* timed,
                               double* total_ms, double* max_ms)
{
    VI_REQUIRE(c, "null context");
    VI_HIP(hipSetDevice(c->device));
    if (c->solve_timing) {
        VI_HIP(hipStreamSynchronize(c->stream));
        const long long n = c->solve_launches < vi_ctx::NSOLVE_EV ? c->solve_launches : vi_ctx::NSOLVE_EV;
        double tot = 0., mx = 0.;
        for (long long i = 0; i < n; ++i) {
            float f = 0.f;
            VI_HIP(hipEventElapsedTime(&f, c->evs[i][0], c->evs[i][1]));
            tot += f;
            if (f > mx) mx = f;
        }
        if (launches) *launches = c->solve_launches;
        if (systems) *systems = c->solve_systems;
        if (timed) *timed = n;
        if (total_ms) *total_ms = tot;
        if (max_ms) *max_ms = mx;
    } else {
        if (launches) *launches = 0;
        if (systems) *systems = 0;
        if (timed) *timed = 0;
        if (total_ms) *total_ms = 0.;
        if (max_ms) *max_ms = 0.;
    }
    if (enable >= 0) {
        if (enable && !c->evs[0][0])
            for (int i = 0; i < vi_ctx::NSOLVE_EV; ++i)
                for (int j = 0; j < 2; ++j) VI_HIP(hipEventCreate(&c->evs[i][j]));
        if (enable && !c->d_rounds) VI_HIP(hipMalloc((void**)&c->d_rounds, sizeof(unsigned long long)));
        if (c->d_rounds) VI_HIP(hipMemsetAsync(c->d_rounds, 0, sizeof(unsigned long long), c->stream));
        c->solve_timing = enable != 0;
        c->solve_launches = 0;
        c->solve_systems = 0;
    }
    return VI_OK;
}

// Jacobi rounds (one round = one pass of the whole LDS-resident matrix through the registers) summed over all systems
// of the launches recorded since vi_solve_timing(enable = 1): the unit the LDS roofline of the kernel is priced in.
extern "C" int vi_solve_rounds(vi_ctx* c, int64_t* rounds)
{
    VI_REQUIRE(c && rounds, "null argument");
    *rounds = 0;
    if (!c->d_rounds) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipStreamSynchronize(c->stream));
    unsigned long long h = 0;
    VI_HIP(hipMemcpy(&h, c->d_rounds, sizeof(h), hipMemcpyDeviceToHost));
    *rounds = (int64_t)h;
    return VI_OK;
}

// ---- model ------------------------------------------------------------------------------------
namespace {
template <class T>
int upload(vi_model* m, const T* h, size_t count, const T** d_out)
{
    void* d = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    VI_HIP(hipMalloc(&d, bytes));
    m->allocs.push_back(d);
    if (count) VI_HIP(hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice));
    *d_out = (const T*)d;
    return VI_OK;
}
}  // namespace

extern "C" void vi_model_destroy(vi_model* m)
{
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->d_coef) (void)hipFree(m->d_coef);
    if (m->d_hull) (void)hipFree(m->d_hull);
    if (m->d_mask) (void)hipFree(m->d_mask);
    delete m;
}

extern "C" int vi_model_create(vi_ctx* c, const vi_model_desc* d, vi_model** out)
{
    VI_REQUIRE(c && d && out, "null argument");
    *out = nullptr;
    VI_HIP(hipSetDevice(c->device));
    vi_model* m = new vi_model();
    m->ctx = c;
    m->kind = d->kind;
    m->N = d->nbasis;
    int rc = VI_OK;
    if (d->kind == VI_MODEL_SPHHARMLAG) {
        if (d->maxk < 1 || d->maxl < 1 || d->nbasis != d->maxk * d->maxl * d->maxl || d->ngroups < 1 || !d->groups ||
            !d->coef_scale) {
            vi_set_error("vi_model_create: inconsistent sphharmlag description");
            delete m;
            return VI_ERR_INVALID;
        }
        SphDev& S = m->sph;
        S.maxk = d->maxk;
        S.maxl = d->maxl;
        S.N = d->nbasis;
        S.ngroups = d->ngroups;
        S.rc = d->rot_cos;
        S.rs = d->rot_sin;
        S.kx = d->rot_kx;
        S.ky = d->rot_ky;
        S.RE = d->earth_radius;
        std::vector<SphGroupDev> hg(d->ngroups);
        for (int g = 0; g < d->ngroups && rc == VI_OK; ++g) {
            const vi_sph_group& G = d->groups[g];
            if (G.nvmax < 0 || !G.pick || !G.c || (G.nterms > 0 && (!G.seed_pref || !G.seed_q))) {
                vi_set_error("vi_model_create: group %d incomplete", g);
                rc = VI_ERR_INVALID;
                break;
            }
            for (int j = 0; j <= G.nvmax + 1; ++j)
                if (G.pick[j] >= d->maxl || (G.pick[j] >= 0 && G.pick[j] > j)) {
                    vi_set_error("vi_model_create: group %d pick[%d]=%d invalid", g, j, G.pick[j]);
                    rc = VI_ERR_INVALID;
                }
            if (rc != VI_OK) break;
            hg[g].v0 = G.v0;
            hg[g].nvmax = G.nvmax;
            if (g == 0) m->nvmax0 = G.nvmax;
            hg[g].nterms = G.nterms;
            const size_t nj = (size_t)G.nvmax + 2;
            if ((rc = upload(m, G.pick, nj, &hg[g].pick)) != VI_OK) break;
            if ((rc = upload(m, G.c, nj * d->maxl, &hg[g].c)) != VI_OK) break;
            if ((rc = upload(m, G.seed_pref, G.nterms ? (size_t)2 * d->maxl : 0, &hg[g].pref)) != VI_OK) break;
            if ((rc = upload(m, G.seed_q, (size_t)2 * d->maxl * G.nterms, &hg[g].q)) != VI_OK) break;
        }
        if (rc == VI_OK) rc = upload(m, hg.data(), hg.size(), &S.groups);
        if (rc == VI_OK) rc = upload(m, d->coef_scale, (size_t)d->maxl * d->maxl, &S.scale);
        if (rc == VI_OK && d->coef_scale1) rc = upload(m, d->coef_scale1, (size_t)d->maxl * d->maxl, &S.scale1);
        if (rc == VI_OK && d->nu) rc = upload(m, d->nu, (size_t)d->maxl, &S.nu);
    } else if (d->kind == VI_MODEL_RADBASFUN) {
        if (d->nbasis < 1 || !d->centers || !(d->eps != 0.0)) {
            vi_set_error("vi_model_create: inconsistent radbasfun description");
            delete m;
            return VI_ERR_INVALID;
        }
        m->rbf.N = d->nbasis;
        m->rbf.inv_eps2 = 1.0 / (d->eps * d->eps);
        rc = upload(m, d->centers, (size_t)3 * d->nbasis, &m->rbf.centers);
    } else {
        vi_set_error("vi_model_create: unknown model kind %d", d->kind);
        rc = VI_ERR_INVALID;
    }
    if (rc != VI_OK) {
        vi_model_destroy(m);
        return rc;
    }
    *out = m;
    return VI_OK;
}

// host-pointer convenience form of vi_eval_f64
extern "C" int vi_eval_f64_host(vi_model* m, int64_t Q, const double* h_lat, const double* h_lon, const double* h_alt,
                                int64_t T, const double* h_C, const double* h_hull_eq, int32_t F, double hull_tol,
                                double* h_out)
{
    VI_REQUIRE(m && h_lat && h_lon && h_alt && h_C && h_out, "null argument");
    VI_REQUIRE(Q >= 0 && T >= 0 && F >= 0, "negative size");
    if (Q == 0 || T == 0) return VI_OK;
    vi_ctx* c = m->ctx;
    VI_HIP(hipSetDevice(c->device));
    const size_t qb = (size_t)Q * sizeof(double);
    double *d_in = nullptr, *d_C = nullptr, *d_hull = nullptr, *d_out = nullptr;
    int rc = VI_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        if (d_in) (void)hipFree(d_in);
        if (d_C) (void)hipFree(d_C);
        if (d_hull) (void)hipFree(d_hull);
        if (d_out) (void)hipFree(d_out);
    };
#define VI_TRY(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            vi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));    \
            cleanup();                                                                            \
            return VI_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)
    VI_TRY(hipMalloc((void**)&d_in, 3 * qb));
    VI_TRY(hipMalloc((void**)&d_C, (size_t)T * m->N * sizeof(double)));
    VI_TRY(hipMalloc((void**)&d_out, (size_t)T * qb));
    if (F > 0) VI_TRY(hipMalloc((void**)&d_hull, (size_t)F * 4 * sizeof(double)));
    VI_TRY(hipMemcpyAsync(d_in, h_lat, qb, hipMemcpyHostToDevice, c->stream));
    VI_TRY(hipMemcpyAsync(d_in + Q, h_lon, qb, hipMemcpyHostToDevice, c->stream));
    VI_TRY(hipMemcpyAsync(d_in + 2 * Q, h_alt, qb, hipMemcpyHostToDevice, c->stream));
    VI_TRY(hipMemcpyAsync(d_C, h_C, (size_t)T * m->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (F > 0) VI_TRY(hipMemcpyAsync(d_hull, h_hull_eq, (size_t)F * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    rc = vi_eval_f64(m, Q, d_in, d_in + Q, d_in + 2 * Q, T, d_C, d_hull, F, hull_tol, d_out);
    if (rc == VI_OK) {
        VI_TRY(hipMemcpyAsync(h_out, d_out, (size_t)T * qb, hipMemcpyDeviceToHost, c->stream));
        VI_TRY(hipStreamSynchronize(c->stream));
    }
#undef VI_TRY
    cleanup();
    return rc;
}
