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
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
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

// A download beside the stream.  vi_d2h_side_mark (caller's thread, after the launches that produce the data): an event on the
// context's stream.  vi_d2h_side (any host thread, blocking): a second stream waits for that event, copies, and is drained -
// the context's stream meanwhile runs what is queued after the mark (FitEngine: the covariances of a batch, 166 MB per 1000
// records, come down while the consistency guard's solves run; the source buffer must stay untouched until the call returns).
extern "C" int vi_d2h_side_mark(vi_ctx* c)
{
    VI_REQUIRE(c, "null context");
    VI_HIP(hipSetDevice(c->device));
    if (!c->side) VI_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    if (!c->ev_side) VI_HIP(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    VI_HIP(hipEventRecord(c->ev_side, c->stream));
    return VI_OK;
}

extern "C" int vi_d2h_side(vi_ctx* c, void* h_dst, const void* d_src, size_t bytes)
{
    VI_REQUIRE(c && (bytes == 0 || (h_dst && d_src)), "null argument");
    VI_REQUIRE(c->side && c->ev_side, "vi_d2h_side without vi_d2h_side_mark");
    if (!bytes) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipStreamWaitEvent(c->side, c->ev_side, 0));
    VI_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->side));
    VI_HIP(hipStreamSynchronize(c->side));
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

// HIP events around the evaluation kernel launches of every vi_eval_f64 / vi_eval_resident_f64 call on this context: off by
// default (the two records cost a call of 0.2 ms about 7 us), on for whoever reads vi_eval_kernel_ms
extern "C" int vi_ctx_set_eval_timing(vi_ctx* c, int32_t on)
{
    VI_REQUIRE(c, "null argument");
    c->evk_enabled = on != 0;
    if (!c->evk_enabled) c->evk_valid = false;
    return VI_OK;
}

// duration of the evaluation kernel launches of the last vi_eval_f64 call (HIP events recorded on the
// context's stream right around them, excluding the coefficient / hull preparation kernels)
extern "C" int vi_eval_kernel_ms(vi_ctx* c, double* ms)
{
    VI_REQUIRE(c && ms, "null argument");
    VI_REQUIRE(c->evk_enabled, "evaluation timing is off on this context: vi_ctx_set_eval_timing(ctx, 1) first");
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
    if (m->h_din) (void)hipFree(m->h_din);
    if (m->h_dC) (void)hipFree(m->h_dC);
    if (m->h_dhull) (void)hipFree(m->h_dhull);
    if (m->h_dout) (void)hipFree(m->h_dout);
    if (m->h_stream2) (void)hipStreamDestroy(m->h_stream2);
    for (int i = 0; i < 2; ++i) {
        if (m->h_ev[i]) (void)hipEventDestroy(m->h_ev[i]);
        if (m->h_evdown[i]) (void)hipEventDestroy(m->h_evdown[i]);
    }
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

// host-pointer form of vi_eval_f64 (what Estimate.__call__ uses: the reference hands over NumPy arrays).
// The device staging buffers live on the model and only grow (the first version paid four hipMalloc / hipFree pairs
// per call).  The grid is processed in chunks: the coordinates of chunk c+1 go up on the context's stream while the
// densities of chunk c come down on a second stream, so the two directions of the link overlap - fully when the caller's
// arrays are page-locked (vi_host_alloc), as far as the runtime's own staging allows when they are pageable.
namespace {
int grow(double** p, size_t* have, size_t need)
{
    if (need <= *have) return VI_OK;
    if (*p) VI_HIP(hipFree(*p));
    *p = nullptr;
    *have = 0;
    VI_HIP(hipMalloc((void**)p, need));
    *have = need;
    return VI_OK;
}
}  // namespace

extern "C" int vi_eval_f64_host(vi_model* m, int64_t Q, const double* h_lat, const double* h_lon, const double* h_alt,
                                int64_t T, const double* h_C, const double* h_hull_eq, int32_t F, double hull_tol,
                                double* h_out)
{
    VI_REQUIRE(m && h_lat && h_lon && h_alt && h_C && h_out, "null argument");
    VI_REQUIRE(Q >= 0 && T >= 0 && F >= 0, "negative size");
    if (Q == 0 || T == 0) return VI_OK;
    vi_ctx* c = m->ctx;
    VI_HIP(hipSetDevice(c->device));
    // chunk: large enough to keep the evaluation kernel efficient (>= 2048 workgroups), small enough for several chunks
    // to be in flight at the BASELINE grid sizes; the whole per-chunk output (T rows) must fit the staging buffer
    int64_t chunk = (int64_t)1 << 19;
    while (chunk > 4096 && (size_t)T * chunk * sizeof(double) > ((size_t)1 << 30)) chunk >>= 1;
    if (chunk > Q) chunk = Q;
    const int64_t nchunk = (Q + chunk - 1) / chunk;
    VI_HIP(hipStreamSynchronize(c->stream));          // nothing of an earlier call may still use the staging buffers
    int rc = VI_OK;
    if ((rc = grow(&m->h_din, &m->h_din_bytes, (size_t)2 * 3 * chunk * sizeof(double))) != VI_OK) return rc;
    if ((rc = grow(&m->h_dC, &m->h_dC_bytes, (size_t)T * m->N * sizeof(double))) != VI_OK) return rc;
    if ((rc = grow(&m->h_dout, &m->h_dout_bytes, (size_t)2 * T * chunk * sizeof(double))) != VI_OK) return rc;
    if (F > 0 && (rc = grow(&m->h_dhull, &m->h_dhull_bytes, (size_t)F * 4 * sizeof(double))) != VI_OK) return rc;
    if (!m->h_stream2) {
        VI_HIP(hipStreamCreateWithFlags(&m->h_stream2, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) VI_HIP(hipEventCreateWithFlags(&m->h_ev[i], hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) VI_HIP(hipEventCreateWithFlags(&m->h_evdown[i], hipEventDisableTiming));
    }
    // From here on every failure leaves through ONE exit that waits for both streams: copies into the caller's arrays and
    // out of the staging buffers may be in flight, and control must not return to Python while they are.
#define VI_TRY(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            vi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            rc = VI_ERR_HIP;                                                                  \
            goto done;                                                                        \
        }                                                                                     \
    } while (0)
    {
        bool used[2] = {false, false};                // "a download out of slot s has been issued" (stream2, h_evdown[s])
        rc = VI_OK;
        VI_TRY(hipMemcpyAsync(m->h_dC, h_C, (size_t)T * m->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (F > 0)
            VI_TRY(hipMemcpyAsync(m->h_dhull, h_hull_eq, (size_t)F * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        for (int64_t k = 0; k < nchunk; ++k) {
            const int s = (int)(k & 1);
            const int64_t q0 = k * chunk;
            const int64_t qc = (Q - q0) < chunk ? (Q - q0) : chunk;
            double* din = m->h_din + (size_t)s * 3 * chunk;
            double* dout = m->h_dout + (size_t)s * T * chunk;
            // slot s was last used by chunk k - 2: its download must be over before the kernel of chunk k overwrites dout
            if (used[s]) VI_TRY(hipStreamWaitEvent(c->stream, m->h_evdown[s], 0));
            VI_TRY(hipMemcpyAsync(din, h_lat + q0, (size_t)qc * sizeof(double), hipMemcpyHostToDevice, c->stream));
            VI_TRY(hipMemcpyAsync(din + chunk, h_lon + q0, (size_t)qc * sizeof(double), hipMemcpyHostToDevice, c->stream));
            VI_TRY(hipMemcpyAsync(din + 2 * chunk, h_alt + q0, (size_t)qc * sizeof(double), hipMemcpyHostToDevice, c->stream));
            rc = vi_eval_f64(m, qc, din, din + chunk, din + 2 * chunk, T, m->h_dC, F > 0 ? m->h_dhull : nullptr, F, hull_tol, dout);
            if (rc != VI_OK) goto done;
            VI_TRY(hipEventRecord(m->h_ev[s], c->stream));
            VI_TRY(hipStreamWaitEvent(m->h_stream2, m->h_ev[s], 0));
            // device rows have length qc (the kernel wrote out[t * qc + q]); host rows have length Q
            VI_TRY(hipMemcpy2DAsync(h_out + q0, (size_t)Q * sizeof(double), dout, (size_t)qc * sizeof(double),
                                    (size_t)qc * sizeof(double), (size_t)T, hipMemcpyDeviceToHost, m->h_stream2));
            VI_TRY(hipEventRecord(m->h_evdown[s], m->h_stream2));
            used[s] = true;
        }
    }
done:
#undef VI_TRY
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(m->h_stream2);
    return rc;
}

// Free / total device memory of the context's GPU (tests: repeated fits must not leak contexts or workspaces).
extern "C" int vi_mem_info(vi_ctx* c, size_t* free_bytes, size_t* total_bytes)
{
    VI_REQUIRE(c && free_bytes && total_bytes, "null argument");
    VI_HIP(hipSetDevice(c->device));
    VI_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return VI_OK;
}

// Page-locked host memory for callers that want the two directions of vi_eval_f64_host to overlap completely.
extern "C" int vi_host_alloc(size_t bytes, void** out)
{
    VI_REQUIRE(out, "null argument");
    *out = nullptr;
    VI_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return VI_OK;
}

extern "C" int vi_host_free(void* p)
{
    if (p) VI_HIP(hipHostFree(p));
    return VI_OK;
}
