// Internal definitions shared by the translation units of libvinterp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vinterp.h"

void vi_set_error(const char* fmt, ...);

#define VI_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            vi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return VI_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define VI_ROCBLAS(call)                                                                      \
    do {                                                                                      \
        rocblas_status s_ = (call);                                                           \
        if (s_ != rocblas_status_success) {                                                   \
            vi_set_error("%s:%d: %s -> rocblas status %d", __FILE__, __LINE__, #call, (int)s_); \
            return VI_ERR_ROCBLAS;                                                            \
        }                                                                                     \
    } while (0)

#define VI_REQUIRE(cond, msg)                                  \
    do {                                                       \
        if (!(cond)) {                                         \
            vi_set_error("%s: %s", __func__, msg);             \
            return VI_ERR_INVALID;                             \
        }                                                      \
    } while (0)

struct vi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    rocblas_handle blas = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evk0 = nullptr, evk1 = nullptr;   // around the dominant kernel of the last vi_eval_f64 call
    bool evk_valid = false;
    bool evk_enabled = false;                    // vi_ctx_set_eval_timing: the event pair costs ~7 us per call
    // grow-only device workspace for the fit entry points
    void* ws = nullptr;
    size_t ws_bytes = 0;
    int n_cu = 256;
    void* rccl_comm = nullptr;   // ncclComm_t, created by vi_rccl_init
    // opt-in timing of the eigen-solve kernel launches (vi_solve_timing): ring of HIP event pairs on the stream
    static constexpr int NSOLVE_EV = 2048;
    bool solve_timing = false;
    hipEvent_t evs[NSOLVE_EV][2] = {};
    long long solve_launches = 0;      // launches recorded since the last reset
    long long solve_systems = 0;       // systems in those launches
    unsigned long long* d_rounds = nullptr;   // device counter: Jacobi rounds (LDS passes) of the recorded launches
    // downloads that do not hold the stream (vi_d2h_side_mark / vi_d2h_side): a second stream and the event it waits for
    hipStream_t side = nullptr;
    hipEvent_t ev_side = nullptr;
};

int vi_ctx_workspace(vi_ctx* ctx, size_t bytes, void** out);

// ---- device-side model tables ----------------------------------------------------------------
struct SphGroupDev {
    double v0;
    int nvmax;
    int nterms;
    const int* pick;        // [nvmax+1]
    const double* c;        // [(nvmax+1) x maxl]
    const double* pref;     // [2 x maxl]
    const double* q;        // [2 x maxl x nterms]
};

struct SphDev {
    int maxk, maxl, N, ngroups;
    double rc, rs, kx, ky, RE;
    const SphGroupDev* groups;
    const double* scale;    // [maxl^2]
    const double* scale1;   // [maxl^2] for degree nu_l + 1 (gradient basis)
    const double* nu;       // [maxl]
};

struct RbfDev {
    int N;
    double inv_eps2;
    const double* centers;  // [N x 3]
};

struct vi_model {
    vi_ctx* ctx = nullptr;
    int kind = 0;
    int N = 0;
    SphDev sph{};
    int nvmax0 = 0;              // nvmax of the first degree group (host copy, for LDS sizing)
    RbfDev rbf{};
    std::vector<void*> allocs;   // device allocations owned by the model
    double* d_coef = nullptr;    // reordered + scaled coefficient staging for vi_eval (grow-only)
    size_t coef_bytes = 0;
    double* d_hull = nullptr;    // internal hull buffer of vi_eval (fp64 facets + fp32 prefilter), grow-only
    size_t hull_bytes = 0;
    unsigned char* d_mask = nullptr;   // inside-hull byte mask of the last vi_eval grid, grow-only
    size_t mask_bytes = 0;
    bool chain_f32 = false;      // evaluate the Legendre degree recurrences in fp32 (vi_model_set_eval_precision)
    // staging of vi_eval_f64_host, grow-only: device coordinates / coefficients / facets / output, second stream + events
    double* h_din = nullptr;     size_t h_din_bytes = 0;
    double* h_dC = nullptr;      size_t h_dC_bytes = 0;
    double* h_dhull = nullptr;   size_t h_dhull_bytes = 0;
    double* h_dout = nullptr;    size_t h_dout_bytes = 0;
    hipStream_t h_stream2 = nullptr;
    hipEvent_t h_ev[2] = {nullptr, nullptr};
    hipEvent_t h_evdown[2] = {nullptr, nullptr};   // "the download out of staging slot s has finished" (h_stream2)
};
