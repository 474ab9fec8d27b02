// Internal definitions shared by the translation units of libhelios_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/helios_hip.h"

struct hx_context {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    char err[512];
    unsigned long long* diag;  // device record behind hx_diag_read: HX_DIAG_* slots
};

// slots of hx_context::diag (the layout of hx_diag in include/helios_hip.h)
enum { HX_DIAG_NEG_DOWN = 0, HX_DIAG_NEG_UP = 1, HX_DIAG_G_LIMITED = 2, HX_DIAG_RO_REBIN = 3, HX_DIAG_ENERGY = 4, HX_DIAG_RO_FIXUP = 5,
       HX_DIAG_SLOTS = 8 };

// post-pass counters used by the debug = 1 variants of the per-stage and fused solvers (context.hip)
extern "C" int hx_internal_count_negative(hx_context* ctx, const double* a, size_t n, int slot);
extern "C" int hx_internal_count_abs_ge(hx_context* ctx, const double* a, size_t n, double limit, int slot);

inline int hx_fail(hx_context* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HX_HIP(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return hx_fail((ctx), -(int)e_, "%s failed: %s (%s:%d)", #call,                \
                           hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

// after a kernel launch
#define HX_LAUNCH_CHECK(ctx) HX_HIP(ctx, hipGetLastError())

#define HX_REQUIRE(ctx, cond, code, msg)                                                   \
    do {                                                                                   \
        if (!(cond)) return hx_fail((ctx), (code), "%s: %s", __func__, (msg));             \
    } while (0)

inline int hx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// device-side physical constants: source/kernels.cu:36-41 (values are physical constants)
#define HX_PI 3.141592653589793
#define HX_HCONST 6.62607004e-27
#define HX_CSPEED 29979245800.0
#define HX_KBOLTZMANN 1.38064852e-16
#define HX_STEFANBOLTZMANN 5.6703669999999995e-5
#define HX_AMU 1.6605390666e-24

// internal cross-translation-unit helpers (not exported through include/helios_hip.h)
extern "C" int hx_internal_planck_star_row(hx_context* ctx, double* row, const double* lambda_edge,
                                           const double* deltalambda, int nwave, double Tstar);
extern "C" int hx_internal_fdir_noniso(hx_context* ctx, double* F_dir_wg, double* Fc_dir_wg,
                                       const double* star, int star_stride,
                                       const double* delta_tau_wg_upper,
                                       const double* delta_tau_wg_lower, const double* z_lay,
                                       double mu_star, double R_planet, double R_star, double a,
                                       int dir_beam, int geom_zenith_corr, int ninterface, int nbin,
                                       int ny);
// the matrix solver of the per-stage path with the fused loop's per-column "done" flag (stage_matrix.hip)
extern "C" int hx_internal_fband_matrix_iso(
    hx_context* ctx, const int* skip, double* F_down_wg, double* F_up_wg, const double* F_dir_wg,
    const double* planckband_lay, const double* w_0, const double* M_term, const double* N_term, const double* P_term,
    const double* G_plus, const double* G_minus, const double* g_0_tot_lay, double* alpha, double* beta,
    double* source_term_down, double* source_term_up, double* c_prime, double* d_prime, const int* scat_trigger,
    const double* trans_wg, const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a,
    int numinterfaces, int nbin, double f_factor, double mu_star, int ny, double epsi, int dir_beam, int clouds,
    int scat_corr, int debug, double i2s_transition);
extern "C" int hx_internal_fband_matrix_noniso(
    hx_context* ctx, const int* skip, double* F_down_wg, double* F_up_wg, double* Fc_down_wg, double* Fc_up_wg,
    const double* F_dir_wg, const double* Fc_dir_wg, const double* planckband_lay, const double* planckband_int,
    const double* w_0_upper, const double* w_0_lower, const double* delta_tau_wg_upper,
    const double* delta_tau_wg_lower, const double* delta_tau_all_clouds_upper,
    const double* delta_tau_all_clouds_lower, const double* M_upper, const double* M_lower, const double* N_upper,
    const double* N_lower, const double* P_upper, const double* P_lower, const double* G_plus_upper,
    const double* G_plus_lower, const double* G_minus_upper, const double* G_minus_lower, const double* g_0_tot_lay,
    const double* g_0_tot_int, double* alpha, double* beta, double* source_term_down, double* source_term_up,
    double* c_prime, double* d_prime, const int* scat_trigger, const double* trans_wg_upper,
    const double* trans_wg_lower, const double* surf_albedo, double g_0, int singlewalk, double Rstar, double a,
    int numinterfaces, int nbin, double f_factor, double mu_star, int ny, double epsi, double delta_tau_limit,
    int dir_beam, int clouds, int scat_corr, int debug, double i2s_transition);
extern "C" int hx_internal_fdir_noniso_batch(hx_context* ctx, double* F_dir_wg, double* Fc_dir_wg, const double* star,
                                             const double* delta_tau_wg_upper, const double* delta_tau_wg_lower,
                                             const double* z_lay, const hx_rt_column* colpar, const int* done,
                                             int ncol, int dir_beam, int geom_zenith_corr, int ninterface, int nbin,
                                             int ny);
