// Temperature steps (reference source/kernels.cu:2606-2884), executed by ONE
// workgroup per atmosphere column.  Shared by the per-stage entry points and the fused path.
#pragma once
#include "two_stream.h"

namespace hx {

struct RadTempArgs {
    const double* F_down_tot;
    const double* F_net;
    double* F_net_diff;
    double* tlay;
    const double* play;
    const double* pint;
    int* abrt;
    double* T_store;
    double* deltat_prefactor;
    const double* F_add_heat_lay;
    const double* F_add_heat_sum;
    double* F_smooth;
    double* F_smooth_sum;
    const double* c_p_lay;
    const double* meanmolmass_lay;
    int* conv_count;  // optional: number of converged layers (incl. ghost layer) of this column
    int itervalue, foreplay;
    double g;
    int nlayer;
    double physical_tstep, local_limit;
    int adapt_interval, smooth, dim, step;
    double F_intern;
    int no_atmo;
};

// smoothing flux from the temperatures BEFORE the step and its prefix sum.  The reference does this
// inside the per-layer threads with a block-local barrier under divergent control flow (racy,
// SURVEY.md Q11); here it is two barrier-separated phases over the whole column.
__device__ inline void smoothing_flux(double* F_smooth, double* F_smooth_sum, const double* tlay,
                                      const double* play, int nlayer, int tid, int nthr) {
    for (int i = tid; i < nlayer; i += nthr) {
        double t_mid = tlay[i];
        if (play[i] < 1e6 && i < nlayer - 1 && i > 0) t_mid = (tlay[i - 1] + tlay[i + 1]) / 2.0;
        F_smooth[i] = pow(t_mid - tlay[i], 7.0);
    }
    __syncthreads();
    for (int i = tid; i < nlayer; i += nthr) {
        double s = 0.0;
        for (int j = 0; j <= i; j++) s += F_smooth[j];
        F_smooth_sum[i] = s;
    }
    __syncthreads();
}

__device__ inline void rad_temp_step(const RadTempArgs& a, int tid, int nthr) {
    __shared__ int s_count;
    if (tid == 0) s_count = 0;
    if (a.smooth == 1) smoothing_flux(a.F_smooth, a.F_smooth_sum, a.tlay, a.play, a.nlayer, tid, nthr);
    __syncthreads();
    const int L = a.nlayer;
    const double F_toa = a.F_down_tot[L];
    int local_count = 0;
    for (int i = tid; i <= L; i += nthr) {
        double dF, delta_T = 0.0;
        if (i < L) {
            const double d = a.F_net[i] - a.F_net[i + 1] + a.F_add_heat_lay[i];
            a.F_net_diff[i] = d;
            dF = d + a.F_smooth[i];
        } else {
            dF = a.F_intern - a.F_net[0];
            if (fabs(a.F_intern - a.F_net[1]) / (F_toa + a.F_intern) > 0.5 * a.local_limit)
                dF = a.F_intern - a.F_net[1];
        }
        const double T_old = a.tlay[i];
        if (a.physical_tstep == 0) {
            double pref = a.deltat_prefactor[i];
            if (a.itervalue == a.foreplay) pref = 1e0;
            if (a.itervalue == 10000) pref = 1e-1;
            if (dF != 0) {
                const double delta_t = pref * a.play[0] / pow(fabs(dF), 0.9);
                delta_T = dF / (a.pint[0] - a.pint[1]) * delta_t;
            }
            if (fabs(delta_T) > 500.0) delta_T = 500.0 * dF / fabs(dF);
            if (a.itervalue % a.adapt_interval == 0) a.T_store[i] = T_old;
            if (a.itervalue % a.adapt_interval == a.adapt_interval - 1) {
                if (fabs(T_old - a.T_store[i]) < a.adapt_interval / 2.0 * fabs(delta_T))
                    pref /= 1.5;
                else
                    pref *= 1.1;
            }
            a.deltat_prefactor[i] = pref;
        } else {
            const int j = i < L ? i : 0;
            delta_T = a.g / (a.c_p_lay[j] / (a.meanmolmass_lay[j] / HX_AMU)) * dF /
                      (a.pint[j] - a.pint[j + 1]) * a.physical_tstep;
        }
        double T = T_old + delta_T;
        if (a.no_atmo == 1 && i != L) T = 1.001;
        a.tlay[i] = dmin(dmax(T, 1.001), a.dim * a.step - 1.001);
        bool ok;
        if (i < L)
            ok = fabs(a.F_intern + a.F_add_heat_sum[i] + a.F_smooth_sum[i] - a.F_net[i + 1]) /
                     (F_toa + a.F_intern) < a.local_limit;
        else
            ok = fabs(a.F_intern - a.F_net[0]) / (F_toa + a.F_intern) < a.local_limit;
        a.abrt[i] = ok ? 1 : 0;
        local_count += ok ? 1 : 0;
    }
    if (a.conv_count) {
        if (local_count) atomicAdd(&s_count, local_count);
        __syncthreads();
        if (tid == 0) *a.conv_count = s_count;
    }
}

struct ConvTempArgs {
    const double* F_net;
    double* F_net_diff;
    double* tlay;
    const double* play;
    const double* pint;
    double* T_store;
    double* deltat_prefactor;
    const int* marked_red;
    const double* F_add_heat_lay;
    double* F_smooth;
    double* F_smooth_sum;
    int nlayer, itervalue, adapt_interval, smooth;
    double F_intern;
};

__device__ inline void conv_temp_step(const ConvTempArgs& a, int tid, int nthr) {
    if (a.smooth == 1) smoothing_flux(a.F_smooth, a.F_smooth_sum, a.tlay, a.play, a.nlayer, tid, nthr);
    __syncthreads();
    const int L = a.nlayer;
    for (int i = tid; i <= L; i += nthr) {
        double dF;
        if (i < L) {
            const double d = a.F_net[i] - a.F_net[i + 1] + a.F_add_heat_lay[i];
            a.F_net_diff[i] = d;
            dF = d + a.F_smooth[i];
        } else {
            dF = a.F_intern - a.F_net[0];
            for (int j = 0; j < L; j++)
                if (a.marked_red[j] == 1) {
                    dF = a.F_intern - a.F_net[j + 1];
                    break;
                }
        }
        double pref = a.deltat_prefactor[i];
        if (a.itervalue == 0) pref = 1e-2;
        if (a.itervalue == 6000) pref = 1e-3;
        double delta_T = 0.0;
        if (dF != 0) {
            const double delta_t = pref * a.play[0] / pow(fabs(dF), 0.5);
            delta_T = dF / (a.pint[0] - a.pint[1]) * delta_t;
        }
        if (fabs(delta_T) > 20.0) delta_T = 20.0 * dF / fabs(dF);
        const double T_old = a.tlay[i];
        if (a.itervalue % a.adapt_interval == 0) a.T_store[i] = T_old;
        if (a.itervalue % a.adapt_interval == a.adapt_interval - 1) {
            if (fabs(T_old - a.T_store[i]) < a.adapt_interval / 2.0 * fabs(delta_T))
                pref /= 1.5;
            else
                pref *= 1.1;
        }
        a.deltat_prefactor[i] = pref;
        a.tlay[i] = dmax(T_old + delta_T, 1.001);
    }
}

}  // namespace hx
