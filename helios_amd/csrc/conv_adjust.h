// Convective adjustment on the device: the reference does this on the host between two
// kernel launches of every convection-loop iteration (source/host_functions.py:337-635, :251-286; driver
// source/computation.py:992-1174).  Here one workgroup per column runs it, so that the fused path needs no
// host round trip inside the convection loop either.
//
// The logic is sequential over at most a few hundred layers and is executed by thread 0 exactly in the
// order of helios_amd/host_functions.py (which tests/test_host_golden.py pins to the reference's Python);
// only the pressure-ratio powers -- independent of temperature -- are tabulated by all threads first:
//     lim+(i) = (T[i] * f1p[i]) * f2p[i]   adiabat through layer i, kappa * (1 + 1e-6)   (conv_check)
//     lim-(i) = (T[i] * f1m[i]) * f2m[i]   the same with kappa * (1 - 1e-6)              (mark_convective_layers)
//     up[i]   = (p_lay[i]/p_int[i])^kappa_int[i],  nxt[i] = (p_int[i+1]/p_lay[i])^kappa_lay[i]   (conv_correct)
#pragma once
#include "hx_common.h"

namespace hx {

struct ConvTables {  // all in LDS, carved by conv_stage_in
    double *f1p, *f2p, *f1m, *f2m, *up, *nxt;
    double* fac;                        // scratch of conv_correct
    double surf_p, surf_m;
    int* in_zone;                       // scratch: flagged layers, index i + 1 (the surface is i = -1)
    int *starts, *ends;
};

struct ConvColumn {  // pointers of ONE column
    double* T;                           // [L+1], surface last
    const double *p_lay, *p_int;         // [L], [L+1]
    const double *kappa_lay, *kappa_int; // [L], [L+1]
    const double *c_p, *mmm;             // [L]
    const double *F_add_heat_sum, *F_smooth_sum;  // [L]
    const double *F_down_tot, *F_up_tot, *F_net;  // [L+1]
    int *conv_unstable, *conv_layer, *marked_red; // [L+1]
    int L, itervalue;
    double F_intern, T_star, dampara;    // dampara <= 0: automatic
    double rad_convergence_limit;
};

// LDS demand in bytes for a column of L layers
__host__ __device__ inline size_t conv_smem_bytes(int L) {
    return (size_t)(17 * (L + 2)) * sizeof(double) + (size_t)(5 * (L + 2)) * sizeof(int);
}

// All threads.  Copies the column's small arrays into LDS (the sequential walk of thread 0 would otherwise pay a
// global-memory round trip per access), carves the tables and fills them; `c` then points into LDS, `g` keeps the
// global pointers for conv_stage_out.  Ends with a barrier.
__device__ inline void conv_stage_in(ConvColumn& c, ConvColumn& g, ConvTables& t, double* smem, int tid, int nthr) {
    g = c;
    const int L = c.L, n1 = L + 2;
    double* d = smem;
    t.f1p = d; d += n1; t.f2p = d; d += n1; t.f1m = d; d += n1; t.f2m = d; d += n1;
    t.up = d; d += n1; t.nxt = d; d += n1; t.fac = d; d += n1;
    double* T = d; d += n1;
    double* p_lay = d; d += n1; double* p_int = d; d += n1; double* c_p = d; d += n1; double* mmm = d; d += n1;
    double* fah = d; d += n1; double* fss = d; d += n1; double* fd = d; d += n1; double* fu = d; d += n1;
    double* fn = d; d += n1;
    int* ip = reinterpret_cast<int*>(d);
    int* unstable = ip; ip += n1; int* layer = ip; ip += n1; int* red = ip; ip += n1;
    t.in_zone = ip; ip += n1; t.starts = ip; ip += n1 / 2 + 1; t.ends = ip;
    for (int i = tid; i <= L; i += nthr) {
        T[i] = g.T[i];
        p_int[i] = g.p_int[i];
        fd[i] = g.F_down_tot[i];
        fu[i] = g.F_up_tot[i];
        fn[i] = g.F_net[i];
        unstable[i] = g.conv_unstable[i];
        layer[i] = g.conv_layer[i];
        red[i] = g.marked_red[i];
        if (i < L) {
            p_lay[i] = g.p_lay[i];
            c_p[i] = g.c_p[i];
            mmm[i] = g.mmm[i];
            fah[i] = g.F_add_heat_sum[i];
            fss[i] = g.F_smooth_sum[i];
        }
    }
    for (int i = tid; i < L; i += nthr) {
        t.up[i] = pow(g.p_lay[i] / g.p_int[i], g.kappa_int[i]);
        t.nxt[i] = pow(g.p_int[i + 1] / g.p_lay[i], g.kappa_lay[i]);
        if (i < L - 1) {
            const double r1 = g.p_int[i + 1] / g.p_lay[i], r2 = g.p_lay[i + 1] / g.p_int[i + 1];
            t.f1p[i] = pow(r1, g.kappa_lay[i] * (1 + 1e-6));
            t.f2p[i] = pow(r2, g.kappa_int[i + 1] * (1 + 1e-6));
            t.f1m[i] = pow(r1, g.kappa_lay[i] * (1 - 1e-6));
            t.f2m[i] = pow(r2, g.kappa_int[i + 1] * (1 - 1e-6));
        }
    }
    if (tid == 0) {
        t.surf_p = pow(g.p_lay[0] / g.p_int[0], g.kappa_int[0] * (1 + 1e-6));
        t.surf_m = pow(g.p_lay[0] / g.p_int[0], g.kappa_int[0] * (1 - 1e-6));
    }
    c.T = T; c.p_lay = p_lay; c.p_int = p_int; c.c_p = c_p; c.mmm = mmm;
    c.F_add_heat_sum = fah; c.F_smooth_sum = fss; c.F_down_tot = fd; c.F_up_tot = fu; c.F_net = fn;
    c.conv_unstable = unstable; c.conv_layer = layer; c.marked_red = red;
    __syncthreads();
}

// All threads, after a barrier: profile and flags back to global memory
__device__ inline void conv_stage_out(const ConvColumn& c, const ConvColumn& g, int tid, int nthr) {
    for (int i = tid; i <= c.L; i += nthr) {
        g.T[i] = c.T[i];
        g.conv_unstable[i] = c.conv_unstable[i];
        g.conv_layer[i] = c.conv_layer[i];
        g.marked_red[i] = c.marked_red[i];
    }
}

// ---- everything below: one thread -------------------------------------------------------------------------

// host_functions.py:337-365; returns the number of flagged entries
__device__ inline int conv_check(const ConvColumn& c, const ConvTables& t) {
    const int L = c.L;
    int n = 0;
    for (int i = 0; i <= L; i++) c.conv_unstable[i] = 0;
    for (int i = 0; i < L - 1; i++) {
        if (c.p_lay[i] <= 1e1) break;  // the top of the atmosphere is left alone
        if (c.T[i + 1] < (c.T[i] * t.f1p[i]) * t.f2p[i]) {
            c.conv_unstable[i] = 1;
            c.conv_unstable[i + 1] = 1;
        }
    }
    if (c.T[0] < c.T[L] * t.surf_p) {
        c.conv_unstable[L] = 1;
        c.conv_unstable[0] = 1;
    }
    for (int i = 0; i <= L; i++) n += c.conv_unstable[i];
    return n;
}

// host_functions.py:585-635: radiative gaps thinner than one scale height between two zones are closed
__device__ inline void conv_stitch_holes(const ConvColumn& c, ConvTables& t) {
    const int L = c.L;
    int ns = 0, ne = 0;
    if (c.conv_layer[L] == 1) {
        t.starts[ns++] = -1;
        if (c.conv_layer[0] == 0) t.ends[ne++] = -1;
    }
    for (int i = 0; i < L; i++) {
        if (c.conv_layer[i] != 1) continue;
        const int below = i > 0 ? c.conv_layer[i - 1] : c.conv_layer[L];
        if (below == 0) t.starts[ns++] = i;
        if (i == L - 1 || c.conv_layer[i + 1] == 0) t.ends[ne++] = i;
    }
    if (ns != ne) return;  // the reference aborts here; cannot happen for consistent flags
    for (int n = 0; n + 1 < ns; n++) {
        const double p_top = c.p_lay[t.starts[n + 1]];
        const double p_bot = t.ends[n] != -1 ? c.p_lay[t.ends[n]] : c.p_int[0];
        if (p_top / p_bot > 1 / 2.718281828459045)
            for (int m = t.ends[n] + 1; m < t.starts[n + 1]; m++) c.conv_layer[m] = 1;
    }
}

// host_functions.py:545-582
__device__ inline void conv_mark_layers(const ConvColumn& c, ConvTables& t, int stitching) {
    const int L = c.L;
    c.conv_layer[L] = 0;
    c.conv_layer[0] = 0;
    for (int i = 0; i < L - 1; i++) {
        if (c.p_lay[i] <= 1e1) break;
        if (c.T[i + 1] < (c.T[i] * t.f1m[i]) * t.f2m[i]) {
            c.conv_layer[i] = 1;
            c.conv_layer[i + 1] = 1;
        } else {
            c.conv_layer[i + 1] = 0;
        }
    }
    for (int i = 0; i < L - 1; i++)  // no temperature kinks at the top edge of a zone
        if (c.T[i + 1] > c.T[i]) c.conv_layer[i] = 0;
    if (c.T[0] < c.T[L] * t.surf_m) {
        c.conv_layer[L] = 1;
        c.conv_layer[0] = 1;
    }
    if (stitching == 1 && c.itervalue > 5000) conv_stitch_holes(c, t);
}

// contiguous runs of flagged layers, the surface "ghost layer" (index L) counted as layer -1
__device__ inline int conv_zones(const ConvColumn& c, ConvTables& t) {
    const int L = c.L;
    for (int i = 0; i < L; i++) t.in_zone[i + 1] = (c.conv_unstable[i] == 1 || c.conv_layer[i] == 1) ? 1 : 0;
    t.in_zone[0] = (c.conv_unstable[L] == 1 || c.conv_layer[L] == 1) ? 1 : 0;
    t.in_zone[L + 1] = 0;
    int ns = 0, ne = 0;
    for (int i = -1; i < L; i++) {
        if (!t.in_zone[i + 1]) continue;
        if (i == -1 || !t.in_zone[i]) t.starts[ns++] = i;
        if (!t.in_zone[i + 2]) t.ends[ne++] = i;
    }
    return ns == ne ? ns : 0;
}

// host_functions.py:368-506: every zone goes onto the adiabat of its enthalpy-conserving mean potential temperature
__device__ inline void conv_correct_zones(const ConvColumn& c, ConvTables& t, int fudging, int nz) {
    const int L = c.L;
    for (int n = 0; n < nz; n++) {
        double fudge = 1.0;
        if (fudging == 1) {
            // flux test at an interface inside the radiative zone above zone n (or well above the top zone)
            int test = 0;
            for (int m = n; m < nz; m++) {
                if (m != nz - 1) {
                    const double p_top = c.p_lay[t.starts[m + 1]];
                    const double p_bot = t.ends[m] != -1 ? c.p_lay[t.ends[m]] : c.p_int[0];
                    if (p_top / p_bot < 1 / 2.718281828459045) {
                        test = (int)((t.ends[m] + t.starts[m + 1]) / 2.0);
                        break;
                    }
                } else {
                    test = (int)(0.8 * t.ends[m] + 0.2 * L);  // ninterface - 1 = L
                }
            }
            double dampara = c.dampara;
            if (!(dampara > 0)) dampara = c.T_star > 10 ? (n < nz - 1 ? 0.5 : 4.0) : 8.0;
            const int below = test - 1 >= 0 ? test - 1 : test - 1 + L;  // numpy wraps a negative index
            const double ratio = (c.F_intern + c.F_add_heat_sum[below] + c.F_smooth_sum[below] + c.F_down_tot[test]) /
                                 c.F_up_tot[test];
            const double f = pow(ratio, 1.0 / dampara);
            const double lo = f > 0.99 ? f : 0.99;  // Python's max(0.99, f): NaN -> 0.99
            fudge = lo < 1.01 ? lo : 1.01;
        }
        const int a = t.starts[n] > 0 ? t.starts[n] : 0, b = t.ends[n] > 0 ? t.ends[n] : 0;
        double num = 0.0, den = 0.0, chain = 1.0;
        for (int i = a; i <= b; i++) {
            const double cp_mu = c.c_p[i] / c.mmm[i], dp = c.p_int[i] - c.p_int[i + 1];
            const double fac = chain * t.up[i];
            t.fac[i] = fac;
            num += cp_mu * c.T[i] * dp;
            den += fac * cp_mu * dp;
            chain *= t.up[i] * t.nxt[i];
        }
        const double theta = num / den * fudge;
        for (int i = a; i <= b; i++) c.T[i] = theta * t.fac[i];
        if (t.starts[n] == -1) c.T[L] = theta;
    }
}

__device__ inline void conv_correct(const ConvColumn& c, ConvTables& t, int fudging) {
    conv_correct_zones(c, t, fudging, conv_zones(c, t));
}

// host_functions.py:509-542
__device__ inline void convective_adjustment(const ConvColumn& c, ConvTables& t) {
    int unstable = conv_check(c, t);
    int guard = 0;
    while (unstable > 0 && guard++ < 100000) {
        conv_mark_layers(c, t, 0);
        conv_correct(c, t, 0);
        unstable = conv_check(c, t);
    }
    conv_mark_layers(c, t, 1);
    conv_correct(c, t, 1);
}

// ---- workgroup-cooperative forms (all threads call them; they contain barriers) -------------------------------
// The element-wise parts of conv_check / conv_mark_layers are evaluated by all threads; what the reference's
// sequential loops leave behind is reproduced exactly:
//   check: unstable[j] = cond+[j-1] | cond+[j]                      (cond[i] for i < lim, lim = the loop's break index)
//   mark : layer[0] = cond-[0]; layer[j] = cond-[j-1] | cond-[j] (1 <= j < lim); layer[lim] = cond-[lim-1];
//          entries above keep their old value; then the kink rule, then the surface rule.
struct ConvShared {
    int lim;       // number of layer pairs the check / mark loops visit before p_lay <= 10
    int count;
};

// the first layer pair at p_lay <= 10 ends the check / mark loops of the reference (a `break`): the smallest such index
__device__ inline void conv_find_lim(const ConvColumn& c, ConvShared& sh, int tid, int nthr) {
    if (tid == 0) sh.lim = c.L - 1 < 0 ? 0 : c.L - 1;
    __syncthreads();
    int mine = 1 << 30;
    for (int i = tid; i < c.L - 1; i += nthr)
        if (c.p_lay[i] <= 1e1) { mine = i; break; }  // ascending i per thread: its first hit is its smallest
    if (mine < (1 << 30)) atomicMin(&sh.lim, mine);
    __syncthreads();
}

// conv_zones by the whole workgroup: the flags by all threads, the (few) zone boundaries compacted by the first
// wavefront with ballots, in ascending order as the sequential walk finds them.  Returns the number of zones (uniform).
__device__ inline int conv_zones_wg(const ConvColumn& c, ConvTables& t, ConvShared& sh, int tid, int nthr) {
    const int L = c.L;
    for (int i = tid; i < L; i += nthr) t.in_zone[i + 1] = (c.conv_unstable[i] == 1 || c.conv_layer[i] == 1) ? 1 : 0;
    if (tid == 0) {
        t.in_zone[0] = (c.conv_unstable[L] == 1 || c.conv_layer[L] == 1) ? 1 : 0;
        t.in_zone[L + 1] = 0;
    }
    __syncthreads();
    if (tid < 64) {
        int ns = 0, ne = 0;
        for (int base = -1; base < L; base += 64) {  // i = base + lane runs over -1 .. L-1
            const int i = base + tid;
            const bool in = i < L && t.in_zone[i + 1] != 0;
            const bool st = in && (i == -1 || !t.in_zone[i]);
            const bool en = in && !t.in_zone[i + 2];
            const unsigned long long ms = __ballot(st), me = __ballot(en);
            const unsigned long long below = tid == 0 ? 0ull : (~0ull >> (64 - tid));
            if (st) t.starts[ns + __popcll(ms & below)] = i;
            if (en) t.ends[ne + __popcll(me & below)] = i;
            ns += __popcll(ms);
            ne += __popcll(me);
        }
        if (tid == 0) sh.count = ns == ne ? ns : 0;
    }
    __syncthreads();
    const int nz = sh.count;
    __syncthreads();
    return nz;
}

__device__ inline int conv_check_wg(const ConvColumn& c, ConvTables& t, ConvShared& sh, int tid, int nthr) {
    const int L = c.L, lim = sh.lim;
    int* cond = t.in_zone;
    for (int i = tid; i <= L; i += nthr) cond[i] = (i < lim && c.T[i + 1] < (c.T[i] * t.f1p[i]) * t.f2p[i]) ? 1 : 0;
    if (tid == 0) sh.count = 0;
    __syncthreads();
    const int surf = c.T[0] < c.T[L] * t.surf_p ? 1 : 0;
    int mine = 0;
    for (int j = tid; j <= L; j += nthr) {
        int u = 0;
        if (j < L) u = cond[j] | (j > 0 ? cond[j - 1] : 0);
        if (surf && (j == 0 || j == L)) u = 1;
        c.conv_unstable[j] = u;
        mine += u;
    }
    if (mine) atomicAdd(&sh.count, mine);
    __syncthreads();
    const int n = sh.count;
    __syncthreads();
    return n;
}

__device__ inline void conv_mark_layers_wg(const ConvColumn& c, ConvTables& t, ConvShared& sh, int stitching, int tid,
                                           int nthr) {
    const int L = c.L, lim = sh.lim;
    int* cond = t.in_zone;
    for (int i = tid; i <= L; i += nthr) cond[i] = (i < lim && c.T[i + 1] < (c.T[i] * t.f1m[i]) * t.f2m[i]) ? 1 : 0;
    __syncthreads();
    const int surf = c.T[0] < c.T[L] * t.surf_m ? 1 : 0;
    for (int j = tid; j <= L; j += nthr) {
        int v = c.conv_layer[j];
        if (j == L || j == 0) v = 0;
        if (lim >= 1) {
            if (j == 0) v = cond[0];
            else if (j < lim) v = cond[j - 1] | cond[j];
            else if (j == lim) v = cond[lim - 1];
        }
        if (j < L - 1 && c.T[j + 1] > c.T[j]) v = 0;   // no temperature kinks at the top edge of a zone
        if (surf && (j == 0 || j == L)) v = 1;
        c.conv_layer[j] = v;
    }
    __syncthreads();
    if (stitching == 1 && c.itervalue > 5000) {
        if (tid == 0) conv_stitch_holes(c, t);
        __syncthreads();
    }
}

__device__ inline void convective_adjustment_wg(const ConvColumn& c, ConvTables& t, ConvShared& sh, int tid, int nthr) {
    conv_find_lim(c, sh, tid, nthr);
    int unstable = conv_check_wg(c, t, sh, tid, nthr);
    int guard = 0;
    while (unstable > 0 && guard++ < 100000) {
        conv_mark_layers_wg(c, t, sh, 0, tid, nthr);
        const int nz = conv_zones_wg(c, t, sh, tid, nthr);
        if (tid == 0) conv_correct_zones(c, t, 0, nz);
        __syncthreads();
        unstable = conv_check_wg(c, t, sh, tid, nthr);
    }
    conv_mark_layers_wg(c, t, sh, 1, tid, nthr);
    const int nz = conv_zones_wg(c, t, sh, tid, nthr);
    if (tid == 0) conv_correct_zones(c, t, 1, nz);
    __syncthreads();
}

// check_for_radiative_eq by all threads; returns the criterion (uniform)
__device__ inline int conv_radiative_eq_wg(const ConvColumn& c, ConvShared& sh, int* s_convective, int tid, int nthr) {
    const int L = c.L;
    const double norm = c.F_down_tot[L] + c.F_intern;
    if (tid == 0) { sh.count = 0; *s_convective = 0; }
    __syncthreads();
    int conv = 0, ok = 0;
    for (int i = tid; i <= L; i += nthr) {
        int red = 0;
        conv += c.conv_layer[i];
        if (c.conv_layer[i] == 0) {
            const double dF = i < L ? fabs(c.F_intern + c.F_add_heat_sum[i] + c.F_smooth_sum[i] - c.F_net[i + 1])
                                    : fabs(c.F_intern - c.F_net[0]);
            if (dF < c.rad_convergence_limit * norm) ok++; else red = 1;
        }
        c.marked_red[i] = red;
    }
    if (ok) atomicAdd(&sh.count, ok);
    if (conv) atomicAdd(s_convective, conv);
    __syncthreads();
    return sh.count == (L + 1) - *s_convective ? 1 : 0;
}

// host_functions.py:251-286: local radiative equilibrium of the non-convective layers; fills marked_red
__device__ inline int conv_radiative_eq(const ConvColumn& c) {
    const int L = c.L;
    const double norm = c.F_down_tot[L] + c.F_intern;
    int converged = 0, convective = 0;
    for (int i = 0; i <= L; i++) {
        c.marked_red[i] = 0;
        convective += c.conv_layer[i];
        if (c.conv_layer[i] != 0) continue;
        const double dF = i < L ? fabs(c.F_intern + c.F_add_heat_sum[i] + c.F_smooth_sum[i] - c.F_net[i + 1])
                                : fabs(c.F_intern - c.F_net[0]);
        if (dF < c.rad_convergence_limit * norm)
            converged++;
        else
            c.marked_red[i] = 1;
    }
    return converged == (L + 1) - convective ? 1 : 0;
}

}  // namespace hx
