// On-the-fly opacity mixing of the fused refresh: the species loop of
// calculate_total_opacity_and_scat_cross_sections_from_species (computation.py:1454-1501) and the host steps around it
// (host_functions.py:874-959, :1050-1056) as THREE launches for all columns, levels and species of a batch.
//
//   k_rt_species_prep   per (column, level): mean molecular mass from the mixing ratios, fractional (T, log10 P)
//                       table indices with the species clamp [0, n-1] (kernels.cu:3233, :3238), and the factor
//                       vmr * mass / mu of every absorber (:3293)
//   k_rt_mix_species    per (column, level, bin) -- one wavefront at a time: the running k-distribution stays in 20 lanes'
//                       registers while the absorbers are folded in one after the other (correlated-k for the first
//                       species and for CIA pairs, random overlap via ro::mix otherwise).  Each absorber's 20 coefficients
//                       are interpolated from the four table corners as they are needed (opac_species_interpol, :3209-3259:
//                       the reference writes them to a 161 MB array per species and reads them back); the next absorber's
//                       corners are in flight while the present one is mixed.  opac_wg_lay / opac_wg_int are written once.
//   k_rt_scat_species   per (column, level, bin): sum of vmr * sigma over the scattering species (add_to_mixed_scat,
//                       :3444-3459; water vapour through calc_h2o_scat, :3404-3440)
//
// Columns whose loop has ended (done[c]) are skipped on the device: no host round trip inside a refresh.
#pragma once
#include "random_overlap.h"
#include "random_overlap_lean.h"
#include "two_stream.h"

namespace hx {

struct SpeciesDev {
    const double* pretab;      // [t][p][x][y] flat (reference order) or null
    const double* scat_cross;  // [x] or null
    const double* vmr_tab;     // mixing ratio on the opacity tables' (T, P) grid, one table per column: [column][p + npress * t],
                               // or null: profile given by the host
    double weight;             // molar weight
    int absorbing, scattering, is_h2o, ro, in_mu, pad;
};

// Mixing ratio of a tabulated species at one level: bilinear in (T, log10 P) on the nodes of the opacity tables, clamped to
// the table edges -- interpolate_grid_to_lay_or_int (host_functions.py:904-910: scipy's RectBivariateSpline(kx = 1,
// ky = 1) evaluated point by point), in the term order of helios_amd/host_functions.py.  The nodes are searched (no
// uniform-grid assumption, as in scipy).
__device__ __forceinline__ int last_node_not_above(const double* grid, int n, double v, bool log_nodes) {
    int lo = 0, hi = n - 1;  // largest k in [0, n-2] with node(k) <= v (v is clamped into the grid)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const double node = log_nodes ? log10(grid[mid]) : grid[mid];
        if (node <= v) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ double vmr_from_table(const double* tab, double T, double P, const double* ktemp, int ntemp,
                                                 const double* kpress, int npress) {
    const double t = dmin(ktemp[ntemp - 1], dmax(ktemp[0], T));
    const double p = dmin(log10(kpress[npress - 1]), dmax(log10(kpress[0]), log10(P)));
    const int it = last_node_not_above(ktemp, ntemp, t, false), ip = last_node_not_above(kpress, npress, p, true);
    const double t0 = ktemp[it], t1 = ktemp[it + 1], p0 = log10(kpress[ip]), p1 = log10(kpress[ip + 1]);
    const double ft = (t - t0) / (t1 - t0), fp = (p - p0) / (p1 - p0);
    const double v00 = tab[ip + npress * it], v10 = tab[ip + npress * (it + 1)], v01 = tab[ip + 1 + npress * it],
                 v11 = tab[ip + 1 + npress * (it + 1)];
    return v00 * (1 - ft) * (1 - fp) + v10 * ft * (1 - fp) + v01 * (1 - ft) * fp + v11 * ft * fp;
}

struct MixArgs {
    int X, Y, L, I, C, S, ntemp, npress, nabs;
    int carry_on;          // 0: the mix starts from zero; 1: from what opac_wg_* holds -- the launch folds in the NEXT block of at most
                           // MIX_MAX_ABSORBERS absorbers of a longer species list (round 6)
    const SpeciesDev* sp;  // [S]
    const int* abs_list;   // [nabs] indices of the absorbing species of this launch, ascending
    const double *T_lay, *T_int, *p_lay, *p_int;  // column strides L+1, I, L, I
    const double *vmr_lay, *vmr_int;              // [C][S][I]
    const double *ktemp, *kpress, *gauss_w, *gauss_y, *wave;
    double *mmm_lay, *mmm_int;          // [C][I]
    TPIndex *tp_lay, *tp_int;           // [C][I]
    double *fac_lay, *fac_int;          // [C][I][S]
    double *opac_wg_lay, *opac_wg_int;  // [C][Y X I]
    double *scat_lay, *scat_int;        // [C][X I]
    const int* done;
    unsigned long long* diag;
};

__global__ void k_rt_species_prep(MixArgs a) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.I || a.done[col]) return;
    for (int pass = 0; pass < 2; pass++) {
        const bool lay = pass == 0;
        if (lay && i >= a.L) continue;
        double* vmr = const_cast<double*>(lay ? a.vmr_lay : a.vmr_int) + (size_t)col * a.S * a.I;
        const double T = lay ? a.T_lay[(size_t)col * (a.L + 1) + i] : a.T_int[(size_t)col * a.I + i];
        const double P = lay ? a.p_lay[(size_t)col * a.L + i] : a.p_int[(size_t)col * a.I + i];
        // calculate_vmr_for_all_species (host_functions.py:874-901) for the species that come with a (T, P) table: the
        // profile follows the temperatures of this refresh, on the device
        for (int s = 0; s < a.S; s++)
            if (a.sp[s].vmr_tab)
                vmr[(size_t)s * a.I + i] = vmr_from_table(a.sp[s].vmr_tab + (size_t)col * a.ntemp * a.npress, T, P, a.ktemp, a.ntemp,
                                                          a.kpress, a.npress);
        double num = 0.0, tot = 0.0;  // host_functions.py:927-959
        for (int s = 0; s < a.S; s++)
            if (a.sp[s].in_mu) {
                num += vmr[(size_t)s * a.I + i] * a.sp[s].weight;
                tot += vmr[(size_t)s * a.I + i];
            }
        const double mmm = num / tot * HX_AMU;
        (lay ? a.mmm_lay : a.mmm_int)[(size_t)col * a.I + i] = mmm;
        (lay ? a.tp_lay : a.tp_int)[(size_t)col * a.I + i] = locate_tp(T, P, a.ktemp, a.ntemp, a.kpress, a.npress, false, false);
        double* fac = (lay ? a.fac_lay : a.fac_int) + ((size_t)col * a.I + i) * a.S;
        for (int s = 0; s < a.S; s++) {
            const double mass = a.sp[s].weight * HX_AMU;
            fac[s] = vmr[(size_t)s * a.I + i] * mass / mmm;  // (vmr * mass) / mu, then times kappa (:3293)
        }
    }
}

// mean molecular mass of the layers only (the convection loop refreshes it ahead of the adjustment, computation.py:1030-1036)
// (the mixing ratios of tabulated species are brought up to the present layer temperatures first, :1035)
__global__ void k_rt_mmm_from_vmr(const SpeciesDev* __restrict__ sp, int S, double* __restrict__ vmr_lay,
                                  double* __restrict__ mmm_lay, int L, int I, const double* __restrict__ T_lay,
                                  const double* __restrict__ p_lay, const double* __restrict__ ktemp, int ntemp,
                                  const double* __restrict__ kpress, int npress) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    double* vmr = vmr_lay + (size_t)col * S * I;
    for (int s = 0; s < S; s++)
        if (sp[s].vmr_tab)
            vmr[(size_t)s * I + i] = vmr_from_table(sp[s].vmr_tab + (size_t)col * ntemp * npress, T_lay[(size_t)col * (L + 1) + i],
                                                    p_lay[(size_t)col * L + i], ktemp, ntemp, kpress, npress);
    double num = 0.0, tot = 0.0;
    for (int s = 0; s < S; s++)
        if (sp[s].in_mu) {
            num += vmr[(size_t)s * I + i] * sp[s].weight;
            tot += vmr[(size_t)s * I + i];
        }
    mmm_lay[(size_t)col * I + i] = num / tot * HX_AMU;
}

constexpr int MIX_MAX_ABSORBERS = 48;  // LDS images of the species list: 1 KB next to the mixing images; a longer list takes
                                       // one launch per block of 48 (MixArgs::carry_on)

// HX_MIX_LEAN = 1 (default): rol::mix (random_overlap_lean.h) -- 7.9 KB of LDS and at most 96 VGPRs, five wavefronts per SIMD;
// 0: ro::mix as until round 5 (10.2 KB, 128 VGPRs, four), kept for the same-box A/B (tools/ab_mix_lean.sh)
#ifndef HX_MIX_LEAN
#define HX_MIX_LEAN 1
#endif
#if HX_MIX_LEAN
namespace mixro = rol;
#define HX_MIX_WAVES_PER_EU 5
#else
namespace mixro = ro;
#define HX_MIX_WAVES_PER_EU 4
#endif

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HX_MIX_WAVES_PER_EU))) k_rt_mix_species(MixArgs a) {
    __shared__ mixro::Shared sh;
    // what the species loop needs per absorber, staged once per wavefront: table base, correlated-k flag, and -- per
    // level -- the factor vmr * mass / mu.  (Read straight from the argument block these were chains of dependent
    // global loads inside the loop: the scalar registers are all taken, so the compiler fetched them through the
    // vector memory path and waited for each; same-box A/B 68.6 -> 67.7 ms per refresh at config 3.)
    __shared__ const double* s_tab[MIX_MAX_ABSORBERS];
    __shared__ double s_fac[MIX_MAX_ABSORBERS];
    __shared__ int s_info[MIX_MAX_ABSORBERS];  // species index << 1 | random-overlap flag
    // the level's bilinear weights (pup - p, p - pdown, tup - t, t - tdown): wave-uniform fp64 values that the compiler
    // otherwise carries in ten VGPRs through the species loop and, at the 128-register cap, parks in scratch around it --
    // 16 bytes per lane and point written and read back (2.1 GB of the 2.5 GB WRITE_SIZE of round 2's launch)
    __shared__ double s_blend[4];
    const int lane = threadIdx.x;
#if HX_MIX_LEAN
    const rol::LaneConst ln = rol::init(sh, lane, a.gauss_w, a.gauss_y);
#else
    ro::Lane ln;
    ro::init(sh, ln, lane, a.gauss_w, a.gauss_y);
#endif
    const int nabs = a.nabs;
    if (lane < nabs) {
        const int s = a.abs_list[lane];
        s_tab[lane] = a.sp[s].pretab;
        s_info[lane] = s << 1 | ((s == 0 || a.sp[s].ro == 0) ? 0 : 1);  // correlated-k: first species, CIA (:3302-3310)
    }
    ro::Counters cnt;
    const long long nlev = a.L + a.I, per_col = nlev * a.X, npair = per_col * a.C;
    const long long chunk = (npair + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = min(npair, p0 + chunk);
    const size_t nc = (size_t)a.Y * a.X, st_p = nc, st_t = nc * a.npress;
    long long lev_cur = -1;
    int tcase = 0;  // bit 0: two pressure nodes, bit 1: two temperature nodes (the four cases of :617-644)
    TPIndex tp = {};
    double* out_level = nullptr;
    size_t pl_dd = 0, pl_ud = 0, pl_du = 0, pl_uu = 0;  // table planes of the level's four (T, P) corners: wave-uniform
    bool skip = false;
    // Three groups of 20 lanes fetch three absorbers at a time: lane 20 g + y holds Gauss point y of absorber kb + g.
    // The next triple's table corners are requested before the present triple is mixed, so that even a run of negligible
    // absorbers (a few dozen instructions each) does not wait for memory.
    const int grp = lane / ro::NY, y = lane - grp * ro::NY;
    const bool loader = grp < 3 && y < a.Y;
    // (bin, level, column) of the run's first point by division, then counted up: no 64-bit division per point
    long long cl = p0 < p1 ? p0 / a.X : 0;  // column * nlev + level
    int x = (int)(p0 - cl * a.X) - 1;
    int col = (int)(cl / nlev), lev = (int)(cl - (long long)col * nlev);
    for (long long pair = p0; pair < p1; pair++) {
        if (++x == a.X) {
            x = 0;
            cl++;
            if (++lev == (int)nlev) {
                lev = 0;
                col++;
            }
        }
        if (cl != lev_cur) {  // wave-uniform: a new level (or column)
            lev_cur = cl;
            const bool lay = lev < a.L;
            const int i = lay ? lev : lev - a.L;
            skip = __builtin_amdgcn_readfirstlane(a.done[col]) != 0;
            tp = (lay ? a.tp_lay : a.tp_int)[(size_t)col * a.I + i];
            const double* fac = (lay ? a.fac_lay : a.fac_int) + ((size_t)col * a.I + i) * a.S;
            out_level = (lay ? a.opac_wg_lay : a.opac_wg_int) + (size_t)col * nc * a.I + nc * i;
            // the level's table nodes are the same in every lane: as scalars (the loads above came through the vector path), so
            // that the four plane offsets are scalar arithmetic and live in scalar registers
            const int pdown = __builtin_amdgcn_readfirstlane(tp.pdown), pup = __builtin_amdgcn_readfirstlane(tp.pup);
            const int tdown = __builtin_amdgcn_readfirstlane(tp.tdown), tup = __builtin_amdgcn_readfirstlane(tp.tup);
            pl_dd = st_p * pdown + st_t * tdown;
            pl_ud = st_p * pup + st_t * tdown;
            pl_du = st_p * pdown + st_t * tup;
            pl_uu = st_p * pup + st_t * tup;
            ro::sync();
            int ls = lane;
            asm volatile("" : "+v"(ls));  // addresses derived from the lane id are rebuilt here, once per level, instead of
                                          // occupying a register (at the cap: a scratch slot) through the whole kernel
            if (ls < nabs) s_fac[ls] = fac[s_info[ls] >> 1];
            if (lane == 0) {
                s_blend[0] = tp.pup - tp.p; s_blend[1] = tp.p - tp.pdown;
                s_blend[2] = tp.tup - tp.t; s_blend[3] = tp.t - tp.tdown;
            }
            tcase = (pdown != pup ? 1 : 0) | (tdown != tup ? 2 : 0);
            ro::sync();
        }
        if (skip) continue;
        const unsigned off = (unsigned)(a.Y * x + y);  // inside a table plane: the same for all corners of all species
        double c_dd = 0.0, c_ud = 0.0, c_du = 0.0, c_uu = 0.0;
        int gl = grp;
        asm volatile("" : "+v"(gl));  // the LDS address of this lane's table pointer is rebuilt per point (no register, no scratch slot)
        if (loader && grp < nabs) {
            const double* tab = s_tab[gl];
            c_dd = (tab + pl_dd)[off]; c_ud = (tab + pl_ud)[off]; c_du = (tab + pl_du)[off]; c_uu = (tab + pl_uu)[off];
        }
        double mixv = 0.0;  // nullify_opac_scat_arrays (host_functions.py:1050-1056)
        if (a.carry_on && lane < a.Y) mixv = out_level[off];   // (wave-uniform flag: a species list of more than 48 absorbers)
        for (int kb = 0; kb < nabs; kb += 3) {
            // blend_tp(..., species = true) with the level's weights re-read from LDS (same products in the same order)
            asm volatile("" ::: "memory");  // the weights are not to be carried in registers across the mixes
            double raw_mine = c_dd;
            if (tcase == 3)
                raw_mine = c_dd * s_blend[0] * s_blend[2] + c_ud * s_blend[1] * s_blend[2] + c_du * s_blend[0] * s_blend[3] +
                           c_uu * s_blend[1] * s_blend[3];
            else if (tcase == 1) raw_mine = c_dd * s_blend[0] + c_ud * s_blend[1];
            else if (tcase == 2) raw_mine = c_du * s_blend[3] + c_dd * s_blend[2];
            if (loader && kb + 3 + grp < nabs) {
                const double* tab = s_tab[kb + 3 + grp];
                c_dd = (tab + pl_dd)[off]; c_ud = (tab + pl_ud)[off]; c_du = (tab + pl_du)[off]; c_uu = (tab + pl_uu)[off];
            }
            for (int j = 0; j < 3 && kb + j < nabs; j++) {
                const double raw = ro::shfl((ro::NY * j + y) << 2, raw_mine);  // lanes 0..19: from group j
                const double add = s_fac[kb + j] * raw;
                if ((s_info[kb + j] & 1) == 0) mixv += add;
                else mixv = mixro::mix(sh, ln, lane, mixv, add, cnt);
            }
        }
        if (lane < a.Y) out_level[off] = mixv;
    }
    ro::flush(cnt, lane, a.diag);
}

__global__ void __launch_bounds__(256) k_rt_scat_species(MixArgs a) {
    const int col = blockIdx.z, lev = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.X || a.done[col]) return;
    const bool lay = lev < a.L;
    const int i = lay ? lev : lev - a.L;
    const double* vmr = (lay ? a.vmr_lay : a.vmr_int) + (size_t)col * a.S * a.I;
    const double T = lay ? a.T_lay[(size_t)col * (a.L + 1) + i] : a.T_int[(size_t)col * a.I + i];
    const double P = lay ? a.p_lay[(size_t)col * a.L + i] : a.p_int[(size_t)col * a.I + i];
    double sum = 0.0;
    for (int s = 0; s < a.S; s++) {
        if (!a.sp[s].scattering) continue;
        const double f = vmr[(size_t)s * a.I + i];
        const double sigma = a.sp[s].is_h2o ? h2o_rayleigh_cross(T, P, f, a.wave[x], a.sp[s].weight * HX_AMU)
                                            : a.sp[s].scat_cross[x];
        sum += f * sigma;
    }
    (lay ? a.scat_lay : a.scat_int)[(size_t)col * a.X * a.I + x + (size_t)a.X * i] = sum;
}

}  // namespace hx
