// Fused fast path (include/helios_hip.h section 4): device-resident state for a batch of
// atmosphere columns, one call per opacity refresh and one per iteration.
//
// Replaces the body of Compute.radiation_loop (reference source/computation.py:851-984):
//   refresh  = :860-879  (opacities, mean molecular mass, [species mixing], transmission, dz, beam)
//   step     = :856-857 + :880-888 + :926-932 (T_int, Planck, 3*scat+1 sweeps, integrate, T step)
#include "rt_kernels.h"
#include "rt_species.h"

#include <algorithm>
#include <cmath>

using namespace hx;

namespace {

int rt_fail(hx_rt* rt, int code, const char* msg) { return hx_fail(rt->ctx, code, "%s", msg); }

template <class T>
int dev_alloc(hx_rt* rt, T** p, size_t n, bool zero = true) {
    void* q = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess)
        return hx_fail(rt->ctx, -(int)e, "hipMalloc of %zu bytes failed: %s", n * sizeof(T),
                       hipGetErrorString(e));
    if (zero) {
        e = hipMemsetAsync(q, 0, n * sizeof(T), rt->ctx->stream);
        if (e != hipSuccess) return hx_fail(rt->ctx, -(int)e, "hipMemset failed");
    }
    rt->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}

#define RT_ALLOC(ptr, n)                        \
    do {                                        \
        int rc_ = dev_alloc(rt, &(ptr), (n));   \
        if (rc_) return rc_;                    \
    } while (0)

int h2d(hx_rt* rt, void* dst, const void* src, size_t bytes) {
    return hx_h2d(rt->ctx, dst, src, bytes);
}

// ypb = Gauss points per pass (a divisor of ny), nxb = bins per workgroup: fill the lanes of a workgroup of at most
// max_threads
void choose_workgroup(int Y, int X, int max_threads, TileGeom& g) {
    g.ypb = 0;
    double best_util = -1.0;
    for (int v = 1; v <= Y; v++) {
        if (Y % v != 0 || v * g.k > max_threads) continue;
        const int nx = std::min(std::max(1, max_threads / (v * g.k)), X);
        const int lanes = nx * v * g.k;
        const double util = (double)lanes / (((lanes + 63) / 64) * 64);
        if (util >= best_util) {
            best_util = util;
            g.ypb = v;
            g.nxb = nx;
        }
    }
}

// k_rt_flux<ROWS, K> instantiations that do not hold their register image in 256 VGPRs at two wavefronts per SIMD (K = 0 stands
// for k = 8): 14 rows spill 2-3 registers (k != 16), 15 rows and more run as single-wavefront workgroups with the overflow in AGPRs
// (round 6, rt_kernels.h flux_one_wave) -- chosen only where no other lane count avoids them
bool flux_variant_spills(int rows, int k) { return rows >= 15 || (rows == 14 && k != 16); }

bool choose_geometry(int H, int Y, int X, int C, int dir_beam, int scat_corr, TileGeom& g, int matrix = 0) {
    // lanes per spectral point: the fewest padded nodes, with the kernels whose scans are written for a compile-time
    // lane count (k = 16, 32, 64) preferred -- measured at 10 000 bins: 50 layers 0.172 ms (k = 16, 12 % padding) against
    // 0.210 ms (k = 8, 4 %), 60 layers 0.194 against 0.256 ms
    // Instantiations that keep part of their register image in scratch are not chosen while another lane count avoids
    // it: k_rt_flux<15, *> spills 15-22 VGPRs, <16, *> 32-37, <14, K != 16> 2-3 (tools/code_object_notes.py;
    // tests/test_abi.py reads the same notes), and scratch round trips inside the Gauss-group loop cost this kernel a
    // quarter of its time in round 2.  Every column of up to 416 layers has a spill-free tiling
    // (k = 64, 13 rows); beyond that only k = 64 with 14-16 rows exists.  Measured: profiles/r03_geometry_ab.txt.
    int best_k = 0, best_rows = 0, best_cost = 1 << 30;
    int force_k = 0;
    if (const char* e = getenv("HELIOS_RT_K")) force_k = atoi(e);  // tuning knob
    for (int pass = 0; pass < 2 && !best_k; pass++)
        for (int k = 8; k <= 64; k <<= 1) {
            int rows = (H + k - 1) / k;
            if (rows > 16) {
                // columns beyond 16 rows x 64 lanes (512 layers, 1024 isothermal ones): 20, 24, 28 or 32 rows on 64 lanes -- up
                // to 1024 layers (2048 isothermal).  These instantiations keep part of their register image in scratch
                // (k_rt_flux<32, 64>: 1.2 KB per lane); still one launch per iteration on compact planes instead of the
                // per-stage path's sixteen arrays per stage (round 6)
                if (k != 64 || rows > 32) continue;
                rows = (rows + 3) / 4 * 4;
            }
            if (force_k && k != force_k) continue;
            if (pass == 0 && flux_variant_spills(rows, k)) continue;
            int cost = k * rows * (k >= 16 ? 100 : 125);
            // (Until round 4 a tiling of more than ten rows paid a 30 % penalty here when the direct beam adds its two planes: the
            // beam planes were then a second, dependent request per tile.  Since their rows are requested in groups that are
            // in flight together -- k_rt_flux, HX_BEAM_GROUP -- the tiling with the fewest padded slots wins with the beam as
            // without: same-box A/B of round 5, k_rt_flux per launch: 10 000 x 100 with beam 0.465 ms (k = 32, 7 rows) ->
            // 0.434 ms (k = 16, 13 rows); 30 000 x 200 with beam and I2S 3.12-3.40 ms (k = 64, 7 rows) against 3.08-3.13 ms
            // (k = 32, 13 rows), its coefficient kernel 6.4 -> 5.85 ms.  profiles/r05_flux_tilings_beam.txt.)
            (void)dir_beam;
            if (cost < best_cost) {
                best_cost = cost;
                best_k = k;
                best_rows = rows;
            }
        }
    if (!best_k) return false;
    g.k = best_k;
    g.ROWS = best_rows;
    g.S = 64 / g.k;
    // Workgroup shape.  With enough bins, single-wavefront workgroups that walk through the Gauss-point groups of their
    // bins one after the other measured fastest on MI355X (8 independent wavefronts per CU drift out of phase, so
    // loads of one overlap the sweeps of another: 0.40 vs 0.66 ms at 10 000 bins).  A small spectral grid cannot fill
    // the 1024 SIMDs that way (300 bins x 50 layers: 150 wavefronts, 50 us); then 5-wavefront workgroups take all
    // Gauss points of their bins at once (18 us).  Measured cross-over: about one single-wavefront workgroup per SIMD.
    // HELIOS_RT_MAXTHREADS (64..320) overrides the choice.
    int max_threads = 64;
    if (flux_one_wave(g.ROWS)) {
        choose_workgroup(Y, X, 64, g);   // (these kernels are built for single-wavefront workgroups: rt_kernels.h)
    } else if (const char* e = getenv("HELIOS_RT_MAXTHREADS")) {
        max_threads = std::max(64, std::min(320, atoi(e)));
        choose_workgroup(Y, X, max_threads, g);
    } else {
        choose_workgroup(Y, X, 64, g);
        if (g.ypb && (long long)((X + g.nxb - 1) / g.nxb) * C < 1024) choose_workgroup(Y, X, 320, g);
    }
    if (!g.ypb) return false;
    g.nparts = Y / g.ypb;
    g.G = g.nxb * g.ypb;
    g.NW = (g.G * g.k + 63) / 64;
    g.threads = g.NW * 64;
    g.nblk_x = (X + g.nxb - 1) / g.nxb;
    g.nblk = g.nblk_x * g.nparts;
    // planes: alpha, beta, u' always; v' only when E != 1 can occur; dd, du only with the beam
    g.has_vp = scat_corr ? 1 : 0;
    g.pl_vp = 3;
    g.pl_dd = 3 + g.has_vp;
    g.nplane = 3 + g.has_vp + (dir_beam ? 2 : 0);
    (void)matrix;   // (the matrix method's direct solve reads the sweeps' planes, nothing more)
    g.tile_rows = g.ROWS;
    g.coef_elems_per_col = (size_t)g.nblk * g.NW * g.nplane * g.ROWS * 64;
    g.flux_elems_per_col = (size_t)g.nblk * g.NW * g.ROWS * 64;
    return true;
}

KArgs make_args(hx_rt* rt) {
    KArgs a;
    memset(&a, 0, sizeof(a));
    const TileGeom& g = rt->g;
    a.X = rt->X; a.Y = rt->Y; a.L = rt->L; a.I = rt->I; a.H = rt->H; a.C = rt->C;
    a.k = g.k; a.ROWS = g.ROWS; a.S = g.S; a.nxb = g.nxb; a.ypb = g.ypb;
    a.nparts = g.nparts; a.G = g.G; a.NW = g.NW; a.nblk_x = g.nblk_x; a.nblk = g.nblk;
    a.nplane = g.nplane; a.nchunk = rt->nchunk;
    a.has_vp = g.has_vp; a.pl_vp = g.pl_vp; a.pl_dd = g.pl_dd;
    a.matrix = rt->matrix_scan ? 1 : 0; a.trigger = rt->mx.trigger;
    a.Kconst = 2.0 * HX_PI * rt->f.epsi;
    a.scat = rt->f.scat; a.dir_beam = rt->f.dir_beam; a.clouds = rt->f.clouds;
    a.scat_corr = rt->f.scat_corr; a.nsweep = rt->nsweep; a.keep_down = rt->keep_down ? 1 : 0;
    a.real_star = rt->f.real_star;
    a.iso = rt->f.iso;
    a.dim = rt->d.plancktable_dim; a.step = rt->d.plancktable_step;
    a.epsi = rt->f.epsi; a.epsi2 = rt->f.epsi2; a.g_0 = rt->f.g_0; a.i2s = rt->f.i2s_transition;
    a.w_0_limit = rt->f.w_0_limit; a.w_0_scat_limit = rt->f.w_0_scat_limit;
    a.dtau_limit = rt->f.delta_tau_limit;
    a.colpar = rt->colpar;
    a.T_lay = rt->T_lay; a.p_lay = rt->p_lay; a.p_int = rt->p_int; a.dcol_u = rt->dcol_u;
    a.dcol_l = rt->dcol_l; a.surf_albedo = rt->surf_albedo; a.Bstar = rt->Bstar;
    a.planck_grid = rt->planck_grid;
    a.opac_wg_lay = rt->opac_wg_lay; a.opac_wg_int = rt->opac_wg_int;
    a.scat_cross_lay = rt->scat_cross_lay; a.scat_cross_int = rt->scat_cross_int;
    a.mmm_lay = rt->mmm_lay; a.mmm_int = rt->mmm_int;
    a.cl_abs_lay = rt->cl_abs_lay; a.cl_abs_int = rt->cl_abs_int; a.cl_sc_lay = rt->cl_sc_lay;
    a.cl_sc_int = rt->cl_sc_int; a.g0_tot_lay = rt->g0_tot_lay; a.g0_tot_int = rt->g0_tot_int;
    a.half_ray = rt->half_ray; a.half_g0 = rt->half_g0; a.half_cab = rt->half_cab; a.half_csc = rt->half_csc;
    a.F_dir_wg = rt->F_dir_wg; a.Fc_dir_wg = rt->Fc_dir_wg; a.F_dir_band_n = rt->F_dir_band_n;
    a.gauss_w = rt->gauss_w; a.deltawave = rt->deltawave;
    a.diag = rt->f.debug == 1 ? rt->ctx->diag : nullptr;
    a.T_int = rt->T_int; a.Bn = rt->Bn; a.coef = rt->coef; a.Utile = rt->Utile; a.Dtile = rt->Dtile;
    a.U0 = rt->U0; a.boaK = rt->boaK; a.Fdir0 = rt->Fdir0;
    a.dtau_u = rt->dtau_u; a.dtau_l = rt->dtau_l;
    a.F_down_band_n = rt->F_down_band_n; a.F_up_band_n = rt->F_up_band_n; a.tot_part = rt->tot_part;
    a.F_up_tot = rt->F_up_tot; a.F_down_tot = rt->F_down_tot; a.F_net = rt->F_net;
    a.coef_col = g.coef_elems_per_col; a.flux_col = g.flux_elems_per_col;
    a.done = rt->done;
    a.ktable = rt->opac_k; a.crosstable = rt->opac_scat_cross; a.ktemp = rt->ktemp; a.kpress = rt->kpress;
    a.tp_lay = (const TPIndex*)rt->tp_lay; a.tp_int = (const TPIndex*)rt->tp_int;
    a.ntemp = rt->d.ntemp; a.npress = rt->d.npress; a.from_table = 0;
    a.iter_dev = rt->iter_dev;
    return a;
}

// ---- profiling helpers ------------------------------------------------------------------------
struct ProfScope {
    hx_rt* rt;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char* name;
    ProfScope(hx_rt* r, const char* n) : rt(r), name(n) {
        if (!rt->profiling) return;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, rt->ctx->stream);
    }
    ~ProfScope() {
        if (!rt->profiling) return;
        (void)hipEventRecord(e1, rt->ctx->stream);
        rt->prof.push_back({name, e0, e1});
        if (rt->prof.size() > 8192) flush(rt);
    }
    static void flush(hx_rt* rt) {
        for (auto& p : rt->prof) {
            (void)hipEventSynchronize(p.e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, p.e0, p.e1);
            bool found = false;
            for (auto& acc : rt->prof_acc)
                if (acc.first == p.name) {
                    acc.second.first += ms;
                    acc.second.second += 1;
                    found = true;
                }
            if (!found) rt->prof_acc.push_back({p.name, {ms, 1}});
            (void)hipEventDestroy(p.e0);
            (void)hipEventDestroy(p.e1);
        }
        rt->prof.clear();
    }
};

size_t flux_shmem_bytes(hx_rt* rt) {
    const TileGeom& g = rt->g;
    return ((size_t)g.nxb * (rt->H + 3) + (size_t)g.nxb * 2 * rt->I + (size_t)g.ypb * g.nxb * 2 * rt->I) *
           sizeof(double);
}

template <int ROWS>
void launch_flux(hx_rt* rt, const KArgs& a) {
    const TileGeom& g = rt->g;
    const size_t shmem = flux_shmem_bytes(rt);
    FluxArgs f;
    memset(&f, 0, sizeof(f));
    f.X = a.X; f.Y = a.Y; f.L = a.L; f.I = a.I; f.H = a.H;
    f.k = a.k; f.nxb = a.nxb; f.ypb = a.ypb; f.nparts = a.nparts; f.G = a.G; f.NW = a.NW;
    f.dir_beam = a.dir_beam; f.nsweep = a.nsweep; f.keep_down = a.keep_down; f.has_vp = a.has_vp;
    f.pl_vp = a.pl_vp; f.pl_dd = a.pl_dd; f.nplane = a.nplane; f.iso = a.iso;
#ifdef HX_PROFILING  // profiling builds only (make PROFILING=1): switches parts of k_rt_flux off, the results are wrong
    static const int debug_skip = [] { const char* e = getenv("HELIOS_RT_DEBUG_SKIP"); return e ? atoi(e) : 0; }();
    f.debug_skip = debug_skip;
#endif
    f.Kconst = a.Kconst;
    f.trigger = a.trigger;
    f.keep_up = rt->matrix_keep_state ? 1 : 0;
    f.colpar = a.colpar;
    f.Bn = a.Bn; f.coef = a.coef; f.U0_in = a.U0; f.boaK = a.boaK; f.Fdir0 = a.Fdir0;
    f.surf_albedo = a.surf_albedo; f.gauss_w = a.gauss_w;
    f.Utile = a.Utile; f.Dtile = a.Dtile; f.U0 = a.U0; f.F_down_band_n = a.F_down_band_n;
    f.F_up_band_n = a.F_up_band_n;
    f.coef_col = a.coef_col; f.flux_col = a.flux_col;
    f.done = a.done;
    // Serpentine order over the iterations: the workgroups of an odd launch take the tiles from the far end, so the
    // tiles the previous launch touched last -- still in the 256 MiB Infinity Cache -- are the ones this launch asks
    // for first.  Same bits (workgroups are independent).
    f.reverse = rt->serpentine ? (rt->flux_launches++ & 1) : 0;
    f.cache_state_from = INT_MAX;
    if (rt->serpentine && rt->state_cache_mb > 0) {
        const double per_wg = (double)g.nparts * g.NW * ROWS * 64 * sizeof(double);
        const long long total = (long long)g.nblk_x * rt->C;
        const long long keep = (long long)(rt->state_cache_mb * 1048576.0 / per_wg);
        f.cache_state_from = (int)std::max(0LL, total - keep);
    }
    const bool generic = rt->generic_scans;
    if constexpr (ROWS > 16) {   // (only on 64 lanes: choose_geometry)
        if (rt->matrix_scan) {
            f.reverse = 0;
            f.cache_state_from = INT_MAX;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 64, true>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                               rt->ctx->stream, f);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 64>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                               rt->ctx->stream, f);
        }
        return;
    } else {
    if (rt->matrix_scan) {   // the direct solve: no state to leave in the cache for a next launch, no launch order to alternate
        f.reverse = 0;
        f.cache_state_from = INT_MAX;
        if (g.k == 16 && !generic)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 16, true>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                               rt->ctx->stream, f);
        else if (g.k == 32 && !generic)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 32, true>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                               rt->ctx->stream, f);
        else if (g.k == 64 && !generic)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 64, true>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                               rt->ctx->stream, f);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 0, true>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                               rt->ctx->stream, f);
        return;
    }
    if (g.k == 16 && !generic)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 16>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                           rt->ctx->stream, f);
    else if (g.k == 32 && !generic)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 32>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                           rt->ctx->stream, f);
    else if (g.k == 64 && !generic)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS, 64>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                           rt->ctx->stream, f);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_flux<ROWS>), dim3(g.nblk_x, rt->C), dim3(g.threads), shmem,
                           rt->ctx->stream, f);
    }
}
template <int ROWS, int TPB>
void launch_coef_tpb(hx_rt* rt, KArgs a) {
    const TileGeom& g = rt->g;
    const int ntiles = g.nblk_x * g.nparts * g.NW;
    const int TS = TPB * g.S, TSP = TS;
    const int NBX = g.nxb * ((TPB - 1) / (g.NW * g.nparts) + 2);
    a.coef_nbx = NBX;
    size_t shmem = ((size_t)(rt->L + rt->I) * TSP + (size_t)rt->H * (NBX + 2)) * sizeof(double) + 2 * TS * sizeof(int);
    const size_t cloud_image = 3 * (size_t)rt->H * NBX * sizeof(double);
    a.cloud_lds = a.clouds == 1 && rt->cloud_lds && shmem + cloud_image <= 80 * 1024;  // keep two workgroups per CU
    if (a.cloud_lds) shmem += cloud_image;
    if (shmem > 64 * 1024 && !rt->coef_shmem_raised) {
        (void)hipFuncSetAttribute((const void*)k_rt_coef<ROWS, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem);
        rt->coef_shmem_raised = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rt_coef<ROWS, TPB>), dim3((ntiles + TPB - 1) / TPB, rt->C), dim3(64 * TPB),
                       shmem, rt->ctx->stream, a);
}
template <int ROWS>
void launch_coef(hx_rt* rt, const KArgs& a) {
    switch (rt->coef_tpb) {
        case 1: launch_coef_tpb<ROWS, 1>(rt, a); break;
        case 2: launch_coef_tpb<ROWS, 2>(rt, a); break;
        case 8:
            if constexpr (ROWS <= 16) { launch_coef_tpb<ROWS, 8>(rt, a); break; }   // (big columns: at most four tiles fit the LDS)
            [[fallthrough]];
        default: launch_coef_tpb<ROWS, 4>(rt, a); break;
    }
}

// LDS demand of k_rt_coef with `tpb` tiles per workgroup (launch_coef_tpb's formula, without the optional cloud image)
size_t coef_shmem_bytes(const hx_rt* rt, int tpb) {
    const TileGeom& g = rt->g;
    const int TS = tpb * g.S;
    const int NBX = g.nxb * ((tpb - 1) / (g.NW * g.nparts) + 2);
    return ((size_t)(rt->L + rt->I) * TS + (size_t)rt->H * (NBX + 2)) * sizeof(double) + 2 * TS * sizeof(int);
}

#define DISPATCH_ROWS(fn, rt, a)                  \
    switch ((rt)->g.ROWS) {                       \
        case 1: fn<1>(rt, a); break;              \
        case 2: fn<2>(rt, a); break;              \
        case 3: fn<3>(rt, a); break;              \
        case 4: fn<4>(rt, a); break;              \
        case 5: fn<5>(rt, a); break;              \
        case 6: fn<6>(rt, a); break;              \
        case 7: fn<7>(rt, a); break;              \
        case 8: fn<8>(rt, a); break;              \
        case 9: fn<9>(rt, a); break;              \
        case 10: fn<10>(rt, a); break;            \
        case 11: fn<11>(rt, a); break;            \
        case 12: fn<12>(rt, a); break;            \
        case 13: fn<13>(rt, a); break;            \
        case 14: fn<14>(rt, a); break;            \
        case 15: fn<15>(rt, a); break;            \
        case 20: fn<20>(rt, a); break;            \
        case 24: fn<24>(rt, a); break;            \
        case 28: fn<28>(rt, a); break;            \
        case 32: fn<32>(rt, a); break;            \
        default: fn<16>(rt, a); break;            \
    }

template <int ROWS>
void raise_flux_shmem(hx_rt* rt, int shmem) {
    hipError_t e = hipSuccess;
    if constexpr (ROWS <= 16) {
        e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    }
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    if (rt->matrix_scan) {
        if constexpr (ROWS <= 16) {
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        }
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)k_rt_flux<ROWS, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    }
    rt->shmem_rc = e;
}

int set_flux_shmem_limits(hx_rt* rt) {
    const size_t shmem = flux_shmem_bytes(rt);
    if (shmem > 160 * 1024) return rt_fail(rt, HX_E_UNSUPPORTED, "workgroup LDS demand exceeds 160 KiB");
    if (shmem <= 64 * 1024) return 0;
    DISPATCH_ROWS(raise_flux_shmem, rt, (int)shmem);
    HX_HIP(rt->ctx, rt->shmem_rc);
    return 0;
}

}  // namespace

extern "C" {

int hx_rt_struct_sizes(int* dims_size, int* flags_size, int* column_size) {
    if (dims_size) *dims_size = (int)sizeof(hx_rt_dims);
    if (flags_size) *flags_size = (int)sizeof(hx_rt_flags);
    if (column_size) *column_size = (int)sizeof(hx_rt_column);
    return 0;
}

static int rt_create_into(hx_rt* rt, hx_context* ctx, const hx_rt_dims* dims, const hx_rt_flags* flags,
                          const hx_rt_column* columns);

int hx_rt_create(hx_context* ctx, const hx_rt_dims* dims, const hx_rt_flags* flags,
                 const hx_rt_column* columns, hx_rt** out_rt) {
    if (!ctx || !dims || !flags || !columns || !out_rt) return HX_E_ARG;
    *out_rt = nullptr;
    HX_REQUIRE(ctx, dims->nbin > 0 && dims->ny > 0 && dims->nlayer >= 2 && dims->ncol > 0, HX_E_ARG,
               "bad dimensions");
    HX_REQUIRE(ctx, dims->plancktable_dim >= 10 && dims->plancktable_step > 0, HX_E_ARG,
               "bad Planck-table dimensions");
    HX_HIP(ctx, hipSetDevice(ctx->device));
    hx_rt* rt = new hx_rt();
    const int rc_create = rt_create_into(rt, ctx, dims, flags, columns);
    if (rc_create) {  // frees every device allocation made so far (a failed 100 GB batch must not leak them)
        hx_rt_destroy(rt);
        return rc_create;
    }
    *out_rt = rt;
    return 0;
}

static int rt_create_into(hx_rt* rt, hx_context* ctx, const hx_rt_dims* dims, const hx_rt_flags* flags,
                          const hx_rt_column* columns) {
    rt->ctx = ctx;
    rt->d = *dims;
    rt->f = *flags;
    rt->cols.assign(columns, columns + dims->ncol);
    rt->X = dims->nbin; rt->Y = dims->ny; rt->L = dims->nlayer; rt->I = rt->L + 1;
    // segments of the sweeps: two half-layers per layer (read.py:888-895), or the layers themselves when isothermal
    rt->H = flags->iso ? rt->L : 2 * rt->L;
    rt->C = dims->ncol;
    rt->have_albedo.assign(dims->ncol, 0);
    RT_ALLOC(rt->iter_dev, 2);
    // computation.py:531-537: 3*scat+1 sweeps per iteration, 1000*scat+1 in the post-processing run type
    rt->nsweep = (flags->singlewalk ? 1000 : 3) * (flags->scat ? 1 : 0) + 1;
#ifdef HX_PROFILING
    if (const char* e = getenv("HELIOS_RT_DEBUG_NSWEEP")) rt->nsweep = atoi(e);  // profiling experiments only
#endif
    rt->matrix = flags->matrix != 0;
    rt->matrix_scan = rt->matrix;
    if (const char* e = getenv("HELIOS_RT_MATRIX")) rt->matrix_scan = rt->matrix && std::string(e) != "stage";
    rt->matrix_keep_state = rt->matrix_scan && flags->debug == 1;   // count_negative_fluxes reads the up-flux tiles
    if (!choose_geometry(rt->H, rt->Y, rt->X, rt->C, flags->dir_beam, flags->scat_corr, rt->g, rt->matrix_scan ? 1 : 0))
        return hx_fail(ctx, HX_E_UNSUPPORTED, "fused path supports nlayer <= 1024 (2048 isothermal layers); use the per-stage API");
    // bin chunks of the totals reduction: k_rt_totals_a wants many, _b few.  nbin/48 measured best at 10 000 bins;
    // a small grid keeps at least 32 chunks (of >= 8 bins) so that the first level still spreads over the chip
    rt->nchunk = std::max(1, std::min(512, std::max((rt->X + 47) / 48, std::min(32, (rt->X + 7) / 8))));
    if (const char* e = getenv("HELIOS_RT_GENERIC_SCANS")) rt->generic_scans = atoi(e) != 0;  // read per batch: tests
    if (const char* e = getenv("HELIOS_RT_CLOUD_LDS")) rt->cloud_lds = atoi(e) != 0;  // 0: k_rt_coef's fallback path (tests)
    // tiles per workgroup of k_rt_coef: 16 spectral points staged side by side (128-byte runs of the k-table) -- 4 tiles
    // at k = 16, 8 at k = 32 (config 5, same box: 2 tiles 6.4 ms, 4 tiles 4.5 ms, 8 tiles 3.4 ms per refresh)
    rt->coef_tpb = std::max(1, std::min(8, 16 / std::max(1, rt->g.S)));    // (16 tiles per workgroup at k = 64: measured, no faster)
    if (const char* e = getenv("HELIOS_RT_COEF_TPB")) rt->coef_tpb = atoi(e);   // tuning knobs
    while (rt->coef_tpb > 1 && coef_shmem_bytes(rt, rt->coef_tpb) > 150 * 1024) rt->coef_tpb /= 2;   // (deep columns: the staged layers of fewer tiles)
    if (coef_shmem_bytes(rt, rt->coef_tpb) > 160 * 1024)
        return hx_fail(ctx, HX_E_UNSUPPORTED, "k_rt_coef's staging of one tile exceeds the 160 KiB of LDS");
    if (const char* e = getenv("HELIOS_RT_NCHUNK")) rt->nchunk = std::max(1, std::min(4096, atoi(e)));  // tuning knob
    {
        // The up-flux state is the one array a k_rt_flux launch writes and the next one reads.  Where it is larger than
        // the 256 MiB Infinity Cache but the arrays of the small kernels in between (node Planck values, band fluxes) are
        // not, the launches walk the grid back and forth and the workgroups dispatched last keep 240 MiB of state in the
        // cache (write-through stores instead of non-temporal ones): the next launch starts with exactly those tiles and
        // neither reads them from HBM nor -- the cache is write-back -- have they been written there.  Same-box A/B at
        // BASELINE config 2: k_rt_flux 0.322 -> 0.299 ms, the step 0.405 -> 0.389 ms (profiles/r04_ab_state_cache.txt);
        // four columns per batch or config 5's grid: no gain (the arrays in between displace the state), small grids:
        // the write-through stores cost 1.6 us per launch -- hence the two conditions.
        const double state_mb = (double)rt->C * rt->g.flux_elems_per_col * sizeof(double) / 1048576.0;
        const double between_mb = (double)rt->C * rt->X * ((rt->H + 3) + 4.0 * rt->I) * sizeof(double) / 1048576.0;
        rt->serpentine = state_mb >= 64.0 && between_mb <= 64.0;
        rt->state_cache_mb = rt->serpentine ? 240.0 : 0.0;
    }
    if (const char* e = getenv("HELIOS_RT_SERPENTINE")) rt->serpentine = atoi(e) != 0;                   // tuning knobs:
    if (const char* e = getenv("HELIOS_RT_STATE_CACHE_MB")) rt->state_cache_mb = atof(e);                // same results
    rt->species.resize(dims->nspecies > 0 ? dims->nspecies : 0);
    int rc = set_flux_shmem_limits(rt);
    if (rc) return rc;

    const size_t X = rt->X, Y = rt->Y, L = rt->L, I = rt->I, C = rt->C, nc = X * Y;
    RT_ALLOC(rt->interwave, X + 1); RT_ALLOC(rt->deltawave, X); RT_ALLOC(rt->wave, X);
    RT_ALLOC(rt->gauss_y, Y); RT_ALLOC(rt->gauss_w, Y);
    RT_ALLOC(rt->ktemp, dims->ntemp); RT_ALLOC(rt->kpress, dims->npress);
    RT_ALLOC(rt->planck_grid, (size_t)(dims->plancktable_dim + 1) * X);
    RT_ALLOC(rt->colpar, C);
    RT_ALLOC(rt->p_lay, C * L); RT_ALLOC(rt->p_int, C * I); RT_ALLOC(rt->dcol_u, C * L);
    RT_ALLOC(rt->dcol_l, C * L); RT_ALLOC(rt->T_lay, C * (L + 1)); RT_ALLOC(rt->T_int, C * I);
    RT_ALLOC(rt->surf_albedo, C * X); RT_ALLOC(rt->starflux, C * X); RT_ALLOC(rt->Bstar, C * X);
    RT_ALLOC(rt->opac_wg_lay, C * nc * I); RT_ALLOC(rt->opac_wg_int, C * nc * I);
    RT_ALLOC(rt->scat_cross_lay, C * X * I); RT_ALLOC(rt->scat_cross_int, C * X * I);
    RT_ALLOC(rt->mmm_lay, C * I); RT_ALLOC(rt->mmm_int, C * I);
    RT_ALLOC(rt->cl_abs_lay, C * X * I); RT_ALLOC(rt->cl_abs_int, C * X * I);
    RT_ALLOC(rt->cl_sc_lay, C * X * I); RT_ALLOC(rt->cl_sc_int, C * X * I);
    if (flags->clouds) {
        RT_ALLOC(rt->cl_g0_lay, C * X * I); RT_ALLOC(rt->cl_g0_int, C * X * I);
    }
    RT_ALLOC(rt->g0_tot_lay, C * X * I); RT_ALLOC(rt->g0_tot_int, C * X * I);
    RT_ALLOC(rt->half_ray, C * X * rt->H);
    if (flags->clouds) {
        RT_ALLOC(rt->half_g0, C * X * rt->H); RT_ALLOC(rt->half_cab, C * X * rt->H); RT_ALLOC(rt->half_csc, C * X * rt->H);
    }
    if (dims->nspecies > 0) {
        RT_ALLOC(rt->vmr_lay, C * dims->nspecies * I); RT_ALLOC(rt->vmr_int, C * dims->nspecies * I);
        RT_ALLOC(rt->spec_lay, nc * I); RT_ALLOC(rt->spec_int, nc * I);
        RT_ALLOC(rt->sc_spec_lay, X * I); RT_ALLOC(rt->sc_spec_int, X * I);
    }
    RT_ALLOC(rt->delta_z, C * L); RT_ALLOC(rt->z_lay, C * L);
    const bool stage_matrix = rt->matrix && !rt->matrix_scan;
    if (flags->dir_beam || stage_matrix) {   // (the per-stage matrix solver reads the beam arrays unconditionally: zeros without a beam)
        RT_ALLOC(rt->dtau_u, C * nc * L); RT_ALLOC(rt->dtau_l, C * nc * L);
        RT_ALLOC(rt->F_dir_wg, C * nc * I); RT_ALLOC(rt->Fc_dir_wg, C * nc * I);
    }
    RT_ALLOC(rt->F_dir_band_n, C * X * I);
    RT_ALLOC(rt->Bn, C * X * (rt->H + 3));
    if (rt->matrix_scan) RT_ALLOC(rt->mx.trigger, C * nc);
    if (stage_matrix) {
        // `flux calculation method = matrix` through the per-stage kernels: the reference's per-half-layer arrays instead of the coefficient tiles
        // and the persistent up-flux state (a direct solve has none)
        MatrixArrays& m = rt->mx;
        const size_t wgL = C * nc * L, halves = flags->iso ? 1 : 2;
        for (double** q : {&m.trans_u, &m.M_u, &m.N_u, &m.P_u, &m.Gp_u, &m.Gm_u, &m.w0_u}) RT_ALLOC(*q, wgL);
        if (!flags->iso)
            for (double** q : {&m.trans_l, &m.M_l, &m.N_l, &m.P_l, &m.Gp_l, &m.Gm_l, &m.w0_l}) RT_ALLOC(*q, wgL);
        RT_ALLOC(m.dtc_u, C * X * L); RT_ALLOC(m.dtc_l, C * X * L); RT_ALLOC(m.dcol_iso, C * L);
        if (!m.trigger) RT_ALLOC(m.trigger, C * nc);
        RT_ALLOC(m.F_down, C * nc * I); RT_ALLOC(m.F_up, C * nc * I);
        RT_ALLOC(m.Fc_down, C * nc * I); RT_ALLOC(m.Fc_up, C * nc * I);
        RT_ALLOC(m.pb_lay, C * X * (L + 2)); RT_ALLOC(m.pb_int, C * X * I);
        RT_ALLOC(m.c_prime, nc * (halves * 2 * I)); RT_ALLOC(m.d_prime, nc * (halves * 2 * I));
    } else {
        RT_ALLOC(rt->coef, C * rt->g.coef_elems_per_col);
        RT_ALLOC(rt->Utile, C * rt->g.flux_elems_per_col);
    }
    RT_ALLOC(rt->U0, C * nc); RT_ALLOC(rt->boaK, C * nc); RT_ALLOC(rt->Fdir0, C * nc);
    RT_ALLOC(rt->F_down_band_n, C * X * I); RT_ALLOC(rt->F_up_band_n, C * X * I);
    RT_ALLOC(rt->tot_part, C * rt->nchunk * 2 * I);
    RT_ALLOC(rt->F_up_tot, C * I); RT_ALLOC(rt->F_down_tot, C * I); RT_ALLOC(rt->F_net, C * I);
    RT_ALLOC(rt->F_net_diff, C * L);
    RT_ALLOC(rt->T_store, C * (L + 1)); RT_ALLOC(rt->prefactor, C * (L + 1));
    RT_ALLOC(rt->F_add_heat_lay, C * L); RT_ALLOC(rt->F_add_heat_sum, C * L);
    RT_ALLOC(rt->F_smooth, C * L); RT_ALLOC(rt->F_smooth_sum, C * L); RT_ALLOC(rt->c_p_lay, C * L);
    RT_ALLOC(rt->abort_flags, C * (L + 1)); RT_ALLOC(rt->conv_count, C); RT_ALLOC(rt->done, C);
    RT_ALLOC(rt->kappa_lay, C * L); RT_ALLOC(rt->kappa_int, C * I); RT_ALLOC(rt->dampara, C);
    RT_ALLOC(rt->conv_unstable, C * (L + 1)); RT_ALLOC(rt->conv_layer, C * (L + 1)); RT_ALLOC(rt->marked_red, C * (L + 1));
    RT_ALLOC(rt->iters_done, C);
    {
        TPIndex* t1 = nullptr; TPIndex* t2 = nullptr;
        RT_ALLOC(t1, C * I); RT_ALLOC(t2, C * I);
        rt->tp_lay = t1; rt->tp_int = t2;
    }
    RT_ALLOC(rt->T_lay_ref, C * (L + 1)); RT_ALLOC(rt->T_int_ref, C * I);
    if (dims->nspecies > 0) {
        SpeciesDev* sd = nullptr;
        RT_ALLOC(sd, (size_t)dims->nspecies);
        rt->species_dev = sd;
        RT_ALLOC(rt->abs_list, (size_t)dims->nspecies);
        RT_ALLOC(rt->fac_lay, C * I * dims->nspecies); RT_ALLOC(rt->fac_int, C * I * dims->nspecies);
    }
    return h2d(rt, rt->colpar, rt->cols.data(), C * sizeof(hx_rt_column));
}

int hx_rt_destroy(hx_rt* rt) {
    if (!rt) return 0;
    if (rt->ctx) (void)hipStreamSynchronize(rt->ctx->stream);
    ProfScope::flush(rt);
    if (rt->iter_graph) (void)hipGraphExecDestroy(rt->iter_graph);
    if (rt->decade_graph) (void)hipGraphExecDestroy(rt->decade_graph);
    for (void* p : rt->allocs) (void)hipFree(p);
    delete rt;
    return 0;
}

// A setter changed what the launches are given: captured graphs hold older arguments (each compares its own generation),
// and spectral flux tiles re-created for hx_rt_get are no longer the last solve's.
static inline void rt_touch(hx_rt* rt) {
    rt->graph_gen++;
    rt->solve_serial++;
}

int hx_rt_set_grid(hx_rt* rt, const double* opac_interwave, const double* opac_deltawave,
                   const double* opac_wave, const double* gauss_y, const double* gauss_weight,
                   const double* ktemp, const double* kpress) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int rc = 0;
    rc |= h2d(rt, rt->interwave, opac_interwave, (rt->X + 1) * 8);
    rc |= h2d(rt, rt->deltawave, opac_deltawave, rt->X * 8);
    rc |= h2d(rt, rt->wave, opac_wave, rt->X * 8);
    rc |= h2d(rt, rt->gauss_y, gauss_y, rt->Y * 8);
    rc |= h2d(rt, rt->gauss_w, gauss_weight, rt->Y * 8);
    rc |= h2d(rt, rt->ktemp, ktemp, rt->d.ntemp * 8);
    rc |= h2d(rt, rt->kpress, kpress, rt->d.npress * 8);
    rt->have_grid = rc == 0;
    return rc;
}

int hx_rt_set_premixed_tables(hx_rt* rt, const double* opac_k, const double* opac_scat_cross,
                              const double* opac_meanmass) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    HX_REQUIRE(rt->ctx, rt->d.nspecies == 0, HX_E_STATE, "object was created for on-the-fly mixing");
    const size_t ntp = (size_t)rt->d.ntemp * rt->d.npress;
    if (!rt->opac_k) {
        RT_ALLOC(rt->opac_k, ntp * rt->X * rt->Y);
        RT_ALLOC(rt->opac_scat_cross, ntp * rt->X);
        RT_ALLOC(rt->opac_meanmass, ntp);
    }
    int rc = 0;
    rc |= h2d(rt, rt->opac_k, opac_k, ntp * rt->X * rt->Y * 8);
    rc |= h2d(rt, rt->opac_scat_cross, opac_scat_cross, ntp * rt->X * 8);
    rc |= h2d(rt, rt->opac_meanmass, opac_meanmass, ntp * 8);
    rt->have_tables = rc == 0;
    return rc;
}

int hx_rt_set_species(hx_rt* rt, int s, const double* opacity_pretab, const double* scat_cross,
                      double weight, int is_h2o, int is_cia, int in_mu) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    HX_REQUIRE(rt->ctx, s >= 0 && s < (int)rt->species.size(), HX_E_ARG, "species index out of range");
    Species& sp = rt->species[s];
    const size_t ntp = (size_t)rt->d.ntemp * rt->d.npress;
    sp.weight = weight; sp.is_h2o = is_h2o; sp.is_cia = is_cia; sp.in_mu = in_mu;
    if (opacity_pretab) {
        if (!sp.pretab) RT_ALLOC(sp.pretab, ntp * rt->X * rt->Y);
        int rc = h2d(rt, sp.pretab, opacity_pretab, ntp * rt->X * rt->Y * 8);
        if (rc) return rc;
        sp.absorbing = true;
    }
    if (scat_cross) {
        if (!sp.scat_cross) RT_ALLOC(sp.scat_cross, rt->X);
        int rc = h2d(rt, sp.scat_cross, scat_cross, rt->X * 8);
        if (rc) return rc;
        sp.scattering = true;
    }
    if (is_h2o == 2) sp.scattering = true;  // H2O with computed Rayleigh cross-section
    rt->have_tables = true;
    rt->species_dev_stale = true;
    return 0;
}

// Synthetic tables (bench.py, tests): kappa[t][p][x][y] = kxy[y + ny x] * ftp[p + npress t], formed on the device -- the
// shape helios_amd/synthetic.py:ktable gives its tables (one fp64 product per entry, the same bits as its numpy form) without a
// 0.96 GB host array per table and its PCIe copy: a rank's set-up of 20 species at config 3's size took 85 core-seconds of
// single-threaded numpy.  The readers of real k-tables do not come here.
__global__ void __launch_bounds__(256) k_rt_table_outer(double* __restrict__ out, const double* __restrict__ kxy,
                                                       const double* __restrict__ ftp, size_t nxy) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nxy) __builtin_nontemporal_store(kxy[i] * ftp[blockIdx.y], out + (size_t)blockIdx.y * nxy + i);
}

static int fill_outer(hx_rt* rt, double* table, const double* kxy, const double* ftp) {
    const size_t nxy = (size_t)rt->X * rt->Y, ntp = (size_t)rt->d.ntemp * rt->d.npress;
    double* stage = nullptr;   // the two factors, freed right after the launch (stream order)
    HX_HIP(rt->ctx, hipMalloc((void**)&stage, (nxy + ntp) * 8));
    int rc = h2d(rt, stage, kxy, nxy * 8);
    if (!rc) rc = h2d(rt, stage + nxy, ftp, ntp * 8);
    if (!rc) {
        k_rt_table_outer<<<dim3((unsigned)((nxy + 255) / 256), (unsigned)ntp), 256, 0, rt->ctx->stream>>>(table, stage, stage + nxy, nxy);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = hx_fail(rt->ctx, -(int)e, "k_rt_table_outer: %s", hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(rt->ctx->stream);
    (void)hipFree(stage);
    return rc;
}

int hx_rt_set_species_separable(hx_rt* rt, int s, const double* kxy, const double* ftp, const double* scat_cross,
                                double weight, int is_h2o, int is_cia, int in_mu) {
    if (!rt) return HX_E_ARG;
    HX_REQUIRE(rt->ctx, kxy && ftp, HX_E_ARG, "both factors of the table are needed");
    int rc = hx_rt_set_species(rt, s, nullptr, scat_cross, weight, is_h2o, is_cia, in_mu);
    if (rc) return rc;
    Species& sp = rt->species[s];
    if (!sp.pretab) RT_ALLOC(sp.pretab, (size_t)rt->d.ntemp * rt->d.npress * rt->X * rt->Y);
    rc = fill_outer(rt, sp.pretab, kxy, ftp);
    if (rc) return rc;
    sp.absorbing = true;
    rt->species_dev_stale = true;
    return 0;
}

int hx_rt_set_premixed_separable(hx_rt* rt, const double* kxy, const double* ftp, const double* opac_scat_cross,
                                 const double* opac_meanmass) {
    if (!rt) return HX_E_ARG;
    rt_touch(rt);         
    HX_REQUIRE(rt->ctx, rt->d.nspecies == 0, HX_E_STATE, "object was created for on-the-fly mixing");
    HX_REQUIRE(rt->ctx, kxy && ftp && opac_scat_cross && opac_meanmass, HX_E_ARG, "null table");
    const size_t ntp = (size_t)rt->d.ntemp * rt->d.npress;
    if (!rt->opac_k) {
        RT_ALLOC(rt->opac_k, ntp * rt->X * rt->Y);
        RT_ALLOC(rt->opac_scat_cross, ntp * rt->X);
        RT_ALLOC(rt->opac_meanmass, ntp);
    }
    int rc = fill_outer(rt, rt->opac_k, kxy, ftp);
    if (!rc) rc = h2d(rt, rt->opac_scat_cross, opac_scat_cross, ntp * rt->X * 8);
    if (!rc) rc = h2d(rt, rt->opac_meanmass, opac_meanmass, ntp * 8);
    rt->have_tables = rc == 0;
    return rc;
}

static int for_cols(hx_rt* rt, int col, int* c0, int* c1);

// calculate_vmr_for_all_species on the device (host_functions.py:874-910): a species whose mixing ratio is tabulated on the
// opacity tables' (T, P) grid -- vmr_pretab[p + npress * t], what read.py keeps per FastChem species -- follows the
// temperatures of every refresh without a host step.  One table per COLUMN (a sweep over FastChem directories -- metallicity,
// C/O -- gives every column its own chemistry, read.py:577-606); col < 0 hands the same table to all columns.  NULL (any
// col): the species goes back to the profiles of hx_rt_set_column_vmr in every column.
int hx_rt_set_column_vmr_table(hx_rt* rt, int col, int s, const double* vmr_pretab) {
    if (!rt) return HX_E_ARG;
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    HX_REQUIRE(rt->ctx, s >= 0 && s < (int)rt->species.size(), HX_E_ARG, "species index out of range");
    HX_REQUIRE(rt->ctx, rt->d.ntemp >= 2 && rt->d.npress >= 2, HX_E_ARG, "a mixing-ratio table needs at least 2 x 2 nodes");
    int c0, c1;
    int rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    Species& sp = rt->species[s];
    rt->species_dev_stale = true;
    if (!vmr_pretab) {
        sp.vmr_from_tab = false;
        return 0;
    }
    const size_t ntp = (size_t)rt->d.ntemp * rt->d.npress;
    // (zero-initialised: a column that is never given a table of its own reads zeros -- its species is absent)
    if (!sp.vmr_tab) RT_ALLOC(sp.vmr_tab, ntp * rt->C);
    sp.vmr_from_tab = true;
    for (int c = c0; c < c1 && !rc; c++) rc = h2d(rt, sp.vmr_tab + ntp * c, vmr_pretab, ntp * 8);
    return rc;
}

int hx_rt_set_species_vmr_table(hx_rt* rt, int s, const double* vmr_pretab) {
    return hx_rt_set_column_vmr_table(rt, -1, s, vmr_pretab);
}

static int for_cols(hx_rt* rt, int col, int* c0, int* c1) {
    if (col < 0) { *c0 = 0; *c1 = rt->C; return 0; }
    if (col >= rt->C) return rt_fail(rt, HX_E_ARG, "column index out of range");
    *c0 = col; *c1 = col + 1;
    return 0;
}

int hx_rt_set_column_profile(hx_rt* rt, int col, const double* p_lay, const double* p_int,
                             const double* T_lay, const double* surf_albedo,
                             const double* starflux) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    const size_t L = rt->L, I = rt->I, X = rt->X;
    if (rt->matrix && surf_albedo)  // row 0 of the tridiagonal system divides by it; the reader keeps it >= 1e-8 (read.py:1261)
        for (size_t x = 0; x < X; x++)
            if (!(surf_albedo[x] > 0.0))
                return rt_fail(rt, HX_E_ARG, "the matrix method needs a surface albedo > 0 in every bin (the reference's reader sets at least 1e-8)");
    for (int c = c0; c < c1; c++) {
        std::vector<double> du(L), dl(L);
        const double g = rt->cols[c].g;
        for (size_t i = 0; i < L; i++) {  // host_functions.py:731-735
            du[i] = (p_lay[i] - p_int[i + 1]) / g;
            dl[i] = (p_int[i] - p_lay[i]) / g;
        }
        rc |= h2d(rt, rt->p_lay + c * L, p_lay, L * 8);
        rc |= h2d(rt, rt->p_int + c * I, p_int, I * 8);
        rc |= h2d(rt, rt->dcol_u + c * L, du.data(), L * 8);
        rc |= h2d(rt, rt->dcol_l + c * L, dl.data(), L * 8);
        if (rt->matrix && rt->mx.dcol_iso) {  // whole layers (host_functions.py:733), calc_trans_iso's delta_colmass
            for (size_t i = 0; i < L; i++) du[i] = (p_int[i] - p_int[i + 1]) / g;
            rc |= h2d(rt, rt->mx.dcol_iso + c * L, du.data(), L * 8);
        }
        rc |= h2d(rt, rt->T_lay + c * (L + 1), T_lay, (L + 1) * 8);
        if (surf_albedo) {
            rc |= h2d(rt, rt->surf_albedo + c * X, surf_albedo, X * 8);
            rt->have_albedo[c] = 1;
        }
        if (starflux) rc |= h2d(rt, rt->starflux + c * X, starflux, X * 8);
    }
    return rc;
}

int hx_rt_set_column_vmr(hx_rt* rt, int col, const double* vmr_lay, const double* vmr_int) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    HX_REQUIRE(rt->ctx, rt->d.nspecies > 0, HX_E_STATE, "no species in a premixed object");
    const size_t S = rt->d.nspecies, I = rt->I, L = rt->L;
    // internal stride is I for both (layer rows use the first L entries)
    std::vector<double> tmp(S * I, 0.0);
    for (int c = c0; c < c1; c++) {
        for (size_t s = 0; s < S; s++) std::copy(vmr_lay + s * L, vmr_lay + (s + 1) * L, tmp.begin() + s * I);
        rc |= h2d(rt, rt->vmr_lay + c * S * I, tmp.data(), S * I * 8);
        rc |= h2d(rt, rt->vmr_int + c * S * I, vmr_int, S * I * 8);
    }
    return rc;
}

int hx_rt_set_column_clouds(hx_rt* rt, int col, const double* abs_cross_lay,
                            const double* abs_cross_int, const double* scat_cross_lay,
                            const double* scat_cross_int, const double* g_0_lay,
                            const double* g_0_int) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    HX_REQUIRE(rt->ctx, rt->f.clouds == 1, HX_E_STATE, "object was created with clouds = 0");
    const size_t XL = (size_t)rt->X * rt->L, XI = (size_t)rt->X * rt->I;
    for (int c = c0; c < c1; c++) {
        rc |= h2d(rt, rt->cl_abs_lay + c * XI, abs_cross_lay, XL * 8);
        rc |= h2d(rt, rt->cl_abs_int + c * XI, abs_cross_int, XI * 8);
        rc |= h2d(rt, rt->cl_sc_lay + c * XI, scat_cross_lay, XL * 8);
        rc |= h2d(rt, rt->cl_sc_int + c * XI, scat_cross_int, XI * 8);
        rc |= h2d(rt, rt->cl_g0_lay + c * XI, g_0_lay, XL * 8);
        rc |= h2d(rt, rt->cl_g0_int + c * XI, g_0_int, XI * 8);
    }
    return rc;
}

int hx_rt_set_column_heating(hx_rt* rt, int col, const double* F_add_heat_lay,
                             const double* F_add_heat_sum) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    for (int c = c0; c < c1; c++) {
        rc |= h2d(rt, rt->F_add_heat_lay + (size_t)c * rt->L, F_add_heat_lay, rt->L * 8);
        rc |= h2d(rt, rt->F_add_heat_sum + (size_t)c * rt->L, F_add_heat_sum, rt->L * 8);
    }
    return rc;
}

int hx_rt_set_temperatures(hx_rt* rt, int col, const double* T_lay) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    for (int c = c0; c < c1; c++) rc |= h2d(rt, rt->T_lay + (size_t)c * (rt->L + 1), T_lay, (rt->L + 1) * 8);
    return rc;
}

int hx_rt_set_convergence_limit(hx_rt* rt, int col, double limit) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    for (int c = c0; c < c1; c++) {
        rt->cols[c].rad_convergence_limit = limit;
        rc |= h2d(rt, &rt->colpar[c], &rt->cols[c], sizeof(hx_rt_column));
    }
    return rc;
}


int hx_rt_build_planck_table(hx_rt* rt, int energy_correction) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    HX_REQUIRE(rt->ctx, rt->have_grid, HX_E_STATE, "hx_rt_set_grid first");
    hx_context* ctx = rt->ctx;
    const int X = rt->X, dim = rt->d.plancktable_dim, step = rt->d.plancktable_step;
    // rows 0..dim-1 depend on the wavelength grid only and are shared by all columns; row `dim`
    // (stellar row of column 0) is kept for layout compatibility with the reference's table
    int rc = hx_plancktable(ctx, rt->planck_grid, rt->interwave, rt->deltawave, X, rt->cols[0].T_star,
                            dim, step);
    if (rc) return rc;
    for (int c = 0; c < rt->C; c++) {
        const double Ts = rt->cols[c].T_star;
        double* brow = rt->Bstar + (size_t)c * X;
        double* sflux = rt->starflux + (size_t)c * X;
        rc = hx_internal_planck_star_row(ctx, brow, rt->interwave, rt->deltawave, X, Ts);
        if (rc) return rc;
        // computation.py:65: only if energy_correction == 1 and T_star > 10
        if (energy_correction == 1 && Ts > 10) {
            // dim = 0: the "table" passed is just the stellar row
            rc = hx_corr_inc_energy(ctx, brow, sflux, rt->deltawave, rt->f.real_star, X, Ts, 0);
            if (rc) return rc;
        }
        if (rt->f.real_star == 1) {
            // planckband_lay[star] = starflux / pi (kernels.cu:945)
            std::vector<double> tmp(X);
            rc = hx_d2h(ctx, tmp.data(), sflux, X * 8);
            if (rc) return rc;
            for (auto& v : tmp) v = v / HX_PI;
            rc = h2d(rt, brow, tmp.data(), X * 8);
            if (rc) return rc;
        }
        if (c == 0) {
            rc = hx_d2d(ctx, rt->planck_grid + (size_t)dim * X, brow, X * 8);
            if (rc) return rc;
        }
    }
    rt->have_planck = true;
    return 0;
}

// device image of the species list for the batched mixing kernels (rt_species.h)
static int upload_species_table(hx_rt* rt) {
    if (!rt->species_dev_stale) return 0;
    const int S = rt->d.nspecies;
    std::vector<SpeciesDev> sd(S);
    std::vector<int> abs;
    for (int s = 0; s < S; s++) {
        const Species& sp = rt->species[s];
        sd[s] = SpeciesDev{sp.pretab, sp.scat_cross, sp.vmr_from_tab ? sp.vmr_tab : nullptr, sp.weight, sp.absorbing ? 1 : 0,
                           sp.scattering ? 1 : 0,
                           // random overlap unless CIA (computation.py:1343) or one Gauss point per bin (opacity sampling:
                           // condition_for_correlated_k includes ny == 1, kernels.cu:3302)
                           sp.is_h2o ? 1 : 0, (rt->f.kcoeff_mixing_ro && !sp.is_cia && rt->Y != 1) ? 1 : 0,
                           sp.in_mu, 0};
        if (sp.absorbing) abs.push_back(s);
    }
    int rc = h2d(rt, rt->species_dev, sd.data(), S * sizeof(SpeciesDev));
    if (!rc && !abs.empty()) rc = h2d(rt, rt->abs_list, abs.data(), abs.size() * sizeof(int));
    rt->nabs = (int)abs.size();
    rt->species_dev_stale = rc != 0;
    return rc;
}

// on-the-fly mixing of every column, level and bin of the batch (computation.py:865-869, :1454-1501)
static int refresh_species(hx_rt* rt) {
    hx_context* ctx = rt->ctx;
    int rc = upload_species_table(rt);
    if (rc) return rc;
    MixArgs m;
    m.X = rt->X; m.Y = rt->Y; m.L = rt->L; m.I = rt->I; m.C = rt->C; m.S = rt->d.nspecies;
    m.ntemp = rt->d.ntemp; m.npress = rt->d.npress; m.nabs = rt->nabs; m.carry_on = 0;
    m.sp = (const SpeciesDev*)rt->species_dev; m.abs_list = rt->abs_list;
    m.T_lay = rt->T_lay; m.T_int = rt->T_int; m.p_lay = rt->p_lay; m.p_int = rt->p_int;
    m.vmr_lay = rt->vmr_lay; m.vmr_int = rt->vmr_int;
    m.ktemp = rt->ktemp; m.kpress = rt->kpress; m.gauss_w = rt->gauss_w; m.gauss_y = rt->gauss_y; m.wave = rt->wave;
    m.mmm_lay = rt->mmm_lay; m.mmm_int = rt->mmm_int;
    m.tp_lay = (TPIndex*)rt->tp_lay; m.tp_int = (TPIndex*)rt->tp_int;
    m.fac_lay = rt->fac_lay; m.fac_int = rt->fac_int;
    m.opac_wg_lay = rt->opac_wg_lay; m.opac_wg_int = rt->opac_wg_int;
    m.scat_lay = rt->scat_cross_lay; m.scat_int = rt->scat_cross_int;
    m.done = rt->done; m.diag = ctx->diag;
    k_rt_species_prep<<<dim3(hx_cdiv(rt->I, 64), rt->C), 64, 0, ctx->stream>>>(m);
    HX_LAUNCH_CHECK(ctx);
    {
        ProfScope ps(rt, "add_to_mixed_opac");
        const long long npair = (long long)rt->C * (rt->L + rt->I) * rt->X;
        // short runs (about ten points, each folding in every absorber): a wavefront lives for a fraction of a
        // millisecond, so the last round of workgroups does not leave the chip half empty
        static const int waves = [] { const char* e = getenv("HELIOS_RT_MIX_WAVES"); return e ? atoi(e) : 256 * 16 * 48; }();
        const int grid = (int)std::min(npair, (long long)waves);
        // occupancy experiment (profiles/r05_mix_occupancy.txt): bytes of unused dynamic LDS per wavefront on top of the
        // kernel's 9.95 KB -- 16 wavefronts share a CU's 160 KB as built, +3.4 KB leaves 12, +10 KB 8.  Results unchanged.
        static const int extra_lds = [] { const char* e = getenv("HELIOS_RT_MIX_EXTRA_LDS"); return e ? atoi(e) : 0; }();
        // the species list on chip holds MIX_MAX_ABSORBERS entries: a longer list is folded in block by block, every launch but
        // the first starting from the mix the one before it wrote (the absorbers' order, and so every sum, is the reference's)
        const int nabs_all = rt->nabs;
        for (int first = 0; first == 0 || first < nabs_all; first += MIX_MAX_ABSORBERS) {
            m.abs_list = rt->abs_list + first;
            m.nabs = std::min(MIX_MAX_ABSORBERS, nabs_all - first);
            m.carry_on = first > 0 ? 1 : 0;
            k_rt_mix_species<<<grid, 64, extra_lds, ctx->stream>>>(m);
            HX_LAUNCH_CHECK(ctx);
        }
        m.abs_list = rt->abs_list;
        m.nabs = nabs_all;
    }
    {
        ProfScope ps(rt, "mixed_scat");
        k_rt_scat_species<<<dim3(hx_cdiv(rt->X, 256), rt->L + rt->I, rt->C), 256, 0, ctx->stream>>>(m);
        HX_LAUNCH_CHECK(ctx);
    }
    return 0;
}

// ---- `flux calculation method = matrix` ------------------------------------------------------------------------------
// The per-stage kernels of calc_trans_* and fband_matrix_* (the ones the goldens pin), launched column by column from the
// device-resident loop on the arrays of rt->mx.  A column whose loop has ended recomputes its coefficients from unchanged
// inputs (same values) and skips the solve (`done`), so its fluxes stay those of its last iteration.
static int matrix_calc_trans(hx_rt* rt) {
    ProfScope ps(rt, "matrix_calc_trans");
    hx_context* ctx = rt->ctx;
    const hx_rt_flags& f = rt->f;
    const MatrixArrays& m = rt->mx;
    const size_t X = rt->X, L = rt->L, I = rt->I, nc = X * rt->Y, wgL = nc * L, wgI = nc * I, bI = X * I, bL = X * L;
    HX_HIP(ctx, hipMemsetAsync(m.trigger, 0, (size_t)rt->C * nc * sizeof(int), ctx->stream));  // computation.py:368
    for (size_t c = 0; c < (size_t)rt->C; c++) {
        const double mu_star = rt->cols[c].mu_star;
        int rc;
        if (f.iso)
            rc = hx_calc_trans_iso(ctx, m.trans_u + c * wgL, rt->dtau_u + c * wgL, m.M_u + c * wgL, m.N_u + c * wgL,
                                   m.P_u + c * wgL, m.Gp_u + c * wgL, m.Gm_u + c * wgL, m.dcol_iso + c * L,
                                   rt->opac_wg_lay + c * wgI, rt->mmm_lay + c * I, rt->scat_cross_lay + c * bI,
                                   rt->cl_abs_lay + c * bI, rt->cl_sc_lay + c * bI, m.dtc_u + c * bL, m.w0_u + c * wgL,
                                   rt->g0_tot_lay + c * bI, m.trigger + c * nc, f.g_0, f.epsi, f.epsi2, mu_star,
                                   f.w_0_limit, f.w_0_scat_limit, f.scat, rt->X, rt->Y, rt->L, f.clouds, f.scat_corr, 0,
                                   f.i2s_transition);
        else
            rc = hx_calc_trans_noniso(
                ctx, m.trans_u + c * wgL, m.trans_l + c * wgL, rt->dtau_u + c * wgL, rt->dtau_l + c * wgL, m.M_u + c * wgL,
                m.M_l + c * wgL, m.N_u + c * wgL, m.N_l + c * wgL, m.P_u + c * wgL, m.P_l + c * wgL, m.Gp_u + c * wgL,
                m.Gp_l + c * wgL, m.Gm_u + c * wgL, m.Gm_l + c * wgL, rt->dcol_u + c * L, rt->dcol_l + c * L,
                rt->opac_wg_lay + c * wgI, rt->opac_wg_int + c * wgI, rt->mmm_lay + c * I, rt->mmm_int + c * I,
                rt->scat_cross_lay + c * bI, rt->scat_cross_int + c * bI, rt->cl_abs_lay + c * bI, rt->cl_abs_int + c * bI,
                rt->cl_sc_lay + c * bI, rt->cl_sc_int + c * bI, m.dtc_u + c * bL, m.dtc_l + c * bL, m.w0_u + c * wgL,
                m.w0_l + c * wgL, rt->g0_tot_lay + c * bI, rt->g0_tot_int + c * bI, m.trigger + c * nc, f.g_0, f.epsi,
                f.epsi2, mu_star, f.w_0_limit, f.w_0_scat_limit, f.scat, rt->X, rt->Y, rt->L, f.clouds, f.scat_corr, 0,
                f.i2s_transition);
        if (rc) return rc;
    }
    return 0;
}

static int matrix_solve(hx_rt* rt) {
    ProfScope ps(rt, "matrix_solve");
    hx_context* ctx = rt->ctx;
    const hx_rt_flags& f = rt->f;
    const MatrixArrays& m = rt->mx;
    const size_t X = rt->X, L = rt->L, I = rt->I, nc = X * rt->Y, wgL = nc * L, wgI = nc * I, bI = X * I, bL = X * L;
    {   // the node values in the reference's layouts
        const long long n = (long long)X * (L + 2 + I);
        k_rt_matrix_planck<<<dim3(hx_cdiv(n, 256), rt->C), 256, 0, ctx->stream>>>(rt->Bn, m.pb_lay, m.pb_int, rt->X, rt->L,
                                                                               rt->H, f.iso, rt->done);
        HX_LAUNCH_CHECK(ctx);
    }
    for (size_t c = 0; c < (size_t)rt->C; c++) {
        const hx_rt_column& cp = rt->cols[c];
        const double* pbl = m.pb_lay + c * X * (L + 2);
        int rc;
        if (f.iso)
            rc = hx_internal_fband_matrix_iso(
                ctx, rt->done + c, m.F_down + c * wgI, m.F_up + c * wgI, rt->F_dir_wg + c * wgI, pbl, m.w0_u + c * wgL,
                m.M_u + c * wgL, m.N_u + c * wgL, m.P_u + c * wgL, m.Gp_u + c * wgL, m.Gm_u + c * wgL,
                rt->g0_tot_lay + c * bI, nullptr, nullptr, nullptr, nullptr, m.c_prime, m.d_prime, m.trigger + c * nc,
                m.trans_u + c * wgL, rt->surf_albedo + c * X, f.g_0, f.singlewalk, cp.R_star, cp.a, rt->I, rt->X,
                cp.f_factor, cp.mu_star, rt->Y, f.epsi, f.dir_beam, f.clouds, f.scat_corr, f.debug, f.i2s_transition);
        else
            rc = hx_internal_fband_matrix_noniso(
                ctx, rt->done + c, m.F_down + c * wgI, m.F_up + c * wgI, m.Fc_down + c * wgI, m.Fc_up + c * wgI,
                rt->F_dir_wg + c * wgI, rt->Fc_dir_wg + c * wgI, pbl, m.pb_int + c * bI, m.w0_u + c * wgL, m.w0_l + c * wgL,
                rt->dtau_u + c * wgL, rt->dtau_l + c * wgL, m.dtc_u + c * bL, m.dtc_l + c * bL, m.M_u + c * wgL,
                m.M_l + c * wgL, m.N_u + c * wgL, m.N_l + c * wgL, m.P_u + c * wgL, m.P_l + c * wgL, m.Gp_u + c * wgL,
                m.Gp_l + c * wgL, m.Gm_u + c * wgL, m.Gm_l + c * wgL, rt->g0_tot_lay + c * bI, rt->g0_tot_int + c * bI,
                nullptr, nullptr, nullptr, nullptr, m.c_prime, m.d_prime, m.trigger + c * nc, m.trans_u + c * wgL,
                m.trans_l + c * wgL, rt->surf_albedo + c * X, f.g_0, f.singlewalk, cp.R_star, cp.a, rt->I, rt->X,
                cp.f_factor, cp.mu_star, rt->Y, f.epsi, f.delta_tau_limit, f.dir_beam, f.clouds, f.scat_corr, f.debug,
                f.i2s_transition);
        if (rc) return rc;
    }
    k_rt_matrix_bands<<<dim3(hx_cdiv(rt->X, QUAD_BINS), rt->I, rt->C), 256, 2 * QUAD_BINS * (rt->Y + 1) * sizeof(double), ctx->stream>>>(
        m.F_down, m.F_up, rt->F_down_band_n, rt->F_up_band_n, rt->gauss_w, rt->X, rt->Y, rt->I, rt->done);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_rt_refresh(hx_rt* rt) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    hx_context* ctx = rt->ctx;
    rt->solve_serial++;
    HX_REQUIRE(ctx, rt->have_grid && rt->have_tables && rt->have_planck, HX_E_STATE,
               "set grid, tables and build the Planck table first");
    HX_REQUIRE(ctx, !rt->matrix || std::all_of(rt->have_albedo.begin(), rt->have_albedo.end(), [](char v) { return v != 0; }),
               HX_E_STATE, "the matrix method needs a surface albedo (> 0) for every column of the batch");
    const int X = rt->X, Y = rt->Y, L = rt->L, I = rt->I, C = rt->C;
    const size_t nc = (size_t)X * Y;
    int rc = 0;
    if (rt->d.nspecies > 0) {  // computation.py:1343: random overlap is written for 20 Gauss points
        bool any_ro = false;
        for (const Species& sp : rt->species) any_ro = any_ro || (sp.absorbing && rt->f.kcoeff_mixing_ro && !sp.is_cia);
        if (Y != ro::NY && Y != 1 && any_ro)  // ny == 1 mixes correlated-k in the reference too (kernels.cu:3302)
            return hx_fail(ctx, HX_E_RO_NY, "random-overlap mixing needs ny == 20 or 1 (got %d)", Y);
        HX_REQUIRE(ctx, Y <= ro::NY, HX_E_UNSUPPORTED, "on-the-fly mixing in the fused path holds at most 20 Gauss points");
        int nabs = 0;
        for (const Species& sp : rt->species) nabs += sp.absorbing ? 1 : 0;
        (void)nabs;   // (any number: the species loop takes them in blocks of MIX_MAX_ABSORBERS, refresh_species)
    }
    KArgs a = make_args(rt);
    rt->iter_dev_expected = -1;   // k_rt_nodes below moves the device's iteration counter on (hx_rt_step sets it right again)
    {   // interface temperatures (and node Planck values) of the CURRENT layer temperatures
        ProfScope ps(rt, "rt_nodes");
        dim3 grid(hx_cdiv(X, 32), hx_cdiv(rt->H + 3, 32), C);
        k_rt_nodes<<<grid, 256, 0, ctx->stream>>>(a);
        HX_LAUNCH_CHECK(ctx);
    }
    ProfScope ps_all(rt, "refresh_total");
    // Every launch below covers all columns; a column whose loop has ended (done[c], set on the device) is skipped
    // inside the kernels and keeps the state of its last real refresh.  No host round trip.
    // premixed table without the beam: the k-table look-up is fused into k_rt_coef (the beam needs the
    // materialised opacities for its optical depths); HELIOS_RT_FUSED_LOOKUP=0 switches it off
    const bool stage_matrix = rt->matrix && !rt->matrix_scan;
    bool fused_lookup = rt->d.nspecies == 0 && !rt->f.dir_beam && !stage_matrix;
    if (const char* e = getenv("HELIOS_RT_FUSED_LOOKUP")) fused_lookup = fused_lookup && atoi(e) != 0;
    rt->opac_stale = fused_lookup;
    if (rt->d.nspecies == 0) {
        k_rt_tp_index<<<dim3(hx_cdiv(I, 64), C), 64, 0, ctx->stream>>>(a, (TPIndex*)rt->tp_lay, (TPIndex*)rt->tp_int);
        k_rt_scat_interp<<<dim3(hx_cdiv(X, 256), I, C), 256, 0, ctx->stream>>>(a, rt->scat_cross_lay, rt->scat_cross_int);
        k_rt_mmm_table<<<dim3(hx_cdiv(I, 64), C), 64, 0, ctx->stream>>>(a, rt->opac_meanmass, rt->mmm_lay, rt->mmm_int);
        HX_LAUNCH_CHECK(ctx);
        if (fused_lookup) {
            // opacities are interpolated inside k_rt_coef; remember the temperatures this refresh used, so that
            // the arrays can be rebuilt on demand
            k_rt_keep_ref_T<<<dim3(hx_cdiv(I + 1, 64), C), 64, 0, ctx->stream>>>(a, rt->T_lay_ref, rt->T_int_ref);
        } else {
            ProfScope ps(rt, "opac_interpol");
            const int chunks = (int)std::min<long long>(hx_cdiv((long long)nc, 256), 1024);
            k_rt_opac_table<<<dim3(chunks, I, C), 256, 0, ctx->stream>>>(a, rt->opac_wg_lay, rt->opac_wg_int);
        }
        HX_LAUNCH_CHECK(ctx);
    } else {
        rc = refresh_species(rt);
        if (rc) return rc;
    }
    if (rt->f.clouds) {
        k_rt_total_g0<<<dim3(hx_cdiv(X, 256), I, C), 256, 0, ctx->stream>>>(a, rt->cl_g0_lay, rt->cl_g0_int,
                                                                          rt->g0_tot_lay, rt->g0_tot_int);
        HX_LAUNCH_CHECK(ctx);
    }
    k_rt_delta_z<<<dim3(hx_cdiv(L, 64), C), 64, 0, ctx->stream>>>(a, rt->delta_z);
    k_rt_height<<<C, 64, 0, ctx->stream>>>(rt->p_lay, rt->delta_z, rt->z_lay, L, rt->f.planet_type_gas, (size_t)L, rt->done);
    HX_LAUNCH_CHECK(ctx);
    if (rt->has_heating) {   // computation.py:913-918: refreshed with the layer heights, every 10th iteration
        k_rt_heating<<<C, 64, 0, ctx->stream>>>(rt->add_heat_dens, rt->delta_z, rt->F_add_heat_lay,
                                                rt->F_add_heat_sum, L, rt->done);
        HX_LAUNCH_CHECK(ctx);
    }
    if (stage_matrix) {
        rc = matrix_calc_trans(rt);
        if (rc) return rc;
    }
    if (rt->f.dir_beam) {
        ProfScope ps(rt, "direct_beam");
        if (!stage_matrix) {  // (calc_trans_* has just written the same optical depths)
            k_rt_dtau_halves<<<dim3(hx_cdiv((long long)nc, 256), L, C), 256, 0, ctx->stream>>>(a);
            HX_LAUNCH_CHECK(ctx);
        }
        rc = hx_internal_fdir_noniso_batch(ctx, rt->F_dir_wg, rt->f.iso ? nullptr : rt->Fc_dir_wg, rt->Bstar, rt->dtau_u, rt->dtau_l,
                                           rt->z_lay, rt->colpar, rt->done, C, rt->f.dir_beam,
                                           rt->f.geom_zenith_corr, I, X, Y);
        if (rc) return rc;
        k_rt_fdir_band<<<dim3(hx_cdiv(X, 32), hx_cdiv(I, 32), C), 256, 0, ctx->stream>>>(rt->F_dir_wg, rt->F_dir_band_n,
                                                                           rt->gauss_w, X, Y, I, rt->done);
        HX_LAUNCH_CHECK(ctx);
    }
    if (!stage_matrix) {
        ProfScope ps(rt, "rt_coef");
        k_rt_half_bands<<<dim3(hx_cdiv(X, 32), hx_cdiv(rt->H, 32), C), 256, 0, ctx->stream>>>(a);
        HX_LAUNCH_CHECK(ctx);
        a.from_table = fused_lookup ? 1 : 0;
        DISPATCH_ROWS(launch_coef, rt, a);
        HX_LAUNCH_CHECK(ctx);
    }
    rt->refreshed = true;
    return 0;
}

namespace { int kappa_cp_from_table(hx_rt* rt, bool refresh_T_int); }

// debug = 1: the negative-flux warnings of fband_noniso (kernels.cu:1663 ff.) as counts over the state the sweeps leave
// behind (up-fluxes always, down-fluxes when they are kept: hx_rt_set_state "keep_down")
static int count_negative_fluxes(hx_rt* rt) {
    const size_t n = (size_t)rt->C * rt->g.flux_elems_per_col;
    int rc = hx_internal_count_negative(rt->ctx, rt->Utile, n, HX_DIAG_NEG_UP);
    if (!rc && rt->keep_down && rt->Dtile) rc = hx_internal_count_negative(rt->ctx, rt->Dtile, n, HX_DIAG_NEG_DOWN);
    return rc;
}

// the spectral fluxes of one iteration: the register-resident sweeps, or one tridiagonal solve per spectral point
static int spectral_fluxes(hx_rt* rt, const KArgs& a) {
    if (rt->matrix && !rt->matrix_scan) return matrix_solve(rt);
    {
        ProfScope ps(rt, rt->matrix_scan ? "matrix_solve" : "rt_flux");
        DISPATCH_ROWS(launch_flux, rt, a);
        HX_LAUNCH_CHECK(rt->ctx);
    }
    return rt->f.debug == 1 ? count_negative_fluxes(rt) : 0;   // (debug = 1 keeps the solve's stores on: rt_create_into)
}

static int rt_step_kernels(hx_rt* rt, int itervalue, int step_temperature, bool nodes_done) {
    hx_context* ctx = rt->ctx;
    rt->solve_serial++;
    KArgs a = make_args(rt);
    if (!nodes_done) {
        ProfScope ps(rt, "rt_nodes");
        dim3 grid(hx_cdiv(rt->X, 32), hx_cdiv(rt->H + 3, 32), rt->C);
        k_rt_nodes<<<grid, 256, 0, ctx->stream>>>(a);
        HX_LAUNCH_CHECK(ctx);
    }
    {
        int rc = spectral_fluxes(rt, a);
        if (rc) return rc;
    }
    // (the two levels of the wavelength sum stay two launches: merged into one -- the workgroup that draws a column's last ticket
    // going on with the second level and the temperature step -- measured slower with device-scope fences and not bit-identical
    // without them, profiles/r06_ab_totals_merged.txt)
    {
        ProfScope ps(rt, "rt_totals_a");
        k_rt_totals_a<<<dim3(rt->nchunk, rt->C), 256, 0, ctx->stream>>>(a);
        HX_LAUNCH_CHECK(ctx);
    }
    if (rt->entr_kappa && rt->cols[0].physical_tstep != 0 && itervalue % 10 == 0) {   // computation.py:921-923
        int rc = kappa_cp_from_table(rt, false);    // (reads temperatures and pressures, not the fluxes: before or behind the first level alike)
        if (rc) return rc;
    }
    {
        ProfScope ps(rt, "rt_totals_b");
        TotalsBArgs q;
        q.a = a;
        memset(&q.rt, 0, sizeof(q.rt));
        q.rt.F_net_diff = rt->F_net_diff;
        q.rt.tlay = rt->T_lay;
        q.rt.play = rt->p_lay;
        q.rt.pint = rt->p_int;
        q.rt.abrt = rt->abort_flags;
        q.rt.T_store = rt->T_store;
        q.rt.deltat_prefactor = rt->prefactor;
        q.rt.F_add_heat_lay = rt->F_add_heat_lay;
        q.rt.F_add_heat_sum = rt->F_add_heat_sum;
        q.rt.F_smooth = rt->F_smooth;
        q.rt.F_smooth_sum = rt->F_smooth_sum;
        q.rt.c_p_lay = rt->c_p_lay;
        q.rt.conv_count = rt->conv_count;
        q.rt.itervalue = itervalue;
        q.rt.nlayer = rt->L;
        q.rt.smooth = rt->f.smooth;
        q.rt.dim = rt->d.plancktable_dim;
        q.rt.step = rt->d.plancktable_step;
        q.step_temperature = step_temperature;
        q.done_w = rt->done;
        q.iters_done = rt->iters_done;
        q.iter_dev = rt->iter_dev;
        k_rt_totals_b<<<rt->C, 1024, 0, ctx->stream>>>(q);
        HX_LAUNCH_CHECK(ctx);
    }
    return 0;
}

__global__ void k_rt_set_iteration(int* iter_dev, int next) { iter_dev[0] = next; }

// the device's iteration counter shows `itervalue` as the next iteration (one tiny launch, only when the host's calls were
// not consecutive: the first iteration of a loop, a restart, a step behind the convection loop)
static int sync_iteration_counter(hx_rt* rt, int itervalue) {
    if (rt->iter_dev_expected == itervalue) return 0;
    k_rt_set_iteration<<<1, 1, 0, rt->ctx->stream>>>(rt->iter_dev, itervalue);
    HX_LAUNCH_CHECK(rt->ctx);
    rt->iter_dev_expected = itervalue;
    return 0;
}

int hx_rt_step(hx_rt* rt, int itervalue, int step_temperature) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    int rc = sync_iteration_counter(rt, itervalue);
    if (rc) return rc;
    bool nodes_done = false;
    if (itervalue % 10 == 0 || !rt->refreshed) {  // computation.py:860
        rc = hx_rt_refresh(rt);                    // (also evaluates the nodes for this iteration)
        if (rc) return rc;
        nodes_done = true;
    }
    rc = rt_step_kernels(rt, itervalue, step_temperature, nodes_done);
    rt->iter_dev_expected = rc ? -1 : itervalue + 1;   // k_rt_nodes has moved the counter on
    return rc;
}

// ---- convection loop (reference computation.py:992-1174) -------------------------------------------------
namespace {

ConvKArgs make_conv_args(hx_rt* rt, int itervalue) {
    ConvKArgs c;
    memset(&c, 0, sizeof(c));
    c.L = rt->L; c.C = rt->C; c.itervalue = itervalue;
    c.colpar = rt->colpar;
    c.T_lay = rt->T_lay; c.p_lay = rt->p_lay; c.p_int = rt->p_int;
    c.kappa_lay = rt->kappa_lay; c.kappa_int = rt->kappa_int;
    c.c_p = rt->c_p_lay; c.mmm_lay = rt->mmm_lay;
    c.F_add_heat_sum = rt->F_add_heat_sum; c.F_smooth_sum = rt->F_smooth_sum;
    c.F_down_tot = rt->F_down_tot; c.F_up_tot = rt->F_up_tot; c.F_net = rt->F_net;
    c.conv_unstable = rt->conv_unstable; c.conv_layer = rt->conv_layer; c.marked_red = rt->marked_red;
    c.dampara = rt->dampara;
    c.done = rt->done;
    return c;
}

}  // namespace

namespace {

// kappa_lay, kappa_int and c_p_lay from the table at the current temperatures (computation.py:199-250)
int kappa_cp_from_table(hx_rt* rt, bool refresh_T_int) {
    hx_context* ctx = rt->ctx;
    const int L = rt->L, I = rt->I;
    if (refresh_T_int) {
        k_rt_tint<<<dim3(hx_cdiv(I, 64), rt->C), 64, 0, ctx->stream>>>(rt->T_lay, rt->T_int, L, rt->done);
        HX_LAUNCH_CHECK(ctx);
    }
    for (int c = 0; c < rt->C; c++) {
        const double* T_lay = rt->T_lay + (size_t)c * (L + 1);
        const double* T_int = rt->T_int + (size_t)c * I;
        const double* p_lay = rt->p_lay + (size_t)c * L;
        const double* p_int = rt->p_int + (size_t)c * I;
        int rc = hx_kappa_interpol(ctx, T_lay, rt->entr_temp, p_lay, rt->entr_press, rt->kappa_lay + (size_t)c * L,
                                   rt->entr_kappa, rt->entr_npress, rt->entr_ntemp, L);
        if (rc) return rc;
        rc = hx_kappa_interpol(ctx, T_int, rt->entr_temp, p_int, rt->entr_press, rt->kappa_int + (size_t)c * I,
                               rt->entr_kappa, rt->entr_npress, rt->entr_ntemp, I);
        if (rc) return rc;
        rc = hx_cp_interpol(ctx, T_lay, rt->entr_temp, p_lay, rt->entr_press, rt->c_p_lay + (size_t)c * L,
                            rt->entr_c_p, rt->entr_npress, rt->entr_ntemp, L);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

// kappa and c_p of every column at its current temperatures (what interpolate_kappa_and_cp does for one column,
// computation.py:199-250); needs hx_rt_set_kappa_table
int hx_rt_kappa_cp_refresh(hx_rt* rt) {
    if (!rt) return HX_E_ARG;
    HX_REQUIRE(rt->ctx, rt->entr_kappa != nullptr, HX_E_STATE, "no kappa / c_p table set");
    return kappa_cp_from_table(rt, true);
}

int hx_rt_set_kappa_table(hx_rt* rt, const double* entr_temp, int entr_ntemp, const double* entr_press,
                          int entr_npress, const double* entr_kappa, const double* entr_c_p) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    HX_REQUIRE(rt->ctx, entr_ntemp > 1 && entr_npress > 1 && entr_temp && entr_press && entr_kappa && entr_c_p,
               HX_E_ARG, "hx_rt_set_kappa_table: need a (T, P) grid of at least 2 x 2");
    const size_t n = (size_t)entr_ntemp * entr_npress;
    RT_ALLOC(rt->entr_temp, entr_ntemp); RT_ALLOC(rt->entr_press, entr_npress);
    RT_ALLOC(rt->entr_kappa, n); RT_ALLOC(rt->entr_c_p, n);
    int rc = h2d(rt, rt->entr_temp, entr_temp, entr_ntemp * 8);
    rc |= h2d(rt, rt->entr_press, entr_press, entr_npress * 8);
    rc |= h2d(rt, rt->entr_kappa, entr_kappa, n * 8);
    rc |= h2d(rt, rt->entr_c_p, entr_c_p, n * 8);
    rt->entr_ntemp = entr_ntemp;
    rt->entr_npress = entr_npress;
    return rc;
}

// first half of one iteration: [every 10th: mean molecular mass at the current temperatures] and the convective
// adjustment of the temperature profile (computation.py:1027-1047)
int hx_rt_conv_adjust(hx_rt* rt, int itervalue) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    hx_context* ctx = rt->ctx;
    HX_REQUIRE(ctx, rt->have_grid && rt->have_tables, HX_E_STATE, "hx_rt_conv_adjust before the tables are set");
    if (itervalue % 10 == 0) {   // computation.py:1030-1036: mu of the profile BEFORE the adjustment
        if (rt->d.nspecies == 0) {
            for (int c = 0; c < rt->C; c++) {
                int rc = hx_meanmolmass_interpol(ctx, rt->T_lay + (size_t)c * (rt->L + 1), rt->ktemp,
                                                 rt->mmm_lay + (size_t)c * rt->I, rt->opac_meanmass,
                                                 rt->p_lay + (size_t)c * rt->L, rt->kpress, rt->d.npress,
                                                 rt->d.ntemp, rt->L);
                if (rc) return rc;
            }
        } else {
            int rc = upload_species_table(rt);
            if (rc) return rc;
            k_rt_mmm_from_vmr<<<dim3(hx_cdiv(rt->L, 64), rt->C), 64, 0, ctx->stream>>>(
                (const SpeciesDev*)rt->species_dev, rt->d.nspecies, rt->vmr_lay, rt->mmm_lay, rt->L, rt->I, rt->T_lay,
                rt->p_lay, rt->ktemp, rt->d.ntemp, rt->kpress, rt->d.npress);
            HX_LAUNCH_CHECK(ctx);
        }
    }
    if (rt->entr_kappa) {   // computation.py:1037: kappa and c_p of the profile before the adjustment
        int rc = kappa_cp_from_table(rt, true);
        if (rc) return rc;
    }
    ProfScope ps(rt, "rt_conv_adjust");
    const size_t shmem = conv_smem_bytes(rt->L);
    if (shmem > 48 * 1024 && !rt->conv_shmem_raised) {
        (void)hipFuncSetAttribute((const void*)k_rt_conv_adjust, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        (void)hipFuncSetAttribute((const void*)k_rt_totals_c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        rt->conv_shmem_raised = true;
    }
    k_rt_conv_adjust<<<rt->C, 256, shmem, ctx->stream>>>(make_conv_args(rt, itervalue));
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

// second half: interface temperatures and Planck function of the adjusted profile, [every 10th: refresh], the
// two-stream sweeps, totals, convective-layer marking, equilibrium test and -- unless that ends the loop -- the
// temperature step (computation.py:1048-1145).  A column whose loop has ended is frozen (`done`, `iters_done`).
int hx_rt_conv_advance(hx_rt* rt, int itervalue) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    hx_context* ctx = rt->ctx;
    bool nodes_done = false;
    if (itervalue % 10 == 0 || !rt->refreshed) {
        int rc = hx_rt_refresh(rt);
        if (rc) return rc;
        nodes_done = true;
    }
    KArgs a = make_args(rt);
    rt->solve_serial++;
    rt->iter_dev_expected = -1;   // (the convection loop passes its own iteration index; k_rt_nodes still counts)
    if (!nodes_done) {
        ProfScope ps(rt, "rt_nodes");
        dim3 grid(hx_cdiv(rt->X, 32), hx_cdiv(rt->H + 3, 32), rt->C);
        k_rt_nodes<<<grid, 256, 0, ctx->stream>>>(a);
        HX_LAUNCH_CHECK(ctx);
    }
    {
        int rc = spectral_fluxes(rt, a);
        if (rc) return rc;
    }
    {
        ProfScope ps(rt, "rt_totals_a");
        k_rt_totals_a<<<dim3(rt->nchunk, rt->C), 256, 0, ctx->stream>>>(a);
        HX_LAUNCH_CHECK(ctx);
    }
    if (rt->entr_kappa) {   // computation.py:1088: once more for the adjusted profile, before the layers are marked
        int rc = kappa_cp_from_table(rt, false);
        if (rc) return rc;
    }
    {
        ProfScope ps(rt, "rt_totals_c");
        TotalsCArgs q;
        q.a = a;
        q.cv = make_conv_args(rt, itervalue);
        memset(&q.ct, 0, sizeof(q.ct));
        q.ct.F_net_diff = rt->F_net_diff;
        q.ct.tlay = rt->T_lay;
        q.ct.play = rt->p_lay;
        q.ct.pint = rt->p_int;
        q.ct.T_store = rt->T_store;
        q.ct.deltat_prefactor = rt->prefactor;
        q.ct.F_add_heat_lay = rt->F_add_heat_lay;
        q.ct.F_smooth = rt->F_smooth;
        q.ct.F_smooth_sum = rt->F_smooth_sum;
        q.ct.nlayer = rt->L;
        q.ct.itervalue = itervalue;
        q.ct.smooth = rt->f.smooth;
        q.done_w = rt->done;
        q.iters_done = rt->iters_done;
        q.physical_tstep_on = rt->cols[0].physical_tstep != 0 ? 1 : 0;
        k_rt_totals_c<<<rt->C, 1024, conv_smem_bytes(rt->L), ctx->stream>>>(q);
        HX_LAUNCH_CHECK(ctx);
    }
    return 0;
}

int hx_rt_conv_run(hx_rt* rt, int itervalue, int nsteps) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    for (int n = 0; n < nsteps; n++) {
        int rc = hx_rt_conv_adjust(rt, itervalue + n);
        if (rc) return rc;
        rc = hx_rt_conv_advance(rt, itervalue + n);
        if (rc) return rc;
    }
    return 0;
}

// Iterations as hipGraphs.  With the iteration index on the device the kernels' arguments never change, so the launches
// between two opacity refreshes are captured once and replayed with one call -- nine refresh-free iterations (entered
// mid-decade), or the whole decade with its refresh (entered at a refresh boundary, round 5).  Where an iteration is a few
// microseconds of GPU work (the reference's default problem is 386 bins x 105 layers; BASELINE config 1 is 300 x 50) the
// host's four launches per iteration were what bounded the loop; on large grids the device time is the same either way
// (config 2: 0.386 ms per iteration with and without, same-box A/B) and the host does one call per ten iterations instead of
// forty-odd launches -- which is what eight ranks sharing a few host cores need.  On for every grid since round 5.
constexpr int GRAPH_ITERATIONS = 9;

static bool graph_wanted(hx_rt* rt) {
    if (rt->use_graph < 0) {
        rt->use_graph = 1;
        if (const char* e = getenv("HELIOS_RT_GRAPH")) rt->use_graph = atoi(e) != 0 ? 1 : 0;   // tuning knob
    }
    // not while the event profiler brackets every launch, not with the per-iteration host decisions of the time-stepped
    // kappa refresh (computation.py:921-923)
    bool time_stepped = false;   // any column of the batch
    for (const auto& c : rt->cols) time_stepped = time_stepped || c.physical_tstep != 0;
    // (the matrix method's direct solve replays like the sweeps; its per-stage form -- HELIOS_RT_MATRIX=stage -- launches
    // column by column and stays outside)
    return rt->use_graph == 1 && !rt->profiling && !(rt->matrix && !rt->matrix_scan) && !(rt->entr_kappa && time_stepped);
}

// `with_refresh`: the whole decade -- the opacity refresh with the iteration that carries it, then the nine refresh-free
// ones -- as ONE graph (round 5).  The refresh is launches with fixed arguments too (its host decisions -- species tables
// uploaded, dynamic-LDS limits raised -- are taken by the first, uncaptured refresh); at 300 x 50 its ~19 launches cost the
// host more than the nine replayed iterations cost the device.
static int build_iteration_graph(hx_rt* rt, bool with_refresh) {
    hx_context* ctx = rt->ctx;
    hipGraphExec_t& exec = with_refresh ? rt->decade_graph : rt->iter_graph;
    if (exec) {
        (void)hipGraphExecDestroy(exec);
        exec = nullptr;
    }
    HX_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    if (with_refresh) {
        rc = hx_rt_refresh(rt);                               // (k_rt_nodes inside moves the device's iteration counter on)
        if (!rc) rc = rt_step_kernels(rt, 1, 1, true);        // (the index comes from the device)
    }
    for (int n = 0; n < GRAPH_ITERATIONS && !rc; n++) rc = rt_step_kernels(rt, 1, 1, false);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        // this batch iterates launch by launch from here on: hx_rt_run goes on through hx_rt_step, where a launch that
        // really fails fails again, outside a capture and with its own message
        rt->use_graph = 0;
        (void)hipGetLastError();
        return 0;
    }
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) {
        exec = nullptr;
        rt->use_graph = 0;
        (void)hipGetLastError();
        return 0;
    }
    // each capture remembers the generation of arguments it holds: building one does not vouch for the other
    (with_refresh ? rt->decade_graph_gen : rt->iter_graph_gen) = rt->graph_gen;
    (with_refresh ? rt->decade_graph_builds : rt->iter_graph_builds)++;
    return 0;
}

int hx_rt_run(hx_rt* rt, int itervalue, int nsteps) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    for (int n = 0; n < nsteps;) {
        const int it = itervalue + n;
        const bool decade = it % 10 == 0 && nsteps - n >= GRAPH_ITERATIONS + 1 && (rt->d.nspecies == 0 || !rt->species_dev_stale);
        if ((decade || (it % 10 == 1 && nsteps - n >= GRAPH_ITERATIONS)) && rt->refreshed && graph_wanted(rt)) {
            hipGraphExec_t& exec = decade ? rt->decade_graph : rt->iter_graph;
            if ((decade ? rt->decade_graph_gen : rt->iter_graph_gen) != rt->graph_gen || !exec) {   // a setter ran since its capture
                int rc = build_iteration_graph(rt, decade);
                if (rc) return rc;
            }
            if (exec) {
                int rc = sync_iteration_counter(rt, it);
                if (rc) return rc;
                HX_HIP(rt->ctx, hipGraphLaunch(exec, rt->ctx->stream));
                rt->solve_serial++;
                (decade ? rt->decade_graph_replays : rt->iter_graph_replays)++;
                const int done = GRAPH_ITERATIONS + (decade ? 1 : 0);
                rt->iter_dev_expected = it + done;
                n += done;
                continue;
            }
        }
        int rc = hx_rt_step(rt, it, 1);
        if (rc) return rc;
        n++;
    }
    return 0;
}

int hx_rt_converged_layers(hx_rt* rt, int* out_counts) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    return hx_d2h(rt->ctx, out_counts, rt->conv_count, rt->C * sizeof(int));
}

// ---- read-back in the reference's layouts ------------------------------------------------------
namespace {

int get_plain(hx_rt* rt, const void* dptr, size_t bytes, void* out, size_t out_bytes) {
    if (bytes != out_bytes)
        return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: buffer is %zu bytes, array has %zu", out_bytes, bytes);
    return hx_d2h(rt->ctx, out, dptr, bytes);
}

// internal band layout [x][i] -> reference [x + X*i]
int get_band(hx_rt* rt, const double* dptr, void* out, size_t out_bytes) {
    const size_t X = rt->X, I = rt->I;
    if (out_bytes != X * I * 8) return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: wrong buffer size");
    std::vector<double> tmp(X * I);
    int rc = hx_d2h(rt->ctx, tmp.data(), dptr, X * I * 8);
    if (rc) return rc;
    double* o = (double*)out;
    for (size_t x = 0; x < X; x++)
        for (size_t i = 0; i < I; i++) o[x + X * i] = tmp[x * I + i];
    return 0;
}

// flux tiles -> reference wg layout.  `interface_nodes`: pick the even (interface) or odd (centre)
// nodes; `up`: tile row r of lane j holds the flux at node h+1 (up) or h (down), h = j*ROWS + r.
int get_flux_wg(hx_rt* rt, int col, const double* tiles, const double* bc, bool up, bool interface_nodes,
                void* out, size_t out_bytes) {
    const TileGeom& g = rt->g;
    const size_t X = rt->X, Y = rt->Y, I = rt->I, nc = X * Y;
    if (out_bytes != nc * I * 8) return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: wrong buffer size");
    std::vector<double> t(g.flux_elems_per_col), b(nc);
    int rc = hx_d2h(rt->ctx, t.data(), tiles + (size_t)col * g.flux_elems_per_col, t.size() * 8);
    if (rc) return rc;
    if (bc) {
        rc = hx_d2h(rt->ctx, b.data(), bc + (size_t)col * nc, nc * 8);
        if (rc) return rc;
    }
    double* o = (double*)out;
    std::fill(o, o + nc * I, 0.0);
    for (int blk = 0; blk < g.nblk; blk++) {
        const int bx = blk / g.nparts, part = blk % g.nparts;
        for (int s = 0; s < g.G; s++) {
            const int xl = s / g.ypb, yl = s % g.ypb;
            const size_t x = (size_t)bx * g.nxb + xl, y = (size_t)part * g.ypb + yl;
            if (x >= X) continue;
            const size_t c = y + Y * x;
            for (int j = 0; j < g.k; j++) {
                const int tid = s * g.k + j, wv = tid / 64, lane = tid % 64;
                const double* tile = t.data() + ((size_t)blk * g.NW + wv) * g.ROWS * 64;
                for (int r = 0; r < g.ROWS; r++) {
                    const int h = j * g.ROWS + r;
                    if (h >= rt->H) continue;
                    const int node = up ? h + 1 : h;
                    if (rt->f.iso) {  // isothermal layers: every node is an interface, there are no centre fluxes
                        if (interface_nodes) o[c + nc * node] = tile[plane_off(r, lane, g.ROWS)];
                        continue;
                    }
                    const bool is_int = (node % 2) == 0;
                    if (is_int != interface_nodes) continue;
                    o[c + nc * (node / 2)] = tile[plane_off(r, lane, g.ROWS)];
                }
            }
            if (up && interface_nodes && bc) o[c] = b[c];  // U at node 0 = BOA boundary value
        }
    }
    return 0;
}

}  // namespace

// rebuild opac_wg_lay/int (reference layout) from the temperatures of the last refresh
static int materialize_opac(hx_rt* rt) {
    if (!rt->opac_stale) return 0;
    hx_context* ctx = rt->ctx;
    const size_t nc = (size_t)rt->X * rt->Y, wgI = nc * rt->I, bandI = (size_t)rt->X * rt->I;
    for (int c = 0; c < rt->C; c++) {
        int rc = hx_opac_interpol(ctx, rt->T_lay_ref + (size_t)c * (rt->L + 1), rt->ktemp, rt->p_lay + (size_t)c * rt->L,
                                  rt->kpress, rt->opac_k, rt->opac_wg_lay + c * wgI, rt->opac_scat_cross,
                                  rt->scat_cross_lay + c * bandI, rt->d.npress, rt->d.ntemp, rt->Y, rt->X, rt->L);
        if (rc) return rc;
        rc = hx_opac_interpol(ctx, rt->T_int_ref + (size_t)c * rt->I, rt->ktemp, rt->p_int + (size_t)c * rt->I,
                              rt->kpress, rt->opac_k, rt->opac_wg_int + c * wgI, rt->opac_scat_cross,
                              rt->scat_cross_int + c * bandI, rt->d.npress, rt->d.ntemp, rt->Y, rt->X, rt->I);
        if (rc) return rc;
    }
    rt->opac_stale = false;
    return 0;
}

int hx_rt_get(hx_rt* rt, int col, const char* name, void* out, size_t out_bytes) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    HX_REQUIRE(rt->ctx, col >= 0 && col < rt->C, HX_E_ARG, "column index out of range");
    if (strncmp(name, "opac_wg_", 8) == 0) {
        int rc = materialize_opac(rt);
        if (rc) return rc;
    }
    const size_t X = rt->X, Y = rt->Y, L = rt->L, I = rt->I, nc = X * Y, c = col;
    const std::string n(name);
    if (n == "T_lay") return get_plain(rt, rt->T_lay + c * (L + 1), (L + 1) * 8, out, out_bytes);
    if (n == "T_int") return get_plain(rt, rt->T_int + c * I, I * 8, out, out_bytes);
    if (n == "F_up_band") return get_band(rt, rt->F_up_band_n + c * X * I, out, out_bytes);
    if (n == "F_down_band") return get_band(rt, rt->F_down_band_n + c * X * I, out, out_bytes);
    if (n == "F_dir_band") return get_band(rt, rt->F_dir_band_n + c * X * I, out, out_bytes);
    if (n == "F_up_tot") return get_plain(rt, rt->F_up_tot + c * I, I * 8, out, out_bytes);
    if (n == "F_down_tot") return get_plain(rt, rt->F_down_tot + c * I, I * 8, out, out_bytes);
    if (n == "F_net") return get_plain(rt, rt->F_net + c * I, I * 8, out, out_bytes);
    if (n == "F_net_diff") return get_plain(rt, rt->F_net_diff + c * L, L * 8, out, out_bytes);
    if (n == "abort") return get_plain(rt, rt->abort_flags + c * (L + 1), (L + 1) * 4, out, out_bytes);
    if (n == "delta_t_prefactor") return get_plain(rt, rt->prefactor + c * (L + 1), (L + 1) * 8, out, out_bytes);
    if (n == "T_store") return get_plain(rt, rt->T_store + c * (L + 1), (L + 1) * 8, out, out_bytes);
    if (n == "delta_z_lay") return get_plain(rt, rt->delta_z + c * L, L * 8, out, out_bytes);
    if (n == "z_lay") return get_plain(rt, rt->z_lay + c * L, L * 8, out, out_bytes);
    if (n == "meanmolmass_lay") return get_plain(rt, rt->mmm_lay + c * I, L * 8, out, out_bytes);
    if (n == "meanmolmass_int") return get_plain(rt, rt->mmm_int + c * I, I * 8, out, out_bytes);
    if (n == "opac_wg_lay") return get_plain(rt, rt->opac_wg_lay + c * nc * I, nc * L * 8, out, out_bytes);
    if (n == "opac_wg_int") return get_plain(rt, rt->opac_wg_int + c * nc * I, nc * I * 8, out, out_bytes);
    if (n == "scat_cross_lay") return get_plain(rt, rt->scat_cross_lay + c * X * I, X * L * 8, out, out_bytes);
    if (n == "scat_cross_int") return get_plain(rt, rt->scat_cross_int + c * X * I, X * I * 8, out, out_bytes);
    if (n == "g_0_tot_lay") return get_plain(rt, rt->g0_tot_lay + c * X * I, X * L * 8, out, out_bytes);
    if (n == "g_0_tot_int") return get_plain(rt, rt->g0_tot_int + c * X * I, X * I * 8, out, out_bytes);
    // mixing-ratio profiles as the last refresh used them, [nspecies][ninterface] (layer rows: the first nlayer entries)
    if ((n == "vmr_lay" || n == "vmr_int") && rt->d.nspecies > 0) {
        const size_t S = rt->d.nspecies;
        return get_plain(rt, (n == "vmr_lay" ? rt->vmr_lay : rt->vmr_int) + c * S * I, S * I * 8, out, out_bytes);
    }
    if (n == "iters_done") return get_plain(rt, rt->iters_done + c, 4, out, out_bytes);
    if (n == "flux_launch_policy") {   // host-side: {launch order back and forth (0/1), MiB of up-flux state kept cached}
        HX_REQUIRE(rt->ctx, out_bytes == 2 * sizeof(double), HX_E_ARG, "flux_launch_policy is two doubles");
        const double v[2] = {rt->serpentine ? 1.0 : 0.0, rt->serpentine ? rt->state_cache_mb : 0.0};
        memcpy(out, v, sizeof(v));
        return 0;
    }
    if (n == "graph_replays") {   // host-side: {replays of the nine refresh-free iterations, replays of a whole decade, graphs in use (0/1)}
        HX_REQUIRE(rt->ctx, out_bytes == 3 * sizeof(double), HX_E_ARG, "graph_replays is three doubles");
        const double v[3] = {(double)rt->iter_graph_replays, (double)rt->decade_graph_replays, rt->use_graph == 1 ? 1.0 : 0.0};
        memcpy(out, v, sizeof(v));
        return 0;
    }
    if (n == "graph_builds") {   // host-side: {captures of the nine-iteration graph, captures of the decade graph}
        HX_REQUIRE(rt->ctx, out_bytes == 2 * sizeof(double), HX_E_ARG, "graph_builds is two doubles");
        const double v[2] = {(double)rt->iter_graph_builds, (double)rt->decade_graph_builds};
        memcpy(out, v, sizeof(v));
        return 0;
    }
    if (n == "planck_grid")
        return get_plain(rt, rt->planck_grid, (size_t)(rt->d.plancktable_dim + 1) * X * 8, out, out_bytes);
    if (n == "done") return get_plain(rt, rt->done + c, 4, out, out_bytes);
    if (n == "conv_layer") return get_plain(rt, rt->conv_layer + c * (L + 1), (L + 1) * 4, out, out_bytes);
    if (n == "conv_unstable") return get_plain(rt, rt->conv_unstable + c * (L + 1), (L + 1) * 4, out, out_bytes);
    if (n == "marked_red") return get_plain(rt, rt->marked_red + c * (L + 1), (L + 1) * 4, out, out_bytes);
    if (n == "F_smooth_sum") return get_plain(rt, rt->F_smooth_sum + c * L, L * 8, out, out_bytes);
    if (n == "kappa_lay") return get_plain(rt, rt->kappa_lay + c * L, L * 8, out, out_bytes);
    if (n == "kappa_int") return get_plain(rt, rt->kappa_int + c * (L + 1), (L + 1) * 8, out, out_bytes);
    if (n == "c_p_lay") return get_plain(rt, rt->c_p_lay + c * L, L * 8, out, out_bytes);
    if (n == "F_add_heat_lay") return get_plain(rt, rt->F_add_heat_lay + c * L, L * 8, out, out_bytes);
    if (n == "F_add_heat_sum") return get_plain(rt, rt->F_add_heat_sum + c * L, L * 8, out, out_bytes);
    if (n == "planckband_lay" || n == "planckband_int") {
        // from the node array Bn[x][H+3]: layers = odd nodes, then star, surface; interfaces = even
        const bool lay = n == "planckband_lay";
        const size_t NN = rt->H + 3, per = lay ? L + 2 : I;
        if (out_bytes != X * per * 8) return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: wrong buffer size");
        std::vector<double> tmp(X * NN);
        int rc = hx_d2h(rt->ctx, tmp.data(), rt->Bn + c * X * NN, X * NN * 8);
        if (rc) return rc;
        double* o = (double*)out;
        for (size_t x = 0; x < X; x++) {
            if (lay) {
                for (size_t i = 0; i < L; i++) o[i + x * per] = tmp[x * NN + (rt->f.iso ? i : 2 * i + 1)];
                o[L + x * per] = tmp[x * NN + rt->H + 1];
                o[L + 1 + x * per] = tmp[x * NN + rt->H + 2];
            } else {  // isothermal layers: the reference does not compute interface values (computation.py:315-329)
                for (size_t i = 0; i < I; i++) o[i + x * per] = rt->f.iso ? 0.0 : tmp[x * NN + 2 * i];
            }
        }
        return 0;
    }
    if (rt->matrix && !rt->matrix_scan && (n == "F_up_wg" || n == "F_down_wg" || n == "Fc_up_wg" || n == "Fc_down_wg")) {
        // the solver's own arrays, already in the reference's layout (centre fluxes: nlayer slabs, the rest stays zero)
        const MatrixArrays& m = rt->mx;
        if (out_bytes != nc * I * 8) return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: wrong buffer size");
        const bool centre = n[1] == 'c';
        const double* src = n == "F_up_wg" ? m.F_up : n == "F_down_wg" ? m.F_down : n == "Fc_up_wg" ? m.Fc_up : m.Fc_down;
        memset(out, 0, out_bytes);
        if (centre && rt->f.iso) return 0;
        return hx_d2h(rt->ctx, out, src + c * nc * I, nc * (centre ? L : I) * 8);
    }
    if (rt->matrix_scan && (n == "F_up_wg" || n == "F_down_wg" || n == "Fc_up_wg" || n == "Fc_down_wg")) {
        // The direct solve keeps no spectral fluxes between iterations (nothing reads them: the band fluxes are summed inside
        // the kernel).  Asked for, they are the last solve's: the coefficient tiles and node Planck values it read are still in
        // place, so the same launch -- this time with its stores, and for every column, also those whose loop has ended --
        // reproduces them bit for bit (and rewrites the same band fluxes).
        // ONE such launch serves every name and column asked for until the next solve or setter (`solve_serial`).
        HX_REQUIRE(rt->ctx, rt->refreshed, HX_E_STATE, "no iteration has been run yet");
        const bool first = !rt->Dtile;
        if (!rt->Dtile) RT_ALLOC(rt->Dtile, (size_t)rt->C * rt->g.flux_elems_per_col);
        if (!rt->zero_flags) RT_ALLOC(rt->zero_flags, (size_t)rt->C);
        if (first || rt->matrix_tiles_serial != rt->solve_serial) {
            const bool kd = rt->keep_down, ks = rt->matrix_keep_state;
            rt->keep_down = true;
            rt->matrix_keep_state = true;
            KArgs a = make_args(rt);
            a.done = rt->zero_flags;
            DISPATCH_ROWS(launch_flux, rt, a);
            rt->keep_down = kd;
            rt->matrix_keep_state = ks;
            HX_LAUNCH_CHECK(rt->ctx);
            rt->matrix_tiles_serial = rt->solve_serial;
        }
        if (first) rt->graph_gen++;   // the captures were taken without the down-flux tiles' address
    }
    if (n == "F_up_wg") return get_flux_wg(rt, col, rt->Utile, rt->U0, true, true, out, out_bytes);
    if (n == "Fc_up_wg") return get_flux_wg(rt, col, rt->Utile, nullptr, true, false, out, out_bytes);
    if (n == "F_down_wg" || n == "Fc_down_wg") {
        HX_REQUIRE(rt->ctx, (rt->keep_down || rt->matrix_scan) && rt->Dtile, HX_E_STATE,
                   "down-flux tiles are only kept after hx_rt_set_state(rt, col, \"keep_down\", ...)");
        int rc = get_flux_wg(rt, col, rt->Dtile, nullptr, false, n == "F_down_wg", out, out_bytes);
        if (rc || n != "F_down_wg") return rc;
        // TOA boundary value D[H] is not a tile row: (1-dir_beam) f (R*/a)^2 pi B*  (kernels.cu:1601)
        std::vector<double> bs(X);
        rc = hx_d2h(rt->ctx, bs.data(), rt->Bstar + c * X, X * 8);
        if (rc) return rc;
        const hx_rt_column& cp = rt->cols[col];
        double* o = (double*)out;
        const double rs = cp.R_star / cp.a;
        for (size_t x = 0; x < X; x++)
            for (size_t y = 0; y < Y; y++)
                o[y + Y * x + nc * L] = (1.0 - rt->f.dir_beam) * cp.f_factor * (rs * rs) * HX_PI * bs[x];
        return 0;
    }
    if (n == "F_dir_wg" || n == "Fc_dir_wg") {
        if (!rt->f.dir_beam) {
            if (out_bytes != nc * I * 8) return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: wrong buffer size");
            memset(out, 0, out_bytes);
            return 0;
        }
        return get_plain(rt, (n == "F_dir_wg" ? rt->F_dir_wg : rt->Fc_dir_wg) + c * nc * I, nc * I * 8, out,
                         out_bytes);
    }
    return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_get: unknown array name '%s'", name);
}

int hx_rt_set_state(hx_rt* rt, int col, const char* name, const void* in, size_t in_bytes) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    rt_touch(rt);            // a captured iteration graph holds the arguments of before this call
    const std::string n(name);
    if (n == "keep_down") {
        if (in_bytes != 4) return hx_fail(rt->ctx, HX_E_ARG, "keep_down expects one int32");
        const int v = *(const int*)in;
        if (v && !rt->Dtile) RT_ALLOC(rt->Dtile, (size_t)rt->C * rt->g.flux_elems_per_col);
        rt->keep_down = v != 0;
        return 0;
    }
    if (n == "planck_grid") {  // a table built elsewhere (tests: the reference's own), shared by all columns
        const size_t want = (size_t)(rt->d.plancktable_dim + 1) * rt->X * 8;
        if (in_bytes != want) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for planck_grid");
        int rc = h2d(rt, rt->planck_grid, in, in_bytes);
        for (int c = 0; c < rt->C && !rc; c++)  // its last row is the stellar spectrum the sweeps use
            rc = hx_d2d(rt->ctx, rt->Bstar + (size_t)c * rt->X,
                        rt->planck_grid + (size_t)rt->d.plancktable_dim * rt->X, (size_t)rt->X * 8);
        return rc;
    }
    if (n == "restart") {
        // every column back to the state of a fresh batch: flux state of the sweeps (which persists between iterations,
        // SURVEY.md Q9), time-step state, convergence flags.  Temperatures are the caller's (hx_rt_set_temperatures).
        if (in_bytes != 4) return hx_fail(rt->ctx, HX_E_ARG, "restart expects one int32");
        const size_t C = rt->C, Ln = rt->L, nc = (size_t)rt->X * rt->Y;
        struct { void* p; size_t bytes; } z[] = {
            {rt->Utile, C * rt->g.flux_elems_per_col * 8}, {rt->Dtile, rt->Dtile ? C * rt->g.flux_elems_per_col * 8 : 0},
            {rt->U0, C * nc * 8}, {rt->T_store, C * (Ln + 1) * 8}, {rt->prefactor, C * (Ln + 1) * 8},
            {rt->abort_flags, C * (Ln + 1) * 4}, {rt->done, C * 4}, {rt->iters_done, C * 4}, {rt->conv_count, C * 4},
            {rt->F_smooth, C * Ln * 8}, {rt->F_smooth_sum, C * Ln * 8}};
        for (auto& e : z)
            if (e.p && e.bytes) {
                hipError_t he = hipMemsetAsync(e.p, 0, e.bytes, rt->ctx->stream);
                if (he != hipSuccess) return hx_fail(rt->ctx, -(int)he, "hipMemset failed");
            }
        rt->refreshed = false;
        return 0;
    }
    int c0, c1, rc = for_cols(rt, col, &c0, &c1);
    if (rc) return rc;
    const size_t L = rt->L;
    for (int c = c0; c < c1; c++) {
        if (n == "T_lay") {
            if (in_bytes != (L + 1) * 8) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for T_lay");
            rc |= h2d(rt, rt->T_lay + c * (L + 1), in, in_bytes);
        } else if (n == "c_p_lay") {
            if (in_bytes != L * 8) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for c_p_lay");
            rc |= h2d(rt, rt->c_p_lay + c * L, in, in_bytes);
        } else if (n == "delta_t_prefactor" || n == "T_store") {
            if (in_bytes != (L + 1) * 8) return hx_fail(rt->ctx, HX_E_ARG, "wrong size");
            rc |= h2d(rt, (n == "T_store" ? rt->T_store : rt->prefactor) + c * (L + 1), in, in_bytes);
        } else if (n == "done") {
            if (in_bytes != 4) return hx_fail(rt->ctx, HX_E_ARG, "done expects one int32");
            rc |= h2d(rt, rt->done + c, in, 4);
        } else if (n == "kappa_lay") {
            if (in_bytes != L * 8) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for kappa_lay");
            rc |= h2d(rt, rt->kappa_lay + c * L, in, in_bytes);
        } else if (n == "kappa_int") {
            if (in_bytes != (L + 1) * 8) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for kappa_int");
            rc |= h2d(rt, rt->kappa_int + c * (L + 1), in, in_bytes);
        } else if (n == "conv_layer" || n == "conv_unstable") {
            if (in_bytes != (L + 1) * 4) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for %s", name);
            rc |= h2d(rt, (n == "conv_layer" ? rt->conv_layer : rt->conv_unstable) + c * (L + 1), in, in_bytes);
        } else if (n == "add_heat_dens") {
            if (in_bytes != L * 8) return hx_fail(rt->ctx, HX_E_ARG, "wrong size for add_heat_dens");
            if (!rt->add_heat_dens) RT_ALLOC(rt->add_heat_dens, (size_t)rt->C * L);
            rc |= h2d(rt, rt->add_heat_dens + c * L, in, in_bytes);
            rt->has_heating = true;
        } else if (n == "dampara") {
            if (in_bytes != 8) return hx_fail(rt->ctx, HX_E_ARG, "dampara expects one double (<= 0: automatic)");
            rc |= h2d(rt, rt->dampara + c, in, 8);
        } else {
            return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_set_state: unknown name '%s'", name);
        }
    }
    return rc;
}

int hx_rt_device_ptr(hx_rt* rt, int col, const char* name, void** out_dptr) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    HX_REQUIRE(rt->ctx, col >= 0 && col < rt->C && out_dptr, HX_E_ARG, "bad arguments");
    const size_t X = rt->X, Y = rt->Y, L = rt->L, I = rt->I, nc = X * Y, c = col;
    const std::string n(name);
    void* p = nullptr;
    if (n == "T_lay") p = rt->T_lay + c * (L + 1);
    else if (n == "T_int") p = rt->T_int + c * I;
    else if (n == "p_lay") p = rt->p_lay + c * L;
    else if (n == "p_int") p = rt->p_int + c * I;
    else if (n == "opac_wg_lay") { int rc = materialize_opac(rt); if (rc) return rc; p = rt->opac_wg_lay + c * nc * I; }
    else if (n == "opac_wg_int") { int rc = materialize_opac(rt); if (rc) return rc; p = rt->opac_wg_int + c * nc * I; }
    else if (n == "F_dir_wg" && rt->f.dir_beam) p = rt->F_dir_wg + c * nc * I;
    else if (n == "Fc_dir_wg" && rt->f.dir_beam) p = rt->Fc_dir_wg + c * nc * I;
    else if (n == "scat_cross_lay") p = rt->scat_cross_lay + c * X * I;
    else if (n == "scat_cross_int") p = rt->scat_cross_int + c * X * I;
    else if (n == "meanmolmass_lay") p = rt->mmm_lay + c * I;
    else if (n == "meanmolmass_int") p = rt->mmm_int + c * I;
    else if (n == "planck_grid") p = rt->planck_grid;
    // band fluxes in the internal layout [bin][interface] (hx_rt_get returns the reference's [interface][bin])
    else if (n == "F_up_band_n") p = rt->F_up_band_n + c * X * I;
    else if (n == "F_down_band_n") p = rt->F_down_band_n + c * X * I;
    else if (n == "F_net") p = rt->F_net + c * I;
    else if (n == "F_up_tot") p = rt->F_up_tot + c * I;
    else if (n == "F_down_tot") p = rt->F_down_tot + c * I;
    else if (n == "gauss_weight") p = rt->gauss_w;
    else if (n == "gauss_y") p = rt->gauss_y;
    else if (n == "opac_deltawave") p = rt->deltawave;
    else if (n == "opac_interwave") p = rt->interwave;
    else if (n == "abs_cross_all_clouds_lay") p = rt->cl_abs_lay + c * X * I;
    else if (n == "delta_col_upper") p = rt->dcol_u + c * L;
    else if (n == "delta_col_lower") p = rt->dcol_l + c * L;
    else return hx_fail(rt->ctx, HX_E_ARG, "hx_rt_device_ptr: unknown name '%s'", name);
    *out_dptr = p;
    return 0;
}

int hx_rt_flux_geometry(int nlayer, int iso, int dir_beam, int ny, int nbin, int ncol, int* out_lanes, int* out_rows) {
    TileGeom g;
    memset(&g, 0, sizeof(g));
    if (nlayer < 1 || ny < 1 || nbin < 1 || ncol < 1 || !out_lanes || !out_rows) return HX_E_ARG;
    if (!choose_geometry(iso ? nlayer : 2 * nlayer, ny, nbin, ncol, dir_beam, 0, g)) return HX_E_ARG;
    *out_lanes = g.k;
    *out_rows = g.ROWS;
    return 0;
}

int hx_rt_traffic_model(hx_rt* rt, double* step_alg, double* step_act, double* refresh_alg,
                        double* refresh_act) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    // algorithmic bytes per iteration and column
    const double X = rt->X, Y = rt->Y, L = rt->L;
    double S = 0;  // absorbers: the species that bring a k-table
    for (const Species& sp : rt->species) S += sp.absorbing ? 1.0 : 0.0;
    const double cl = rt->f.clouds ? 3.0 * (2 * L + 1) : 0.0;
    const double BE = 8.0 * X * (3.0 * Y * (2 * L + 1) + (2 * L + 1) + 2.0 * (2 * L + 3) + 3.0 * (L + 1) + cl);
    const double BT = 8.0 * Y * X * (2 * L + 1) * (S > 0 ? 4.0 * S + 1.0 : 5.0);
    // what this implementation actually moves (per column)
    const TileGeom& g = rt->g;
    const double tiles = (double)g.nblk * g.NW * 64.0 * g.ROWS * 8.0;  // one plane
    // (the matrix method's direct solve reads the same planes and keeps no flux state)
    const double state_planes = rt->matrix_scan ? (rt->matrix_keep_state ? 1.0 : 0.0) : 2.0;
    const double flux_k = tiles * (g.nplane + state_planes + (rt->keep_down ? 1.0 : 0.0))   // coef + U read/write
                          + 8.0 * X * (rt->H + 3) * 2.0                            // node Planck write+read
                          + 8.0 * X * 2.0 * (L + 1) * 2.0                          // band arrays w + r
                          + 8.0 * X * Y * 2.0;                                     // U0
    const double premixed = 8.0 * Y * X * (2 * L + 1) * 5.0;
    const double species = 8.0 * Y * X * (2 * L + 1) * (4.0 * S + 1.0);  // table corners of every absorber + one write
    const double coef_k = tiles * g.nplane + 8.0 * Y * X * (2 * L + 1);
    if (step_alg) *step_alg = BE * rt->C;
    if (step_act) *step_act = flux_k * rt->C;
    if (refresh_alg) *refresh_alg = BT * rt->C;
    if (refresh_act) *refresh_act = ((S > 0 ? species : premixed) + coef_k) * rt->C;
    return 0;
}

int hx_rt_profile(hx_rt* rt, int enable) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    if (!enable) ProfScope::flush(rt);
    rt->profiling = enable != 0;
    return 0;
}

int hx_rt_profile_read(hx_rt* rt, const char* kernel, double* out_avg_ms, int* out_count) {
    if (!rt) return HX_E_ARG;  // e.g. a call after hx_rt_destroy
    ProfScope::flush(rt);
    for (auto& acc : rt->prof_acc)
        if (acc.first == kernel) {
            if (out_avg_ms) *out_avg_ms = acc.second.second ? acc.second.first / acc.second.second : 0.0;
            if (out_count) *out_count = acc.second.second;
            return 0;
        }
    if (out_avg_ms) *out_avg_ms = 0.0;
    if (out_count) *out_count = 0;
    return 0;
}

}  // extern "C"
