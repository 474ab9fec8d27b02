// Per-stage entry points, part 1: Planck table, incident-energy correction, temperature / Planck /
// opacity-table interpolation.  Layouts are the reference's; thread maps are chosen so that a
// wavefront's 64 lanes touch consecutive addresses (the reference puts x on threadIdx.x although y is
// the fastest index of its wg arrays).
#include "two_stream.h"

using namespace hx;

namespace {

// ---------------------------------------------------------------------------------------------
// Planck table: tab[x + r*nbin] = (D/dlambda) * sum_{n=1}^{199} [Phi_n(y_top) - Phi_n(y_bot)],
//     Phi_n(y) = exp(-n y) (y^3/n + 3 y^2/n^2 + 6 y/n^3 + 6/n^4)
// (kernels.cu:95-105, :362-416).  8 001 rows x nbin bins x 199 terms, each with two `exp` and eight
// divisions in the reference's formula: 70 ms at 10 000 bins, 200 ms at 30 000 -- set-up time a run pays before its first
// iteration.  Same values bit for bit, a quarter of the instructions:
//   * a bin's upper edge is its neighbour's lower edge and y(edge, T) is the same expression for both, so Phi_n is evaluated
//     once per EDGE -- lane l of a wavefront holds edge x0 + l, the 63 bins x0 ... x0 + 62 take their second value from the
//     next lane (DPP wave_shl:1; one edge per wavefront is evaluated twice);
//   * a / n^k with the correctly rounded reciprocal r = RN(1 / n^k) from a table (wave-uniform: scalar loads): q = RN(a r),
//     q' = fma(fma(-q, n^k, a), r, q) is the correctly rounded quotient (Markstein's theorem: r correctly rounded, q within
//     one ulp) -- three instructions instead of the division's scale / reciprocal / Newton / fix-up sequence.  On the CPU the
//     whole table of config 2's grid and the test grids: not one entry differs from the divisions;
//   * the series is left where no later term can change the sum.  The n-th term of a bin is D times the integral of
//     x^3 exp(-n x) over the bin's [y_top, y_bot]: positive and falling with n, and so is the rounding noise of its two computed
//     halves (a few ulp of Phi_n(y_top)).  Once D (|term_n| + 2^-47 Phi_n(y_top)) < 2^-55 |sum| -- a quarter of an ulp of the sum,
//     with the noise bound counted in -- every later addition would round back to the sum it was added to, in every bin of the
//     wavefront: same bits, a tenth to a half of the 199 terms outside the Rayleigh-Jeans corner.  (exp(-n y) = 0 exactly,
//     beyond n y = 745.2, is the special case of it.)
// A multiply-recurrence for exp(-n y) (re-anchored or in double-double) is NOT used: at 10 000 bins the Rayleigh-Jeans corner
// of the table cancels eleven digits inside the formula and the entries move by 1e-4 relative -- the table's own noise, but
// two hundred times the tolerance the table is held to against the reference.
// ---------------------------------------------------------------------------------------------
struct PlanckSeriesRow {   // one term of the series: what depends on n alone (64 bytes: one scalar load)
    double d1, d2, d3;     // n, n^2, n^3 (exact)
    double r1, r2, r3;     // their reciprocals, correctly rounded (host division)
    double c4;             // 6 / n^4, correctly rounded (host division): the constant term of Phi_n's polynomial
    double pad;
};
struct PlanckSeriesTable {
    PlanckSeriesRow row[200];
};
__constant__ PlanckSeriesTable c_planck_series;

// a row of the table in sixteen scalar registers, loaded without the compiler's help (see planck_row_entry)
typedef int int16v __attribute__((ext_vector_type(16)));
union RowBits {
    int16v v;
    PlanckSeriesRow r;
    __device__ RowBits() {}
};
__device__ __forceinline__ RowBits load_row(const PlanckSeriesRow* p) {
    RowBits b;
    // (early clobber: the sixteen registers are written asynchronously -- they must not double as the address operand; that no
    // instruction touches them before wait_row() is held by tests/test_abi.py on the built code object)
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(b.v) : "s"(p));
    return b;
}
__device__ __forceinline__ void wait_row(RowBits& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b.v)); }

__device__ __forceinline__ double quotient(double a, double d, double r) {
    const double q = a * r;
    return fma(fma(-q, d, a), r, q);
}

__device__ __forceinline__ double next_lane(double v) {  // the value of lane + 1 (lane 63: unspecified, unused)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

constexpr int PLANCK_BINS_PER_WAVE = 63, PLANCK_WAVES = 4;

__device__ __forceinline__ void planck_row_entry(double* __restrict__ grid, const double* __restrict__ lambda_edge,
                                                 const double* __restrict__ dlambda, int nbin, double Tstar, int nrow_T,
                                                 int step, const PlanckSeriesTable* __restrict__ tab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (blockIdx.x * PLANCK_WAVES + wave) * PLANCK_BINS_PER_WAVE + lane;   // this lane's edge, and its bin
    const int r = blockIdx.y;
    if (x - lane >= nbin) return;   // (whole wavefronts only: the lanes exchange values below)
    const double T = (r < nrow_T) ? (double)(r * step + 1) : Tstar;
    const bool bin = lane < PLANCK_BINS_PER_WAVE && x < nbin;
    double acc = 0.0;
    if (T > 0.01) {
        const double kh = HX_KBOLTZMANN / HX_HCONST;
        const double D = 2.0 * (kh * kh * kh * HX_KBOLTZMANN * (T * T * T * T)) / (HX_CSPEED * HX_CSPEED);
        const double y = x <= nbin ? HX_HCONST * HX_CSPEED / (lambda_edge[x] * HX_KBOLTZMANN * T) : 1e6;  // (beyond the grid: e = 0)
        // the reference orders the two edges of a bin so that y_top < y_bot (:399-403) and sums D * (Phi(y_top) - Phi(y_bot)):
        // D (other - mine) where this lane's edge is the bin's y_bot, D (mine - other) = (-D) (other - mine) otherwise (exact)
        const double Ds = !(y < next_lane(y)) ? D : -D;
        const double y2 = y * y, y3 = y2 * y, a2 = 3.0 * y2, a1 = 6.0 * y;
        const PlanckSeriesRow* __restrict__ rows = tab->row;
        RowBits c = load_row(rows + 1);
        wait_row(c);
        for (int n = 1; n < 200; n++) {
            // the next term's row: ONE 64-byte scalar load, requested here and waited for behind this term's arithmetic (written
            // as a plain struct load, the compiler sinks it to its first use and every term begins with a scalar-memory round trip)
            RowBits nx = load_row(rows + (n < 199 ? n + 1 : 199));
            const PlanckSeriesRow& cr = c.r;
            const double e = exp(-cr.d1 * y);
            const double phi = e * (quotient(y3, cr.d1, cr.r1) + quotient(a2, cr.d2, cr.r2) + quotient(a1, cr.d3, cr.r3) + cr.c4);
            const double other = next_lane(phi);
            const double diff = other - phi;
            acc += Ds * diff;
            // may a later term still move this bin's sum?  (lanes without a bin do not hold the wavefront back; asked every
            // fourth term: leaving a few terms late changes nothing)
            if ((n & 3) == 0) {
                const bool live = bin && !(D * (fabs(diff) + 0x1p-47 * fmax(phi, other)) < 0x1p-55 * fabs(acc));
                if (__ballot(live) == 0ull) break;
            }
            wait_row(nx);
            c = nx;
        }
    }
    if (bin) grid[x + (size_t)r * nbin] = acc / dlambda[x];
}

__global__ void __launch_bounds__(64 * PLANCK_WAVES)
k_plancktable(double* __restrict__ grid, const double* __restrict__ lambda_edge,
              const double* __restrict__ dlambda, int nbin, double Tstar, int nrow_T, int step) {
    planck_row_entry(grid, lambda_edge, dlambda, nbin, Tstar, nrow_T, step, &c_planck_series);
}

// one row at the stellar temperature (a name of its own, so that kernel statistics do not average it with the table)
__global__ void __launch_bounds__(64 * PLANCK_WAVES)
k_planck_star_row(double* __restrict__ row, const double* __restrict__ lambda_edge,
                  const double* __restrict__ dlambda, int nbin, double Tstar) {
    planck_row_entry(row, lambda_edge, dlambda, nbin, Tstar, 0, 1, &c_planck_series);
}

// The series as the reference writes it (kernels.cu:95-105, :362-416): one thread per (bin, row), 199 terms, two `exp` and
// eight divisions each -- the form this file had until round 5.  Kept for ONE purpose: tests/test_gpu_stages.py holds
// k_plancktable to it bit for bit at full size (hx_internal_plancktable_plain; not part of the C-ABI, not used by the product).
__global__ void __launch_bounds__(256)
k_plancktable_plain(double* __restrict__ grid, const double* __restrict__ lambda_edge, const double* __restrict__ dlambda,
                    int nbin, double Tstar, int nrow_T, int step) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (x >= nbin) return;
    const double T = (r < nrow_T) ? (double)(r * step + 1) : Tstar;
    double acc = 0.0;
    if (T > 0.01) {
        const double kh = HX_KBOLTZMANN / HX_HCONST;
        const double D = 2.0 * (kh * kh * kh * HX_KBOLTZMANN * (T * T * T * T)) / (HX_CSPEED * HX_CSPEED);
        double y2 = HX_HCONST * HX_CSPEED / (lambda_edge[x + 1] * HX_KBOLTZMANN * T);   // y_top
        double y1 = HX_HCONST * HX_CSPEED / (lambda_edge[x] * HX_KBOLTZMANN * T);       // y_bot
        if (y1 < y2) {
            const double t = y2;
            y2 = y1;
            y1 = t;
        }
        for (int n = 1; n < 200; n++) {
            const double dn = n;
            acc += D * (exp(-dn * y2) * ((y2 * y2 * y2) / dn + 3.0 * (y2 * y2) / (dn * dn) + 6.0 * y2 / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn)) -
                        exp(-dn * y1) * ((y1 * y1 * y1) / dn + 3.0 * (y1 * y1) / (dn * dn) + 6.0 * y1 / (dn * dn * dn) + 6.0 / (dn * dn * dn * dn)));
        }
    }
    grid[x + (size_t)r * nbin] = acc / dlambda[x];
}

int upload_planck_series(hx_context* ctx) {
    static const PlanckSeriesTable tab = [] {
        PlanckSeriesTable t{};
        for (int n = 1; n < 200; n++) {
            const double dn = n;
            PlanckSeriesRow& r = t.row[n];
            r.d1 = dn; r.d2 = dn * dn; r.d3 = dn * dn * dn;
            r.r1 = 1.0 / r.d1; r.r2 = 1.0 / r.d2; r.r3 = 1.0 / r.d3;
            r.c4 = 6.0 / (dn * dn * dn * dn);
        }
        return t;
    }();
    // (per call: the symbol belongs to the device that is current, a few KB on the stream)
    HX_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_planck_series), &tab, sizeof(tab), 0, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

// incident-energy correction (kernels.cu:420-468): ONE block sums dlambda*F over all bins in a
// fixed order (the reference lets every thread redo the whole sum), then rescales.
__global__ void __launch_bounds__(1024)
k_corr_inc_energy(double* __restrict__ spec, const double* __restrict__ dlambda, int realstar,
                  int nbin, double Tstar, double* __restrict__ factor_out) {
    __shared__ double part[1024];
    double s = 0.0;
    for (int x = threadIdx.x; x < nbin; x += blockDim.x)
        s += realstar == 1 ? dlambda[x] * spec[x] : dlambda[x] * HX_PI * spec[x];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    const double corr = HX_STEFANBOLTZMANN * pow(Tstar, 4.0) / part[0];
    if (threadIdx.x == 0) *factor_out = corr;  // the reference prints it (kernels.cu:455); here: hx_diag_read
    for (int x = threadIdx.x; x < nbin; x += blockDim.x) spec[x] *= corr;
}

__global__ void k_temp_inter(const double* __restrict__ tlay, double* __restrict__ tint, int ni) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ni) return;
    if (i == 0)
        tint[i] = tlay[i] - 0.5 * (tlay[i + 1] - tlay[i]);
    else if (i == ni - 1)
        tint[i] = tlay[i - 1] + 0.5 * (tlay[i - 1] - tlay[i - 2]);
    else
        tint[i] = tlay[i - 1] + 0.5 * (tlay[i] - tlay[i - 1]);
}

// Planck interpolation.  The table is x-fastest, the outputs are level-fastest ([i + x*stride]):
// a 32x32 tile goes through LDS so that both the table reads and the output writes are coalesced.
template <bool LAYER>
__global__ void __launch_bounds__(256)
k_planck_interpol(const double* __restrict__ temp, double* __restrict__ out,
                  const double* __restrict__ planck_grid, const double* __restrict__ starflux,
                  int realstar, int nlev_out, int nlayer, int nbin, int dim, int step) {
    __shared__ double tile[32][33];
    const int x0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, x = x0 + tx;
        if (i < nlev_out && x < nbin) {
            double v;
            if (LAYER && i == nlayer) {
                v = realstar == 1 ? starflux[x] / HX_PI : planck_grid[x + (size_t)dim * nbin];
            } else {
                const double T = (LAYER && i == nlayer + 1) ? temp[nlayer] : temp[i];
                v = planck_lookup(planck_grid, T, x, nbin, dim, step);
            }
            tile[r][tx] = v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, i = i0 + tx;
        if (i < nlev_out && x < nbin) out[i + (size_t)x * nlev_out] = tile[tx][r];
    }
}

// k-table look-up (premixed: kernels.cu:524-610, per species: :3209-3259).  One thread per
// (c = y + ny*x, level): the four table corners and the output are contiguous in c.
template <bool SPECIES>
__global__ void __launch_bounds__(256)
k_opac_interpol(const double* __restrict__ temp, const double* __restrict__ opactemp,
                const double* __restrict__ press, const double* __restrict__ opacpress,
                const double* __restrict__ ktable, double* __restrict__ opac,
                const double* __restrict__ crosstable, double* __restrict__ scat_cross, int npress,
                int ntemp, int ny, int nbin, int nlev) {
    const int i = blockIdx.y;
    const size_t nc = (size_t)ny * nbin;
    const TPIndex k = locate_tp(temp[i], press[i], opactemp, ntemp, opacpress, npress, !SPECIES, false);
    const size_t sp = nc, st = nc * npress;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc;
         c += (size_t)gridDim.x * blockDim.x) {
        opac[c + nc * i] =
            blend_tp(ktable[c + sp * k.pdown + st * k.tdown], ktable[c + sp * k.pup + st * k.tdown],
                     ktable[c + sp * k.pdown + st * k.tup], ktable[c + sp * k.pup + st * k.tup], k,
                     SPECIES);
        if (!SPECIES && c < (size_t)nbin) {
            const size_t x = c, cp = nbin, ct = (size_t)nbin * npress;
            scat_cross[x + (size_t)nbin * i] = blend_tp(
                crosstable[x + cp * k.pdown + ct * k.tdown], crosstable[x + cp * k.pup + ct * k.tdown],
                crosstable[x + cp * k.pdown + ct * k.tup], crosstable[x + cp * k.pup + ct * k.tup], k,
                false);
        }
    }
}

// scalar tables (mean molecular mass :649, kappa :703, c_p :761)
__global__ void k_scalar_table(const double* __restrict__ temp, const double* __restrict__ tgrid,
                               const double* __restrict__ press, const double* __restrict__ pgrid,
                               double* __restrict__ out, const double* __restrict__ table, int npress,
                               int ntemp, int nlev, int log_t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlev) return;
    const TPIndex k = locate_tp(temp[i], press[i], tgrid, ntemp, pgrid, npress, true, log_t != 0);
    out[i] = blend_tp(table[k.pdown + npress * k.tdown], table[k.pup + npress * k.tdown],
                      table[k.pdown + npress * k.tup], table[k.pup + npress * k.tup], k, false);
}

}  // namespace

extern "C" {

int hx_plancktable(hx_context* ctx, double* planck_grid, const double* lambda_edge,
                   const double* deltalambda, int nwave, double Tstar, int dim, int step) {
    HX_REQUIRE(ctx, nwave > 0 && dim >= 10 && step > 0, HX_E_ARG, "bad dimensions");
    const int nrow_T = 10 * (dim / 10);  // the reference fills rows in ten launches of dim/10
    int rc = upload_planck_series(ctx);
    if (rc) return rc;
    dim3 grid(hx_cdiv(nwave, PLANCK_BINS_PER_WAVE * PLANCK_WAVES), nrow_T + 1);
    k_plancktable<<<grid, 64 * PLANCK_WAVES, 0, ctx->stream>>>(planck_grid, lambda_edge, deltalambda, nwave, Tstar,
                                                             nrow_T, step);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

// internal (not part of the C-ABI): the table by the reference's plain formula, for the bit-for-bit test of hx_plancktable
int hx_internal_plancktable_plain(hx_context* ctx, double* planck_grid, const double* lambda_edge, const double* deltalambda,
                                  int nwave, double Tstar, int dim, int step) {
    HX_REQUIRE(ctx, nwave > 0 && dim >= 10 && step > 0, HX_E_ARG, "bad dimensions");
    const int nrow_T = 10 * (dim / 10);
    k_plancktable_plain<<<dim3(hx_cdiv(nwave, 256), nrow_T + 1), 256, 0, ctx->stream>>>(planck_grid, lambda_edge, deltalambda, nwave,
                                                                                     Tstar, nrow_T, step);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

// internal (not part of the C-ABI): the stellar row only, for per-column stars of a batch
int hx_internal_planck_star_row(hx_context* ctx, double* row, const double* lambda_edge,
                                const double* deltalambda, int nwave, double Tstar) {
    int rc = upload_planck_series(ctx);
    if (rc) return rc;
    k_planck_star_row<<<dim3(hx_cdiv(nwave, PLANCK_BINS_PER_WAVE * PLANCK_WAVES), 1), 64 * PLANCK_WAVES, 0, ctx->stream>>>(
        row, lambda_edge, deltalambda, nwave, Tstar);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_corr_inc_energy(hx_context* ctx, double* planck_grid, double* starflux,
                       const double* deltalambda, int realstar, int nwave, double Tstar, int dim) {
    double* spec = realstar == 1 ? starflux : planck_grid + (size_t)dim * nwave;
    k_corr_inc_energy<<<1, 1024, 0, ctx->stream>>>(spec, deltalambda, realstar, nwave, Tstar,
                                                   reinterpret_cast<double*>(ctx->diag + HX_DIAG_ENERGY));
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_temp_inter(hx_context* ctx, const double* tlay, double* tint, int numinterfaces,
                  int itervalue) {
    (void)itervalue;
    HX_REQUIRE(ctx, numinterfaces >= 3, HX_E_ARG, "needs at least 2 layers");
    k_temp_inter<<<hx_cdiv(numinterfaces, 64), 64, 0, ctx->stream>>>(tlay, tint, numinterfaces);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_planck_interpol_layer(hx_context* ctx, const double* temp, double* planckband_lay,
                             const double* planck_grid, const double* starflux, int realstar,
                             int numlayers, int nwave, int dim, int step) {
    dim3 grid(hx_cdiv(nwave, 32), hx_cdiv(numlayers + 2, 32));
    k_planck_interpol<true><<<grid, 256, 0, ctx->stream>>>(temp, planckband_lay, planck_grid, starflux,
                                                          realstar, numlayers + 2, numlayers, nwave,
                                                          dim, step);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_planck_interpol_interface(hx_context* ctx, const double* temp, double* planckband_int,
                                 const double* planck_grid, int numinterfaces, int nwave, int dim,
                                 int step) {
    dim3 grid(hx_cdiv(nwave, 32), hx_cdiv(numinterfaces, 32));
    k_planck_interpol<false><<<grid, 256, 0, ctx->stream>>>(temp, planckband_int, planck_grid, nullptr,
                                                           0, numinterfaces, -5, nwave, dim, step);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_opac_interpol(hx_context* ctx, const double* temp, const double* opactemp,
                     const double* press, const double* opacpress, const double* ktable,
                     double* opac, const double* crosstable, double* scat_cross, int npress,
                     int ntemp, int ny, int nbin, int nlay_or_nint) {
    const long long nc = (long long)ny * nbin;
    dim3 grid((unsigned)min((long long)hx_cdiv(nc, 256), 4096LL), nlay_or_nint);
    k_opac_interpol<false><<<grid, 256, 0, ctx->stream>>>(temp, opactemp, press, opacpress, ktable,
                                                         opac, crosstable, scat_cross, npress, ntemp,
                                                         ny, nbin, nlay_or_nint);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_opac_species_interpol(hx_context* ctx, const double* temp, const double* opactemp,
                             const double* press, const double* opacpress,
                             const double* opac_opacity_pretab, double* opac_spec_wg, int npress,
                             int ntemp, int ny, int nbin, int nlay_or_nint) {
    const long long nc = (long long)ny * nbin;
    dim3 grid((unsigned)min((long long)hx_cdiv(nc, 256), 4096LL), nlay_or_nint);
    k_opac_interpol<true><<<grid, 256, 0, ctx->stream>>>(temp, opactemp, press, opacpress,
                                                        opac_opacity_pretab, opac_spec_wg, nullptr,
                                                        nullptr, npress, ntemp, ny, nbin, nlay_or_nint);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_meanmolmass_interpol(hx_context* ctx, const double* temp, const double* opactemp,
                            double* meanmolmass, const double* opac_meanmass, const double* press,
                            const double* opacpress, int npress, int ntemp, int ninterface) {
    k_scalar_table<<<hx_cdiv(ninterface, 64), 64, 0, ctx->stream>>>(
        temp, opactemp, press, opacpress, meanmolmass, opac_meanmass, npress, ntemp, ninterface, 0);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_kappa_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                      const double* press, const double* entr_press, double* kappa,
                      const double* entr_kappa, int entr_npress, int entr_ntemp, int nlay_or_nint) {
    k_scalar_table<<<hx_cdiv(nlay_or_nint, 64), 64, 0, ctx->stream>>>(
        temp, entr_temp, press, entr_press, kappa, entr_kappa, entr_npress, entr_ntemp, nlay_or_nint, 0);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_cp_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                   const double* press, const double* entr_press, double* cp_lay,
                   const double* entr_cp, int entr_npress, int entr_ntemp, int nlayer) {
    k_scalar_table<<<hx_cdiv(nlayer, 64), 64, 0, ctx->stream>>>(
        temp, entr_temp, press, entr_press, cp_lay, entr_cp, entr_npress, entr_ntemp, nlayer, 1);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_entropy_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                        const double* press, const double* entr_press, double* entropy,
                        const double* entr_entropy, int entr_npress, int entr_ntemp, int nlayer) {
    k_scalar_table<<<hx_cdiv(nlayer, 64), 64, 0, ctx->stream>>>(
        temp, entr_temp, press, entr_press, entropy, entr_entropy, entr_npress, entr_ntemp, nlayer, 1);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_phase_number_interpol(hx_context* ctx, const double* temp, const double* entr_temp,
                             const double* press, const double* entr_press, double* state,
                             const double* entr_state, int entr_npress, int entr_ntemp, int nlayer) {
    k_scalar_table<<<hx_cdiv(nlayer, 64), 64, 0, ctx->stream>>>(
        temp, entr_temp, press, entr_press, state, entr_state, entr_npress, entr_ntemp, nlayer, 0);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
