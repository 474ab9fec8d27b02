// Per-stage entry points, part 4: on-the-fly opacity mixing (correlated-k and random overlap),
// H2O Rayleigh scattering, scattering-cross-section accumulation, total asymmetry parameter.
#include "two_stream.h"

using namespace hx;

namespace {

constexpr int RO_NY = 20;
constexpr int RO_N = RO_NY * RO_NY;  // 400 pair sums
constexpr int RO_PER_LANE = 7;       // 64 * 7 = 448 >= 400


// add_to_mixed_opac (kernels.cu:3263-3399).  ONE wavefront (= one 64-thread
// workgroup) per (bin x, level i): the 20+20 k-coefficients, the 400 pair sums and their sorted
// copies live in LDS (13 KB per workgroup) instead of the reference's 9.9 KB of per-thread scratch.
//
// Sorting: the reference repeats adjacent-swap passes with a strict '<' (a stable sort of the
// fill-ordered array).  Here every pair sum gets its rank directly: rank(e) = #{f : K_f < K_e} +
// #{f < e : K_f == K_e} with e, f the positions in the reference's fill order -- the same permutation
// (the order inside a group of equal sums matters: it decides which weight sits at the group's edge).
// The sums are read from LDS as broadcasts (all lanes, one address, two sums per 128-bit read) and each
// lane ranks its 7 sums with one fp64 compare + add per pair.  If two sums are exactly equal their ranks
// collide (detected by writing the positions into the rank slots and reading them back, a wave-uniform
// decision); only then a second pass adds the tie-break.  (Measured alternatives, both slower on gfx950:
// 64-bit integer keys -- v_cmp_lt_u64 issues at a fraction of the fp64 compare rate; a first pass on the
// upper 32 key bits -- pair sums of a dominant and a minor absorber agree to < 1e-6 far too often.)
__global__ void __launch_bounds__(64)
k_add_to_mixed_opac(const double* __restrict__ vmr, const double* __restrict__ opac_spec,
                    double* __restrict__ opac_wg, const double* __restrict__ meanmolmass,
                    const double* __restrict__ gauss_weight, const double* __restrict__ gauss_y,
                    double mass_spec, int s, int ro_method, int ny, int nbin, int nlev) {
    __shared__ double s_mix[RO_NY], s_add[RO_NY], s_hw[RO_NY], s_gy[RO_NY];
    __shared__ double s_G[RO_N], s_Ks[RO_N + 64], s_Y[RO_N + 64];
    __shared__ __align__(16) double s_K[RO_N];
    __shared__ int s_w[RO_NY];
    int* s_slot = (int*)s_Ks;  // rank slots alias the sorted-sum buffer (used before it is filled)
    const int lane = threadIdx.x;
    const long long npair = (long long)nbin * nlev;
    if (lane < ny && lane < RO_NY) {
        s_hw[lane] = 0.5 * gauss_weight[lane];
        s_gy[lane] = gauss_y[lane];
    }
    for (long long pair = blockIdx.x; pair < npair; pair += gridDim.x) {
        const int i = (int)(pair / nbin);
        const size_t base = (size_t)ny * pair;  // = ny*x + ny*nbin*i
        __syncthreads();
        const double scale_num = vmr[i] * mass_spec;
        const double mmm = meanmolmass[i];
        if (ny > RO_NY || ny == 1 || ro_method == 0 || s == 0) {
            // correlated-k for any ny (kernels.cu:3302-3310)
            for (int y = lane; y < ny; y += 64) opac_wg[base + y] += scale_num / mmm * opac_spec[base + y];
            continue;
        }
        if (lane < ny) {
            s_mix[lane] = opac_wg[base + lane];
            s_add[lane] = scale_num / mmm * opac_spec[base + lane];
        }
        __syncthreads();
        // negligibility test (:3297): wave-uniform
        const bool negligible = (0.01 * s_mix[0] > s_add[ny - 1]) || (0.01 * s_add[0] > s_mix[ny - 1]);
        if (negligible) {
            if (lane < ny) opac_wg[base + lane] = s_mix[lane] + s_add[lane];
            continue;
        }
        // last crossing of the two curves (:3321-3329)
        bool cross = false;
        if (lane >= 1 && lane < ny)
            cross = (s_mix[lane] > s_add[lane]) != (s_mix[lane - 1] > s_add[lane - 1]);
        const unsigned long long cmask = __ballot(cross);
        const int yx = cmask ? 63 - __clzll((long long)cmask) : ny;
        const bool mix_first = s_mix[0] > s_add[0];
        // fill in the reference's order (:3332-3365)
        double ke[RO_PER_LANE];
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int e = lane + 64 * r;
            ke[r] = 0.0;
            if (e < RO_N) {
                int y1, y2;  // y1 indexes the running mix, y2 the new species
                const int nfirst = ny * yx;
                if (mix_first) {
                    if (e < nfirst) { y1 = e / yx; y2 = e - yx * y1; }
                    else            { y2 = e / ny; y1 = e - ny * y2; }
                } else {
                    if (e < nfirst) { y2 = e / yx; y1 = e - yx * y2; }
                    else            { y1 = e / ny; y2 = e - ny * y1; }
                }
                ke[r] = s_mix[y1] + s_add[y2];
                s_G[e] = s_hw[y1] * s_hw[y2];
                s_K[e] = ke[r];
            }
        }
        __syncthreads();
        // ranks
        int rank[RO_PER_LANE];
#pragma unroll
        for (int r = 0; r < RO_PER_LANE; r++) rank[r] = 0;
        {
            const double2* keys2 = reinterpret_cast<const double2*>(s_K);
#pragma unroll 8
            for (int f2 = 0; f2 < RO_N / 2; f2++) {
                const double2 kf = keys2[f2];
#pragma unroll
                for (int r = 0; r < RO_PER_LANE; r++) rank[r] += (kf.x < ke[r] ? 1 : 0) + (kf.y < ke[r] ? 1 : 0);
            }
        }
        // equal sums collide on a rank slot
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int e = lane + 64 * r;
            if (e < RO_N) s_slot[rank[r]] = e;
        }
        __syncthreads();
        bool clash = false;
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int e = lane + 64 * r;
            if (e < RO_N && s_slot[rank[r]] != e) clash = true;
        }
        const bool any_clash = __ballot(clash) != 0;
        __syncthreads();
        if (any_clash) {  // rare: exact ties -> stable order by fill position
            for (int f = 0; f < RO_N; f++) {
                const double kf = s_K[f];
#pragma unroll
                for (int r = 0; r < RO_PER_LANE; r++) rank[r] += (kf == ke[r] && f < lane + 64 * r) ? 1 : 0;
            }
        }
        // scatter into sorted order; s_Y temporarily holds the sorted weights
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int e = lane + 64 * r;
            if (e < RO_N) {
                s_Ks[rank[r]] = ke[r];
                s_Y[rank[r]] = s_G[e];
            }
        }
        __syncthreads();
        // cumulative mid-point abscissae Y_w = sum_{v<w} g_v + g_w/2 (:3371-3376): lane-contiguous
        // chunks of 7 + wave exclusive scan
        double g[RO_PER_LANE], csum = 0.0;
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int w = lane * RO_PER_LANE + r;
            g[r] = w < RO_N ? s_Y[w] : 0.0;
            csum += g[r];
        }
        double incl = csum;
        for (int d = 1; d < 64; d <<= 1) {
            const double up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        double run = incl - csum;
        __syncthreads();
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int w = lane * RO_PER_LANE + r;
            if (w < RO_N) s_Y[w] = run + 0.5 * g[r];
            run += g[r];
        }
        __syncthreads();
        // re-binning (:3379-3396): first w >= 1 with Y_w > y_q, at most one Gauss point per w
        if (lane < ny) {
            const double yq = s_gy[lane];
            int lo = 1, hi = RO_N;  // first index in [1, 400) with Y > yq, else 400
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_Y[mid] > yq) hi = mid; else lo = mid + 1;
            }
            s_w[lane] = lo;
        }
        __syncthreads();
        if (lane == 0)
            for (int q = 1; q < ny; q++)
                if (s_w[q] <= s_w[q - 1]) s_w[q] = s_w[q - 1] + 1;
        __syncthreads();
        if (lane < ny) {
            const int w = s_w[lane];
            if (w < RO_N) {
                const double yq = s_gy[lane];
                opac_wg[base + lane] =
                    (s_Ks[w - 1] * (s_Y[w] - yq) + s_Ks[w] * (yq - s_Y[w - 1])) / (s_Y[w] - s_Y[w - 1]);
            }
        }
    }
}

// calc_index_h2o / calc_h2o_scat (kernels.cu:3174-3205, :3404-3440)
__global__ void __launch_bounds__(256)
k_calc_h2o_scat(const double* __restrict__ temp, const double* __restrict__ press,
                const double* __restrict__ wave, double* __restrict__ scat_cross,
                const double* __restrict__ vmr, double mass_h2o, int nbin, int nlev) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (x >= nbin) return;
    const double T = temp[i], P = press[i], f = vmr[i], lam_cm = wave[x];
    const double dens = f * P * mass_h2o / (HX_KBOLTZMANN * T);
    const double lamda = lam_cm / 0.589e-4;
    const double delta = dmin(1.0, dens) / 1.0;
    const double theta = T / 273.15;
    const double lamda_UV = 0.229202, lamda_IR = 5.432937;
    const double a0 = 0.244257733, a1 = 0.974634476e-2, a2 = -0.373234996e-2, a3 = 0.268678472e-3,
                 a4 = 0.158920570e-2, a5 = 0.245934259e-2, a6 = 0.900704920, a7 = -0.166626219e-1;
    const double l2 = lamda * lamda;
    const double A = delta * (a0 + a1 * delta + a2 * theta + a3 * l2 * theta + a4 / l2 +
                              a5 / (l2 - lamda_UV * lamda_UV) + a6 / (l2 - lamda_IR * lamda_IR) +
                              a7 * (delta * delta));
    const double index = sqrt((2.0 * A + 1.0) / (1.0 - A));
    const double n_ref = f * P / (HX_KBOLTZMANN * T);
    const double King = (6.0 + 3.0 * 3e-4) / (6.0 - 7.0 * 3e-4);
    double sc = 0.0;
    if (lam_cm < 2.5e-4) {
        const double n2 = index * index;
        const double lor = (n2 - 1.0) / (n2 + 2.0);
        const double lam2 = lam_cm * lam_cm;
        sc = 24.0 * (HX_PI * HX_PI * HX_PI) / ((n_ref * n_ref) * (lam2 * lam2)) * (lor * lor) * King;
    }
    scat_cross[x + (size_t)nbin * i] = sc;
}

__global__ void __launch_bounds__(256)
k_add_to_mixed_scat(const double* __restrict__ vmr, const double* __restrict__ spec,
                    double* __restrict__ total, int nbin, int nlev) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (x >= nbin) return;
    const size_t b = x + (size_t)nbin * i;
    total[b] += vmr[i] * spec[b];
}

__global__ void __launch_bounds__(256)
k_calc_total_g0(const double* __restrict__ scat_cross, const double* __restrict__ g_cl,
                const double* __restrict__ scat_cl, double* __restrict__ g_tot, double g_0, size_t n) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double num = g_0 * scat_cross[k] + g_cl[k] * scat_cl[k];
    const double den = scat_cross[k] + scat_cl[k];
    g_tot[k] = num / den;
}

}  // namespace

extern "C" {

int hx_add_to_mixed_opac(hx_context* ctx, const double* vmr, const double* opac_spec,
                         double* opac_wg, const double* meanmolmass, const double* gauss_weight,
                         const double* gauss_y, double mass_spec, int s, int ro_method, int ny,
                         int nbin, int nlay_or_nint) {
    const bool ro_possible = ro_method != 0 && s != 0 && ny != 1;
    if (ro_possible && ny != RO_NY)
        return hx_fail(ctx, HX_E_RO_NY, "random-overlap mixing needs ny == 20 (got %d)", ny);
    const long long npair = (long long)nbin * nlay_or_nint;
    const int grid = (int)min(npair, (long long)256 * 12 * 16);
    k_add_to_mixed_opac<<<grid, 64, 0, ctx->stream>>>(vmr, opac_spec, opac_wg, meanmolmass,
                                                     gauss_weight, gauss_y, mass_spec, s, ro_method,
                                                     ny, nbin, nlay_or_nint);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_calc_h2o_scat(hx_context* ctx, const double* temp, const double* press, const double* wave,
                     double* scat_cross, const double* vmr, double mass_h2o, int nbin,
                     int nlay_or_nint) {
    k_calc_h2o_scat<<<dim3(hx_cdiv(nbin, 256), nlay_or_nint), 256, 0, ctx->stream>>>(
        temp, press, wave, scat_cross, vmr, mass_h2o, nbin, nlay_or_nint);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_add_to_mixed_scat(hx_context* ctx, const double* vmr, const double* scat_cross_spec,
                         double* scat_cross, int nbin, int nlay_or_nint) {
    k_add_to_mixed_scat<<<dim3(hx_cdiv(nbin, 256), nlay_or_nint), 256, 0, ctx->stream>>>(
        vmr, scat_cross_spec, scat_cross, nbin, nlay_or_nint);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_calc_total_g_0_of_gas_and_clouds(hx_context* ctx, const double* scat_cross,
                                        const double* g_0_all_clouds,
                                        const double* scat_cross_all_clouds, double* g_0_tot,
                                        double g_0, int nbin, int nlay_or_nint) {
    const size_t n = (size_t)nbin * nlay_or_nint;
    k_calc_total_g0<<<hx_cdiv((long long)n, 256), 256, 0, ctx->stream>>>(
        scat_cross, g_0_all_clouds, scat_cross_all_clouds, g_0_tot, g_0, n);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
