// Per-stage entry points, part 4: on-the-fly opacity mixing (correlated-k and random overlap),
// H2O Rayleigh scattering, scattering-cross-section accumulation, total asymmetry parameter.
#include "two_stream.h"
#include "random_overlap.h"
#include "random_overlap_lean.h"
#include <cstdlib>
#include <cstring>

using namespace hx;

namespace {

constexpr int RO_NY = 20;
constexpr int RO_N = RO_NY * RO_NY;  // 400 pair sums
constexpr int RO_PER_LANE = 7;       // 64 * 7 = 448 >= 400


// add_to_mixed_opac (kernels.cu:3263-3399).  ONE wavefront (= one 64-thread
// workgroup) per (bin x, level i): the 20+20 k-coefficients, the 400 pair sums and their sorted
// copies live in LDS (12 KB per workgroup) instead of the reference's 9.9 KB of per-thread scratch.
//
// Sorting: the reference repeats adjacent-swap passes with a strict '<' (a stable sort of the
// fill-ordered array), i.e. position(e) = #{f : K_f < K_e} + #{f < e : K_f == K_e} with e, f the positions
// in the reference's fill order (the order inside a group of equal sums matters: it decides which weight
// sits at the group's edge).  Two implementations of that permutation:
//  * BITONIC (default): a 512-slot bitonic network over (sum, fill position), 8 slots per lane in registers
//    (slot p = 8*lane + s; 112 pads of +inf).  24 of its 45 steps stay inside a lane, 21 exchange with lane^m
//    through ds_bpermute (the LDS crossbar, which leaves the VALU free).  Every phase opens with the mirrored
//    step (p against p^(k-1)), so all compare-exchanges put the smaller sum at the lower slot and no direction
//    masks are needed.  The exchanges keep both entries when the sums are equal, so the result is a permutation
//    with equal sums adjacent; only if such a pair exists (wave-uniform test) the fill positions inside each
//    group of equal sums are put in ascending order afterwards.  ~1.4 k VALU instructions per problem.
//  * RANK (HELIOS_RO_SORT=rank; the cross-check): every pair sum is ranked against all 400 (LDS broadcasts,
//    one fp64 compare + add per pair): 5.6 k VALU instructions per problem, 160 000 compares.
// (Measured alternatives for the rank form, both slower on gfx950: 64-bit integer keys -- v_cmp_lt_u64 issues
// at a fraction of the fp64 compare rate; a first pass on the upper 32 key bits -- pair sums of a dominant and
// a minor absorber agree to < 1e-6 far too often.)
constexpr int BT_PER_LANE = 8;
constexpr int BT_N = 64 * BT_PER_LANE;  // 512 slots

struct BtSlots {
    double k[BT_PER_LANE];
    int id[BT_PER_LANE];
};

__device__ __forceinline__ void bt_ce(BtSlots& v, int lo, int hi) {  // inside a lane: smaller sum to slot lo
    const double a = v.k[lo], b = v.k[hi];
    const int ia = v.id[lo], ib = v.id[hi];
    const bool sw = b < a;
    // v_min/v_max agree with the swap decision for ordered, unequal sums and change nothing for equal ones
    asm("v_min_f64 %0, %1, %2" : "=v"(v.k[lo]) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(v.k[hi]) : "v"(a), "v"(b));
    v.id[lo] = sw ? ib : ia;
    v.id[hi] = sw ? ia : ib;
}

template <int J>
__device__ __forceinline__ void bt_lane_step(BtSlots& v) {  // slot s against s^J
#pragma unroll
    for (int s = 0; s < BT_PER_LANE; s++)
        if ((s & J) == 0) bt_ce(v, s, s | J);
}

template <int W>
__device__ __forceinline__ void bt_lane_mirror(BtSlots& v) {  // slot s against s^(W-1) inside blocks of W
#pragma unroll
    for (int s = 0; s < BT_PER_LANE; s++)
        if ((s & (W - 1)) < W / 2) bt_ce(v, s, s ^ (W - 1));
}

// the value lane^M holds: DPP moves (VALU) where one or two of them express the permutation, ds_bpermute (LDS
// crossbar) for M = 16, 31, 63.  With every exchange on the crossbar the kernel was bound by it (PMC: LDS unit 68 % busy
// at 4.6 cycles per ds_bpermute, VALU 48 %).
template <int M>
__device__ __forceinline__ int bt_xor_lane(int addr, int x) {
    if constexpr (M == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return __builtin_amdgcn_mov_dpp(x, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (M == 15) return __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true); // row_mirror
    else if constexpr (M == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (M == 4)                                                              // 7 ^ 3
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
    else return __builtin_amdgcn_ds_bpermute(addr, x);
}

template <int M>
__device__ __forceinline__ double bt_xor_lane(int addr, double x) {
    const int lo = bt_xor_lane<M>(addr, __double2loint(x));
    const int hi = bt_xor_lane<M>(addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bt_fetch(int addr, double x) { return bt_xor_lane<0>(addr, x); }

// exchange with lane^M: slot s meets the partner's slot s (plain step) or 7-s (MIRROR, first step of a phase)
template <int M, bool MIRROR>
__device__ __forceinline__ void bt_cross_step(BtSlots& v, int lane) {
    constexpr int TOP = MIRROR ? (M + 1) / 2 : M;  // the lane bit that tells the upper partner from the lower
    constexpr unsigned long long UPPER = TOP == 1 ? 0xAAAAAAAAAAAAAAAAull : TOP == 2 ? 0xCCCCCCCCCCCCCCCCull :
                                         TOP == 4 ? 0xF0F0F0F0F0F0F0F0ull : TOP == 8 ? 0xFF00FF00FF00FF00ull :
                                         TOP == 16 ? 0xFFFF0000FFFF0000ull : 0xFFFFFFFF00000000ull;
    const int addr = (lane ^ M) << 2;
    BtSlots n;
#pragma unroll
    for (int s = 0; s < BT_PER_LANE; s++) {
        const int ps = MIRROR ? BT_PER_LANE - 1 - s : s;
        const double pk = bt_xor_lane<M>(addr, v.k[ps]);
        const int pid = bt_xor_lane<M>(addr, v.id[ps]);
        // lower partner takes the smaller sum, upper partner the larger one; equal sums stay where they are.
        // The lane masks are combined on the scalar unit.
        const unsigned long long ge = __ballot(pk >= v.k[s]), le = __ballot(pk <= v.k[s]);
        const bool keep = __builtin_amdgcn_inverse_ballot_w64((le & UPPER) | (ge & ~UPPER));
        n.k[s] = keep ? v.k[s] : pk;
        n.id[s] = keep ? v.id[s] : pid;  // partner first: lets the DPP move fold into the v_cndmask
    }
    v = n;
}

__device__ __forceinline__ void bt_sort(BtSlots& v, int lane) {
    bt_lane_step<1>(v);                                                            // k = 2
    bt_lane_mirror<4>(v); bt_lane_step<1>(v);                                      // k = 4
    bt_lane_mirror<8>(v); bt_lane_step<2>(v); bt_lane_step<1>(v);                  // k = 8
#define BT_LANE_TAIL bt_lane_step<4>(v); bt_lane_step<2>(v); bt_lane_step<1>(v);
    bt_cross_step<1, true>(v, lane); BT_LANE_TAIL                                  // k = 16
    bt_cross_step<3, true>(v, lane); bt_cross_step<1, false>(v, lane); BT_LANE_TAIL  // k = 32
    bt_cross_step<7, true>(v, lane); bt_cross_step<2, false>(v, lane); bt_cross_step<1, false>(v, lane);
    BT_LANE_TAIL                                                                   // k = 64
    bt_cross_step<15, true>(v, lane); bt_cross_step<4, false>(v, lane); bt_cross_step<2, false>(v, lane);
    bt_cross_step<1, false>(v, lane); BT_LANE_TAIL                                 // k = 128
    bt_cross_step<31, true>(v, lane); bt_cross_step<8, false>(v, lane); bt_cross_step<4, false>(v, lane);
    bt_cross_step<2, false>(v, lane); bt_cross_step<1, false>(v, lane); BT_LANE_TAIL  // k = 256
    bt_cross_step<63, true>(v, lane); bt_cross_step<16, false>(v, lane); bt_cross_step<8, false>(v, lane);
    bt_cross_step<4, false>(v, lane); bt_cross_step<2, false>(v, lane); bt_cross_step<1, false>(v, lane);
    BT_LANE_TAIL                                                                   // k = 512
#undef BT_LANE_TAIL
}

template <bool BITONIC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BITONIC ? 3 : 1)))
k_add_to_mixed_opac(const double* __restrict__ vmr, const double* __restrict__ opac_spec,
                    double* __restrict__ opac_wg, const double* __restrict__ meanmolmass,
                    const double* __restrict__ gauss_weight, const double* __restrict__ gauss_y,
                    double mass_spec, int s, int ro_method, int ny, int nbin, int nlev,
                    unsigned long long* __restrict__ rebin_skipped) {
    __shared__ double s_mix[RO_NY], s_add[RO_NY], s_hw[RO_NY], s_gy[RO_NY];
    __shared__ double s_G[BITONIC ? 1 : RO_N], s_Ks[BT_N], s_Y[BT_N];
    __shared__ __align__(16) double s_K[BITONIC ? 2 : RO_N];
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
        // e / yx and e / 20 for e < 512 as multiply-shift (exact: e * d < 2^20 / d for d <= 20), full-rate 24-bit
        // multiplies instead of the generic 32-bit division sequence
        const int nfirst = RO_NY * yx;
        const int inv_yx = (1048576 + yx - 1) / yx;  // yx is wave-uniform: once per problem
        auto pair_of = [&](int e, int& y1, int& y2) {  // y1 indexes the running mix, y2 the new species
            const bool first = e < nfirst;
            const int q = (int)(__umul24(e, first ? inv_yx : 52429) >> 20);  // e / yx  or  e / 20
            const int rem = e - __umul24(q, first ? yx : RO_NY);
            const bool q_is_mix = mix_first == first;
            y1 = q_is_mix ? q : rem;
            y2 = q_is_mix ? rem : q;
        };
        if constexpr (BITONIC) {
            BtSlots v;
#pragma unroll
            for (int r = 0; r < BT_PER_LANE; r++) {
                const int e = lane * BT_PER_LANE + r;
                v.id[r] = e << 10;  // payload: fill position (the tie-break) above the two Gauss indices
                v.k[r] = __builtin_inf();
                if (e < RO_N) {
                    int y1, y2;
                    pair_of(e, y1, y2);
                    v.k[r] = s_mix[y1] + s_add[y2];
                    v.id[r] = e << 10 | y1 << 5 | y2;
                }
            }
            bt_sort(v, lane);
            // equal sums next to each other (slots 0..400)?
            bool tie = false;
#pragma unroll
            for (int r = 0; r + 1 < BT_PER_LANE; r++)
                tie = tie || (lane * BT_PER_LANE + r + 1 <= RO_N && v.k[r] == v.k[r + 1]);
            const double knext = bt_fetch(((lane + 1) & 63) << 2, v.k[0]);
            tie = tie || (lane * BT_PER_LANE + BT_PER_LANE <= RO_N && v.k[BT_PER_LANE - 1] == knext);
            if (__ballot(tie) != 0) {  // rare: inside each group of equal sums, ascending fill position
                int* s_id = (int*)s_Y;
                int* s_id2 = s_id + BT_N;
#pragma unroll
                for (int r = 0; r < BT_PER_LANE; r++) {
                    s_Ks[lane * BT_PER_LANE + r] = v.k[r];
                    s_id[lane * BT_PER_LANE + r] = v.id[r];
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < BT_PER_LANE; r++) {
                    const int p = lane * BT_PER_LANE + r;
                    int gs = p, below = 0;
                    while (gs > 0 && s_Ks[gs - 1] == v.k[r]) gs--;
                    for (int f = gs; f < BT_N && s_Ks[f] == v.k[r]; f++) below += s_id[f] < v.id[r] ? 1 : 0;
                    s_id2[gs + below] = v.id[r];
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < BT_PER_LANE; r++) v.id[r] = s_id2[lane * BT_PER_LANE + r];
            }
#pragma unroll
            for (int r = 0; r < BT_PER_LANE; r++) {
                const int w = lane * BT_PER_LANE + r;
                s_Ks[w] = v.k[r];
                s_Y[w] = w < RO_N ? s_hw[(v.id[r] >> 5) & 31] * s_hw[v.id[r] & 31] : 0.0;
            }
        } else {
        double ke[RO_PER_LANE];
        for (int r = 0; r < RO_PER_LANE; r++) {
            const int e = lane + 64 * r;
            ke[r] = 0.0;
            if (e < RO_N) {
                int y1, y2;
                pair_of(e, y1, y2);
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
        // equal sums collide on a rank slot (detected by writing the positions into the rank slots and reading
        // them back, a wave-uniform decision); only then a second pass adds the tie-break
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
        }
        __syncthreads();
        // cumulative mid-point abscissae Y_w = sum_{v<w} g_v + g_w/2 (:3371-3376): lane-contiguous
        // chunks of 8 (the slots of a lane in the sorting networks, so that all three variants add in the same
        // order and agree bit for bit) + wave exclusive scan
        // (rank w at position ro::RANK0 + w, as in the default kernel's network: the wave scan then adds in the same order)
        double g[BT_PER_LANE], csum = 0.0;
        for (int r = 0; r < BT_PER_LANE; r++) {
            const int w = lane * BT_PER_LANE + r - ro::RANK0;
            g[r] = (w >= 0 && w < RO_N) ? s_Y[w] : 0.0;
            csum += g[r];
        }
        double run = ro::wave_inclusive_sum(csum) - csum;
        __syncthreads();
        for (int r = 0; r < BT_PER_LANE; r++) {
            const int w = lane * BT_PER_LANE + r - ro::RANK0;
            if (w >= 0 && w < RO_N) s_Y[w] = run + 0.5 * g[r];
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
        if (lane == 0) {
            // a Gauss point that falls into the interval of its predecessor takes the next one; the reference reports
            // this as a malfunction of the re-binning (:3383-3387), here it is counted (hx_diag_read)
            int skipped = 0;
            for (int q = 1; q < ny; q++)
                if (s_w[q] <= s_w[q - 1]) {
                    s_w[q] = s_w[q - 1] + 1;
                    skipped++;
                }
            if (skipped) atomicAdd(rebin_skipped, (unsigned long long)skipped);
        }
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

// ---- random overlap, default kernel: 32-bit quantised keys + exact finish (random_overlap.h) ---------------------
__global__ void __launch_bounds__(64)
k_add_to_mixed_opac_q32(const double* __restrict__ vmr, const double* __restrict__ opac_spec,
                        double* __restrict__ opac_wg, const double* __restrict__ meanmolmass,
                        const double* __restrict__ gauss_weight, const double* __restrict__ gauss_y,
                        double mass_spec, int nbin, int nlev, unsigned long long* __restrict__ diag) {
    __shared__ ro::Shared sh;
    const int lane = threadIdx.x;
    ro::Lane ln;
    ro::init(sh, ln, lane, gauss_weight, gauss_y);
    // a contiguous run of (bin, level) problems per wavefront: the level -- and with it the factor
    // vmr * mass / mu, a division -- changes once or twice per run
    const long long npair = (long long)nbin * nlev;
    const long long chunk = (npair + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = min(npair, p0 + chunk);
    int i_cur = -1;
    double fac = 0.0;
    ro::Counters cnt;
    for (long long pair = p0; pair < p1; pair++) {
        const int i = (int)(pair / nbin);
        if (i != i_cur) {
            i_cur = i;
            fac = vmr[i] * mass_spec / meanmolmass[i];  // (vmr * mass) / mu, then times kappa (:3293)
        }
        const size_t base = (size_t)RO_NY * pair;  // = ny*x + ny*nbin*i
        double my_mix = 0.0, my_add = 0.0;
        if (lane < RO_NY) {
            my_mix = opac_wg[base + lane];
            my_add = fac * opac_spec[base + lane];
        }
        const double out = ro::mix(sh, ln, lane, my_mix, my_add, cnt);
        if (lane < RO_NY) opac_wg[base + lane] = out;
    }
    ro::flush(cnt, lane, diag);
}

// ---- random overlap, default since round 6: the same network on keys that carry their cell (random_overlap_lean.h) --
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5)))
k_add_to_mixed_opac_lean(const double* __restrict__ vmr, const double* __restrict__ opac_spec,
                         double* __restrict__ opac_wg, const double* __restrict__ meanmolmass,
                         const double* __restrict__ gauss_weight, const double* __restrict__ gauss_y,
                         double mass_spec, int nbin, int nlev, unsigned long long* __restrict__ diag) {
    __shared__ rol::Shared sh;
    const int lane = threadIdx.x;
    const rol::LaneConst lc = rol::init(sh, lane, gauss_weight, gauss_y);
    const long long npair = (long long)nbin * nlev;
    const long long chunk = (npair + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = min(npair, p0 + chunk);
    int i_cur = -1;
    double fac = 0.0;
    ro::Counters cnt;
    for (long long pair = p0; pair < p1; pair++) {
        const int i = (int)(pair / nbin);
        if (i != i_cur) {
            i_cur = i;
            fac = vmr[i] * mass_spec / meanmolmass[i];  // (vmr * mass) / mu, then times kappa (:3293)
        }
        const size_t base = (size_t)RO_NY * pair;  // = ny*x + ny*nbin*i
        double my_mix = 0.0, my_add = 0.0;
        if (lane < RO_NY) {
            my_mix = opac_wg[base + lane];
            my_add = fac * opac_spec[base + lane];
        }
        const double out = rol::mix(sh, lc, lane, my_mix, my_add, cnt);
        if (lane < RO_NY) opac_wg[base + lane] = out;
    }
    ro::flush(cnt, lane, diag);
}

__global__ void __launch_bounds__(256)
k_add_correlated_k(const double* __restrict__ vmr, const double* __restrict__ opac_spec, double* __restrict__ opac_wg,
                   const double* __restrict__ meanmolmass, double mass_spec, int ny, int nbin, int nlev) {
    // kernels.cu:3302-3310 for any ny: opac += (vmr * mass / mu) * kappa
    const size_t per_level = (size_t)ny * nbin;
    const int i = blockIdx.y;
    const double fac = vmr[i] * mass_spec / meanmolmass[i];
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < per_level; k += (size_t)gridDim.x * blockDim.x)
        opac_wg[per_level * i + k] += fac * opac_spec[per_level * i + k];
}

// calc_index_h2o / calc_h2o_scat (kernels.cu:3174-3205, :3404-3440)
__global__ void __launch_bounds__(256)
k_calc_h2o_scat(const double* __restrict__ temp, const double* __restrict__ press,
                const double* __restrict__ wave, double* __restrict__ scat_cross,
                const double* __restrict__ vmr, double mass_h2o, int nbin, int nlev) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (x >= nbin) return;
    const double sc = h2o_rayleigh_cross(temp[i], press[i], vmr[i], wave[x], mass_h2o);
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
    static const int sort_kind = [] {  // cross-check / A-B knob, read once: lean (default), q32 (rounds 2-5), bitonic (fp64 network), rank
        const char* e = getenv("HELIOS_RO_SORT");
        if (e != nullptr && strcmp(e, "rank") == 0) return 2;
        if (e != nullptr && strcmp(e, "bitonic") == 0) return 1;
        if (e != nullptr && strcmp(e, "q32") == 0) return 3;
        return 0;
    }();
    if (sort_kind == 2)
        k_add_to_mixed_opac<false><<<grid, 64, 0, ctx->stream>>>(vmr, opac_spec, opac_wg, meanmolmass, gauss_weight,
                                                                gauss_y, mass_spec, s, ro_method, ny, nbin,
                                                                nlay_or_nint, ctx->diag + HX_DIAG_RO_REBIN);
    else if (sort_kind == 1)
        k_add_to_mixed_opac<true><<<grid, 64, 0, ctx->stream>>>(vmr, opac_spec, opac_wg, meanmolmass, gauss_weight,
                                                               gauss_y, mass_spec, s, ro_method, ny, nbin,
                                                               nlay_or_nint, ctx->diag + HX_DIAG_RO_REBIN);
    else if (!ro_possible)
        k_add_correlated_k<<<dim3(hx_cdiv((long long)ny * nbin, 1024), nlay_or_nint), 256, 0, ctx->stream>>>(
            vmr, opac_spec, opac_wg, meanmolmass, mass_spec, ny, nbin, nlay_or_nint);
    else if (sort_kind == 3)
        k_add_to_mixed_opac_q32<<<grid, 64, 0, ctx->stream>>>(vmr, opac_spec, opac_wg, meanmolmass, gauss_weight,
                                                             gauss_y, mass_spec, nbin, nlay_or_nint, ctx->diag);
    else
        k_add_to_mixed_opac_lean<<<grid, 64, 0, ctx->stream>>>(vmr, opac_spec, opac_wg, meanmolmass, gauss_weight,
                                                              gauss_y, mass_spec, nbin, nlay_or_nint, ctx->diag);
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
