// Device kernels of the fused fast path.  Included by rt_fused.hip only.
//
// Unknowns of one spectral point (x,y) live on NODES n = 0..H (H = 2*nlayer): node 2i = interface i,
// node 2i+1 = centre of layer i.  Half-layer h spans node h (bottom) .. h+1 (top).  With
//   alpha = P/M, beta = -N/M, u' = K u / M, v' = K v / M,  K = 2 pi eps (1-w)/(E-w),
//   u = (M+N) + q, v = -P - q, q = eps/(E(1-w g)) (P-M+N)/dtau   (u = v = (N+M-P)/2 if dtau < limit)
// both sweeps of SURVEY.md 10.3 take the same form
//   down:  D[h]   = alpha D[h+1] + beta U[h]   + u' B[h]   + v' B[h+1] + dd
//   up  :  U[h+1] = alpha U[h]   + beta D[h+1] + u' B[h+1] + v' B[h]   + du
// (B = Planck function at the node; dd/du = min(0, direct-beam term)/M).  The coefficient planes are
// built once per opacity refresh (k_rt_coef) and streamed once per iteration (k_rt_flux), where
// each lane keeps its ROWS half-layers in registers across all 3*scat+1 sweeps.
// Since u + v = M + N - P in both branches, v' = K (1 - alpha - beta) - u'; without the improved
// two-stream correction E == 1 and K = 2 pi eps is a constant, so the v' plane is not stored then.
#pragma once
#include "rt_fused.h"
#include "conv_adjust.h"
#include "temp_step.h"
#include "two_stream.h"

namespace hx {

struct KArgs {
    int X, Y, L, I, H, C;
    int k, ROWS, S, nxb, ypb, nparts, G, NW, nblk_x, nblk, nplane, nchunk, coef_nbx;
    int cloud_lds;             // k_rt_coef stages the clouds' half-layer terms in LDS (when they fit)
    int scat, dir_beam, clouds, scat_corr, nsweep, keep_down, real_star;
    int iso;                   // isothermal layers: H = L segments between the interfaces, one coefficient set per layer
    int has_vp, pl_vp, pl_dd;  // v' plane stored? plane indices of v' and of dd (du = dd + 1)
    int matrix;                // `flux calculation method = matrix` as scans (k_rt_flux<.., true>): k_rt_coef serves its two branches
    int* trigger;              // [C][Y X] scat_trigger of calc_trans_* (kernels.cu:1102, :1240), matrix method only
    double Kconst;             // 2 pi eps: source prefactor when E == 1 (scat_corr == 0)
    int dim, step;
    double epsi, epsi2, g_0, i2s, w_0_limit, w_0_scat_limit, dtau_limit;
    const hx_rt_column* colpar;
    const double *T_lay, *p_lay, *p_int, *dcol_u, *dcol_l, *surf_albedo, *Bstar, *planck_grid;
    const double *opac_wg_lay, *opac_wg_int, *scat_cross_lay, *scat_cross_int, *mmm_lay, *mmm_int;
    const double *cl_abs_lay, *cl_abs_int, *cl_sc_lay, *cl_sc_int, *g0_tot_lay, *g0_tot_int;
    double *half_ray, *half_g0, *half_cab, *half_csc;  // [C][X][H] half-layer band quantities, bin-major (k_rt_half_bands)
    const double *F_dir_wg, *Fc_dir_wg, *F_dir_band_n, *gauss_w, *deltawave;
    double *T_int, *Bn, *coef, *Utile, *Dtile, *U0, *boaK, *Fdir0;
    double *dtau_u, *dtau_l;
    double *F_down_band_n, *F_up_band_n, *tot_part, *F_up_tot, *F_down_tot, *F_net;
    size_t coef_col, flux_col;  // per-column strides (doubles) of coef / Utile / Dtile
    const int* done;
    // premixed table look-up fused into the coefficient kernel
    const double *ktable, *crosstable, *ktemp, *kpress;
    const TPIndex *tp_lay, *tp_int;  // [C][I] fractional table indices of the levels
    int ntemp, npress, from_table;
    unsigned long long* diag;  // hx_context::diag when the batch runs with debug = 1, else nullptr
    int* iter_dev;             // [0] index of the next iteration, [1] index of the iteration under way (k_rt_nodes)
};

// the subset k_rt_flux needs (a leaner argument block keeps its SGPR pressure -- and with it the VGPR
// count, which sits at the 256-register / 2-waves-per-SIMD edge -- down)
struct FluxArgs {
    int X, Y, L, I, H;
    int k, nxb, ypb, nparts, G, NW;
    int dir_beam, nsweep, keep_down, has_vp, pl_vp, pl_dd, nplane, iso, debug_skip;
    int keep_up;           // matrix method: store the up-fluxes too (the sweeps always do: their state); 0 inside the loop
    int reverse;           // walk the grid from its far end (see launch_flux)
    int cache_state_from;  // dispatch index from which the state stores stay cached
    const int* trigger;    // matrix method: scat_trigger per spectral point
    double Kconst;
    const hx_rt_column* colpar;
    const double *Bn, *coef, *U0_in, *boaK, *Fdir0, *surf_albedo, *gauss_w;
    double *Utile, *Dtile, *U0, *F_down_band_n, *F_up_band_n;
    size_t coef_col, flux_col;
    const int* done;
};

__device__ __forceinline__ double interface_T(const double* T, int i, int L) {
    if (i == 0) return T[0] - 0.5 * (T[1] - T[0]);
    if (i == L) return T[L - 1] + 0.5 * (T[L - 1] - T[L - 2]);
    return T[i - 1] + 0.5 * (T[i] - T[i - 1]);
}

// Tile planes hold ROWS rows of 64 lanes, [row][lane]: every wavefront load/store of a row is one
// contiguous 512-byte segment.  (A paired-row layout with 16-byte accesses per lane was measured: the
// 16-byte register alignment pushed k_rt_flux<13> from 218 to 256+ VGPRs with scratch spills and made
// it 40 % slower, so rows stay scalar.)  Offset of (row r, lane) inside a plane:
__host__ __device__ __forceinline__ size_t plane_off(int r, int lane, int ROWS) {
    (void)ROWS;
    return (size_t)r * 64 + lane;
}

// ---- per iteration: interface temperatures + Planck function at every node ------------------
// Bn[col][x][n], n in [0, H+3): nodes 0..H, then H+1 = stellar row, H+2 = surface (T_lay[L]).
__global__ void __launch_bounds__(256) k_rt_nodes(KArgs a) {
    __shared__ double tile[32][33];
    const int col = blockIdx.z;
    // first kernel of every iteration: the iteration counter moves on (one thread of the launch; before the `done` test,
    // so that it counts launches, not live columns)
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && a.iter_dev != nullptr) {
        const int cur = a.iter_dev[0];
        a.iter_dev[1] = cur;
        a.iter_dev[0] = cur + 1;
    }
    if (a.done[col]) return;
    const int NN = a.H + 3;
    const double* T = a.T_lay + (size_t)col * (a.L + 1);
    const int x0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < a.I; i += blockDim.x)
            a.T_int[(size_t)col * a.I + i] = interface_T(T, i, a.L);
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, x = x0 + tx;
        if (n < NN && x < a.X) {
            double v;
            if (n == a.H + 1) {
                v = a.Bstar[(size_t)col * a.X + x];
            } else {
                double Tn;
                if (n == a.H + 2) Tn = T[a.L];
                else if (a.iso) Tn = T[min(n, a.L - 1)];  // isothermal layers: "node" n = layer n (slot H is not used)
                else if (n & 1) Tn = T[(n - 1) >> 1];
                else Tn = interface_T(T, n >> 1, a.L);
                v = planck_lookup(a.planck_grid, Tn, x, a.X, a.dim, a.step);
            }
            tile[r][tx] = v;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int x = x0 + r, n = n0 + tx;
        if (n < NN && x < a.X) a.Bn[((size_t)col * a.X + x) * NN + n] = tile[tx][r];
    }
}

// ---- neighbour-lane moves for the scans --------------------------------------------------------
// Distances 1, 2, 4, 8 stay inside a 16-lane DPP row: `row_shr:n` / `row_shl:n` move a register across
// lanes in the VALU (a few cycles) instead of going through the LDS crossbar (ds_bpermute, ~100 cycles
// of dependent latency per scan step).  Values arriving from outside the spectral point's k-lane group
// are discarded by the callers' `j` conditions, so no masking is needed here.  Groups wider than a
// DPP row (k = 32, 64) use __shfl for every distance.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {  // lanes without a source (or outside ROW_MASK) keep their own value
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int N>
__device__ __forceinline__ double from_lane_below(double v, int k) {  // value of lane (id - N)
    if (N == 1 && k == 32) return dpp_move<0x138, 0xf>(v);  // wave_shr:1 crosses the row boundary inside the group
    if (N < 16 && k <= 16) {  // wave-uniform: the k-lane group lies inside one DPP row
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x110 + (N & 15), 0xf, 0xf, false);  // row_shr:N
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x110 + (N & 15), 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl_up(v, N);
    }
}
template <int N>
__device__ __forceinline__ double from_lane_above(double v, int k) {  // value of lane (id + N)
    if (N == 1 && k == 32) return dpp_move<0x130, 0xf>(v);  // wave_shl:1
    if (N < 16 && k <= 16) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + (N & 15), 0xf, 0xf, false);  // row_shl:N
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + (N & 15), 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl_down(v, N);
    }
}

// one Kogge-Stone step of the affine-map scans: (A, B) <- (A, B) o (A, B)[neighbour at distance N]
template <int N>
__device__ __forceinline__ void scan_step_down(double& A, double& Bc, int j, int k) {
    if (N < k) {  // wave-uniform
        const double A2 = from_lane_above<N>(A, k), B2 = from_lane_above<N>(Bc, k);
        if (j + N < k) {
            Bc = fma(A, B2, Bc);
            A *= A2;
        }
    }
}
template <int N>
__device__ __forceinline__ void scan_step_up(double& A, double& Bc, int j, int k) {
    if (N < k) {
        const double A2 = from_lane_below<N>(A, k), B2 = from_lane_below<N>(Bc, k);
        if (j >= N) {
            Bc = fma(A, B2, Bc);
            A *= A2;
        }
    }
}

// k = 32: the group spans two DPP rows.  Kogge-Stone inside each row (row_shl / row_shr by 1, 2, 4, 8), then the rows are
// joined: upwards every lane of the upper row composes with the lower row's total, which `row_bcast:15` delivers; downwards
// the lower row needs the upper row's total, lane 16 of the group, read through the scalar unit.  All in the VALU: with
// __shfl (LDS crossbar) for every distance this tiling ran 45 % slower than k = 16.
__device__ __forceinline__ double group_lane16(double v, int lane) {  // value of lane 16 of this lane's 32-lane group
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo_a = __builtin_amdgcn_readlane(lo, 16), lo_b = __builtin_amdgcn_readlane(lo, 48);
    const int hi_a = __builtin_amdgcn_readlane(hi, 16), hi_b = __builtin_amdgcn_readlane(hi, 48);
    return __hiloint2double(lane < 32 ? hi_a : hi_b, lane < 32 ? lo_a : lo_b);
}

#define HX_SCAN32_ROW_STEP(CTRL, COND)                                              \
    {                                                                               \
        const double A2 = dpp_move<CTRL, 0xf>(A), B2 = dpp_move<CTRL, 0xf>(Bc);     \
        if (COND) {                                                                 \
            Bc = fma(A, B2, Bc);                                                    \
            A *= A2;                                                                \
        }                                                                           \
    }

__device__ __forceinline__ void scan32_down(double& A, double& Bc, int j, int lane) {  // suffix composition over j
    const int jr = j & 15;
    HX_SCAN32_ROW_STEP(0x101, jr + 1 < 16)  // row_shl:1
    HX_SCAN32_ROW_STEP(0x102, jr + 2 < 16)
    HX_SCAN32_ROW_STEP(0x104, jr + 4 < 16)
    HX_SCAN32_ROW_STEP(0x108, jr + 8 < 16)
    const double A2 = group_lane16(A, lane), B2 = group_lane16(Bc, lane);
    if (j < 16) {
        Bc = fma(A, B2, Bc);
        A *= A2;
    }
}

__device__ __forceinline__ void scan32_up(double& A, double& Bc, int j) {  // prefix composition over j
    const int jr = j & 15;
    HX_SCAN32_ROW_STEP(0x111, jr >= 1)  // row_shr:1
    HX_SCAN32_ROW_STEP(0x112, jr >= 2)
    HX_SCAN32_ROW_STEP(0x114, jr >= 4)
    HX_SCAN32_ROW_STEP(0x118, jr >= 8)
    const double A2 = dpp_move<0x142, 0xa>(A), B2 = dpp_move<0x142, 0xa>(Bc);  // row_bcast:15 into rows 1 and 3
    if (j >= 16) {
        Bc = fma(A, B2, Bc);
        A *= A2;
    }
}
#undef HX_SCAN32_ROW_STEP

// which spectral point / layer chunk a thread of a flux workgroup works on
struct LaneMap {
    int lane, wv, j, x, y, xl, yl;
    bool valid;
    size_t sp;        // y + Y*x
    size_t tile;      // index of this lane's wavefront tile within the column
};

// threadIdx.x behind an optimisation barrier: what is derived from it is recomputed where it is used instead of being
// carried (and, in a kernel at the register limit, spilled to scratch) across the sweeps
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// workgroup bx covers bins [bx*nxb, (bx+1)*nxb) and walks the nparts groups of ypb Gauss points
template <class Args>
__device__ __forceinline__ LaneMap lane_map(const Args& a, int bx, int part, int tid) {
    LaneMap m;
    m.lane = tid & 63;
    m.wv = tid >> 6;
    m.j = m.lane & (a.k - 1);                       // k is a power of two (choose_workgroup)
    const int s_local = tid >> (31 - __clz(a.k));
    m.xl = s_local / a.ypb;
    m.yl = s_local - m.xl * a.ypb;
    m.x = bx * a.nxb + m.xl;
    m.y = part * a.ypb + m.yl;
    m.valid = s_local < a.G && m.x < a.X;
    m.sp = (size_t)m.y + (size_t)a.Y * m.x;
    m.tile = ((size_t)bx * a.nparts + part) * a.NW + m.wv;
    return m;
}

// ---- scans with the lanes per spectral point known at compile time (K = 16, 32) ------------------------------
// The neighbour's map arrives through DPP; a lane that has no neighbour inside its 16-lane row receives the identity
// map (A2 = 1: `old` keeps 1.0's high word, bound_ctrl zeroes the rest; B2 = 0), so the composition is unconditional --
// no compare, no selects, no wave-uniform branches on k.  Composing with the identity is exact: same bits as the
// generic path's `if (j + N < k)`.
template <int CTRL>
__device__ __forceinline__ void compose_row_neighbour(double& A, double& Bc) {
    const int a_lo = __builtin_amdgcn_update_dpp(0, __double2loint(A), CTRL, 0xf, 0xf, true);
    const int a_hi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(A), CTRL, 0xf, 0xf, false);
    const int b_lo = __builtin_amdgcn_update_dpp(0, __double2loint(Bc), CTRL, 0xf, 0xf, true);
    const int b_hi = __builtin_amdgcn_update_dpp(0, __double2hiint(Bc), CTRL, 0xf, 0xf, true);
    Bc = fma(A, __hiloint2double(b_hi, b_lo), Bc);
    A *= __hiloint2double(a_hi, a_lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_or_zero(double v) {  // the DPP-selected lane's value, 0.0 where there is none
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// value of lane SRC of the wavefront, through the scalar unit
template <int SRC>
__device__ __forceinline__ double wave_lane(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), SRC), __builtin_amdgcn_readlane(__double2loint(v), SRC));
}

template <int K>
__device__ __forceinline__ void scan_down_fixed(double& A, double& Bc, int j, int lane) {  // suffix composition over j
    compose_row_neighbour<0x101>(A, Bc);  // row_shl:1
    compose_row_neighbour<0x102>(A, Bc);
    compose_row_neighbour<0x104>(A, Bc);
    compose_row_neighbour<0x108>(A, Bc);
    if (K == 32 || K == 64) {  // rows 0 and 2 take the total of the row above them: its first lane, 16 or 48
        const double A2 = group_lane16(A, lane), B2 = group_lane16(Bc, lane);
        if ((lane & 16) == 0) {
            Bc = fma(A, B2, Bc);
            A *= A2;
        }
    }
    if (K == 64) {             // rows 0 and 1 take the total of rows 2 and 3, which lane 32 now holds
        const double A2 = wave_lane<32>(A), B2 = wave_lane<32>(Bc);
        if (lane < 32) {
            Bc = fma(A, B2, Bc);
            A *= A2;
        }
    }
}
template <int K>
__device__ __forceinline__ void scan_up_fixed(double& A, double& Bc, int j) {  // prefix composition over j
    compose_row_neighbour<0x111>(A, Bc);  // row_shr:1
    compose_row_neighbour<0x112>(A, Bc);
    compose_row_neighbour<0x114>(A, Bc);
    compose_row_neighbour<0x118>(A, Bc);
    if (K == 32 || K == 64) {
        const double A2 = dpp_move<0x142, 0xa>(A), B2 = dpp_move<0x142, 0xa>(Bc);  // row_bcast:15 into rows 1 and 3
        if (j & 16) {
            Bc = fma(A, B2, Bc);
            A *= A2;
        }
    }
    if (K == 64) {
        const double A2 = dpp_move<0x143, 0xc>(A), B2 = dpp_move<0x143, 0xc>(Bc);  // row_bcast:31 into rows 2 and 3
        if (j >= 32) {
            Bc = fma(A, B2, Bc);
            A *= A2;
        }
    }
}
// value of the lane below / above inside the group; the group's first / last lane gets something its caller overrides
template <int K>
__device__ __forceinline__ double below_fixed(double v) {
    return K == 16 ? dpp_or_zero<0x111>(v) : dpp_or_zero<0x138>(v);  // row_shr:1 / wave_shr:1
}
template <int K>
__device__ __forceinline__ double above_fixed(double v) {
    return K == 16 ? dpp_or_zero<0x101>(v) : dpp_or_zero<0x130>(v);  // row_shl:1 / wave_shl:1
}
template <int K>
__device__ __forceinline__ double group_first_lane(double v, int lane) {  // lane 0 of this lane's K-lane group
    if (K == 16) return dpp_or_zero<0x150>(v);  // row_newbcast:0
    if (K == 64) return wave_lane<0>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo_a = __builtin_amdgcn_readlane(lo, 0), lo_b = __builtin_amdgcn_readlane(lo, 32);
    const int hi_a = __builtin_amdgcn_readlane(hi, 0), hi_b = __builtin_amdgcn_readlane(hi, 32);
    return __hiloint2double(lane < 32 ? hi_a : hi_b, lane < 32 ? lo_a : lo_b);
}
// tiny_abs with the threshold as data: 1e-100 where the reference applies it, 0.0 (never true) where it does not
__device__ __forceinline__ double tiny_abs_below(double F, double thr) { return fabs(F) < thr ? fabs(F) : F; }

// ---- per refresh: band quantities of the half-layers, bin-major ---------------------------------
// grid (ceil(X/32), ceil(H/32), C), 256 threads.  The band arrays are level-major ([i][x], the reference's layout); a
// workgroup of k_rt_coef needs ALL half-layers of one or two bins, which there is one 64-byte sector per double.  This
// transposes them once per refresh (32 x 32 tiles through LDS) into [x][h] rows that k_rt_coef stages with unit stride:
// Rayleigh cross-section and, with clouds, asymmetry parameter, absorption and scattering cross-sections -- the
// half-layer averages of calc_trans_noniso (kernels.cu:1131-1177), or the layer values themselves (calc_trans_iso).
__global__ void __launch_bounds__(256) k_rt_half_bands(KArgs a) {
    __shared__ double tile[4][32][33];
    const int col = blockIdx.z;
    if (a.done[col]) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t bandI = (size_t)a.X * a.I;
    const double* src_i[4] = {a.scat_cross_int + col * bandI, a.g0_tot_int + col * bandI, a.cl_abs_int + col * bandI,
                              a.cl_sc_int + col * bandI};
    const double* src_l[4] = {a.scat_cross_lay + col * bandI, a.g0_tot_lay + col * bandI, a.cl_abs_lay + col * bandI,
                              a.cl_sc_lay + col * bandI};
    double* dst[4] = {a.half_ray, a.half_g0, a.half_cab, a.half_csc};
    const int nq = a.clouds == 1 ? 4 : 1;
    const int x = blockIdx.x * 32 + tx;
    for (int hh = ty; hh < 32; hh += 8) {
        const int h = blockIdx.y * 32 + hh;
        if (x < a.X && h < a.H) {
            const int i = a.iso ? h : h >> 1, ii = a.iso ? h : i + (h & 1);
            const size_t b_l = x + (size_t)a.X * i, b_i = x + (size_t)a.X * ii;
            for (int q = 0; q < nq; q++) {
                const bool on = q == 1 || q == 2 || a.scat == 1;  // Rayleigh and cloud scattering only with scat = 1
                tile[q][hh][tx] = !on ? 0.0 : a.iso ? src_l[q][b_l] : (src_i[q][b_i] + src_l[q][b_l]) / 2.0;
            }
        }
    }
    __syncthreads();
    const int h = blockIdx.y * 32 + tx;
    for (int xx = ty; xx < 32; xx += 8) {
        const int xo = blockIdx.x * 32 + xx;
        if (xo < a.X && h < a.H)
            for (int q = 0; q < nq; q++) dst[q][((size_t)col * a.X + xo) * a.H + h] = tile[q][tx][xx];
    }
}

// ---- per refresh: compact coefficient tiles ---------------------------------------------------
// grid (ceil(ntiles / COEF_TPB), C), COEF_TPB wavefronts per workgroup, one tile each.  The opacities
// live in the reference's layout [y + ny*x + ny*nbin*level] (level slowest), so the spectral points of
// one tile are only 64/k doubles apart per level: the workgroup first stages the opacities of ALL its
// tiles' spectral points (consecutive in memory when a tile row holds one bin) for every level into
// LDS with >= 128-byte contiguous reads, then every lane builds the coefficients of its half-layers.
template <int ROWS, int COEF_TPB>
__global__ void __launch_bounds__(64 * COEF_TPB) k_rt_coef(KArgs a) {
    extern __shared__ __align__(16) double smem[];
    const int col = blockIdx.y;
    if (a.done[col]) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ntiles = a.nblk_x * a.nparts * a.NW;
    const int TS = COEF_TPB * a.S;                  // spectral points staged per workgroup
    const int TSP = TS;                             // row pitch of the staged opacities (padding it was measured: slower)
    const int NBX = a.coef_nbx;                     // bins this workgroup's tiles can touch (upper bound)
    double* sh_lay = smem;                          // [L][TSP]  opacity at layer centres
    double* sh_int = sh_lay + (size_t)a.L * TSP;    // [I][TSP]  opacity at interfaces
    double* sh_ray = sh_int + (size_t)a.I * TSP;    // [H][NBX]  Rayleigh cross-section of the half-layers
    double* sh_mu = sh_ray + (size_t)a.H * NBX;     // [H]       mean molecular mass of the half-layers
    double* sh_dc = sh_mu + a.H;                    // [H]       column mass of the half-layers
    // clouds: asymmetry parameter, absorption and scattering cross-sections of the half-layers, [H][NBX] each.  They do
    // not depend on the Gauss point; read per lane from the band arrays (level-strided: one 64-byte sector per double)
    // they were 18 scattered loads per half-layer -- config 5's k_rt_coef 7.4 ms per refresh against 2.9 ms expected
    // from config 2's rate
    const size_t ncl = a.cloud_lds ? (size_t)a.H * NBX : 0;
    double* sh_g0 = sh_dc + a.H;
    double* sh_cab = sh_g0 + ncl;
    double* sh_csc = sh_cab + ncl;
    int* c_of_q = (int*)(sh_csc + ncl);             // [TS] global spectral-point index or -1
    int* x_of_q = c_of_q + TS;
    const size_t nc = (size_t)a.Y * a.X;
    const size_t wgI = nc * a.I;
    const int x_base = (blockIdx.x * COEF_TPB) / (a.NW * a.nparts) * a.nxb;  // first bin of the first tile
    // spectral points of this workgroup's tiles
    for (int q = threadIdx.x; q < TS; q += blockDim.x) {
        const int tl = blockIdx.x * COEF_TPB + q / a.S, s_in_wave = q % a.S;
        int c = -1, xq = x_base;
        if (tl < ntiles) {
            const int wv = tl % a.NW, part = (tl / a.NW) % a.nparts, bx = tl / (a.NW * a.nparts);
            const int s_local = wv * a.S + s_in_wave;
            const int xl = s_local / a.ypb, yl = s_local - xl * a.ypb;
            const int x = bx * a.nxb + xl, y = part * a.ypb + yl;
            if (s_local < a.G && x < a.X) { c = y + a.Y * x; xq = x; }
        }
        c_of_q[q] = c;
        x_of_q[q] = xq;
    }
    // per-level quantities of the half-layers.  Lower half h = 2i averages (interface i, centre i),
    // upper half h = 2i+1 (centre i, interface i+1); the sums are commutative, so one form serves both.
    {
        const double* mml = a.mmm_lay + (size_t)col * a.I;
        const double* mmi = a.mmm_int + (size_t)col * a.I;
        const double* dcu = a.dcol_u + (size_t)col * a.L;
        const double* dcl = a.dcol_l + (size_t)col * a.L;
        const double* pint = a.p_int + (size_t)col * a.I;
        const double grav = a.colpar[col].g;
        for (int h = threadIdx.x; h < a.H; h += blockDim.x) {
            if (a.iso) {  // whole layers (calc_trans_iso, kernels.cu:1015-1104): delta_colmass of host_functions.py:733
                sh_mu[h] = mml[h];
                sh_dc[h] = (pint[h] - pint[h + 1]) / grav;
            } else {
                const int i = h >> 1, ii = i + (h & 1);
                sh_mu[h] = (mmi[ii] + mml[i]) / 2.0;
                sh_dc[h] = (h & 1) ? dcu[i] : dcl[i];
            }
        }
        // bin-major rows written by k_rt_half_bands: consecutive threads read consecutive half-layers of a bin
        for (int t = threadIdx.x; t < a.H * NBX; t += blockDim.x) {
            const int xs = t / a.H, h = t - xs * a.H, x = min(x_base + xs, a.X - 1);
            const size_t src = ((size_t)col * a.X + x) * a.H + h;
            sh_ray[(size_t)h * NBX + xs] = a.half_ray[src];
            if (a.cloud_lds) {
                sh_g0[(size_t)h * NBX + xs] = a.half_g0[src];
                sh_cab[(size_t)h * NBX + xs] = a.half_cab[src];
                sh_csc[(size_t)h * NBX + xs] = a.half_csc[src];
            }
        }
    }
    __syncthreads();
    {
        // thread -> (staged point q0, a run of consecutive levels): consecutive levels mostly fall into
        // the same (T, P) cell of the table, whose four corners then stay in registers.  TS divides the
        // workgroup size (both are powers of two).
        const int q0 = threadIdx.x % TS, nrun = blockDim.x / TS, run = threadIdx.x / TS;
        const int cq = c_of_q[q0];
        if (a.from_table) {
            // premixed k-table look-up done while staging (kernels.cu:561-608): the opacity arrays of
            // the reference are not materialised on this path (hx_rt_get rebuilds them on demand)
            const size_t sp = nc, st = nc * a.npress;
            for (int pass = 0; pass < (a.iso ? 1 : 2); pass++) {
                const int nlev = pass == 0 ? a.L : a.I;
                const TPIndex* tp = (pass == 0 ? a.tp_lay : a.tp_int) + (size_t)col * a.I;
                double* dst = pass == 0 ? sh_lay : sh_int;
                const int per = (nlev + nrun - 1) / nrun;
                const int l0 = run * per, l1 = min(nlev, l0 + per);
                int ktd = -1, ktu = -1, kpd = -1, kpu = -1;
                double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
                // (the cell of the next level is requested while this level's corners are on their way: the look-up was a
                // chain of two dependent requests per level)
                TPIndex knext = tp[min(l0, nlev - 1)];
                for (int lev = l0; lev < l1; lev++) {
                    double v = 0.0;
                    const TPIndex k = knext;
                    knext = tp[min(lev + 1, nlev - 1)];
                    if (cq >= 0) {
                        if (k.tdown != ktd || k.tup != ktu || k.pdown != kpd || k.pup != kpu) {
                            const double* t0 = a.ktable + (size_t)cq + st * k.tdown;
                            const double* t1 = a.ktable + (size_t)cq + st * k.tup;
                            c00 = t0[sp * k.pdown];
                            c01 = t0[sp * k.pup];
                            c10 = t1[sp * k.pdown];
                            c11 = t1[sp * k.pup];
                            ktd = k.tdown; ktu = k.tup; kpd = k.pdown; kpu = k.pup;
                        }
                        v = blend_tp(c00, c01, c10, c11, k, false);
                    }
                    dst[(size_t)lev * TSP + q0] = v;
                }
            }
        } else {
            const double* opl = a.opac_wg_lay + col * wgI;
            const double* opi = a.opac_wg_int + col * wgI;
            for (int lev = run; lev < a.L; lev += nrun)
                sh_lay[(size_t)lev * TSP + q0] = cq >= 0 ? opl[(size_t)cq + nc * lev] : 0.0;
            if (!a.iso)
                for (int lev = run; lev < a.I; lev += nrun)
                    sh_int[(size_t)lev * TSP + q0] = cq >= 0 ? opi[(size_t)cq + nc * lev] : 0.0;
        }
    }
    __syncthreads();
    const int tl = blockIdx.x * COEF_TPB + wave;
    if (tl >= ntiles) return;
    const int j = lane % a.k, q = wave * a.S + lane / a.k;
    const int c = c_of_q[q], x = x_of_q[q], xs = x - x_base;
    const bool valid = c >= 0;
    const hx_rt_column cp = a.colpar[col];
    double* ctile = a.coef + col * a.coef_col + (size_t)tl * a.nplane * ROWS * 64;
    const double nmu = -cp.mu_star;
    const bool plain = a.clouds != 1 && a.scat_corr != 1 && a.g_0 == 0.0 && a.dir_beam != 1;
    // The beam at the nodes of this lane's half-layers.  Half-layer h spans the nodes h (bottom) and h + 1 (top) -- even
    // nodes are interfaces (F_dir_wg), odd ones layer centres (Fc_dir_wg); isothermal: node = interface -- so the top value
    // of one row is the bottom value of the next: ONE load per row instead of two, requested a whole row of arithmetic
    // (divisions, exp, sqrt) before it is used.  (Loaded where they were used, the compiler sent all of a tile's beam
    // loads through one register pair, each waited for in turn: DESIGN.md section 4, tools/code_object_notes.py.)
    const double* Fd = a.F_dir_wg + col * wgI;
    const double* Fc = a.Fc_dir_wg + col * wgI;
    auto beam_at_node = [&](int n) -> double {
        if (!(a.dir_beam == 1 && valid && n <= a.H)) return 0.0;
        if (a.iso) return Fd[(size_t)c + nc * n];
        return (n & 1) ? Fc[(size_t)c + nc * (n >> 1)] : Fd[(size_t)c + nc * (n >> 1)];
    };
    double F_here = beam_at_node(j * ROWS), F_above = beam_at_node(j * ROWS + 1);
    // `flux calculation method = matrix`: a spectral point none of whose (half-)layers scatters (w0 <= w_0_scat_limit in all
    // of them: scat_trigger stays 0, kernels.cu:1102, :1240-1241) takes the solver's pure-absorption branch (:1969-2021,
    // :2286-2421): F_out = T F_in + 2 pi eps P' -- the same affine form with alpha = T, beta = 0 and sources that are again
    // u' B_near + v' B_far, so the planes serve both branches.  The trigger is a property of the whole column: the k lanes of a
    // point vote before any of them writes a coefficient.
    bool scatters = true;
    if (a.matrix) {
        bool mine = false;
#pragma unroll 1
        for (int r = 0; r < ROWS; r++) {
            const int h = j * ROWS + r;
            if (valid && h < a.H) {
                const int i = a.iso ? h : h >> 1;
                const bool lower = a.iso || (h & 1) == 0;
                const int ii = lower ? i : i + 1;
                double ray = 0.0, csc = 0.0, cab = 0.0;
                if (a.cloud_lds) {
                    cab = sh_cab[(size_t)h * NBX + xs];
                    csc = sh_csc[(size_t)h * NBX + xs];
                } else if (a.clouds == 1) {
                    const size_t src = ((size_t)col * a.X + x) * a.H + h;
                    cab = a.half_cab[src];
                    csc = a.half_csc[src];
                }
                if (a.scat == 1) ray = sh_ray[(size_t)h * NBX + xs];
                const double o_l = sh_lay[(size_t)i * TSP + q], o_i = a.iso ? o_l : sh_int[(size_t)ii * TSP + q];
                const double kap = a.iso ? o_l : (lower ? (o_i + o_l) / 2.0 : (o_l + o_i) / 2.0);
                mine = mine || single_scat_albedo(ray + csc, kap * sh_mu[h] + cab, a.w_0_limit) > a.w_0_scat_limit;
            }
        }
        const unsigned long long votes = __ballot(mine);
        const unsigned long long group = a.k >= 64 ? ~0ull : ((1ull << a.k) - 1ull) << (lane - j);
        scatters = (votes & group) != 0ull;
        if (valid && j == 0) a.trigger[col * nc + c] = scatters ? 1 : 0;
    }
    for (int r = 0; r < ROWS; r++) {
        const int h = j * ROWS + r;
        double alpha = 1.0, beta = 0.0, up = 0.0, vp = 0.0, dd = 0.0, du = 0.0;
        const double Fbot = F_here, Ftop = F_above;
        F_here = F_above;
        F_above = beam_at_node(h + 2);      // the next row's top node: in flight during this row's arithmetic
        if (valid && h < a.H) {
            const int i = a.iso ? h : h >> 1;
            const bool lower = a.iso || (h & 1) == 0;
            // lower half averages (interface i, centre i); upper half (centre i, interface i+1); isothermal layers take
            // the layer-centre values as they are
            const int ii = lower ? i : i + 1;
            double g0 = a.g_0, ray = 0.0, csc = 0.0, cab = 0.0;
            if (a.cloud_lds) {
                g0 = sh_g0[(size_t)h * NBX + xs];
                cab = sh_cab[(size_t)h * NBX + xs];
                csc = sh_csc[(size_t)h * NBX + xs];
            } else if (a.clouds == 1) {  // the staged image would not fit the LDS: from the bin-major rows
                const size_t src = ((size_t)col * a.X + x) * a.H + h;
                g0 = a.half_g0[src];
                cab = a.half_cab[src];
                csc = a.half_csc[src];
            }
            if (a.scat == 1) ray = sh_ray[(size_t)h * NBX + xs];
            const double o_l = sh_lay[(size_t)i * TSP + q], o_i = a.iso ? o_l : sh_int[(size_t)ii * TSP + q];
            const double kap = a.iso ? o_l : (lower ? (o_i + o_l) / 2.0 : (o_l + o_i) / 2.0);
            const double mu = sh_mu[h], dcol = sh_dc[h];
            const double w0 = single_scat_albedo(ray + csc, kap * mu + cab, a.w_0_limit);
            const double dtau_gas = dcol * (kap + ray / mu);
            // `plain` (wave-uniform): no clouds, no I2S correction, g0 = 0, no beam -- the cloud term is an exact zero and
            // E (1 - w0 g0) an exact one: the general formulas minus their no-ops, same bits (two_stream.h)
            const double dtau = plain ? dtau_gas : dtau_gas + dcol * (cab + csc) / mu;
            const Slab s = plain ? slab_coeffs_plain(w0, dtau, a.epsi)
                                 : slab_coeffs(w0, dtau, g0, a.epsi, a.epsi2, cp.mu_star, a.scat_corr, a.i2s, a.dir_beam == 1);
            if (a.diag != nullptr && a.dir_beam == 1) {  // G_limiter's warning (kernels.cu:217-231) as a count
                const int nlim = (fabs(s.Gp) >= 1e8 ? 1 : 0) + (fabs(s.Gm) >= 1e8 ? 1 : 0);
                if (nlim) atomicAdd(a.diag + HX_DIAG_G_LIMITED, (unsigned long long)nlim);
            }
            double invM = 1.0 / s.M;
            alpha = s.P * invM;
            beta = -s.N * invM;
            double K = 2.0 * HX_PI * a.epsi * (1.0 - w0) / (s.E - w0);
            double u, v;
            if (!scatters) {
                // pure absorption (matrix method, see above): down P' = B_b - T B_t + eps (T - 1) (B_b - B_t) / dtau, up the
                // same with the nodes exchanged (kernels.cu:2300-2316, :2376-2411); thin or isothermal: (1 - T) (B_b + B_t) / 2
                alpha = s.trans;
                beta = 0.0;
                invM = 1.0;
                K = 2.0 * HX_PI * a.epsi;
                if (a.iso || dtau < a.dtau_limit) {
                    u = v = (1.0 - s.trans) / 2.0;
                } else {
                    const double gq = a.epsi * (s.trans - 1.0) / dtau;
                    u = 1.0 + gq;
                    v = -s.trans - gq;
                }
            } else if (a.iso || dtau < a.dtau_limit) {  // isothermal source: B (N + M - P) (kernels.cu:1442, :1640-1643)
                u = v = (s.N + s.M - s.P) / 2.0;
            } else {
                const double qq = (plain ? a.epsi : a.epsi / (s.E * (1.0 - w0 * g0))) * (s.P - s.M + s.N) / dtau;
                u = (s.M + s.N) + qq;
                v = -s.P - qq;
            }
            up = K * u * invM;
            vp = K * v * invM;
            if (a.dir_beam == 1 && scatters) {
                // beam at node h (bottom) and h+1 (top) of this half-layer: Fbot, Ftop from above
                const double dn = Fbot / nmu * (s.Gm * s.M + s.Gp * s.N) - Ftop / nmu * s.Gm * s.P;
                const double upw = Ftop / nmu * (s.Gm * s.N + s.Gp * s.M) - Fbot / nmu * s.P * s.Gp;
                dd = dmin(0.0, dn) * invM;
                du = dmin(0.0, upw) * invM;
            }
            if (h == 0) {
                a.boaK[col * nc + c] = scatters ? (1.0 - w0) / (s.E - w0) : 1.0;   // (pure absorption: (1 - A) pi B_surf, :2349)
                a.Fdir0[col * nc + c] = a.dir_beam == 1 ? (a.F_dir_wg + col * wgI)[c] : 0.0;
            }
        }
        const size_t off = plane_off(r, lane, ROWS);
        // written once per refresh, streamed by k_rt_flux afterwards: past the L2
        __builtin_nontemporal_store(alpha, ctile + 0 * ROWS * 64 + off);
        __builtin_nontemporal_store(beta, ctile + 1 * ROWS * 64 + off);
        __builtin_nontemporal_store(up, ctile + 2 * ROWS * 64 + off);
        if (a.has_vp) __builtin_nontemporal_store(vp, ctile + (size_t)a.pl_vp * ROWS * 64 + off);
        if (a.dir_beam == 1) {
            __builtin_nontemporal_store(dd, ctile + (size_t)a.pl_dd * ROWS * 64 + off);
            __builtin_nontemporal_store(du, ctile + (size_t)(a.pl_dd + 1) * ROWS * 64 + off);
        }
    }
}

// gas optical depths of the half-layers in the reference's layout (needed by the direct beam only)
__global__ void __launch_bounds__(256) k_rt_dtau_halves(KArgs a) {
    const int col = blockIdx.z, i = blockIdx.y;
    const size_t nc = (size_t)a.Y * a.X;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc || a.done[col]) return;
    const int x = (int)(c / a.Y);
    const size_t wgI = nc * a.I, bandI = (size_t)a.X * a.I;
    const double* opl = a.opac_wg_lay + col * wgI;
    const double* opi = a.opac_wg_int + col * wgI;
    const double* scl = a.scat_cross_lay + col * bandI;
    const double* sci = a.scat_cross_int + col * bandI;
    const double* mml = a.mmm_lay + (size_t)col * a.I;
    const double* mmi = a.mmm_int + (size_t)col * a.I;
    const size_t b = x + (size_t)a.X * i, k = c + nc * i;
    if (a.iso) {  // one optical depth per layer (calc_trans_iso, kernels.cu:1082), kept in the "upper" array
        const double* pint = a.p_int + (size_t)col * a.I;
        const double ray = a.scat == 1 ? scl[b] : 0.0;
        a.dtau_u[col * nc * a.L + k] = (pint[i] - pint[i + 1]) / a.colpar[col].g * (opl[k] + ray / mml[i]);
        return;
    }
    double ray_up = 0, ray_low = 0;
    if (a.scat == 1) {
        ray_up = (scl[b] + sci[b + a.X]) / 2.0;
        ray_low = (sci[b] + scl[b]) / 2.0;
    }
    const double kap_up = (opl[k] + opi[k + nc]) / 2.0, kap_low = (opi[k] + opl[k]) / 2.0;
    const double mu_up = (mml[i] + mmi[i + 1]) / 2.0, mu_low = (mmi[i] + mml[i]) / 2.0;
    const size_t wgL = nc * a.L;
    a.dtau_u[col * wgL + k] = a.dcol_u[(size_t)col * a.L + i] * (kap_up + ray_up / mu_up);
    a.dtau_l[col * wgL + k] = a.dcol_l[(size_t)col * a.L + i] * (kap_low + ray_low / mu_low);
}

// ---- `flux calculation method = matrix`: the direct solve as three scans ------------------------------------------------------
// The reference solves, per spectral point, the tridiagonal system of the down and up equations of all half-layers with the
// two boundary conditions (kernels.cu:2109-2284) by Thomas elimination: one thread per point, c' and d' of
// 4 L + 2 rows in two work arrays in HBM.  The same system, eliminated from the surface upwards in its physical variables:
// with U[n] = rho_n D[n] + sigma_n at node n (at the surface rho_0 = albedo, sigma_0 = A F_dir + (1 - A) pi K B_surf),
// half-layer h (nodes h, h + 1) gives
//     rho_{h+1}   = beta + alpha^2 rho_h / (1 - beta rho_h)
//     sigma_{h+1} = alpha sigma_h / (1 - beta rho_h) + s_up + alpha rho_h s_down / (1 - beta rho_h)
//     D[h]        = (alpha D[h+1] + beta sigma_h + s_down) / (1 - beta rho_h),      U[n] = rho_n D[n] + sigma_n
// (Thomas' c' at the even rows is -1 / rho_n: the reference carries the reciprocal form, which loses digits where the
// atmosphere below reflects little -- tests/matrix_referee.py measures it against an extended-precision solve.)  rho -- the
// reflectivity of everything below a node -- follows a Moebius map, i.e. the composition of the 2 x 2 matrices
// [[alpha^2 - beta^2, beta], [-beta, 1]]; sigma and D are affine recurrences.  All three run on the sweeps' tiling inside
// k_rt_flux<ROWS, K, true>: local product of a lane's rows, prefix product over the k lanes of the spectral point (DPP), the
// lane's rows again from its true start value -- on the three coefficient planes the sweeps read, with nothing kept between
// iterations (no flux state, no work arrays): 1.0 GB per launch at config 2 where the reference-shaped solver moved 9.
// 1 / x for the direct solve's thirteen dependent reciprocals per lane: the hardware's estimate and two Newton steps (the last
// bit is not guaranteed -- the solve is held to 1e-9, tests/matrix_referee.py -- at half the instructions of the IEEE division)
#ifndef HX_MATRIX_RCP
#define HX_MATRIX_RCP(x) newton_rcp(x)
#endif
__device__ __forceinline__ double newton_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct Moebius {
    double p11, p12, p21, p22;   // rho -> (p11 rho + p12) / (p21 rho + p22); only the ratios matter
};
// P <- P o Q (Q acts first)
__device__ __forceinline__ void moebius_after(Moebius& P, double q11, double q12, double q21, double q22) {
    const double n11 = fma(P.p11, q11, P.p12 * q21), n12 = fma(P.p11, q12, P.p12 * q22);
    const double n21 = fma(P.p21, q11, P.p22 * q21), n22 = fma(P.p21, q12, P.p22 * q22);
    P.p11 = n11; P.p12 = n12; P.p21 = n21; P.p22 = n22;
}
// the DPP-selected lane's value; a lane without a source inside its row reads the identity's entry (ONE: 1.0, else 0.0)
template <int CTRL, bool ONE>
__device__ __forceinline__ double dpp_or_identity(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = ONE ? __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(v), CTRL, 0xf, 0xf, false)
                       : __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ void moebius_row_neighbour(Moebius& P) {
    moebius_after(P, dpp_or_identity<CTRL, true>(P.p11), dpp_or_identity<CTRL, false>(P.p12), dpp_or_identity<CTRL, false>(P.p21),
                  dpp_or_identity<CTRL, true>(P.p22));
}
// inclusive prefix composition over the lanes of a spectral point (lower lanes = lower half-layers act first)
template <int K>
__device__ __forceinline__ void moebius_scan_up(Moebius& P, int j, int k) {
    if (K) {
        moebius_row_neighbour<0x111>(P);  // row_shr:1
        moebius_row_neighbour<0x112>(P);
        moebius_row_neighbour<0x114>(P);
        moebius_row_neighbour<0x118>(P);
        if (K == 32 || K == 64) {         // row_bcast:15 into rows 1 and 3
            const double q11 = dpp_move<0x142, 0xa>(P.p11), q12 = dpp_move<0x142, 0xa>(P.p12), q21 = dpp_move<0x142, 0xa>(P.p21),
                         q22 = dpp_move<0x142, 0xa>(P.p22);
            if (j & 16) moebius_after(P, q11, q12, q21, q22);
        }
        if (K == 64) {                    // row_bcast:31 into rows 2 and 3
            const double q11 = dpp_move<0x143, 0xc>(P.p11), q12 = dpp_move<0x143, 0xc>(P.p12), q21 = dpp_move<0x143, 0xc>(P.p21),
                         q22 = dpp_move<0x143, 0xc>(P.p22);
            if (j >= 32) moebius_after(P, q11, q12, q21, q22);
        }
    } else {
#define HX_MOEBIUS_STEP(N)                                                                                              \
    if (N < k) {                                                                                                        \
        const double q11 = from_lane_below<N>(P.p11, k), q12 = from_lane_below<N>(P.p12, k),                            \
                     q21 = from_lane_below<N>(P.p21, k), q22 = from_lane_below<N>(P.p22, k);                            \
        if (j >= N) moebius_after(P, q11, q12, q21, q22);                                                               \
    }
        HX_MOEBIUS_STEP(1) HX_MOEBIUS_STEP(2) HX_MOEBIUS_STEP(4) HX_MOEBIUS_STEP(8) HX_MOEBIUS_STEP(16) HX_MOEBIUS_STEP(32)
#undef HX_MOEBIUS_STEP
    }
}

// ---- per iteration: all two-stream sweeps + Gauss quadrature ---------------------------------
// grid (nblk_x, C).  A workgroup owns nxb bins and walks their ny/ypb groups of Gauss points one
// after the other, so the Gauss sum of a bin is completed inside the workgroup (fixed order).
// (k = 32 / ROWS = 7 held to 128 VGPRs for four wavefronts per SIMD through amdgpu_waves_per_eu spills 49 dwords inside
// the sweeps: 0.62 ms against 0.42 ms at its natural 169 registers and 0.40 ms for k = 16 -- not done.)
// K: lanes per spectral point when known at compile time (16, 32, 64: the scans are straight-line DPP code), 0: a.k
#ifndef HX_BEAM_GROUP
#define HX_BEAM_GROUP 4
#endif
// MATRIX: `flux calculation method = matrix` -- instead of the sweeps, the direct solve of the same equations by the two
// scans described above (`the direct solve as three scans`) (no up-flux state is read: a direct solve has none)
#ifndef HX_BIG_ROWS_ONE_WAVE
#define HX_BIG_ROWS_ONE_WAVE 1   // tilings of 15 and more rows (columns of more than 448 layers) run as single-wavefront workgroups: with
                                 // one wavefront per SIMD the part of the register image that does not fit 256 VGPRs lives in AGPRs
                                 // instead of scratch -- 2 000 bins x 1 000 layers 6.73 -> 2.27 ms per iteration, 600 layers 1.29 -> 0.71,
                                 // 512 layers 0.699 -> 0.649; 14 rows (2-3 spilled registers) lose 2 % that way and stay as they were
                                 // (0: 320-thread workgroups, two wavefronts per SIMD, scratch; profiles/r06_deep_columns.txt)
#endif
constexpr bool flux_one_wave(int rows) { return rows >= 15 && HX_BIG_ROWS_ONE_WAVE; }
template <int ROWS, int K = 0, bool MATRIX = false>
__global__ void __launch_bounds__(flux_one_wave(ROWS) ? 64 : 320) k_rt_flux(FluxArgs a) {
    extern __shared__ __align__(16) double smem[];
    const int col = a.reverse ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    if (a.done[col]) return;
    const int bx = a.reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    // the workgroups this launch dispatches last leave their up-flux state in the Infinity Cache for the next launch
    const bool keep_state_cached = (int)(blockIdx.y * gridDim.x + blockIdx.x) >= a.cache_state_from;
    const int NN = a.H + 3, I = a.I;
    double* sB = smem;                               // [nxb][NN]  Planck function at the nodes
    double* acc = sB + (size_t)a.nxb * NN;           // [nxb][2][I] band fluxes being accumulated
    double* stage = acc + (size_t)a.nxb * 2 * I;     // [ypb][nxb][2][I]
    const hx_rt_column cp = a.colpar[col];
    const size_t nc = (size_t)a.Y * a.X;
    const int k = K ? K : a.k;

    for (int t = threadIdx.x; t < a.nxb * NN; t += blockDim.x) {
        const int xl = t / NN, n = t - xl * NN, x = bx * a.nxb + xl;
        sB[t] = x < a.X ? a.Bn[((size_t)col * a.X + x) * NN + n] : 0.0;
    }
    for (int t = threadIdx.x; t < a.nxb * 2 * I; t += blockDim.x) acc[t] = 0.0;
    __syncthreads();

    for (int part = 0; part < a.nparts; part++) {
        const LaneMap m = lane_map(a, bx, part, opaque_tid());
        // coefficient planes and up-flux state -> registers.  The tiles are streamed once per launch: non-temporal
        // loads AND stores together keep them from displacing the node and band arrays the neighbouring kernels and
        // the next workgroups find in the L2 (same-box A/B: k_rt_flux 400 -> 386 us, k_rt_nodes 16 -> 14.4,
        // k_rt_totals_a 16 -> 12.7; either hint alone changes nothing)
        const size_t toff = m.tile * (size_t)ROWS * 64 + m.lane;
        const double* ctile = a.coef + col * a.coef_col + m.tile * (size_t)a.nplane * ROWS * 64 + m.lane;
        double* utile = a.Utile + col * a.flux_col + toff;
        double al[ROWS], be[ROWS], sd[ROWS], su[ROWS], Uo[ROWS], Do[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            al[r] = __builtin_nontemporal_load(ctile + (0 * ROWS + r) * 64);
            be[r] = __builtin_nontemporal_load(ctile + (1 * ROWS + r) * 64);
            sd[r] = __builtin_nontemporal_load(ctile + (2 * ROWS + r) * 64);  // u' for now
            if (!MATRIX) Uo[r] = __builtin_nontemporal_load(utile + r * 64);
        }
        if (a.has_vp) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) su[r] = __builtin_nontemporal_load(ctile + (a.pl_vp * ROWS + r) * 64);  // v' for now
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; r++) su[r] = a.Kconst * ((1.0 - al[r]) - be[r]) - sd[r];
        }
        double U0 = 0.0, boaK = 0.0, Fdir0 = 0.0, albedo = 0.0;
        if (m.valid && m.j == 0) {
            if (!MATRIX) U0 = a.U0[col * nc + m.sp];
            boaK = a.boaK[col * nc + m.sp];
            Fdir0 = a.Fdir0[col * nc + m.sp];
            albedo = a.surf_albedo[(size_t)col * a.X + m.x];
        }
        // the quadrature weight is requested here, with the tiles: asked for after the sweeps it was a dependent load
        // into a saturated memory system, several microseconds per tile with nothing to hide behind
        const double w = m.valid ? 0.5 * a.gauss_w[m.y] : 0.0;
        const double* Bx = sB + (size_t)(m.valid ? m.xl : 0) * NN;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int h = min(m.j * ROWS + r, a.H - 1);
            const double Bb = Bx[h], Bt = a.iso ? Bb : Bx[h + 1], upc = sd[r], vpc = su[r];
            sd[r] = upc * Bb + vpc * Bt;
            su[r] = upc * Bt + vpc * Bb;
        }
        if (a.dir_beam == 1) {
            // The rows of the two beam planes are requested in groups that are in flight together, and only then added: left
            // to the scheduler, the loads came out as ONE register pair loaded and added once per row -- 14 (7 rows) or 26
            // (13 rows) dependent memory round trips per tile; k_rt_flux<7, 64> took 3.70 instead of 3.40 ms per launch at
            // config 5 (whether it happened depended on unrelated code: round 3's build had the 14 in flight together).
            // Up to 8 rows per lane all at once; 13 rows in groups of BEAM_GROUP (the register file is full there).
            constexpr int BEAM_GROUP = ROWS <= 8 ? ROWS : HX_BEAM_GROUP;
#pragma unroll
            for (int r0 = 0; r0 < ROWS; r0 += BEAM_GROUP) {
                double bd[BEAM_GROUP], bu[BEAM_GROUP];
#pragma unroll
                for (int u = 0; u < BEAM_GROUP; u++)
                    if (r0 + u < ROWS) {
                        bd[u] = __builtin_nontemporal_load(ctile + (a.pl_dd * ROWS + r0 + u) * 64);
                        bu[u] = __builtin_nontemporal_load(ctile + ((a.pl_dd + 1) * ROWS + r0 + u) * 64);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < BEAM_GROUP; u++)
                    if (r0 + u < ROWS) {
                        sd[r0 + u] += bd[u];
                        su[r0 + u] += bu[u];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const double rs = cp.R_star / cp.a;
        const double D_toa = (1.0 - a.dir_beam) * cp.f_factor * (rs * rs) * HX_PI * Bx[a.H + 1];
        const double B_surf = Bx[a.H + 2];

        // rows (r even, r odd) of this lane whose up-flux the reference makes positive when it is tiny: the odd nodes
        const bool odd0 = (m.j * ROWS) & 1;
        const double thr_even = (a.iso || odd0) ? 1e-100 : 0.0, thr_odd = (a.iso || !odd0) ? 1e-100 : 0.0;
        if constexpr (MATRIX) {
            // which values the reference makes positive: with scattering (Thomas, non-isothermal) every x < 1e-100 of the back-
            // substitution becomes |x| (kernels.cu:2268), with isothermal layers none (:1967); in the pure-absorption sweeps tiny
            // values do (:2329, :2351, :2418, and -- isothermal -- :1990, :2018), the up-flux at the layer centres excepted (:2394)
            const bool scatters = m.valid ? a.trigger[col * nc + m.sp] != 0 : false;
            const bool flip_negative = scatters && !a.iso;
            const double tiny_d = scatters ? 0.0 : 1e-100;
            const double tiny_u_even = scatters ? 0.0 : ((a.iso || odd0) ? 1e-100 : 0.0), tiny_u_odd = scatters ? 0.0 : ((a.iso || !odd0) ? 1e-100 : 0.0);
            auto patch = [&](double v, double tiny) { return flip_negative ? (v < 1e-100 ? fabs(v) : v) : tiny_abs_below(v, tiny); };
            // ---------------- rho: surface -> TOA ----------------
            // (the tiles' padding rows and lanes hold alpha = 1, beta = 0: the identity)
            Moebius P = {1.0, 0.0, 0.0, 1.0};
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const double ga = fma(al[r], al[r], -(be[r] * be[r]));
                const double n11 = fma(ga, P.p11, be[r] * P.p21), n12 = fma(ga, P.p12, be[r] * P.p22);
                P.p21 = fma(-be[r], P.p11, P.p21);
                P.p22 = fma(-be[r], P.p12, P.p22);
                P.p11 = n11;
                P.p12 = n12;
            }
            {   // entries near one before the lanes are combined: a stack of thick, nearly conservative scatterers shrinks the
                // product by 4e-5 per row, and only the ratios matter
                const double sc = 1.0 / fmax(fmax(fabs(P.p11), fabs(P.p12)), fmax(fabs(P.p21), fabs(P.p22)));
                P.p11 *= sc; P.p12 *= sc; P.p21 *= sc; P.p22 *= sc;
            }
            moebius_scan_up<K>(P, m.j, k);
            double e11 = K ? below_fixed<K>(P.p11) : from_lane_below<1>(P.p11, k), e12 = K ? below_fixed<K>(P.p12) : from_lane_below<1>(P.p12, k);
            double e21 = K ? below_fixed<K>(P.p21) : from_lane_below<1>(P.p21, k), e22 = K ? below_fixed<K>(P.p22) : from_lane_below<1>(P.p22, k);
            const double alb = K ? group_first_lane<K>(albedo, m.lane) : __shfl(albedo, 0, k);
            double rho = fma(e11, alb, e12) / fma(e21, alb, e22);   // at this lane's lowest node
            if (m.j == 0) rho = alb;
            // per row: a = alpha / (1 - beta rho_b) -- the factor of BOTH affine recurrences --, the constant of the sigma
            // recurrence s_up + a rho_b s_down, what the D recurrence needs: beta / (1 - beta rho_b), s_down / (1 - beta rho_b),
            // and rho at the row's top node (kept where the sweeps keep their up-flux)
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const double inv = HX_MATRIX_RCP(1.0 - be[r] * rho), aa = al[r] * inv;
                su[r] = fma(aa * rho, sd[r], su[r]);
                rho = fma(aa * al[r], rho, be[r]);
                Uo[r] = rho;
                be[r] *= inv;
                sd[r] *= inv;
                al[r] = aa;
            }
            // ---------------- sigma: surface -> TOA ----------------
            double sigma0 = 0.0;
            if (m.j == 0) sigma0 = albedo * Fdir0 + (1.0 - albedo) * HX_PI * boaK * B_surf;
            sigma0 = K ? group_first_lane<K>(sigma0, m.lane) : __shfl(sigma0, 0, k);
            {
                double A = 1.0, Bc = 0.0;
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    Bc = fma(al[r], Bc, su[r]);
                    A *= al[r];
                }
                if (K) {
                    scan_up_fixed<K>(A, Bc, m.j);
                } else if (k == 32) {
                    scan32_up(A, Bc, m.j);
                } else {
                    scan_step_up<1>(A, Bc, m.j, k);
                    scan_step_up<2>(A, Bc, m.j, k);
                    scan_step_up<4>(A, Bc, m.j, k);
                    scan_step_up<8>(A, Bc, m.j, k);
                    scan_step_up<16>(A, Bc, m.j, k);
                    scan_step_up<32>(A, Bc, m.j, k);
                }
                double sg = K ? below_fixed<K>(fma(A, sigma0, Bc)) : from_lane_below<1>(fma(A, sigma0, Bc), k);
                if (m.j == 0) sg = sigma0;
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    sd[r] = fma(be[r], sg, sd[r]);      // constant of the D recurrence: (beta sigma_bottom + s_down) / (1 - beta rho_b)
                    sg = fma(al[r], sg, su[r]);
                    su[r] = sg;                         // sigma at the row's top node
                }
            }
            // ---------------- D: TOA -> surface, and U = rho D + sigma ----------------
            {
                double A = 1.0, Bc = 0.0;
#pragma unroll
                for (int r = ROWS - 1; r >= 0; r--) {
                    Bc = fma(al[r], Bc, sd[r]);
                    A *= al[r];
                }
                if (K) {
                    scan_down_fixed<K>(A, Bc, m.j, m.lane);
                } else if (k == 32) {
                    scan32_down(A, Bc, m.j, m.lane);
                } else {
                    scan_step_down<1>(A, Bc, m.j, k);
                    scan_step_down<2>(A, Bc, m.j, k);
                    scan_step_down<4>(A, Bc, m.j, k);
                    scan_step_down<8>(A, Bc, m.j, k);
                    scan_step_down<16>(A, Bc, m.j, k);
                    scan_step_down<32>(A, Bc, m.j, k);
                }
                double D = K ? above_fixed<K>(fma(A, D_toa, Bc)) : from_lane_above<1>(fma(A, D_toa, Bc), k);
                if (m.j == k - 1) D = D_toa;
#pragma unroll
                for (int r = ROWS - 1; r >= 0; r--) {
                    Uo[r] = patch(fma(Uo[r], D, su[r]), (r & 1) ? tiny_u_odd : tiny_u_even);   // U at the top node, D there still in hand
                    D = patch(fma(al[r], D, sd[r]), tiny_d);
                    Do[r] = D;
                }
            }
            if (m.j == 0) U0 = patch(fma(albedo, Do[0], sigma0), 0.0);
        } else
        for (int sweep = 0; sweep < a.nsweep; sweep++) {
            // ---------------- down: TOA -> BOA ----------------
            {
                double Ubelow = K ? below_fixed<K>(Uo[ROWS - 1]) : from_lane_below<1>(Uo[ROWS - 1], k);  // U at the bottom node of this chunk
                if (m.j == 0) Ubelow = U0;
                double A = 1.0, Bc = 0.0;
#pragma unroll
                for (int r = ROWS - 1; r >= 0; r--) {
                    const double Uh = r > 0 ? Uo[r - 1] : Ubelow;
                    const double t = fma(be[r], Uh, sd[r]);
                    Bc = fma(al[r], Bc, t);
                    A *= al[r];
                }
                // inclusive suffix composition over the k lanes of this spectral point
                if (K) {
                    scan_down_fixed<K>(A, Bc, m.j, m.lane);
                } else if (k == 32) {
                    scan32_down(A, Bc, m.j, m.lane);
                } else {
                    scan_step_down<1>(A, Bc, m.j, k);
                    scan_step_down<2>(A, Bc, m.j, k);
                    scan_step_down<4>(A, Bc, m.j, k);
                    scan_step_down<8>(A, Bc, m.j, k);
                    scan_step_down<16>(A, Bc, m.j, k);
                    scan_step_down<32>(A, Bc, m.j, k);
                }
                double Din = K ? above_fixed<K>(fma(A, D_toa, Bc)) : from_lane_above<1>(fma(A, D_toa, Bc), k);
                if (m.j == k - 1) Din = D_toa;
                double D = Din;
#pragma unroll
                for (int r = ROWS - 1; r >= 0; r--) {
                    const double Uh = r > 0 ? Uo[r - 1] : Ubelow;
                    D = tiny_abs(fma(al[r], D, fma(be[r], Uh, sd[r])));
                    Do[r] = D;
                }
            }
            // ---------------- BOA boundary ----------------
            if (m.j == 0) U0 = albedo * (Fdir0 + Do[0]) + (1.0 - albedo) * HX_PI * boaK * B_surf;
            const double Ubc = K ? group_first_lane<K>(U0, m.lane) : __shfl(U0, 0, k);
            // ---------------- up: BOA -> TOA ----------------
            {
                double Dabove = K ? above_fixed<K>(Do[0]) : from_lane_above<1>(Do[0], k);  // D at the top node of this chunk
                if (m.j == k - 1) Dabove = D_toa;
                double A = 1.0, Bc = 0.0;
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const double Dh = r < ROWS - 1 ? Do[r + 1] : Dabove;
                    const double t = fma(be[r], Dh, su[r]);
                    Bc = fma(al[r], Bc, t);
                    A *= al[r];
                }
                if (K) {
                    scan_up_fixed<K>(A, Bc, m.j);
                } else if (k == 32) {
                    scan32_up(A, Bc, m.j);
                } else {
                    scan_step_up<1>(A, Bc, m.j, k);
                    scan_step_up<2>(A, Bc, m.j, k);
                    scan_step_up<4>(A, Bc, m.j, k);
                    scan_step_up<8>(A, Bc, m.j, k);
                    scan_step_up<16>(A, Bc, m.j, k);
                    scan_step_up<32>(A, Bc, m.j, k);
                }
                double Uin = K ? below_fixed<K>(fma(A, Ubc, Bc)) : from_lane_below<1>(fma(A, Ubc, Bc), k);
                if (m.j == 0) Uin = Ubc;
                double U = Uin;
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const double Dh = r < ROWS - 1 ? Do[r + 1] : Dabove;
                    U = fma(al[r], U, fma(be[r], Dh, su[r]));
                    // interface nodes only (reference quirk, kernels.cu:1763; isothermal layers: every node, :1509)
                    if (K) U = tiny_abs_below(U, (r & 1) ? thr_odd : thr_even);
                    else if (a.iso || ((m.j * ROWS + r) & 1)) U = tiny_abs(U);
                    Uo[r] = U;
                }
            }
        }

        // Gauss quadrature of the interface fluxes: stage[yl][xl][dir][i], summed over yl in order.
        // The lane map is derived afresh (e for "epilogue"): carried across the sweeps it lived in scratch
        const LaneMap e = lane_map(a, bx, part, opaque_tid());
#ifdef HX_PROFILING  // HELIOS_RT_DEBUG_SKIP: bit 0 no quadrature, bit 1 no state stores -- not in the shipped library
        const int debug_skip = a.debug_skip;
#else
        constexpr int debug_skip = 0;
#endif
        auto store_state = [&]() {
            const size_t eoff = e.tile * (size_t)ROWS * 64 + e.lane;
            if (!(debug_skip & 2) && (!MATRIX || a.keep_up)) {   // (a direct solve has no state: its spectral fluxes are
                double* ut = a.Utile + col * a.flux_col + eoff;   //  written where somebody asks for them, hx_rt_get)
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    // (write-through store that leaves the line in the Infinity Cache: agent scope = `sc1`; see launch_flux)
                    if (keep_state_cached) __hip_atomic_store(ut + r * 64, Uo[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else __builtin_nontemporal_store(Uo[r], ut + r * 64);
                }
            }
            if (a.keep_down) {
                double* dtile = a.Dtile + col * a.flux_col + eoff;
#pragma unroll
                for (int r = 0; r < ROWS; r++) __builtin_nontemporal_store(Do[r], dtile + r * 64);
            }
            if (e.valid && e.j == 0) a.U0[col * nc + e.sp] = U0;
        };
        if (!(debug_skip & 1)) {
            if (e.valid) {
                // Row r of this lane is node h = h0 + r.  Staggered grid: an even node gives D at interface h/2, an odd
                // one U at interface (h+1)/2 -- with the lane's parity p folded into two base pointers the rows use
                // compile-time offsets (26 separately computed LDS addresses did not fit the register file: they were
                // reloaded from scratch, one memory round trip per row and tile)
                const int h0 = e.j * ROWS, nrow = a.H - h0;
                double* st = stage + ((size_t)e.yl * a.nxb + e.xl) * 2 * I;
                if (a.iso) {                                            // every node is an interface
                    double *pd = st + h0, *pu = st + I + h0 + 1;
#pragma unroll
                    for (int r = 0; r < ROWS; r++)
                        if (r < nrow) {
                            pd[r] = w * Do[r];
                            pu[r] = w * Uo[r];
                        }
                } else {
                    const bool p = h0 & 1;
                    const int q = (h0 + (p ? 1 : 0)) >> 1;
                    double* pe = st + q + (p ? I : 0);                  // rows 0, 2, ...: D (p = 0) or U (p = 1)
                    double* po = st + q + (p ? -1 : I);                 // rows 1, 3, ...: U (p = 0) or D (p = 1)
#pragma unroll
                    for (int r = 0; r < ROWS; r++)
                        if (r < nrow) {
                            if ((r & 1) == 0) pe[r >> 1] = w * (p ? Uo[r] : Do[r]);
                            else po[(r + 1) >> 1] = w * (p ? Do[r] : Uo[r]);
                        }
                }
                if (e.j == 0) st[I + 0] = w * U0;
                if (h0 <= a.H - 1 && a.H - 1 < h0 + ROWS) st[a.L] = w * D_toa;
            }
            __syncthreads();
            for (int t = threadIdx.x; t < a.nxb * 2 * I; t += blockDim.x) {
                const int xl = t / (2 * I), rest = t - xl * 2 * I;
                double s = acc[t];
                for (int yl = 0; yl < a.ypb; yl++) s += stage[((size_t)yl * a.nxb + xl) * 2 * I + rest];
                acc[t] = s;
            }
            __syncthreads();
        }
        store_state();
    }
    // band fluxes of this workgroup's bins, internal layout [x][i]
    for (int t = threadIdx.x; t < a.nxb * 2 * I; t += blockDim.x) {
        const int xl = t / (2 * I), rest = t - xl * 2 * I, x = bx * a.nxb + xl;
        if (x >= a.X) continue;
        const int dir = rest / I, i = rest - dir * I;
        (dir == 0 ? a.F_down_band_n : a.F_up_band_n)[((size_t)col * a.X + x) * I + i] = acc[t];
    }
}

// ---- `flux calculation method = matrix`: glue between the node arrays of the loop and the per-stage solver -------------
// Planck values of the nodes, Bn[x][H+3], in the reference's layouts: planckband_lay[i + x (L+2)] (layers, then the stellar
// row and the surface) and planckband_int[i + x I].  grid (chunks, C).
__global__ void __launch_bounds__(256) k_rt_matrix_planck(const double* __restrict__ Bn, double* __restrict__ pb_lay,
                                                          double* __restrict__ pb_int, int X, int L, int H, int iso,
                                                          const int* __restrict__ done) {
    const int col = blockIdx.y;
    if (done[col]) return;
    const int NN = H + 3, I = L + 1, per = L + 2 + I;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)X * per) return;
    const int x = (int)(t / per), s = (int)(t - (long long)x * per);
    const double* B = Bn + ((size_t)col * X + x) * NN;
    if (s < L + 2) {
        const int n = s < L ? (iso ? s : 2 * s + 1) : H + 1 + (s - L);
        pb_lay[((size_t)col * X + x) * (L + 2) + s] = B[n];
    } else {  // isothermal layers: the reference computes no interface values (computation.py:315-329)
        const int i = s - (L + 2);
        pb_int[((size_t)col * X + x) * I + i] = iso ? 0.0 : B[2 * i];
    }
}

// Gauss quadrature of the solver's interface fluxes (kernels.cu:2474-2476) into the band arrays of the loop, [x][i].
// grid (ceil(X / QUAD_BINS), I, C), 256 threads: the workgroup reads its bins' ny*QUAD_BINS spectral points of one
// interface as one contiguous run, weights them into LDS, and one thread per bin adds its Gauss points in order (a thread
// per bin reading its own ny values took 0.65 ms at 10 000 x 101 x 20: one 64-byte sector per double)
constexpr int QUAD_BINS = 32;
__global__ void __launch_bounds__(256) k_rt_matrix_bands(const double* __restrict__ F_down_wg, const double* __restrict__ F_up_wg,
                                                         double* __restrict__ F_down_band_n, double* __restrict__ F_up_band_n,
                                                         const double* __restrict__ gauss_w, int X, int Y, int I,
                                                         const int* __restrict__ done) {
    extern __shared__ __align__(16) double smem[];
    const int col = blockIdx.z, i = blockIdx.y, x0 = blockIdx.x * QUAD_BINS;
    if (done[col]) return;
    const int nb = min(QUAD_BINS, X - x0), pitch = Y + 1;
    double* su = smem;
    double* sd = smem + QUAD_BINS * pitch;
    const size_t nc = (size_t)X * Y, base = (size_t)col * nc * I + nc * i + (size_t)Y * x0;
    for (int t = threadIdx.x; t < nb * Y; t += blockDim.x) {
        const int xl = t / Y, y = t - xl * Y;
        const double w = 0.5 * gauss_w[y];
        su[xl * pitch + y] = w * F_up_wg[base + t];
        sd[xl * pitch + y] = w * F_down_wg[base + t];
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
        const int xl = threadIdx.x;
        double d = 0.0, u = 0.0;
        for (int y = 0; y < Y; y++) {
            u += su[xl * pitch + y];
            d += sd[xl * pitch + y];
        }
        const size_t b = ((size_t)col * X + x0 + xl) * I + i;
        F_up_band_n[b] = u;
        F_down_band_n[b] = d;
    }
}

// ---- per iteration: wavelength totals, level 1 ---------------------------------------------------
// grid (nchunk, C), 256 threads.  Thread t owns the (dir, i) slots t, t+256, ... (< 2I) and walks the
// bins of its chunk; consecutive threads read consecutive addresses of the [x][i] band arrays.
__global__ void __launch_bounds__(256) k_rt_totals_a(KArgs a) {
    const int col = blockIdx.y, chunk = blockIdx.x;
    if (a.done[col]) return;
    const int I = a.I;
    const int per = (a.X + a.nchunk - 1) / a.nchunk;
    const int x0 = chunk * per, x1 = min(a.X, x0 + per);
    const double* __restrict__ fdir = a.F_dir_band_n + (size_t)col * a.X * I;
    const double* __restrict__ dl = a.deltawave;
    for (int t = threadIdx.x; t < 2 * I; t += blockDim.x) {
        const int dir = t / I, i = t - dir * I;
        const double* __restrict__ band = (dir == 0 ? a.F_down_band_n : a.F_up_band_n) + (size_t)col * a.X * I;
        // loads issued eight bins at a time (the loop is latency-bound otherwise); summed in bin order
        double acc = 0.0;
        for (int xb = x0; xb < x1; xb += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {  // branch-free so that all 16 loads are in flight together
                const int x = min(xb + u, x1 - 1);
                const double f = fdir[(size_t)x * I + i], b = band[(size_t)x * I + i];
                v[u] = dir == 0 ? f + b : b;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (xb + u < x1) acc += v[u] * dl[xb + u];
        }
        a.tot_part[(((size_t)col * a.nchunk + chunk) * 2) * I + t] = acc;
    }
}

struct TotalsBArgs {
    KArgs a;
    RadTempArgs rt;  // pointers of column 0; strides applied below
    int step_temperature;
    int* done_w;
    int* iters_done;
    const int* iter_dev;  // the iteration index on the device (see KArgs); nullptr: rt.itervalue
    size_t sL, sL1, sI;
};

__global__ void __launch_bounds__(1024) k_rt_totals_b(TotalsBArgs q) {
    const KArgs& a = q.a;
    const int col = blockIdx.x;
    if (a.done[col]) return;
    const int I = a.I, L = a.L;
    double* up = a.F_up_tot + (size_t)col * I;
    double* down = a.F_down_tot + (size_t)col * I;
    double* net = a.F_net + (size_t)col * I;
    // chunk partials -> totals: 4 segments of chunks per (dir, i) slot, combined in a fixed order
    __shared__ double seg[4][256];
    for (int t0 = 0; t0 < 2 * I; t0 += 256) {
        const int t = t0 + (threadIdx.x & 255), sgm = threadIdx.x >> 8;
        const int cper = (a.nchunk + 3) / 4, c0 = sgm * cper, c1 = min(a.nchunk, c0 + cper);
        double s = 0.0;
        if (t < 2 * I) {
            const double* __restrict__ part = a.tot_part + ((size_t)col * a.nchunk * 2) * I + t;
            for (int cb = c0; cb < c1; cb += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = part[(size_t)min(cb + u, c1 - 1) * 2 * I];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (cb + u < c1) s += v[u];
            }
        }
        seg[sgm][threadIdx.x & 255] = s;
        __syncthreads();
        if (sgm == 0 && t < 2 * I) {
            const double tot = ((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x];
            if (t < I) down[t] = tot; else up[t - I] = tot;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < I; i += blockDim.x) net[i] = up[i] - down[i];
    __syncthreads();
    if (!q.step_temperature) return;
    const hx_rt_column cp = a.colpar[col];
    const int itervalue = q.iter_dev != nullptr ? q.iter_dev[1] : q.rt.itervalue;
    if (itervalue < cp.foreplay) return;
    RadTempArgs r = q.rt;
    r.itervalue = itervalue;
    r.F_down_tot = down;
    r.F_net = net;
    r.F_net_diff += (size_t)col * L;
    r.tlay += (size_t)col * (L + 1);
    r.play += (size_t)col * L;
    r.pint += (size_t)col * I;
    r.abrt += (size_t)col * (L + 1);
    r.T_store += (size_t)col * (L + 1);
    r.deltat_prefactor += (size_t)col * (L + 1);
    r.F_add_heat_lay += (size_t)col * L;
    r.F_add_heat_sum += (size_t)col * L;
    r.F_smooth += (size_t)col * L;
    r.F_smooth_sum += (size_t)col * L;
    r.c_p_lay += (size_t)col * L;
    r.meanmolmass_lay = a.mmm_lay + (size_t)col * I;
    r.conv_count += col;
    r.foreplay = cp.foreplay;
    r.g = cp.g;
    r.physical_tstep = cp.physical_tstep;
    r.local_limit = cp.rad_convergence_limit;
    r.adapt_interval = cp.adapt_interval;
    r.F_intern = cp.F_intern;
    r.no_atmo = cp.no_atmo;
    rad_temp_step(r, threadIdx.x, blockDim.x);
    __syncthreads();
    if (threadIdx.x == 0 && *r.conv_count == L + 1) {
        q.done_w[col] = 1;  // the reference leaves radiation_loop once every flag is set
        q.iters_done[col] = r.itervalue + 1;
    }
}

// ---- convection loop on the device (reference computation.py:992-1174) ---------------------------------------
struct ConvKArgs {
    int L, C, itervalue;
    const hx_rt_column* colpar;
    double* T_lay;                          // [C][L+1]
    const double *p_lay, *p_int;            // [C][L], [C][L+1]
    const double *kappa_lay, *kappa_int;    // [C][L], [C][L+1]
    const double *c_p, *mmm_lay;            // [C][L], [C][L+1]
    const double *F_add_heat_sum, *F_smooth_sum;
    const double *F_down_tot, *F_up_tot, *F_net;
    int *conv_unstable, *conv_layer, *marked_red;  // [C][L+1]
    const double* dampara;                  // [C], <= 0: automatic
    const int* done;
};

__device__ __forceinline__ ConvColumn conv_column(const ConvKArgs& a, int col) {
    const size_t L = a.L, I = a.L + 1;
    const hx_rt_column cp = a.colpar[col];
    ConvColumn c;
    c.T = a.T_lay + col * I;
    c.p_lay = a.p_lay + col * L;
    c.p_int = a.p_int + col * I;
    c.kappa_lay = a.kappa_lay + col * L;
    c.kappa_int = a.kappa_int + col * I;
    c.c_p = a.c_p + col * L;
    c.mmm = a.mmm_lay + col * I;
    c.F_add_heat_sum = a.F_add_heat_sum + col * L;
    c.F_smooth_sum = a.F_smooth_sum + col * L;
    c.F_down_tot = a.F_down_tot + col * I;
    c.F_up_tot = a.F_up_tot + col * I;
    c.F_net = a.F_net + col * I;
    c.conv_unstable = a.conv_unstable + col * I;
    c.conv_layer = a.conv_layer + col * I;
    c.marked_red = a.marked_red + col * I;
    c.L = a.L;
    c.itervalue = a.itervalue;
    c.F_intern = cp.F_intern;
    c.T_star = cp.T_star;
    c.dampara = a.dampara[col];
    c.rad_convergence_limit = cp.rad_convergence_limit;
    return c;
}

// step E of the loop: check -> mark -> correct until stable, then once more with stitching and flux fudging.
// grid C, 256 threads: all of them tabulate the pressure-ratio powers, thread 0 walks the layers.
__global__ void __launch_bounds__(256) k_rt_conv_adjust(ConvKArgs a) {
    extern __shared__ __align__(16) double conv_smem[];
    __shared__ ConvTables t;
    const int col = blockIdx.x;
    if (a.done[col]) return;
    __shared__ ConvShared sh;
    ConvColumn c = conv_column(a, col), g;
    conv_stage_in(c, g, t, conv_smem, threadIdx.x, blockDim.x);
    convective_adjustment_wg(c, t, sh, threadIdx.x, blockDim.x);
    conv_stage_out(c, g, threadIdx.x, blockDim.x);
}

struct TotalsCArgs {
    KArgs a;
    ConvKArgs cv;
    ConvTempArgs ct;   // pointers of column 0
    int* done_w;
    int* iters_done;
    int physical_tstep_on;
};

// steps H (totals), I, J, K: wavelength totals, mark the convective layers, test the radiative layers for local
// equilibrium, and -- unless the column is done -- the radiative temperature step of the convection loop
__global__ void __launch_bounds__(1024) k_rt_totals_c(TotalsCArgs q) {
    extern __shared__ __align__(16) double conv_smem[];
    __shared__ ConvTables t;
    __shared__ double seg[4][256];
    __shared__ int s_go;
    const KArgs& a = q.a;
    const int col = blockIdx.x;
    if (a.done[col]) return;
    const int I = a.I, L = a.L;
    double* up = a.F_up_tot + (size_t)col * I;
    double* down = a.F_down_tot + (size_t)col * I;
    double* net = a.F_net + (size_t)col * I;
    for (int t0 = 0; t0 < 2 * I; t0 += 256) {
        const int ts = t0 + (threadIdx.x & 255), sgm = threadIdx.x >> 8;
        const int cper = (a.nchunk + 3) / 4, c0 = sgm * cper, c1 = min(a.nchunk, c0 + cper);
        double s = 0.0;
        if (ts < 2 * I) {
            const double* __restrict__ part = a.tot_part + ((size_t)col * a.nchunk * 2) * I + ts;
            for (int cb = c0; cb < c1; cb += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = part[(size_t)min(cb + u, c1 - 1) * 2 * I];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (cb + u < c1) s += v[u];
            }
        }
        seg[sgm][threadIdx.x & 255] = s;
        __syncthreads();
        if (sgm == 0 && ts < 2 * I) {
            const double tot = ((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x];
            if (ts < I) down[ts] = tot; else up[ts - I] = tot;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < I; i += blockDim.x) net[i] = up[i] - down[i];
    __syncthreads();
    ConvColumn c = conv_column(q.cv, col), g;
    conv_stage_in(c, g, t, conv_smem, threadIdx.x, blockDim.x);
    __shared__ ConvShared sh;
    __shared__ int s_convective;
    conv_find_lim(c, sh, threadIdx.x, blockDim.x);
    conv_mark_layers_wg(c, t, sh, 1, threadIdx.x, blockDim.x);
    int go = 0;
    if (!q.physical_tstep_on) {  // with a physical time step the reference leaves the loop here
        const int eq = conv_radiative_eq_wg(c, sh, &s_convective, threadIdx.x, blockDim.x);
        go = (!eq) || (c.itervalue < 400) || (s_convective == 0);
    }
    if (threadIdx.x == 0) {
        if (!go) {
            q.done_w[col] = 1;
            q.iters_done[col] = c.itervalue;   // no temperature step, no increment: the loop exits here
        }
        s_go = go;
    }
    __syncthreads();
    conv_stage_out(c, g, threadIdx.x, blockDim.x);   // (T unchanged here) flags for the temperature step and the host
    __syncthreads();
    if (!s_go) return;
    const hx_rt_column cp = a.colpar[col];
    ConvTempArgs r = q.ct;
    r.F_net = net;
    r.F_net_diff += (size_t)col * L;
    r.tlay += (size_t)col * (L + 1);
    r.play += (size_t)col * L;
    r.pint += (size_t)col * I;
    r.T_store += (size_t)col * (L + 1);
    r.deltat_prefactor += (size_t)col * (L + 1);
    r.marked_red = g.marked_red;
    r.F_add_heat_lay += (size_t)col * L;
    r.F_smooth += (size_t)col * L;
    r.F_smooth_sum += (size_t)col * L;
    r.adapt_interval = cp.adapt_interval;
    r.F_intern = cp.F_intern;
    conv_temp_step(r, threadIdx.x, blockDim.x);
}

// interface temperatures only (the node kernel does the same as part of its work)
__global__ void k_rt_tint(const double* __restrict__ T_lay, double* __restrict__ T_int, int L, const int* done) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > L || done[col]) return;
    T_int[(size_t)col * (L + 1) + i] = interface_T(T_lay + (size_t)col * (L + 1), i, L);
}

// additional heating flux of the layers and its running sum from the TOP of the list down, i.e. index 0 upwards
// (host_functions.py:701-711), one thread per column
__global__ void k_rt_heating(const double* __restrict__ dens, const double* __restrict__ dz,
                             double* __restrict__ F_lay, double* __restrict__ F_sum, int L, const int* done) {
    const int col = blockIdx.x;
    if (threadIdx.x != 0 || done[col]) return;
    dens += (size_t)col * L; dz += (size_t)col * L; F_lay += (size_t)col * L; F_sum += (size_t)col * L;
    double run = 0.0;
    for (int i = 0; i < L; i++) {
        const double f = dens[i] * dz[i];
        F_lay[i] = f;
        run = i == 0 ? f : run + f;
        F_sum[i] = run;
    }
}

// altitude of the layer centres from the layer thicknesses (host_functions.py:673-698), one thread
__global__ void k_rt_height(const double* __restrict__ p_lay, const double* __restrict__ dz,
                            double* __restrict__ z, int L, int gas, size_t stride, const int* __restrict__ done) {
    const int col = blockIdx.x;
    p_lay += col * stride; dz += col * stride; z += col * stride;
    if (threadIdx.x != 0 || done[col]) return;
    if (gas) {
        int i0 = 0;
        for (int i = 0; i < L; i++) if (p_lay[i] >= 1e7) i0 = i;
        z[i0] = 0.0;
        for (int i = i0 + 1; i < L; i++) z[i] = z[i - 1] + 0.5 * dz[i - 1] + 0.5 * dz[i];
        for (int i = i0 - 1; i >= 0; i--) z[i] = z[i + 1] - 0.5 * dz[i + 1] - 0.5 * dz[i];
    } else {
        z[0] = 0.5 * dz[0];
        for (int i = 1; i < L; i++) z[i] = z[i - 1] + 0.5 * dz[i - 1] + 0.5 * dz[i];
    }
}

// mean molecular mass from the species' mixing ratios (host_functions.py:927-959)
__global__ void k_rt_meanmolmass(const double* __restrict__ vmr, const double* __restrict__ weight,
                                 const int* __restrict__ in_mu, double* __restrict__ out, int nspecies,
                                 int nlev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlev) return;
    double num = 0.0, tot = 0.0;
    for (int s = 0; s < nspecies; s++)
        if (in_mu[s]) {
            num += vmr[(size_t)s * nlev + i] * weight[s];
            tot += vmr[(size_t)s * nlev + i];
        }
    out[i] = num / tot * HX_AMU;
}

// F_dir_band_n[x][i] = sum_y w_y/2 F_dir_wg[y + Y x + Y X i]; grid (chunks of x, I, C)
__global__ void __launch_bounds__(256) k_rt_fdir_band(const double* __restrict__ F_dir_wg,
                                                      double* __restrict__ out,
                                                      const double* __restrict__ gauss_w, int X, int Y,
                                                      int I, const int* __restrict__ done) {
    // tile of 32 bins x 32 levels: a lane sums the Gauss points of its bin (160 contiguous bytes, neighbouring lanes
    // neighbouring bins), the sums go through LDS so that the bin-major output rows [x][i] are written along i.  (One
    // thread per (bin, level) with the level as the grid's y index wrote 8 bytes per 1.6 KB: 2.1 ms per refresh at
    // 30 000 x 200, a quarter of what the coefficient kernel takes.)  Same order of additions as before.
    __shared__ double tile[32][33];
    const int col = blockIdx.z;
    if (done[col]) return;
    const int x0 = blockIdx.x * 32, i0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    F_dir_wg += (size_t)col * Y * X * I;
    out += (size_t)col * X * I;
    for (int lv = ty; lv < 32; lv += 8) {
        const int x = x0 + tx, i = i0 + lv;
        double s = 0.0;
        if (x < X && i < I) {
            const double* f = F_dir_wg + (size_t)Y * x + (size_t)Y * X * i;
            if (Y == 20) {  // all twenty values requested before the first is used (with a run-time trip count every load
                            // waited for the one before it: 1.9 ms at 30 000 x 200 for 0.96 GB)
                double v[20];
#pragma unroll
                for (int y = 0; y < 20; y += 2) {
                    const double2 p = *(const double2*)(f + y);
                    v[y] = p.x;
                    v[y + 1] = p.y;
                }
#pragma unroll
                for (int y = 0; y < 20; y++) s += 0.5 * gauss_w[y] * v[y];
            } else {
                for (int y = 0; y < Y; y++) s += 0.5 * gauss_w[y] * f[y];
            }
        }
        tile[lv][tx] = s;
    }
    __syncthreads();
    for (int xx = ty; xx < 32; xx += 8) {
        const int x = x0 + xx, i = i0 + tx;
        if (x < X && i < I) out[(size_t)x * I + i] = tile[tx][xx];
    }
}

// fractional (T, log10 P) table indices of every layer and interface of every column
__global__ void k_rt_tp_index(KArgs a, TPIndex* tp_lay, TPIndex* tp_int) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.I || a.done[col]) return;
    const double* T = a.T_lay + (size_t)col * (a.L + 1);
    tp_int[(size_t)col * a.I + i] = locate_tp(interface_T(T, i, a.L), a.p_int[(size_t)col * a.I + i], a.ktemp,
                                              a.ntemp, a.kpress, a.npress, true, false);
    if (i < a.L)
        tp_lay[(size_t)col * a.I + i] = locate_tp(T[i], a.p_lay[(size_t)col * a.L + i], a.ktemp, a.ntemp, a.kpress,
                                                  a.npress, true, false);
}

// Rayleigh cross-sections of the premixed table at every level (the [x + X*level] half of opac_interpol)
__global__ void __launch_bounds__(256) k_rt_scat_interp(KArgs a, double* scat_lay, double* scat_int) {
    const int col = blockIdx.z, lev = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.X || a.done[col]) return;
    const size_t cp = a.X, ct = (size_t)a.X * a.npress, bandI = (size_t)a.X * a.I;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 0 && lev >= a.L) continue;
        const TPIndex k = (pass == 0 ? a.tp_lay : a.tp_int)[(size_t)col * a.I + lev];
        const double* t0 = a.crosstable + x + ct * k.tdown;
        const double* t1 = a.crosstable + x + ct * k.tup;
        (pass == 0 ? scat_lay : scat_int)[col * bandI + x + (size_t)a.X * lev] =
            blend_tp(t0[cp * k.pdown], t0[cp * k.pup], t1[cp * k.pdown], t1[cp * k.pup], k, false);
    }
}

// ---- per refresh, all columns at once (a column whose loop has ended is skipped on the device) -------------------
// mean molecular mass of a premixed table at every level (meanmolmass_interpol, kernels.cu:649-699) from the table
// indices k_rt_tp_index left behind
__global__ void k_rt_mmm_table(KArgs a, const double* __restrict__ table, double* mmm_lay, double* mmm_int) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.I || a.done[col]) return;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 0 && i >= a.L) continue;
        const TPIndex k = (pass == 0 ? a.tp_lay : a.tp_int)[(size_t)col * a.I + i];
        (pass == 0 ? mmm_lay : mmm_int)[(size_t)col * a.I + i] =
            blend_tp(table[k.pdown + a.npress * k.tdown], table[k.pup + a.npress * k.tdown],
                     table[k.pdown + a.npress * k.tup], table[k.pup + a.npress * k.tup], k, false);
    }
}

// premixed k-table look-up into opac_wg_lay / opac_wg_int (opac_interpol, kernels.cu:524-610); grid (chunks, I, C)
__global__ void __launch_bounds__(256) k_rt_opac_table(KArgs a, double* opac_lay, double* opac_int) {
    const int col = blockIdx.z, lev = blockIdx.y;
    if (a.done[col]) return;
    const size_t nc = (size_t)a.Y * a.X, sp = nc, st = nc * a.npress, wgI = nc * a.I;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 0 && lev >= a.L) continue;
        const TPIndex k = (pass == 0 ? a.tp_lay : a.tp_int)[(size_t)col * a.I + lev];
        double* out = (pass == 0 ? opac_lay : opac_int) + col * wgI + nc * lev;
        for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += (size_t)gridDim.x * blockDim.x)
            out[c] = blend_tp(a.ktable[c + sp * k.pdown + st * k.tdown], a.ktable[c + sp * k.pup + st * k.tdown],
                              a.ktable[c + sp * k.pdown + st * k.tup], a.ktable[c + sp * k.pup + st * k.tup], k, false);
    }
}

// layer heights (calc_delta_z, kernels.cu:1247-1261)
__global__ void k_rt_delta_z(KArgs a, double* __restrict__ dz) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.L || a.done[col]) return;
    const double* pint = a.p_int + (size_t)col * a.I;
    dz[(size_t)col * a.L + i] = HX_KBOLTZMANN * a.T_lay[(size_t)col * (a.L + 1) + i] /
                                (a.mmm_lay[(size_t)col * a.I + i] * a.colpar[col].g) * log(pint[i] / pint[i + 1]);
}

// asymmetry parameter of gas + clouds at the layer centres and interfaces (calc_total_g_0_of_gas_and_clouds, :472-492)
__global__ void __launch_bounds__(256) k_rt_total_g0(KArgs a, const double* __restrict__ g_cl_lay,
                                                     const double* __restrict__ g_cl_int, double* g_tot_lay,
                                                     double* g_tot_int) {
    const int col = blockIdx.z, lev = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.X || a.done[col]) return;
    const size_t k = (size_t)col * a.X * a.I + x + (size_t)a.X * lev;
    if (lev < a.L) {
        const double num = a.g_0 * a.scat_cross_lay[k] + g_cl_lay[k] * a.cl_sc_lay[k];
        g_tot_lay[k] = num / (a.scat_cross_lay[k] + a.cl_sc_lay[k]);
    }
    const double num = a.g_0 * a.scat_cross_int[k] + g_cl_int[k] * a.cl_sc_int[k];
    g_tot_int[k] = num / (a.scat_cross_int[k] + a.cl_sc_int[k]);
}

// remember the temperatures a refresh used (opacities are rebuilt from them on demand)
__global__ void k_rt_keep_ref_T(KArgs a, double* T_lay_ref, double* T_int_ref) {
    const int col = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.done[col]) return;  // a finished column keeps the state of its last real refresh
    if (i <= a.L) T_lay_ref[(size_t)col * (a.L + 1) + i] = a.T_lay[(size_t)col * (a.L + 1) + i];
    if (i < a.I) T_int_ref[(size_t)col * a.I + i] = a.T_int[(size_t)col * a.I + i];
}

// broadcast a scalar-per-bin cross-section to all levels: out[x + X*i] = src[x]
__global__ void __launch_bounds__(256) k_rt_tile_rows(const double* __restrict__ src,
                                                      double* __restrict__ out, int X, int nlev) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (x < X) out[x + (size_t)X * i] = src[x];
}

}  // namespace hx
