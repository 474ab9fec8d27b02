// Random-overlap mixing of one more absorber into the running k-distribution of a (bin, level) point --
// add_to_mixed_opac, kernels.cu:3263-3399 -- as a device function for ONE wavefront, shared by the
// per-stage entry point (stage_mixing.hip) and the species loop of the fused refresh (rt_fused.hip).
//
// The reference forms the 400 pair sums K = mix[y1] + add[y2] with weights (w[y1]/2)(w[y2]/2) in a crossing-dependent
// fill order, sorts them with repeated adjacent swaps (a stable sort), accumulates mid-point abscissae and reads the
// 20 new Gauss points off by linear interpolation.  Here:
//
//  * 512 slots = 64 lanes x 8; slot r of lane l starts with fill position e = 64 r + l, so the per-position LDS images of
//    the sums and weights are written with unit stride (no bank conflicts).
//  * If the sums are already ascending in fill order -- the reference fills the stronger curve on the outer loop for
//    that reason, and a quarter of the problems of a 20-species mix are -- nothing is sorted at all.
//  * Otherwise a bitonic network sorts ONE 32-bit key per slot,
//        key = q(K) << 9 | e,     q(K) = (bits(K) - (hi32(Kmin) << 32)) >> sh     (23 bits, monotone in K)
//    with `sh` chosen per problem so that the 23 bits span exactly [Kmin, Kmax] (18 mantissa bits for sums that cover
//    eight decades).  A compare-exchange is v_min_u32 / v_max_u32 inside a lane and one DPP move + v_med3_u32 per slot
//    across lanes (med3(own, partner, 0) = min for the lower lane, med3(own, partner, ~0) = max for the upper one):
//    about 530 vector instructions for the 45 steps, against 1.4 k for a network that moves fp64 sums with a payload.
//    Keys with equal q are ordered by fill position, which is the reference's stable order for equal sums.
//  * What the quantisation can get wrong -- two sums of different rows closer than 2^-18 relative, in the wrong fill
//    order -- is repaired on the exact fp64 values: the sorted keys fetch (K, weight) from LDS, a wave-uniform test looks
//    for an inversion, and only then odd-even transposition passes (strict '>', hence stable) run until none is left.
//    The permutation is therefore exactly the reference's; the fp64 network and the all-pairs ranking kept in
//    stage_mixing.hip give bit-identical results (tests/test_gpu_stages.py::test_random_overlap_orderings_vs_oracle).
//  * Sorted sums and abscissae go back to LDS at index w + w/8 (stride 9 doubles between lanes: conflict-free), the 20
//    Gauss points find their interval by binary search.
#pragma once
#include "hx_common.h"

#ifndef RO_SWIZZLE_XOR4
#define RO_SWIZZLE_XOR4 0
#endif
#ifndef RO_DPP_MINMAX4
#define RO_DPP_MINMAX4 1   // same-box A/B at config 3: 43.6 -> 43.05 ms per refresh
#endif

// phase markers for the static instruction budget (tools/isa_stats.py --phases): comments in the listing, nothing else
#ifdef RO_MARKERS
#define RO_MARK(name) asm volatile("; RO_MARK " name)
#else
#define RO_MARK(name)
#endif

namespace ro {

constexpr int NY = 20;
constexpr int N = NY * NY;   // 400 pair sums
constexpr int SLOTS = 8;     // per lane: 512 slots
constexpr int LDS_N = 512;   // by fill position (<= 511) or by padded rank (399 + 49)
// LDS images of the run layout: the sum of cell (i, j) of a tableau without a crossing lives at index 21 i + j -- a row
// pitch of 21 instead of the reference's 20 spreads the sixteen lanes of a store over all banks (with pitch 20 they hit
// four bank pairs: same-box A/B +7 % on the whole mixing kernel).  21 i + j ascends with the fill position 20 i + j, so it
// serves as the tie-break of equal quantised sums just as well.  Index PAD holds (inf, 0) for every padding slot.
constexpr int ROW_PITCH = 21;
constexpr int PAD = 511;
// Cells of the two images that no path writes after init: the fill uses the indices 21 i + j <= 418 and PAD, the sorted
// images the padded ranks <= 449.  They hold the re-binning of a PRESORTED tableau, worked out once per wavefront
// (prepare_presorted): rank w of such a tableau is cell (w / 20, w % 20) whatever the two curves are, so the weights in rank
// order, their abscissae and the pair of ranks that brackets each Gauss point do not depend on the problem.  For Gauss
// point q: A[PRE_Y + q], B[PRE_Y + q] = abscissae of the two ranks; the 64 bits of A[PRE_C + q] = LDS byte offsets of their
// cells' table entries (per rank: op entry | ip entry << 16; 0: the walk ran out of sums, :3379-3396); the bits of
// B[PRE_C] = Gauss points moved on by the reference's walk (:3383-3387) per problem.
constexpr int PRE_Y = 470, PRE_C = 490;
// The network's positions 0 ... 15 hold LOW padding (it sorts in front of every sum), so that the left half of the network
// -- rows 0 ... 11 of the tableau, see run_lane -- is exactly 256 positions: rank w sits at position RANK0 + w.
constexpr int RANK0 = 16;
constexpr int LOWPAD = 510;   // cell of the low padding: (-inf, 0); PAD holds (inf, 0)
static_assert(RANK0 + NY * NY - 1 + (RANK0 + NY * NY - 1) / 8 + 1 < PRE_Y && PRE_Y + NY <= PRE_C && PRE_C + NY <= LOWPAD,
              "free cells of the images");

constexpr int NTAB_O = 21, NTAB_I = 28;  // table entries per curve: 20 Gauss points, then constants for the padding slots

struct Pair {
    double v, hw;  // a curve's coefficient at a Gauss point and the half weight of that point, read with one 16-byte load
};

struct Shared {
    // op: the curve that is stronger at y = 0 (outer fill loop), ip: the other one.  The entries from NY on are constants,
    // (inf, 0): what the padding slots of the run layout read, so that their sum is inf and their weight 0 without a
    // branch or a select (inf + inf = inf, 0 * 0 = 0)
    // (A and B first: their byte offsets inside the struct are then multiples of 512 and fold into the offset fields of the
    // paired 64-byte-strided LDS instructions instead of costing an addition per access)
    double A[LDS_N], B[LDS_N];  // by fill position: pair sum / weight; later by padded position: sorted sum / abscissa
    Pair oplow[4];              // op[-4 ... -1]: (-inf, 0), what the low padding in front of a column piece reads
    Pair op[NTAB_O], ip[NTAB_I];
    double gy[NY];
};

struct Lane {
    unsigned c[3];         // c[t] = 0 where lane bit t is clear (lower partner of an exchange over that bit), else ~0;
                           // bits 3-5 (6 of the 35 exchanges) are extracted where they are used: three registers fewer
    // The run layout (see run_lane): a lane's eight network positions are eight consecutive cells of ONE row or ONE
    // column of the tableau, so one operand of its sums is the same for all slots and the other steps by one table entry.
    unsigned fix, var;     // LDS byte offsets (from the start of Shared) of the fixed entry and of slot 0's varying entry
    unsigned e0step;       // padded fill position 21 i + j of slot 0 | its step per slot (1 along a row, 21 down a column) << 16
    unsigned ij;           // i | j << 8 of slot 0 | (1 << 16 if i steps, 1 << 24 if j steps): the crossing case's fill positions
    // padding: a key of the slots 0-3 is (key & aklo) | oklo -- (~0, 0) for sums, (~0, ~0) for high padding, (0, LOWPAD) for
    // low padding -- and one of the slots 4-7 is key | padhi (all ones where they are high padding)
    unsigned aklo, oklo, padhi;
};

struct Counters {
    unsigned skipped = 0, passes = 0;
};

// The run layout.  K[i][j] = outer[i] + inner[j] of two ascending curves ascends along every row and every column, so
// the 400 sums can be dealt out to the network as runs of 16 positions that are ALREADY ascending -- and then the first
// ten steps of the 45-step bitonic network (which only sort inside blocks of 16) have nothing to do.  The blocks of 16
// (block = lane / 2) are arranged so that the rows 0 ... 11 of the tableau fill the LEFT half of the network exactly and
// the rows 12 ... 19 sit in the right half:
//     blocks  0-11: row i = block, columns 0-15          blocks 12-15: 4 x low padding, then column 16 + (block - 12), rows 0-11
//     blocks 16-23: row i = block - 4, columns 0-15      blocks 24-27: column 16 + (block - 24), rows 12-19, then 8 x high padding
//     blocks 28-31: high padding
// Where row 11 ends below the start of row 12 (half of the problems that need the network at all: k-distributions are
// steep at their upper end) the two halves are sorted lists that follow each other, and the last of the five merge
// phases -- 9 of the 35 steps -- has nothing to do either (mix: `split_apart`).  Low padding sorts to the positions
// 0 ... 15 in either case: rank w sits at position RANK0 + w.
// (400 sums cannot be cut into fewer than 20 ascending chains -- the anti-diagonal is an antichain -- so runs of 32 are
// out of reach.)  Keys inside a run ascend too: equal quantised sums are ordered by fill position, and both fill orders
// of the reference (:3332-3365) ascend along rows and along columns.  Position p = 8 lane + s.
__device__ __forceinline__ void run_lane(Lane& ln, int lane) {
    const unsigned OP = (unsigned)offsetof(Shared, op), IP = (unsigned)offsetof(Shared, ip), PS = (unsigned)sizeof(Pair);
    const int blk = lane >> 1, idx0 = 8 * (lane & 1);
    int i = NY, j = NY, di = 0, dj = 0;   // default: high padding everywhere (constant entries)
    ln.aklo = 0xFFFFFFFFu;
    ln.oklo = ln.padhi = 0xFFFFFFFFu;
    if (blk < 12) { i = blk; j = idx0; dj = 1; ln.oklo = ln.padhi = 0u; }                          // row i, columns idx0 ...
    else if (blk < 16) {                                                                           // column j, rows 0 ... 11 behind
        j = 16 + (blk - 12); i = idx0 - 4; di = 1; ln.oklo = ln.padhi = 0u;                        // four positions of low padding
        if (idx0 == 0) { ln.aklo = 0u; ln.oklo = (unsigned)LOWPAD; }
    }
    else if (blk < 24) { i = 12 + (blk - 16); j = idx0; dj = 1; ln.oklo = ln.padhi = 0u; }         // row i, columns idx0 ...
    else if (blk < 28 && idx0 == 0) { j = 16 + (blk - 24); i = 12; di = 1; ln.oklo = ln.padhi = 0u; }   // column j, rows 12 ... 19
    if (di) {  // the column's entry is the fixed operand
        ln.fix = IP + PS * j;
        ln.var = OP + PS * i;     // (i = -4: the four low entries in front of op)
    } else {
        ln.fix = OP + PS * i;
        ln.var = IP + PS * j;
    }
    ln.e0step = (unsigned)((ROW_PITCH * i + j) & 0xFFFF) | (unsigned)(di ? ROW_PITCH : 1) << 16;
    ln.ij = (unsigned)(i & 0xFF) | (unsigned)j << 8 | (unsigned)di << 16 | (unsigned)dj << 24;
}

__device__ __forceinline__ void prepare_presorted(Shared& sh, int lane);

__device__ __forceinline__ void init(Shared& sh, Lane& ln, int lane, const double* gauss_weight, const double* gauss_y) {
    if (lane < NY) {
        sh.op[lane].hw = sh.ip[lane].hw = 0.5 * gauss_weight[lane];
        sh.gy[lane] = gauss_y[lane];
    } else if (lane < NTAB_I) {
        sh.ip[lane].v = __builtin_inf();
        sh.ip[lane].hw = 0.0;
        if (lane < NTAB_O) {
            sh.op[lane].v = __builtin_inf();
            sh.op[lane].hw = 0.0;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; t++) ln.c[t] = (lane >> t) & 1 ? 0xFFFFFFFFu : 0u;
    run_lane(ln, lane);
    if (lane < 4) {
        sh.oplow[lane].v = -__builtin_inf();
        sh.oplow[lane].hw = 0.0;
    }
    if (lane == 0) {
        sh.A[PAD] = __builtin_inf();
        sh.B[PAD] = 0.0;
        sh.A[LOWPAD] = -__builtin_inf();
        sh.B[LOWPAD] = 0.0;
    }
    prepare_presorted(sh, lane);
}

__device__ __forceinline__ void flush(const Counters& c, int lane, unsigned long long* diag) {
    if (lane == 0) {
        if (c.skipped) atomicAdd(diag + HX_DIAG_RO_REBIN, (unsigned long long)c.skipped);
        if (c.passes) atomicAdd(diag + HX_DIAG_RO_FIXUP, (unsigned long long)c.passes);
    }
}

__device__ __forceinline__ unsigned med3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// the value lane^M holds: DPP moves where one or two of them express the permutation, the LDS crossbar for M = 16, 31, 63.
// RO_SWIZZLE_MASK (bit M set: exchange with lane^M through ds_swizzle instead of DPP) moves vector instructions over to
// the LDS pipe -- the kernel is bound by vector issue (VALU active 90 % of the time, the LDS unit 30 %), the exchanges
// then wait for the crossbar; which mix wins is measured
// Measured at config 3 (43.6 ms with none): lane^4 through the crossbar 46.1 ms, lane^4 and ^8 47.1, ^4 ^7 ^8 ^15 48.8 --
// the DPP forms stay.  The exchanges that go through the crossbar anyway (lane^16, lane^31) take ds_swizzle instead of
// ds_bpermute (no address register): 43.1 ms.
#ifndef RO_SWIZZLE_MASK
#define RO_SWIZZLE_MASK ((1u << 16) | (1u << 31) | (RO_SWIZZLE_XOR4 ? (1u << 4) : 0u))
#endif
template <int M>
__device__ __forceinline__ unsigned xor_lane(int addr, unsigned x) {
    const int v = (int)x;
    if constexpr (M < 32 && ((RO_SWIZZLE_MASK >> M) & 1u))
        return __builtin_amdgcn_ds_swizzle(v, (M << 10) | 0x1F);  // bit mode: and 0x1f, or 0, xor M (inside each half of the wavefront)
    else if constexpr (M == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return __builtin_amdgcn_mov_dpp(v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (M == 15) return __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); // row_mirror
    else if constexpr (M == 8) return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (M == 4)   // no single DPP pattern: 7 ^ 3 as two moves
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
    else return __builtin_amdgcn_ds_bpermute(addr, v);
}

struct Keys {
    unsigned k[SLOTS];
};

template <int J>
__device__ __forceinline__ void lane_step(Keys& v) {  // slot s against s^J
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
        if ((s & J) == 0) {
            const unsigned a = v.k[s], b = v.k[s | J];
            v.k[s] = min(a, b);
            v.k[s | J] = max(a, b);
        }
}

template <int W>
__device__ __forceinline__ void lane_mirror(Keys& v) {  // slot s against s^(W-1) inside blocks of W
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
        if ((s & (W - 1)) < W / 2) {
            const unsigned a = v.k[s], b = v.k[s ^ (W - 1)];
            v.k[s] = min(a, b);
            v.k[s ^ (W - 1)] = max(a, b);
        }
}

// exchange with lane^M: slot s meets the partner's slot s (plain step) or 7-s (MIRROR, first step of a phase)
template <int M, bool MIRROR>
__device__ __forceinline__ void cross_step(Keys& v, int lane, const Lane& ln) {
    constexpr int TOP = MIRROR ? (M + 1) / 2 : M;
    constexpr int T = TOP == 1 ? 0 : TOP == 2 ? 1 : TOP == 4 ? 2 : TOP == 8 ? 3 : TOP == 16 ? 4 : 5;
#if RO_DPP_MINMAX4
    if constexpr (M == 4 && !MIRROR) {
        // lane^4 has no single DPP pattern, but its two halves have: the lanes with bit 2 clear sit in the DPP banks 0 and 2
        // and find their partner four lanes up (row_shl:4), the others in banks 1 and 3 four lanes down (row_shr:4).  The
        // min / max themselves take the DPP operand, each writing only its banks: two instructions per slot, no med3.
        // (s_nop: a DPP operand must not be read within two cycles of the instruction that wrote it)
        Keys n;
        asm volatile("s_nop 1\n\t"
                     "v_min_u32_dpp %0, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %0, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %1, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %2, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %2, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %3, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %4, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %4, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %5, %13, %13 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %5, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %6, %14, %14 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %6, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_min_u32_dpp %7, %15, %15 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_max_u32_dpp %7, %15, %15 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(n.k[0]), "=&v"(n.k[1]), "=&v"(n.k[2]), "=&v"(n.k[3]), "=&v"(n.k[4]), "=&v"(n.k[5]), "=&v"(n.k[6]), "=&v"(n.k[7])
                     : "v"(v.k[0]), "v"(v.k[1]), "v"(v.k[2]), "v"(v.k[3]), "v"(v.k[4]), "v"(v.k[5]), "v"(v.k[6]), "v"(v.k[7]));
        v = n;
        return;
    }
#endif
    const int addr = (lane ^ M) << 2;
    unsigned c;
    if constexpr (T < 3) c = ln.c[T];
    else c = (unsigned)__builtin_amdgcn_sbfe(lane, T, 1);  // v_bfe_i32: 0 or ~0
    Keys n;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
        const int ps = MIRROR ? SLOTS - 1 - s : s;
        n.k[s] = med3(v.k[s], xor_lane<M>(addr, v.k[ps]), c);
    }
    v = n;
}

// position p = 8 lane + slot, ascending
__device__ __forceinline__ void sort512(Keys& v, int lane, const Lane& ln) {
    lane_step<1>(v);
    lane_mirror<4>(v); lane_step<1>(v);
    lane_mirror<8>(v); lane_step<2>(v); lane_step<1>(v);
#define RO_LANE_TAIL lane_step<4>(v); lane_step<2>(v); lane_step<1>(v);
    cross_step<1, true>(v, lane, ln); RO_LANE_TAIL
    cross_step<3, true>(v, lane, ln); cross_step<1, false>(v, lane, ln); RO_LANE_TAIL
    cross_step<7, true>(v, lane, ln); cross_step<2, false>(v, lane, ln); cross_step<1, false>(v, lane, ln);
    RO_LANE_TAIL
    cross_step<15, true>(v, lane, ln); cross_step<4, false>(v, lane, ln); cross_step<2, false>(v, lane, ln);
    cross_step<1, false>(v, lane, ln); RO_LANE_TAIL
    cross_step<31, true>(v, lane, ln); cross_step<8, false>(v, lane, ln); cross_step<4, false>(v, lane, ln);
    cross_step<2, false>(v, lane, ln); cross_step<1, false>(v, lane, ln); RO_LANE_TAIL
    cross_step<63, true>(v, lane, ln); cross_step<16, false>(v, lane, ln); cross_step<8, false>(v, lane, ln);
    cross_step<4, false>(v, lane, ln); cross_step<2, false>(v, lane, ln); cross_step<1, false>(v, lane, ln);
    RO_LANE_TAIL
#undef RO_LANE_TAIL
}

// the same network entered behind its tenth step: every block of 16 positions (two lanes) is already ascending
// `halves_apart` (wave-uniform): every key of the left half is below every key of the right half -- the last merge phase is skipped
__device__ __forceinline__ void sort512_from_runs16(Keys& v, int lane, const Lane& ln, bool halves_apart) {
#define RO_LANE_TAIL lane_step<4>(v); lane_step<2>(v); lane_step<1>(v);
    cross_step<3, true>(v, lane, ln); cross_step<1, false>(v, lane, ln); RO_LANE_TAIL
    cross_step<7, true>(v, lane, ln); cross_step<2, false>(v, lane, ln); cross_step<1, false>(v, lane, ln);
    RO_LANE_TAIL
    cross_step<15, true>(v, lane, ln); cross_step<4, false>(v, lane, ln); cross_step<2, false>(v, lane, ln);
    cross_step<1, false>(v, lane, ln); RO_LANE_TAIL
    cross_step<31, true>(v, lane, ln); cross_step<8, false>(v, lane, ln); cross_step<4, false>(v, lane, ln);
    cross_step<2, false>(v, lane, ln); cross_step<1, false>(v, lane, ln); RO_LANE_TAIL
    if (!halves_apart) {
        cross_step<63, true>(v, lane, ln); cross_step<16, false>(v, lane, ln); cross_step<8, false>(v, lane, ln);
        cross_step<4, false>(v, lane, ln); cross_step<2, false>(v, lane, ln); cross_step<1, false>(v, lane, ln);
        RO_LANE_TAIL
    }
#undef RO_LANE_TAIL
}

__device__ __forceinline__ double shfl(int addr, double x) {  // the value of lane addr / 4
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ constexpr int padded(int w) { return w + (w >> 3); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double x) {  // the DPP-selected lane's value, 0.0 where there is none
    // a lane without a source inside its row reads 0 through bound_ctrl (no register to clear beforehand); only the
    // row-masked broadcasts need a zeroed destination for the rows they leave out
    constexpr bool BC = ROW_MASK == 0xF;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, BC);
    return __hiloint2double(hi, lo);
}

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8), then the row
// totals across (row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3).  All variants of the mixing kernel
// add in this order.
__device__ __forceinline__ double wave_inclusive_sum(double x) {
    x += dpp_or_zero<0x111, 0xF>(x);
    x += dpp_or_zero<0x112, 0xF>(x);
    x += dpp_or_zero<0x114, 0xF>(x);
    x += dpp_or_zero<0x118, 0xF>(x);
    x += dpp_or_zero<0x142, 0xA>(x);
    x += dpp_or_zero<0x143, 0xC>(x);
    return x;
}

// Orders this wavefront's LDS traffic.  One wavefront works on one problem and the LDS unit serves a wavefront's
// instructions in issue order, so a fence at wavefront scope (no s_barrier, no wait for acknowledgements) is all that
// is needed between a write by one lane and a read by another; workgroups may therefore hold several wavefronts.
__device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The abscissae of the eight ranks a lane holds (exclusive prefix `run` of the weights before them), and optionally the
// sums, to LDS at pitch 9 per lane (conflict-free stores); the ninth cell of a lane takes the NEXT lane's first abscissa, so
// that the image is a gap-free ascending array and the search needs no index arithmetic.  All 64 lanes call.
template <bool WITH_SUMS>
__device__ __forceinline__ void put_abscissae(Shared& sh, int lane, const double (&K)[SLOTS], const double (&g)[SLOTS], double run) {
    const double y_first = fma(0.5, g[0], run);
    const double y_next = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(y_first), 0x130, 0xF, 0xF, false),
                                           __builtin_amdgcn_update_dpp(0, __double2loint(y_first), 0x130, 0xF, 0xF, false));  // wave_shl:1
    // the lanes that hold ranks: positions RANK0 ... RANK0 + 399, whole lanes (both bounds are multiples of 8)
    const bool mine = SLOTS * lane >= RANK0 && SLOTS * lane < RANK0 + N;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        if (mine) {
            if (WITH_SUMS) sh.A[9 * lane + r] = K[r];  // = padded(position)
            sh.B[9 * lane + r] = fma(0.5, g[r], run);  // = run + 0.5 g bit for bit (0.5 g is exact): one instruction
        }
        run += g[r];
    }
    if (mine) sh.B[9 * lane + SLOTS] = y_next;  // (last lane: the high padding's abscissa, the total weight, above every Gauss point)
}

// re-binning (:3379-3396): the rank w >= 1 whose abscissa is the first above Gauss point `lane`'s (returned in yq), at most
// one Gauss point per rank.  All 64 lanes call; lanes >= NY return a rank beyond the sums.
__device__ __forceinline__ int locate(Shared& sh, int lane, double& yq, unsigned& skipped) {
    int w = N + lane;  // beyond the Gauss points: ascending, so that no skip is seen there
    yq = 0.0;
    if (lane < NY) {
        yq = sh.gy[lane];
        // lower bound over the padded positions of the ranks 1 ... 399 and the duplicates between them, with lengths known
        // at compile time: ten dependent LDS reads at immediate offsets from one running byte offset -- compare, add,
        // select per step
        constexpr int P0 = padded(RANK0 + 1);                       // rank 1
        constexpr int NP = padded(RANK0 + N - 1) + 1 - P0 + 1;      // ... rank 399 and the duplicate behind it
        const char* Bb = (const char*)sh.B;
        unsigned pb = 8 * P0;
#pragma unroll
        for (int len = NP; len > 1; len -= len / 2) {
            const int half = len / 2;
            pb = *(const double*)(Bb + pb + 8 * (half - 1)) > yq ? pb : pb + 8 * half;
        }
        pb += *(const double*)(Bb + pb) > yq ? 0u : 8u;
        const unsigned pi = pb >> 3;
        // padded index -> position: minus pi / 9 (exact below 512; a duplicate cell 9 l + 8 gives 8 l + 8, the position it
        // stands for) -> rank
        w = (int)(pi - (__umul24(pi, 7282u) >> 16)) - RANK0;
    }
    // a Gauss point that falls into the interval of its predecessor takes the next one (the reference's walk advances w
    // before it looks at the next point, and reports a malfunction, :3383-3387): w'_q = max over j <= q of (w_j + q - j)
    int wq = w;
    const int wprev = __shfl_up(w, 1);
    if (__ballot(lane >= 1 && lane < NY && w <= wprev) != 0) {  // never seen with Gauss-Legendre points and weights
        int t = lane < NY ? w - lane : -(1 << 20);
        for (int d = 1; d < 32; d <<= 1) {
            const int up = __shfl_up(t, d);
            if (lane >= d) t = max(t, up);
        }
        wq = t + lane;
        skipped += __popcll(__ballot(lane < NY && wq != w));
    }
    return wq;
}

__device__ __forceinline__ void prepare_presorted(Shared& sh, int lane) {
    sync();  // the half weights and the Gauss points are in LDS
    // exactly what mix() does with a tableau it finds presorted -- weights in rank order, their sum per lane, the wave scan,
    // the abscissae, the interval search, the skip rule -- so that the values kept here are the ones it would compute
    const unsigned OP = (unsigned)offsetof(Shared, op), IP = (unsigned)offsetof(Shared, ip), PS = (unsigned)sizeof(Pair);
    double K[SLOTS], g[SLOTS];
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        const int w = SLOTS * lane + r - RANK0;   // the rank at this position
        K[r] = 0.0;
        g[r] = (w >= 0 && w < N) ? sh.op[w / NY].hw * sh.ip[w % NY].hw : 0.0;
    }
    double csum = 0.0;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) csum += g[r];
    const double run = wave_inclusive_sum(csum) - csum;
    put_abscissae<false>(sh, lane, K, g, run);
    sync();
    double yq;
    unsigned skipped = 0;
    const int wq = locate(sh, lane, yq, skipped);
    double y0 = 0.0, y1 = 0.0;
    unsigned long long cells = 0ull;
    if (lane < NY && wq < N) {
        y0 = sh.B[padded(RANK0 + wq - 1)];
        y1 = sh.B[padded(RANK0 + wq)];
        const unsigned c0 = (OP + PS * ((wq - 1) / NY)) | (IP + PS * ((wq - 1) % NY)) << 16;
        const unsigned c1 = (OP + PS * (wq / NY)) | (IP + PS * (wq % NY)) << 16;
        cells = (unsigned long long)c1 << 32 | c0;
    }
    sync();  // every lane has read its abscissae: the images are free again
    if (lane < NY) {
        sh.A[PRE_Y + lane] = y0;
        sh.B[PRE_Y + lane] = y1;
        sh.A[PRE_C + lane] = __longlong_as_double((long long)cells);
    }
    if (lane == 0) sh.B[PRE_C] = __longlong_as_double((long long)skipped);
}

// One problem.  Lanes 0..19 pass the running mix and the new absorber's (already scaled) k-coefficients at their Gauss
// point and receive the mixed value (kernels.cu:3293-3396, ro_method == 1, s > 0, ny == 20).  All 64 lanes must call.
template <bool MONOTONE, bool CROSSING>
__device__ __forceinline__ void fill(Shared& sh, const Lane& ln, int lane, Keys& v, int yx, int hmin, int sh_bits) {
    asm volatile("" : "+v"(lane));  // rare path (a curve that is not ascending): its per-slot addresses are derived here,
                                    // not kept in fourteen registers through the whole kernel
    const int nfirst = NY * yx;
    const int inv_yx = (1048576 + yx - 1) / yx;
    const char* outer = (const char*)sh.op;
    const char* inner = (const char*)sh.ip;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        // position 64 r + lane: RANK0 positions of low padding, the sums in fill order, high padding
        const int e = 64 * r + lane - RANK0;
        unsigned key = e < 0 ? (unsigned)LOWPAD : 0xFFFFFE00u | (unsigned)PAD;
        if (e >= 0 && e < N) {
            int aq = 16 * (e / 20), ar = 16 * (e % 20);  // byte offsets of op[e / 20], ip[e % 20]
            if (CROSSING) {  // the curves cross: two fill regions (:3332-3365)
                const bool first = e < nfirst;
                // e / yx and e / 20 for e < 512 as multiply-shift (exact: e * d < 2^20 / d for d <= 20)
                const int q = (int)(__umul24(e, first ? inv_yx : 52429) >> 20);
                const int rem = e - __umul24(q, first ? yx : NY);
                // second part: the curves have changed places, the other one is on the outer loop
                aq = 16 * (first ? q : rem);
                ar = 16 * (first ? rem : q);
            }
            const Pair po = *(const Pair*)(outer + aq), pi = *(const Pair*)(inner + ar);
            const double K = po.v + pi.v;
            int dh = __double2hiint(K) - hmin;
            if (!MONOTONE) dh = max(dh, 0);
            unsigned q23 = sh_bits >= 32 ? (unsigned)dh >> (sh_bits - 32)
                                         : __builtin_amdgcn_alignbit((unsigned)dh, (unsigned)__double2loint(K), sh_bits);
            if (!MONOTONE) q23 = min(max(q23, 1u), 0x7FFFFFu);
            key = q23 << 9 | (unsigned)e;
            sh.A[e] = K;
            sh.B[e] = po.hw * pi.hw;
        }
        v.k[r] = key;
    }
}

// fill for two ascending curves: the sums in the run layout (no index arithmetic: the cell of a slot is fixed), their fill
// positions e of the reference's order -- e0 without a crossing, j < yx ? j + yx i : i + 20 j with one (:3332-3365 with the
// stronger curve on the outer loop) -- as the low key bits and as the address of the LDS images.  HI: sh_bits >= 32.
template <bool CROSSING, bool HI>
__device__ __forceinline__ void fill_runs(Shared& sh, const Lane& ln, Keys& v, int yx, int hmin, int sh_bits) {
    const char* base = (const char*)&sh;
    // (the barriers: what is derived from the lane's constants -- eight addresses, eight fill positions -- is derived
    // here, per problem; hoisted out of the problem loop it would sit in two dozen registers)
    unsigned fix = ln.fix, var = ln.var, e0step = ln.e0step;
    asm volatile("" : "+v"(fix), "+v"(var), "+v"(e0step));
    const Pair F = *(const Pair*)(base + fix);
    const int e0 = (int)(short)(e0step & 0xFFFF);   // (negative where the lane starts with low padding)
    const unsigned estep = e0step >> 16;
    // padding has no entry of its own in the images (the cells PAD and LOWPAD hold its values since init)
    const bool sums_lo = ln.oklo == 0u, sums_hi = ln.padhi == 0u;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        const Pair V = *(const Pair*)(base + var + (unsigned)sizeof(Pair) * r);
        const double K = F.v + V.v;      // padding: inf, -inf
        const double g = F.hw * V.hw;    // padding: 0
        unsigned e = (unsigned)(e0 + (int)estep * r);     // index of the LDS images and tie-break of the key
        if (CROSSING) {
            const int i = (int)(signed char)(ln.ij & 0xFF) + (int)((ln.ij >> 16) & 1) * r;
            const int j = (int)((ln.ij >> 8) & 0xFF) + (int)(ln.ij >> 24) * r;
            e = (unsigned)(j < yx ? j + yx * i : i + 20 * j);
        }
        const unsigned dh = (unsigned)(__double2hiint(K) - hmin);
        const unsigned q23 = HI ? dh >> (sh_bits - 32) : __builtin_amdgcn_alignbit(dh, (unsigned)__double2loint(K), sh_bits);
        const unsigned key = q23 << 9 | e;
        v.k[r] = r < SLOTS / 2 ? (key & ln.aklo) | ln.oklo : key | ln.padhi;   // high padding: all ones; low padding: LOWPAD
        if (r < SLOTS / 2 ? sums_lo : sums_hi) {
            *(double*)((char*)sh.A + (e << 3)) = K;
            *(double*)((char*)sh.B + (e << 3)) = g;
        }
    }
}

__device__ __forceinline__ double mix(Shared& sh, const Lane& ln, int lane, double my_mix, double my_add, Counters& cnt) {
    RO_MARK("prologue");
    // corners of the tableau, wave-uniform
    const double m0 = __shfl(my_mix, 0), a0 = __shfl(my_add, 0), m19 = __shfl(my_mix, NY - 1), a19 = __shfl(my_add, NY - 1);
    // less than 1 % of the other everywhere: correlated-k (:3297-3310)
    if ((0.01 * m0 > a19) || (0.01 * a0 > m19)) return my_mix + my_add;
    const bool mix_first = m0 > a0;
    sync();  // the previous problem's readers are done with sh
    if (lane < NY) {
        sh.op[lane].v = mix_first ? my_mix : my_add;
        sh.ip[lane].v = mix_first ? my_add : my_mix;
    }
    sync();
    // last crossing of the two curves (:3321-3329); are both k-distributions (ascending)?
    // ... and do the rows of the tableau overlap?  If row i ends below the start of row i + 1 for every i, the sums are
    // ascending in fill order (without a crossing: e = 20 i + j) and nothing has to be sorted
    bool cross = false, down = false, over = false, touch = false;
    if (lane >= 1 && lane < NY) {
        const double po = sh.op[lane - 1].v, pi = sh.ip[lane - 1].v;
        const double pm = mix_first ? po : pi, pa = mix_first ? pi : po;
        cross = (my_mix > my_add) != (pm > pa);
        down = my_mix < pm || my_add < pa;
        const double mo = mix_first ? my_mix : my_add;  // outer[lane]
        const double row_end = po + (mix_first ? a19 : m19), next_start = mo + (mix_first ? a0 : m0);
        over = row_end > next_start;
        touch = row_end >= next_start;
    }
    const unsigned long long cmask = __ballot(cross);
    const int yx = cmask ? 63 - __clzll((long long)cmask) : NY;
    const bool monotone = __ballot(down) == 0;
    const unsigned long long overlaps = __ballot(over);
    const bool rows_apart = overlaps == 0;
    // row 11 ends below the start of row 12: the two halves of the network's run layout are sorted lists that follow each
    // other (run_lane)
    // (strictly below: with a crossing the reference's fill order does not follow the rows, and equal sums keep fill order)
    const bool halves_apart = ((__ballot(touch) >> 12) & 1ull) == 0;
    double kmin = m0 + a0, kmax = m19 + a19;
    if (!monotone) {  // the extreme sums are not at the corners of the tableau
        double mn1 = sh.op[0].v, mx1 = mn1, mn2 = sh.ip[0].v, mx2 = mn2;
#pragma unroll 1
        for (int j = 1; j < NY; j++) {
            mn1 = fmin(mn1, sh.op[j].v); mx1 = fmax(mx1, sh.op[j].v);
            mn2 = fmin(mn2, sh.ip[j].v); mx2 = fmax(mx2, sh.ip[j].v);
        }
        kmin = mn1 + mn2;
        kmax = mx1 + mx2;
    }
    // key scale (wave-uniform): q = (bits(K) - (hmin << 32)) >> sh with 1 <= q < 2^23 for Kmin <= K <= Kmax (q = 0 is the
    // low padding's): hmin lies one unit of q or more below hi32(Kmin)
    const int hk = __builtin_amdgcn_readfirstlane(__double2hiint(kmin));
    const unsigned long long span =
        ((unsigned long long)(unsigned)(__builtin_amdgcn_readfirstlane(__double2hiint(kmax)) - hk) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(kmax));
    const int bl0 = span ? 64 - __clzll((long long)span) : 0;
    const int bias = 1 << max(0, bl0 - 23 - 31);                       // in units of 2^32: >= 2^sh once sh is settled below
    const int hmin = hk - bias;
    const unsigned long long dmax = span + ((unsigned long long)(unsigned)bias << 32);
    const int bl = 64 - __clzll((long long)dmax);
    const int sh_bits = bl > 23 ? bl - 23 : 0;
    RO_MARK("fill");
    if (monotone && yx == NY && rows_apart) {
        // nothing to sort -- and nothing to scan or to search either: rank w is cell (w / 20, w % 20), the abscissae and
        // the ranks that bracket each Gauss point are those prepare_presorted() worked out (same arithmetic, same bits)
        double out = my_mix;
        if (lane < NY) {
            const unsigned long long cells = (unsigned long long)__double_as_longlong(sh.A[PRE_C + lane]);
            if (cells != 0ull) {
                const char* base = (const char*)&sh;
                const unsigned c0 = (unsigned)cells, c1 = (unsigned)(cells >> 32);
                const double K0 = ((const Pair*)(base + (c0 & 0xFFFFu)))->v + ((const Pair*)(base + (c0 >> 16)))->v;
                const double K1 = ((const Pair*)(base + (c1 & 0xFFFFu)))->v + ((const Pair*)(base + (c1 >> 16)))->v;
                const double yq = sh.gy[lane], y0 = sh.A[PRE_Y + lane], y1 = sh.B[PRE_Y + lane];
                out = (K0 * (y1 - yq) + K1 * (yq - y0)) / (y1 - y0);
            }
        }
        cnt.skipped += (unsigned)__double_as_longlong(sh.B[PRE_C]);
        return out;
    }
    double K[SLOTS], g[SLOTS];
    {
        Keys v;
        if (monotone) {
            if (yx == NY) {
                if (sh_bits >= 32) fill_runs<false, true>(sh, ln, v, yx, hmin, sh_bits);
                else fill_runs<false, false>(sh, ln, v, yx, hmin, sh_bits);
            } else {
                if (sh_bits >= 32) fill_runs<true, true>(sh, ln, v, yx, hmin, sh_bits);
                else fill_runs<true, false>(sh, ln, v, yx, hmin, sh_bits);
            }
            int lv = lane;
            RO_MARK("network");
            asm volatile("" : "+v"(lv));  // what the network derives from the lane id (exchange addresses, bits 3-5) is
                                          // rebuilt per problem and does not sit in registers between the problems
            sort512_from_runs16(v, lv, ln, halves_apart);
        } else {  // a curve that is not a k-distribution: positions in fill order, the whole network
            if (yx == NY) fill<false, false>(sh, ln, lane, v, yx, hmin, sh_bits);
            else fill<false, true>(sh, ln, lane, v, yx, hmin, sh_bits);
            sort512(v, lane, ln);
        }
        RO_MARK("fetch");
        sync();
#pragma unroll
        for (int r = 0; r < SLOTS; r++) {
            const int src = (int)(v.k[r] & 511) << 3;  // byte offset of the LDS images' entry of the element at rank 8 lane + r
            K[r] = *(const double*)((const char*)sh.A + src);
            g[r] = *(const double*)((const char*)sh.B + src);
        }
    }
    RO_MARK("finish");
    // exact finish: any inversion left by the quantisation?
    const int next = (lane < 63 ? lane + 1 : lane) << 2, prev = (lane > 0 ? lane - 1 : lane) << 2;
    int passes = 0;
    for (;;) {
        bool inv = false;
#pragma unroll
        for (int r = 0; r + 1 < SLOTS; r++) inv = inv || K[r] > K[r + 1];
        const double kn = shfl(next, K[0]);
        inv = inv || (lane < 63 && K[SLOTS - 1] > kn);
        if (__ballot(inv) == 0 || passes >= 2 * LDS_N) break;
        passes++;
        auto ce = [&](int a, int b) {
            const bool sw = K[a] > K[b];
            const double ka = K[a], kb = K[b], ga = g[a], gb = g[b];
            K[a] = sw ? kb : ka; K[b] = sw ? ka : kb;
            g[a] = sw ? gb : ga; g[b] = sw ? ga : gb;
        };
        ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);
        ce(1, 2); ce(3, 4); ce(5, 6);
        const double kn0 = shfl(next, K[0]), gn0 = shfl(next, g[0]);
        const double kp7 = shfl(prev, K[SLOTS - 1]), gp7 = shfl(prev, g[SLOTS - 1]);
        const bool sw_hi = lane < 63 && K[SLOTS - 1] > kn0, sw_lo = lane > 0 && kp7 > K[0];
        if (sw_hi) { K[SLOTS - 1] = kn0; g[SLOTS - 1] = gn0; }
        if (sw_lo) { K[0] = kp7; g[0] = gp7; }
    }
    cnt.passes += passes;
    RO_MARK("scan");
    // cumulative mid-point abscissae Y_w = sum_{v<w} g_v + g_w/2 (:3371-3376): 8 per lane + wave exclusive scan
    double csum = 0.0;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) csum += g[r];
    double run = wave_inclusive_sum(csum) - csum;
    sync();  // every lane has fetched its (K, g): A and B change meaning
    put_abscissae<true>(sh, lane, K, g, run);
    sync();
    RO_MARK("search");
    double yq;
    const int wq = locate(sh, lane, yq, cnt.skipped);
    RO_MARK("interpolate");
    double out = my_mix;  // w = 400: the walk ran out of sums, the reference leaves the entry as it was
    if (lane < NY && wq < N) {
        const int i0 = padded(RANK0 + wq - 1), i1 = padded(RANK0 + wq);
        out = (sh.A[i0] * (sh.B[i1] - yq) + sh.A[i1] * (yq - sh.B[i0])) / (sh.B[i1] - sh.B[i0]);
    }
    RO_MARK("end");
    return out;
}

}  // namespace ro
