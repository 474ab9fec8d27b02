// Random-overlap mixing for FIVE wavefronts per SIMD (round 6): the algorithm of random_overlap.h -- run layout, one 32-bit key
// per slot, bitonic network entered behind its tenth step, exact finish, DPP scan, interval search -- with the two things that
// held the mixing kernel at four wavefronts per SIMD removed (profiles/r05_mix_occupancy.txt: t = 13 + 85 / w ms):
//
//  * LDS.  random_overlap.h keeps two 4 KB images: (pair sum, weight) by fill position for the fetch behind the network, then
//    (sorted sum, abscissa) by rank for the search.  Here a key carries its CELL -- the low bits are i << 5 | j instead of the
//    fill position 21 i + j; both ascend with the reference's fill order -- so whoever holds a key recomputes the sum and the
//    weight from the two 20-entry curve tables (two reads, one add / one product: the bits the fill computed).  Nothing is
//    written at fill time, the network's result is kept as 400 keys by rank (1.6 KB), and the only fp64 image left is the
//    abscissae (3.6 KB): 6.9 KB per wavefront with the tables, 7.9 KB with the species list of k_rt_mix_species -- twenty
//    wavefronts share a CU's 160 KB.
//  * Registers.  The compiler hoists everything that depends on the lane alone out of the problem loop -- eight cell addresses,
//    eight fill positions, scan and search offsets, 32-bit literals its VOP3 encodings cannot hold: 55 of the old kernel's 128
//    VGPRs (tools/vgpr_liveness.py).  Here a lane's place in the run layout is seven numbers (LaneConst) passed through an
//    optimisation barrier in every problem, and what the phases derive from the lane id is derived from an opaque copy of it.
//  * One decode per slot -- a key's cell as two table offsets -- serves the exact test on the fp64 sums (always made: an
//    "equal quantised sums?" pre-test in front of it fired in most problems and cost more than it saved) and the weights of
//    the scan.
//
// Same permutation, same arithmetic, same bits as random_overlap.h (tests/test_gpu_stages.py::test_random_overlap_orderings_
// vs_oracle holds the kernels to each other bit for bit).  Reference: kernels.cu:3263-3399, sort :3152-3171.
#pragma once
#include "random_overlap.h"

namespace rol {

using ro::Counters;
using ro::Keys;
using ro::N;
using ro::NY;
using ro::RANK0;
using ro::SLOTS;

constexpr int NTAB = 32;                 // entries per curve table: a 5-bit cell index never leaves it
constexpr int LANE0 = RANK0 / SLOTS;     // the first lane that holds ranks (positions RANK0 ... RANK0 + 399: lanes 2 ... 51)
constexpr int NLANES = N / SLOTS;
constexpr int YBASE = (SLOTS + 1) * LANE0;   // padded index of position RANK0: the abscissa image starts there
constexpr int NYIMG = (SLOTS + 1) * NLANES;
// table entries beyond the 20 Gauss points: (inf, 0) everywhere, op[OP_LOW] = (-inf, 0)
constexpr int OP_LOW = 21;
constexpr unsigned HIGHKEY = 0xFFFFFFFFu;            // cell (31, 31): inf + inf, weight 0 x 0
constexpr unsigned LOWKEY = (unsigned)OP_LOW << 5;   // q = 0 (sums have q >= 1), cell (21, 0): -inf + ip[0], weight 0 x w

// entry k of the curve table: both curves' Gauss point k side by side, 32 bytes -- a key's row field (bits 5-9) IS the byte
// offset of its op entry, the column field shifted by five that of its ip entry (one and two instructions per decode)
struct Cell {
    double ov, ohw;   // op: the curve that is stronger at y = 0 (outer fill loop): coefficient, half weight
    double iv, ihw;   // ip: the other one
};
constexpr unsigned IP_OFF = 16;   // byte offset of the ip half inside a Cell

struct Shared {
    double Y[NYIMG];          // abscissae by padded position - YBASE: nine cells per lane, the ninth repeats the next lane's first
    unsigned E[N];            // the sorted keys by rank
    Cell tab[NTAB];
    double gy[NY];
    // the re-binning of a PRESORTED tableau (random_overlap.h, prepare_presorted): per Gauss point the abscissae of the two
    // ranks that bracket it and the LDS byte offsets of their cells' table entries (per rank: op entry | ip entry << 16)
    double pre_y0[NY], pre_y1[NY];
    unsigned long long pre_cells[NY];
    unsigned pre_skipped, pad_;
};
static_assert(sizeof(Shared) <= 6880, "twenty wavefronts per CU: 8 KB each with the species list of k_rt_mix_species");
static_assert(offsetof(Shared, tab) % 16 == 0 && offsetof(Shared, tab) >= 4 * sizeof(Cell), "aligned table; entry -4 stays inside the struct");

// A lane's place in the run layout (random_overlap.h, run_lane): seven registers kept through the kernel (round 6 first
// packed them into two and unpacked per problem, twelve instructions; with the search's literals out of the register file
// there is room for them as they are -- 93 VGPRs in k_rt_mix_species).
//   fix, var: byte offsets (from the start of Shared) of the fixed table operand and of slot 0's varying operand
//   t0, tstep: cell code of slot 0, i0 << 5 | j0 (i0 = -4 in front of a column piece), and its step per slot (1 along a row,
//              32 down a column: the lane walks down a column instead of along a row)
//   aklo, oklo, padhi: slots 0-3 take (key & aklo) | oklo, slots 4-7 key | padhi -- (~0, 0, 0) for sums, (0, LOWKEY, 0) where the
//              lane's slots 0-3 are low padding, (~0, ~0, ~0) where all its slots are high padding
struct LaneConst {
    unsigned fix, var, t0, tstep, aklo, oklo, padhi;
};

__device__ __forceinline__ LaneConst lane_const(int lane) {
    const unsigned TAB = (unsigned)offsetof(Shared, tab), CS = (unsigned)sizeof(Cell);
    const int blk = lane >> 1, idx0 = 8 * (lane & 1);
    int i = NY, j = NY, col = 0, padlow = 0, padhigh = 1;   // default: high padding (reads the constant entries)
    if (blk < 12) { i = blk; j = idx0; padhigh = 0; }                                              // row i, columns idx0 ...
    else if (blk < 16) { j = 16 + (blk - 12); i = idx0 - 4; col = 1; padhigh = 0; padlow = idx0 == 0; }   // column j, rows 0 ... 11 behind
                                                                                                   // four positions of low padding
    else if (blk < 24) { i = 12 + (blk - 16); j = idx0; padhigh = 0; }                             // row i, columns idx0 ...
    else if (blk < 28 && idx0 == 0) { j = 16 + (blk - 24); i = 12; col = 1; padhigh = 0; }         // column j, rows 12 ... 19
    const unsigned o = TAB + CS * i, p = TAB + CS * j + IP_OFF;   // (i = -4: the four cells in front of the table, inside E: any bits do)
    LaneConst lc;
    lc.fix = col ? p : o;
    lc.var = col ? o : p;
    lc.t0 = (unsigned)((i << 5) + j);
    lc.tstep = col ? 32u : 1u;
    lc.padhi = padhigh ? ~0u : 0u;
    lc.aklo = padlow ? 0u : ~0u;
    lc.oklo = padlow ? LOWKEY : lc.padhi;
    return lc;
}

__device__ __forceinline__ void prepare_presorted(Shared& sh, int lane);

__device__ __forceinline__ LaneConst init(Shared& sh, int lane, const double* gauss_weight, const double* gauss_y) {
    if (lane < NTAB) {
        const bool real = lane < NY;
        const double hw = real ? 0.5 * gauss_weight[lane] : 0.0;
        sh.tab[lane].ohw = sh.tab[lane].ihw = hw;
        if (!real) {
            sh.tab[lane].ov = lane == OP_LOW ? -__builtin_inf() : __builtin_inf();
            sh.tab[lane].iv = __builtin_inf();
        }
        if (real) sh.gy[lane] = gauss_y[lane];
    }
    prepare_presorted(sh, lane);
    return lane_const(lane);
}

using ro::dpp_or_zero;
using ro::padded;
using ro::shfl;
using ro::sync;
using ro::wave_inclusive_sum;

__device__ __forceinline__ double from_next_lane(double x) {   // wave_shl:1 (lane 63: 0)
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ double from_prev_lane(double x) {   // wave_shr:1 (lane 0: 0)
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xF, 0xF, false));
}

template <int L>
__device__ __forceinline__ double lane_value(double x) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), L), __builtin_amdgcn_readlane(__double2loint(x), L));
}

// the table entries of a key's cell: byte offsets from the start of the table (the ip half is IP_OFF further: an immediate).
// CROSSING: bit 10 says that the two fields have changed places (the second fill region of :3332-3365, where the inner curve
// runs on the outer loop)
template <bool CROSSING>
__device__ __forceinline__ void decode(unsigned key, unsigned& ao, unsigned& ai) {
    const unsigned hi = key & 0x3E0u, lo = (key << 5) & 0x3E0u;
    if (CROSSING) {
        const bool swapped = (key & 1024u) != 0;
        ao = swapped ? lo : hi;
        ai = swapped ? hi : lo;
    } else {
        ao = hi;
        ai = lo;
    }
}

template <bool CROSSING>
__device__ __forceinline__ double cell_sum(const Shared& sh, unsigned key) {
    unsigned ao, ai;
    decode<CROSSING>(key, ao, ai);
    const char* tb = (const char*)sh.tab;
    return *(const double*)(tb + ao) + *(const double*)(tb + ai + IP_OFF);
}

// The abscissae of the eight ranks a lane holds (exclusive prefix `run` of the weights before them) to LDS at pitch 9 per
// lane; the ninth cell of a lane takes the NEXT lane's first abscissa, so that the image is a gap-free ascending array
// (random_overlap.h, put_abscissae).  All 64 lanes call.
__device__ __forceinline__ void put_abscissae(Shared& sh, int lane, const double (&g)[SLOTS], double run) {
    const double y_first = fma(0.5, g[0], run);
    const double y_next = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(y_first), 0x130, 0xF, 0xF, false),
                                           __builtin_amdgcn_update_dpp(0, __double2loint(y_first), 0x130, 0xF, 0xF, false));  // wave_shl:1
    const bool mine = lane >= LANE0 && lane < LANE0 + NLANES;
    double* y = sh.Y + (SLOTS + 1) * lane - YBASE;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        if (mine) y[r] = fma(0.5, g[r], run);  // = run + 0.5 g bit for bit (0.5 g is exact)
        run += g[r];
    }
    if (mine) y[SLOTS] = y_next;  // (last lane: the high padding's abscissa, the total weight, above every Gauss point)
}

// re-binning (:3379-3396): the rank w >= 1 whose abscissa is the first above Gauss point `lane`'s (returned in yq), at most one
// Gauss point per rank (random_overlap.h, locate: the same search on the same image, whose first cell is YBASE here)
__device__ __forceinline__ int locate(const Shared& sh, int lane, double& yq, unsigned& skipped) {
    int w = N + lane;
    yq = 0.0;
    if (lane < NY) {
        yq = sh.gy[lane];
        constexpr int P0 = padded(RANK0 + 1);                       // rank 1
        constexpr int NP = padded(RANK0 + N - 1) + 1 - P0 + 1;      // ... rank 399 and the duplicate behind it
        const char* Bb = (const char*)sh.Y;
        unsigned pb = 8 * (P0 - YBASE);      // byte offset inside the image, which begins at padded index YBASE
        // Lower bound over the NP = 449 cells: halving gives the steps 224, 112, 56, 28, 14, 7, 4, 2, 1 cells -- the first five
        // are 7 << k, written as `select 0 or 7, shift-add`: as byte counts they are 32-bit literals, which the select cannot
        // take as an operand and the compiler therefore parks in five registers for the whole kernel.  Same probes as before.
        static_assert(NP == 449, "the step sequence below is halving 449 cells");
        unsigned seven = 7u;
        asm volatile("" : "+v"(seven));     // (opaque: the compiler folds `select 0 or 7, shift` back into a select of parked literals)
#pragma unroll
        for (int k = 8; k >= 4; k--) {      // 7 << (k - 3) cells = 7 << k bytes
            const unsigned m = *(const double*)(Bb + pb + (7u << k) - 8) > yq ? 0u : seven;
            pb += m << k;
        }
        pb += *(const double*)(Bb + pb + 56 - 8) > yq ? 0u : 56u;
#pragma unroll
        for (int st = 32; st >= 8; st >>= 1) pb += *(const double*)(Bb + pb + st - 8) > yq ? 0u : (unsigned)st;
        pb += *(const double*)(Bb + pb) > yq ? 0u : 8u;
        const unsigned pi = (pb >> 3) + YBASE;
        w = (int)(pi - (__umul24(pi, 7282u) >> 16)) - RANK0;   // padded index -> position (minus pi / 9) -> rank
    }
    int wq = w;
    const int wprev = __builtin_amdgcn_update_dpp(0, w, 0x138, 0xF, 0xF, false);   // wave_shr:1: the lane below's rank (no address register)
    if (__ballot(lane >= 1 && lane < NY && w <= wprev) != 0) {  // never seen with Gauss-Legendre points and weights
        int ll = lane, floor_ = -(1 << 20);
        asm volatile("" : "+v"(ll), "+v"(floor_));   // this path's addresses and constants are made here, not kept in registers for it
        int t = ll < NY ? w - ll : floor_;
#pragma unroll 1
        for (int d = 1; d < 32; d <<= 1) {   // (rolled: no per-step constants)
            const int up = __builtin_amdgcn_ds_bpermute(max(ll - d, 0) << 2, t);
            if (ll >= d) t = max(t, up);
        }
        wq = t + ll;
        skipped += __popcll(__ballot(ll < NY && wq != w));
    }
    return wq;
}

__device__ __forceinline__ void prepare_presorted(Shared& sh, int lane) {
    sync();  // the half weights and the Gauss points are in LDS
    // exactly what mix() does with a tableau it finds presorted -- weights in rank order (rank w = cell (w / 20, w % 20)),
    // their sum per lane, the wave scan, the abscissae, the interval search, the skip rule
    const unsigned OP = (unsigned)offsetof(Shared, tab), IP = OP + IP_OFF, PS = (unsigned)sizeof(Cell);
    double g[SLOTS];
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        const int w = SLOTS * lane + r - RANK0;
        g[r] = (w >= 0 && w < N) ? sh.tab[w / NY].ohw * sh.tab[w % NY].ihw : 0.0;
    }
    double csum = 0.0;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) csum += g[r];
    const double run = wave_inclusive_sum(csum) - csum;
    put_abscissae(sh, lane, g, run);
    sync();
    double yq;
    unsigned skipped = 0;
    const int wq = locate(sh, lane, yq, skipped);
    double y0 = 0.0, y1 = 0.0;
    unsigned long long cells = 0ull;
    if (lane < NY && wq < N) {
        y0 = sh.Y[padded(RANK0 + wq - 1) - YBASE];
        y1 = sh.Y[padded(RANK0 + wq) - YBASE];
        const unsigned c0 = (OP + PS * ((wq - 1) / NY)) | (IP + PS * ((wq - 1) % NY)) << 16;
        const unsigned c1 = (OP + PS * (wq / NY)) | (IP + PS * (wq % NY)) << 16;
        cells = (unsigned long long)c1 << 32 | c0;
    }
    if (lane < NY) {
        sh.pre_y0[lane] = y0;
        sh.pre_y1[lane] = y1;
        sh.pre_cells[lane] = cells;
    }
    if (lane == 0) sh.pre_skipped = skipped;
}

// keys of the run layout for two ascending curves: key = q << TB | cell code, where the code orders equal quantised sums as
// the reference's fill order does -- i << 5 | j without a crossing (fill position 20 i + j), and with one (:3332-3365 with the
// stronger curve on the outer loop) i << 5 | j in the first region (j < yx: position j + yx i), 1 << 10 | j << 5 | i in the
// second (position i + 20 j, behind all of the first).  HI: sh_bits >= 32.  Nothing is written: the cell IS the address.
template <bool CROSSING, bool HI>
__device__ __forceinline__ void fill_runs(const Shared& sh, LaneConst lc, Keys& v, int yx, int hmin, int sh_bits) {
    constexpr int TB = CROSSING ? 11 : 10;
    const char* base = (const char*)&sh;
    const unsigned fix = lc.fix, var = lc.var, aklo = lc.aklo, oklo = lc.oklo, padhi = lc.padhi;
    const double F = *(const double*)(base + fix);
    // q = (bits(K) >> sh) - (base >> sh) with the base a multiple of 2^sh (mix): the wave-uniform second term, shifted to the
    // keys' q field, goes into the lane's cell code once instead of into every slot (all of it modulo 2^32: the key fits)
    const unsigned qbase = HI ? (unsigned)hmin >> (sh_bits - 32) : (sh_bits ? (unsigned)hmin << (32 - sh_bits) : 0u);
    const unsigned koff = qbase << TB;
    const unsigned t0 = lc.t0 - koff, tstep = lc.tstep;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        const double K = F + *(const double*)(base + var + (unsigned)sizeof(Cell) * r);   // padding: inf, or any bits (masked below)
        unsigned t = t0 + tstep * (unsigned)r;
        if (CROSSING) {
            // (i, j) of this slot from its code: the low five bits never carry (j0 + r <= 15 along a row, j fixed down a column)
            const unsigned tc = t + koff;
            const int i = (int)tc >> 5, j = (int)(tc & 31u);
            t = j < yx ? t : (1u << 10 | (unsigned)j << 5 | (unsigned)i) - koff;
        }
        const unsigned qraw = HI ? (unsigned)__double2hiint(K) >> (sh_bits - 32)
                                 : __builtin_amdgcn_alignbit((unsigned)__double2hiint(K), (unsigned)__double2loint(K), sh_bits);
        const unsigned key = (qraw << TB) + (unsigned)t;
        v.k[r] = r < SLOTS / 2 ? (key & aklo) | oklo : key | padhi;
    }
}

// rare path (a curve that is not ascending): positions in fill order, the whole network
template <bool CROSSING>
__device__ __forceinline__ void fill_any(const Shared& sh, int lane, Keys& v, int yx, int hmin, int sh_bits) {
    constexpr int TB = CROSSING ? 11 : 10;
    unsigned qmax = (1u << (32 - TB)) - 1u, lowkey = LOWKEY, inv20 = 52429;
    asm volatile("" : "+v"(qmax), "+v"(lowkey), "+v"(inv20));   // a rare path: its constants are made here, not kept in registers for it
    const int nfirst = NY * yx;
    const int inv_yx = (1048576 + yx - 1) / yx;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) {
        const int e = 64 * r + lane - RANK0;   // position 64 r + lane: RANK0 positions of low padding, the sums in fill order, high padding
        unsigned key = e < 0 ? lowkey : HIGHKEY;
        if (e >= 0 && e < N) {
            int i = (int)(__umul24(e, inv20) >> 20), j = e - NY * i;   // e / 20 (exact below 512), e % 20
            bool second = false;
            if (CROSSING) {
                const bool first = e < nfirst;
                const int q = (int)(__umul24(e, first ? (unsigned)inv_yx : inv20) >> 20);   // e / yx, e / 20 (exact below 512)
                const int rem = e - __umul24(q, first ? yx : NY);
                i = first ? q : rem;
                j = first ? rem : q;
                second = !first;
            }
            const double K = sh.tab[i].ov + sh.tab[j].iv;
            const int dh = max(__double2hiint(K) - hmin, 0);
            unsigned q = sh_bits >= 32 ? (unsigned)dh >> (sh_bits - 32)
                                       : __builtin_amdgcn_alignbit((unsigned)dh, (unsigned)__double2loint(K), sh_bits);
            q = min(max(q, 1u), qmax);
            key = q << TB | (second ? (1u << 10 | (unsigned)j << 5 | (unsigned)i) : ((unsigned)i << 5 | (unsigned)j));
        }
        v.k[r] = key;
    }
}

// Behind the network: the exact order, the keys by rank to LDS, the abscissae; returns this lane's Gauss point re-binned.
template <bool CROSSING>
__device__ __forceinline__ double finish_and_rebin(Shared& sh, int lane, Keys& v, double my_mix, Counters& cnt) {
    const bool mine = lane >= LANE0 && lane < LANE0 + NLANES;
    RO_MARK("finish");
    // every slot's cell, decoded once: the byte offsets of its two table entries serve the sums (the exact test below) and
    // the weights (the scan).  Sums and weights are read in two halves each, so that at most eight reads are in flight.
    const char* ob = (const char*)sh.tab;
    const char* ib = (const char*)sh.tab + IP_OFF;
    unsigned ao[SLOTS], ai[SLOTS];
#pragma unroll
    for (int r = 0; r < SLOTS; r++) decode<CROSSING>(v.k[r], ao[r], ai[r]);
    {
        double K[SLOTS];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int r = 4 * h; r < 4 * h + 4; r++) K[r] = *(const double*)(ob + ao[r]) + *(const double*)(ib + ai[r]);   // padding: -inf, inf
            asm volatile("" ::: "memory");
        }
        // exact finish: any inversion left by the quantisation?  (odd-even transposition on the exact sums, strict '>': stable,
        // as random_overlap.h; the keys go along, sums and weights follow from them)
        auto inverted = [&]() {
            bool inv = false;
#pragma unroll
            for (int r = 0; r + 1 < SLOTS; r++) inv = inv || K[r] > K[r + 1];
            const double kn = from_next_lane(K[0]);
            inv = inv || (lane < 63 && K[SLOTS - 1] > kn);
            return __ballot(inv) != 0;
        };
        if (inverted()) {   // wave-uniform, 1-5 % of the problems; the cells' offsets are decoded again behind it: the keys move
            int passes = 0;
            do {
                passes++;
                auto ce = [&](int a, int b) {
                    const bool sw = K[a] > K[b];
                    const double ka = K[a], kb = K[b];
                    const unsigned ea = v.k[a], eb = v.k[b];
                    K[a] = sw ? kb : ka; K[b] = sw ? ka : kb;
                    v.k[a] = sw ? eb : ea; v.k[b] = sw ? ea : eb;
                };
                ce(0, 1); ce(2, 3); ce(4, 5); ce(6, 7);
                ce(1, 2); ce(3, 4); ce(5, 6);
                // across the lanes: every lane looks at its neighbours' values of BEFORE the exchange (DPP: no address registers)
                const double kn0 = from_next_lane(K[0]), kp7 = from_prev_lane(K[SLOTS - 1]);
                const bool sw_hi = lane < 63 && K[SLOTS - 1] > kn0, sw_lo = lane > 0 && kp7 > K[0];
                const unsigned en0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.k[0], 0x130, 0xF, 0xF, false);
                const unsigned ep7 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.k[SLOTS - 1], 0x138, 0xF, 0xF, false);
                if (sw_hi) { K[SLOTS - 1] = kn0; v.k[SLOTS - 1] = en0; }
                if (sw_lo) { K[0] = kp7; v.k[0] = ep7; }
            } while (passes < 2 * ro::LDS_N && inverted());
            cnt.passes += passes;
#pragma unroll
            for (int r = 0; r < SLOTS; r++) decode<CROSSING>(v.k[r], ao[r], ai[r]);
        }
    }
    RO_MARK("scan");
    if (mine) {   // the keys by rank: what the interpolation below reads its two sums from
        uint4* e = (uint4*)(sh.E + SLOTS * (lane - LANE0));
        e[0] = make_uint4(v.k[0], v.k[1], v.k[2], v.k[3]);
        e[1] = make_uint4(v.k[4], v.k[5], v.k[6], v.k[7]);
    }
    // the weights in rank order, from the cells; cumulative mid-point abscissae Y_w = sum_{v<w} g_v + g_w/2 (:3371-3376)
    double g[SLOTS];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int r = 4 * h; r < 4 * h + 4; r++) g[r] = *(const double*)(ob + ao[r] + 8) * *(const double*)(ib + ai[r] + 8);   // padding: 0
        asm volatile("" ::: "memory");
    }
    double csum = 0.0;
#pragma unroll
    for (int r = 0; r < SLOTS; r++) csum += g[r];
    const double run = wave_inclusive_sum(csum) - csum;
    put_abscissae(sh, lane, g, run);
    sync();
    RO_MARK("search");
    double yq;
    const int wq = locate(sh, lane, yq, cnt.skipped);
    RO_MARK("interpolate");
    double out = my_mix;  // w = 400: the walk ran out of sums, the reference leaves the entry as it was
    if (lane < NY && wq < N) {
        const double K0 = cell_sum<CROSSING>(sh, sh.E[wq - 1]), K1 = cell_sum<CROSSING>(sh, sh.E[wq]);
        const double y0 = sh.Y[padded(RANK0 + wq - 1) - YBASE], y1 = sh.Y[padded(RANK0 + wq) - YBASE];
        out = (K0 * (y1 - yq) + K1 * (yq - y0)) / (y1 - y0);
    }
    RO_MARK("end");
    return out;
}

// One problem.  Lanes 0..19 pass the running mix and the new absorber's (already scaled) k-coefficients at their Gauss
// point and receive the mixed value (kernels.cu:3293-3396, ro_method == 1, s > 0, ny == 20).  All 64 lanes must call.
__device__ __forceinline__ double mix(Shared& sh, LaneConst lc, int lane, double my_mix, double my_add, Counters& cnt) {
    RO_MARK("prologue");
    // what depends on the lane alone is derived again in every problem, from these opaque copies: only the seven numbers
    // themselves occupy registers between the problems, not the eight addresses and eight codes that follow from them
    asm volatile("" : "+v"(lane), "+v"(lc.fix), "+v"(lc.var), "+v"(lc.t0), "+v"(lc.tstep));
    // corners of the tableau, wave-uniform (v_readlane: scalar results, no address registers)
    const double m0 = lane_value<0>(my_mix), a0 = lane_value<0>(my_add), m19 = lane_value<NY - 1>(my_mix), a19 = lane_value<NY - 1>(my_add);
    // less than 1 % of the other everywhere: correlated-k (:3297-3310)
    if ((0.01 * m0 > a19) || (0.01 * a0 > m19)) return my_mix + my_add;
    const bool mix_first = m0 > a0;
    sync();  // the previous problem's readers are done with sh
    if (lane < NY) {
        sh.tab[lane].ov = mix_first ? my_mix : my_add;
        sh.tab[lane].iv = mix_first ? my_add : my_mix;
    }
    sync();
    // last crossing of the two curves (:3321-3329); are both k-distributions (ascending)?  Do the rows of the tableau overlap?
    bool cross = false, down = false, over = false, touch = false;
    if (lane >= 1 && lane < NY) {
        const double po = sh.tab[lane - 1].ov, pi = sh.tab[lane - 1].iv;
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
    const bool rows_apart = __ballot(over) == 0;
    // row 11 ends below the start of row 12: the two halves of the run layout are sorted lists that follow each other
    const bool halves_apart = ((__ballot(touch) >> 12) & 1ull) == 0;
    RO_MARK("fill");
    if (monotone && yx == NY && rows_apart) {
        // nothing to sort, to scan or to search: rank w is cell (w / 20, w % 20) (random_overlap.h, prepare_presorted)
        double out = my_mix;
        if (lane < NY) {
            const unsigned long long cells = sh.pre_cells[lane];
            if (cells != 0ull) {
                const char* base = (const char*)&sh;
                const unsigned c0 = (unsigned)cells, c1 = (unsigned)(cells >> 32);
                const double K0 = *(const double*)(base + (c0 & 0xFFFFu)) + *(const double*)(base + (c0 >> 16));
                const double K1 = *(const double*)(base + (c1 & 0xFFFFu)) + *(const double*)(base + (c1 >> 16));
                const double yq = sh.gy[lane], y0 = sh.pre_y0[lane], y1 = sh.pre_y1[lane];
                out = (K0 * (y1 - yq) + K1 * (yq - y0)) / (y1 - y0);
            }
        }
        cnt.skipped += sh.pre_skipped;
        return out;
    }
    double kmin = m0 + a0, kmax = m19 + a19;
    if (!monotone) {  // the extreme sums are not at the corners of the tableau
        double mn1 = sh.tab[0].ov, mx1 = mn1, mn2 = sh.tab[0].iv, mx2 = mn2;
#pragma unroll 1
        for (int j = 1; j < NY; j++) {
            mn1 = fmin(mn1, sh.tab[j].ov); mx1 = fmax(mx1, sh.tab[j].ov);
            mn2 = fmin(mn2, sh.tab[j].iv); mx2 = fmax(mx2, sh.tab[j].iv);
        }
        kmin = mn1 + mn2;
        kmax = mx1 + mx2;
    }
    // key scale (wave-uniform): q = (bits(K) - (hmin << 32)) >> sh with 1 <= q < 2^QB for Kmin <= K <= Kmax (q = 0 is the
    // low padding's), QB = 22 bits without a crossing, 21 with one (random_overlap.h has 23: its tie-break is nine bits)
    const int QB = 32 - (yx == NY ? 10 : 11);
    const int hk = __builtin_amdgcn_readfirstlane(__double2hiint(kmin));
    const unsigned long long span =
        ((unsigned long long)(unsigned)(__builtin_amdgcn_readfirstlane(__double2hiint(kmax)) - hk) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(kmax));
    const int bl0 = span ? 64 - __clzll((long long)span) : 0;
    const int bias = 1 << max(0, bl0 - QB - 30);                       // in units of 2^32: >= 2^(sh + 1) once sh is settled below
    int hmin = hk - bias;
    const unsigned long long dmax = span + ((unsigned long long)(unsigned)bias << 32);
    const int bl = 64 - __clzll((long long)dmax);
    int sh_bits = bl > QB ? bl - QB : 0;
    // The base hmin << 32 is made a multiple of 2^sh, so that a slot's q is (bits(K) >> sh) minus a wave-uniform number and the
    // subtraction moves out of the slots into the lane's cell code (fill_runs).  With sh <= 32 it is one already; beyond, hmin
    // is rounded down to a multiple of 2^(sh - 32) -- every q grows by at most one, which the scale allows for (one more shift
    // where the largest q would touch 2^QB; the bias above keeps the smallest q at one or more either way).
    if (sh_bits > 32) {
        const unsigned long long slack = ((1ull << (sh_bits - 32)) - 1ull) << 32;
        if (64 - __clzll((long long)(dmax + slack)) > bl) sh_bits++;
        hmin &= ~((1 << (sh_bits - 32)) - 1);
    }
    Keys v;
    ro::Lane lnc;   // the network reads the three low lane bits as masks from here
#pragma unroll
    for (int t = 0; t < 3; t++) lnc.c[t] = (unsigned)__builtin_amdgcn_sbfe(lane, t, 1);
    if (monotone) {
        if (yx == NY) {
            if (sh_bits >= 32) fill_runs<false, true>(sh, lc, v, yx, hmin, sh_bits);
            else fill_runs<false, false>(sh, lc, v, yx, hmin, sh_bits);
        } else {
            if (sh_bits >= 32) fill_runs<true, true>(sh, lc, v, yx, hmin, sh_bits);
            else fill_runs<true, false>(sh, lc, v, yx, hmin, sh_bits);
        }
        RO_MARK("network");
        ro::sort512_from_runs16(v, lane, lnc, halves_apart);
    } else {
        if (yx == NY) fill_any<false>(sh, lane, v, yx, hmin, sh_bits);
        else fill_any<true>(sh, lane, v, yx, hmin, sh_bits);
        ro::sort512(v, lane, lnc);
    }
    if (yx == NY) return finish_and_rebin<false>(sh, lane, v, my_mix, cnt);
    return finish_and_rebin<true>(sh, lane, v, my_mix, cnt);
}

}  // namespace rol
