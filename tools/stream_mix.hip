// Ceiling probe for k_rt_flux's traffic mix on one MI355X: streams NR planes in and NW planes out with the access
// pattern of the flux kernel (one wavefront per tile of ROWS x 64 doubles per plane, rows of 512 contiguous bytes,
// all loads of a tile in flight before the first use; a wavefront walks 5 consecutive tiles like the 5 Gauss-point
// groups of a bin).  Options: occupancy limited through the LDS allocation (the flux kernel runs 2 wavefronts per
// SIMD), a dependent fp64 chain between loads and stores that stands for the sweeps, and a prefetch of the next tile
// (one dword per 128-byte line into a register nobody reads).  Standalone:
//   hipcc -O3 --offload-arch=gfx950 tools/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 13;
constexpr int PASSES = 5;

template <int NR, int NW, bool PREFETCH, bool NT = false, bool STRIDED = false>
__global__ void __launch_bounds__(64)
k_stream(const double* __restrict__ in, double* __restrict__ out, size_t plane_elems, int spin) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    int sink = 0;
    for (int pass = 0; pass < PASSES; pass++) {
        const size_t t = STRIDED ? (size_t)pass * gridDim.x + blockIdx.x : (size_t)blockIdx.x * PASSES + pass;
        double v[NR][ROWS];
#pragma unroll
        for (int p = 0; p < NR; p++)
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const double* q = in + p * plane_elems + (t * ROWS + r) * 64 + lane;
                v[p][r] = NT ? __builtin_nontemporal_load(q) : *q;
            }
        double acc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            acc[r] = 0.0;
#pragma unroll
            for (int p = 0; p < NR; p++) acc[r] += v[p][r];
        }
        if (PREFETCH && pass + 1 < PASSES) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) asm volatile("" ::"v"(acc[r]));
            const int bytes = ROWS * 64 * 8;
            const int off = min(lane * 128, bytes - 128);
#pragma unroll
            for (int p = 0; p < NR; p++) {
                const double* nxt = in + p * plane_elems + (t + 1) * ROWS * 64;
                asm volatile("global_load_dword %0, %1, %2" : "+v"(sink) : "v"(off), "s"(nxt) : "memory");
            }
        }
        // stand-in for the sweeps: a dependent fp64 chain per row set, `spin` rounds
        for (int s = 0; s < spin; s++) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) acc[r] = fma(acc[r], 1.0000001, 1e-9);
            acc[0] += __shfl_xor(acc[ROWS - 1], 1);
        }
        if (NW > 0) {
#pragma unroll
            for (int p = 0; p < NW; p++)
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    double* q = out + p * plane_elems + (t * ROWS + r) * 64 + lane;
                    if (NT) __builtin_nontemporal_store(acc[r] + p, q); else *q = acc[r] + p;
                }
        } else {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) s += acc[r];
            if (s == 12345.678) out[t] = s;  // keeps the loads alive, never true for the zero-filled input
        }
    }
    if (PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::"v"(sink));
    if (lds[lane] == 1.5) out[0] = 1.0;  // keeps the LDS allocation
}

template <int NR, int NW, bool PF, bool NT = false, bool ST = false>
void run(const char* what, const double* in, double* out, size_t ntiles, size_t plane_elems, int lds_bytes, int spin) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_stream<NR, NW, PF, NT, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int grid = (int)(ntiles / PASSES);
    for (int i = 0; i < 3; i++) k_stream<NR, NW, PF, NT, ST><<<grid, 64, lds_bytes>>>(in, out, plane_elems, spin);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) k_stream<NR, NW, PF, NT, ST><<<grid, 64, lds_bytes>>>(in, out, plane_elems, spin);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)(NR + NW) * plane_elems * 8.0;
    printf("%-24s waves/SIMD %d  chain %4d  prefetch %d  nontemporal %d  tile order %s  %7.3f ms  %7.1f GB/s\n", what,
           160 * 1024 / lds_bytes / 4, spin, PF ? 1 : 0, NT ? 1 : 0, ST ? "interleaved" : "5 in a row ", ms, bytes / ms * 1e-6);
}

int main() {
    // the C2 plane: 10 000 bins x 5 tiles x 13 rows x 64 lanes (332.8 MB)
    const size_t ntiles = 50000, plane_elems = ntiles * ROWS * 64;
    double *in, *out;
    CK(hipMalloc(&in, 4 * plane_elems * 8));
    CK(hipMalloc(&out, 1 * plane_elems * 8));
    CK(hipMemset(in, 0, 4 * plane_elems * 8));
    CK(hipMemset(out, 0, 1 * plane_elems * 8));
    const int W8 = 5 * 1024, W2 = 20 * 1024, W1 = 40 * 1024;  // LDS per single-wavefront workgroup -> wavefronts per SIMD
    run<1, 1, false>("copy 1 in : 1 out", in, out, ntiles, plane_elems, W8, 0);
    run<1, 1, false, true>("copy 1 in : 1 out", in, out, ntiles, plane_elems, W8, 0);
    run<4, 0, false>("read 4 in : 0 out", in, out, ntiles, plane_elems, W8, 0);
    run<4, 0, false, true>("read 4 in : 0 out", in, out, ntiles, plane_elems, W8, 0);
    for (int spin : {0, 100}) {
        run<4, 1, false, false, false>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W2, spin);
        run<4, 1, false, true, false>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W2, spin);
        run<4, 1, false, false, true>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W2, spin);
        run<4, 1, false, true, true>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W2, spin);
        run<4, 1, true, true, false>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W2, spin);
    }
    run<4, 1, false, true, true>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W1, 100);
    run<4, 1, false, true, true>("flux mix 4 in : 1 out", in, out, ntiles, plane_elems, W8, 100);
    return 0;
}
