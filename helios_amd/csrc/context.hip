// Context, stream and device-memory entry points of the C-ABI (include/helios_hip.h sections 1-2).
// Replaces pycuda.autoinit / gpuarray.to_gpu / .get() / cuda.mem_alloc of the reference
// (source/quantities.py:463-665, source/computation.py:24).
#include "hx_common.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

extern "C" {

int hx_abi_version(void) { return 1; }

int hx_create(int device_id, hx_context** out_ctx) {
    if (!out_ctx) return HX_E_ARG;
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int)e;
    if (device_id < 0 || device_id >= n) return HX_E_ARG;
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return -(int)e;
    hx_context* ctx = new hx_context();
    ctx->device = device_id;
    ctx->err[0] = 0;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return -(int)e;
    }
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    ctx->diag = nullptr;
    e = hipMalloc((void**)&ctx->diag, HX_DIAG_SLOTS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->diag, 0, HX_DIAG_SLOTS * sizeof(unsigned long long), ctx->stream);
    if (e != hipSuccess) {
        (void)hx_destroy(ctx);
        return -(int)e;
    }
    *out_ctx = ctx;
    return 0;
}

int hx_destroy(hx_context* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->diag) (void)hipFree(ctx->diag);
    delete ctx;
    return 0;
}

}  // extern "C"

namespace {
template <bool ABS>
__global__ void __launch_bounds__(256) k_count(const double* __restrict__ a, size_t n, double limit,
                                                unsigned long long* __restrict__ slot) {
    unsigned long long mine = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        mine += ABS ? (fabs(a[k]) >= limit ? 1 : 0) : (a[k] < 0.0 ? 1 : 0);
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(slot, mine);
}
}  // namespace

extern "C" {

int hx_internal_count_negative(hx_context* ctx, const double* a, size_t n, int slot) {
    if (!a || !n) return 0;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    k_count<false><<<grid, 256, 0, ctx->stream>>>(a, n, 0.0, ctx->diag + slot);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_internal_count_abs_ge(hx_context* ctx, const double* a, size_t n, double limit, int slot) {
    if (!a || !n) return 0;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    k_count<true><<<grid, 256, 0, ctx->stream>>>(a, n, limit, ctx->diag + slot);
    HX_LAUNCH_CHECK(ctx);
    return 0;
}

int hx_diag_read(hx_context* ctx, hx_diag* out) {
    static_assert(sizeof(hx_diag) == HX_DIAG_SLOTS * sizeof(unsigned long long), "hx_diag layout");
    if (!ctx || !out) return HX_E_ARG;
    HX_HIP(ctx, hipMemcpyAsync(out, ctx->diag, sizeof(hx_diag), hipMemcpyDeviceToHost, ctx->stream));
    HX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int hx_diag_reset(hx_context* ctx) {
    if (!ctx) return HX_E_ARG;
    HX_HIP(ctx, hipMemsetAsync(ctx->diag, 0, HX_DIAG_SLOTS * sizeof(unsigned long long), ctx->stream));
    return 0;
}

int hx_sync(hx_context* ctx) {
    HX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

const char* hx_last_error(hx_context* ctx) { return ctx ? ctx->err : "null context"; }

int hx_device_name(hx_context* ctx, char* buf, int buflen) {
    hipDeviceProp_t p;
    HX_HIP(ctx, hipGetDeviceProperties(&p, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return 0;
}

void* hx_stream(hx_context* ctx) { return (void*)ctx->stream; }

int hx_timer_start(hx_context* ctx) {
    HX_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

int hx_timer_stop_ms(hx_context* ctx, double* out_ms) {
    HX_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HX_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HX_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *out_ms = ms;
    return 0;
}

int hx_alloc(hx_context* ctx, size_t nbytes, void** out_dptr) {
    HX_HIP(ctx, hipSetDevice(ctx->device));
    if (nbytes == 0) nbytes = 8;
    HX_HIP(ctx, hipMalloc(out_dptr, nbytes));
    return 0;
}

int hx_free(hx_context* ctx, void* dptr) {
    if (!dptr) return 0;
    HX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HX_HIP(ctx, hipFree(dptr));
    return 0;
}

int hx_h2d(hx_context* ctx, void* dptr, const void* hptr, size_t nbytes) {
    // the library never retains host pointers: pageable memory + async copy returns only after
    // the source has been staged, and we additionally wait so that the caller may free at once
    HX_HIP(ctx, hipMemcpyAsync(dptr, hptr, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int hx_d2h(hx_context* ctx, void* hptr, const void* dptr, size_t nbytes) {
    HX_HIP(ctx, hipMemcpyAsync(hptr, dptr, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int hx_d2d(hx_context* ctx, void* dst, const void* src, size_t nbytes) {
    HX_HIP(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int hx_memset0(hx_context* ctx, void* dptr, size_t nbytes) {
    HX_HIP(ctx, hipMemsetAsync(dptr, 0, nbytes, ctx->stream));
    return 0;
}

int hx_mem_info(hx_context* ctx, size_t* out_free, size_t* out_total) {
    HX_HIP(ctx, hipSetDevice(ctx->device));
    HX_HIP(ctx, hipMemGetInfo(out_free, out_total));
    return 0;
}

// ---- host utility (no device call): the rows of the per-bin output tables ---------------------------------------------
// The reference's writers (source/write.py:576-714) put one row per wavelength bin into fourteen text files: the bin's
// geometry, then one cell per level -- 14 million cells at 10 000 bins x 100 layers, two seconds of single-threaded
// number formatting, four times the iteration to equilibrium itself.  printf's conversions are the ones Python's
// `%`-operator performs (both round correctly), so the rows are formatted here by several threads.
static bool cell_format_ok(const char* f) {  // %-<width>[.<precision>](e|g)
    if (!f || f[0] != '%' || f[1] != '-') return false;
    const char* p = f + 2;
    if (!isdigit((unsigned char)*p)) return false;
    // width and precision at most 64: a cell never outgrows the 512-byte buffer it is formatted into
    int width = 0, prec = 0;
    while (isdigit((unsigned char)*p)) { width = std::min(1000, 10 * width + (*p - '0')); p++; }
    if (*p == '.') {
        p++;
        if (!isdigit((unsigned char)*p)) return false;
        while (isdigit((unsigned char)*p)) { prec = std::min(1000, 10 * prec + (*p - '0')); p++; }
    }
    return width <= 64 && prec <= 64 && (*p == 'e' || *p == 'g') && p[1] == 0;
}

int hx_host_format_rows(const double* prefix, const double* values, int nrows, int ncols, const char* cell_format,
                        int nthreads, char** out_text, size_t* out_len) {
    if (!prefix || !values || !out_text || !out_len || nrows < 0 || ncols < 0 || !cell_format_ok(cell_format)) return HX_E_ARG;
    nthreads = std::max(1, std::min(nthreads, std::max(1, nrows / 64)));
    std::vector<std::string> part(nthreads);
    std::vector<char> failed(nthreads, 0);  // an exception (out of memory) must not leave a thread: it would end the process
    auto work = [&](int t) {
      try {
        const int r0 = (int)((long long)nrows * t / nthreads), r1 = (int)((long long)nrows * (t + 1) / nthreads);
        std::string& o = part[t];
        o.reserve((size_t)(r1 - r0) * (72 + (size_t)ncols * 26));
        char buf[512];
        for (int r = r0; r < r1; r++) {
            const double* p = prefix + 4 * (size_t)r;
            o.append(buf, snprintf(buf, sizeof buf, "\n%-8g%-18.9g%-21.9g%-19.9g", p[0], p[1], p[2], p[3]));
            const double* v = values + (size_t)ncols * r;
            for (int c = 0; c < ncols; c++) {
                double x = v[c];
                if (x != x) x = fabs(x);  // printf writes the sign of a NaN ("-nan"), Python does not
                o.append(buf, snprintf(buf, sizeof buf, cell_format, x));
            }
        }
      } catch (...) {
        failed[t] = 1;
      }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    } catch (...) {  // no more threads to be had: the rows of the missing ones are formatted here
        for (int t = (int)th.size() + 1; t < nthreads; t++) work(t);
    }
    work(0);
    for (auto& t : th) t.join();
    for (char f : failed)
        if (f) return HX_E_ARG;
    size_t total = 0;
    for (auto& o : part) total += o.size();
    char* text = (char*)malloc(total ? total : 1);
    if (!text) return HX_E_ARG;
    size_t at = 0;
    for (auto& o : part) {
        memcpy(text + at, o.data(), o.size());
        at += o.size();
    }
    *out_text = text;
    *out_len = total;
    return 0;
}

void hx_host_free(void* p) { free(p); }

}  // extern "C"
