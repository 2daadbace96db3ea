// Context, error reporting and device buffers of libsfm_hip.so.
#include "sfm_internal.h"

#include <cstring>
#include <new>

namespace {
thread_local std::string g_create_error;
}

namespace sfm {

int set_error(sfm_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

static int grow(sfm_ctx* ctx, void** buf, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return SFM_OK;
    // The old buffer may still be in use by work queued on the stream.
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) {
        SFM_HIP_CHECK(ctx, hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    size_t want = align_up(bytes + bytes / 8, size_t(1) << 20);
    hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) {
        *buf = nullptr;
        return set_error(ctx, SFM_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    *cap = want;
    return SFM_OK;
}

static hipEvent_t take_event(sfm_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

KernelTimer::KernelTimer(sfm_ctx* c, int k) : ctx(c), kid(k) {
    if (!ctx->profiling) return;
    start = take_event(ctx);
    stop = take_event(ctx);
    if (start && stop) (void)hipEventRecord(start, ctx->stream);
}

KernelTimer::~KernelTimer() {
    if (!start || !stop) return;
    (void)hipEventRecord(stop, ctx->stream);
    ctx->timed.push_back({kid, start, stop});
}

int workspace(sfm_ctx* ctx, size_t bytes, void** out) {
    SFM_RETURN_IF(grow(ctx, &ctx->ws, &ctx->ws_bytes, bytes));
    *out = ctx->ws;
    return SFM_OK;
}

int staging(sfm_ctx* ctx, size_t bytes, void** out) {
    SFM_RETURN_IF(grow(ctx, &ctx->staging, &ctx->staging_bytes, bytes));
    *out = ctx->staging;
    return SFM_OK;
}

int upload_rows(sfm_ctx* ctx, void* d_dst, const void* src, size_t src_stride, size_t row_bytes,
                size_t rows) {
    if (rows == 0 || row_bytes == 0) return SFM_OK;
    const size_t bytes = rows * row_bytes;
    if (src_stride == row_bytes || rows == 1) {
        SFM_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SFM_OK;
    }
    if (ctx->pinned_used + bytes > ctx->pinned_bytes) {
        // Earlier gathers may still be in flight from the old buffer.
        SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->pinned_used = 0;
        if (bytes > ctx->pinned_bytes) {
            if (ctx->pinned) SFM_HIP_CHECK(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
            const size_t want = align_up(2 * bytes, size_t(1) << 20);
            hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
            if (e != hipSuccess) {
                ctx->pinned = nullptr;
                return set_error(ctx, SFM_E_NOMEM, "hipHostMalloc(%zu) failed: %s", want,
                                 hipGetErrorString(e));
            }
            ctx->pinned_bytes = want;
        }
    }
    char* dst = static_cast<char*>(ctx->pinned) + ctx->pinned_used;
    ctx->pinned_used += align_up(bytes, 256);
    const char* s = static_cast<const char*>(src);
    for (size_t r = 0; r < rows; ++r) memcpy(dst + r * row_bytes, s + r * src_stride, row_bytes);
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, dst, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SFM_OK;
}

}  // namespace sfm

static int create_common(int device, hipStream_t stream, bool own, sfm_ctx** out) {
    if (!out) return sfm::set_error(nullptr, SFM_E_INVALID, "sfm_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return sfm::set_error(nullptr, SFM_E_HIP, "no HIP device available: %s",
                              e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count)
        return sfm::set_error(nullptr, SFM_E_INVALID, "device %d out of range [0,%d)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess)
        return sfm::set_error(nullptr, SFM_E_HIP, "hipSetDevice(%d) failed: %s", device,
                              hipGetErrorString(e));
    sfm_ctx* ctx = new (std::nothrow) sfm_ctx();
    if (!ctx) return sfm::set_error(nullptr, SFM_E_NOMEM, "out of host memory");
    ctx->device = device;
    if (own) {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return sfm::set_error(nullptr, SFM_E_HIP, "hipStreamCreate failed: %s",
                                  hipGetErrorString(e));
        }
        ctx->own_stream = true;
    } else {
        ctx->stream = stream;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->num_cus = prop.multiProcessorCount;
    sfm::triangulate_scratch_setup(device);
    e = hipMalloc(reinterpret_cast<void**>(&ctx->stats), sfm::kStatsBytes);
    if (e == hipSuccess) e = hipMemset(ctx->stats, 0, sfm::kStatsBytes);
    if (e != hipSuccess) {
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return sfm::set_error(nullptr, SFM_E_NOMEM, "hipMalloc of the context's counters failed: %s",
                              hipGetErrorString(e));
    }
    *out = ctx;
    return SFM_OK;
}

extern "C" {

int sfm_ctx_create(int device, sfm_ctx** out) { return create_common(device, nullptr, true, out); }

int sfm_ctx_create_on_stream(int device, void* hip_stream, sfm_ctx** out) {
    return create_common(device, static_cast<hipStream_t>(hip_stream), false, out);
}

void sfm_ctx_destroy(sfm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& t : ctx->timed) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->stats) (void)hipFree(ctx->stats);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->staging) (void)hipFree(ctx->staging);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int sfm_ctx_synchronize(sfm_ctx* ctx) {
    if (!ctx) return sfm::set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

int sfm_ctx_reserve(sfm_ctx* ctx, size_t bytes) {
    if (!ctx) return sfm::set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    void* p;
    return sfm::workspace(ctx, bytes, &p);
}

const char* sfm_last_error(const sfm_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

const char* sfm_version(void) { return "sfm_hip 0.1 gfx950"; }

int sfm_l2_exhaustive_count(sfm_ctx* ctx, int* count) {
    if (!ctx || !count) return sfm::set_error(ctx, SFM_E_INVALID, "NULL argument");
    int32_t v = 0;
    SFM_HIP_CHECK(ctx, hipMemcpyAsync(&v, ctx->stats, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *count = v;
    return SFM_OK;
}

int sfm_ctx_set_profiling(sfm_ctx* ctx, int enabled) {
    if (!ctx) return sfm::set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& t : ctx->timed) {
        ctx->event_pool.push_back(t.start);
        ctx->event_pool.push_back(t.stop);
    }
    ctx->timed.clear();
    ctx->profiling = enabled != 0;
    return SFM_OK;
}

int sfm_profile_kernel_count(void) { return sfm::KID_COUNT; }

const char* sfm_profile_kernel_name(int kernel_id) {
    static const char* names[sfm::KID_COUNT] = {
        "hamming_knn2_kernel", "hamming_pair_kernel", "ratio_cross_kernel", "merge_ratio_kernel", "knn_finalize_kernel", "pack_rows_kernel",
        "ratio_compact_kernel", "l2_candidates_kernel", "l2_rescore_kernel", "l2_pack_kernel",
        "lk_pad_level0_kernel", "lk_pyr_down_kernel", "lk_scharr_kernel", "lk_track_kernel",
        "triangulate_dlt_kernel", "triangulate_list_kernel", "fm_subsets_kernel", "fm_solve_kernel",
        "fm_score_kernel", "fm_select_kernel", "corner_eig_kernel", "corner_pick_kernel",
        "corner_sort", "corner_select_kernel", "em_advance_kernel", "em_solve_kernel", "em_score_kernel",
        "pose_kernels", "hamming_mfma_kernel", "hamming_expand_kernel", "compact_matches_kernels", "l2_exhaustive_kernel"};
    return kernel_id >= 0 && kernel_id < sfm::KID_COUNT ? names[kernel_id] : "";
}

int sfm_ctx_profile_read(sfm_ctx* ctx, int kernel_id, double* total_ms, int* launches) {
    if (!ctx) return sfm::set_error(nullptr, SFM_E_INVALID, "ctx is NULL");
    if (kernel_id < 0 || kernel_id >= sfm::KID_COUNT || !total_ms || !launches)
        return sfm::set_error(ctx, SFM_E_INVALID, "bad profile query");
    SFM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0;
    int n = 0;
    for (auto& t : ctx->timed)
        if (t.kid == kernel_id) {
            float ms = 0.f;
            SFM_HIP_CHECK(ctx, hipEventElapsedTime(&ms, t.start, t.stop));
            sum += ms;
            ++n;
        }
    *total_ms = sum;
    *launches = n;
    return SFM_OK;
}

}  // extern "C"
