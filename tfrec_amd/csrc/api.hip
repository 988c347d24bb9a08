// Context, error reporting, device-memory helpers and parameter initialisation.
#include "common.hpp"

static thread_local char g_err[512] = "";

void tfrec_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *tfrec_last_error(void) { return g_err; }
extern "C" int tfrec_abi_version(void) { return TFREC_ABI_VERSION; }

int tfrec_ctx_enter(tfrec_ctx *ctx)
{
    TF_REQUIRE(ctx != nullptr, TFREC_EINVAL, "null context");
    TF_HIP(hipSetDevice(ctx->device));
    ctx->ws_used = 0;
    return TFREC_OK;
}

int tfrec_ws_require(tfrec_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return TFREC_OK;
    TF_REQUIRE(!ctx->ws_fixed, TFREC_ENOMEM, "the fixed scratch block holds %zu bytes, %zu needed", ctx->ws_bytes, bytes);
    // grow: wait for work that may still use the old block, then replace it
    TF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->ws) TF_HIP(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    const size_t want = tf_align(bytes + bytes / 8, 1 << 20);
    hipError_t e = hipMalloc((void **)&ctx->ws, want);
    if (e != hipSuccess) {
        tfrec_set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return TFREC_ENOMEM;
    }
    ctx->ws_bytes = want;
    return TFREC_OK;
}

void *tfrec_ws_take(tfrec_ctx *ctx, size_t bytes)
{
    const size_t off = tf_align(ctx->ws_used);
    if (off + bytes > ctx->ws_bytes) return nullptr;   // callers size with tfrec_ws_require first
    ctx->ws_used = off + bytes;
    return ctx->ws + off;
}

int tfrec_pinned_require(tfrec_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->h_pinned_bytes) return TFREC_OK;
    void *old = ctx->h_pinned;
    ctx->h_pinned = nullptr;          // whatever hipHostFree says, the old block is not used again
    ctx->h_pinned_bytes = 0;
    ctx->pinned_tab_k = 0;
    if (old) TF_HIP(hipHostFree(old));
    const size_t want = tf_align(bytes, 4096);
    TF_HIP(hipHostMalloc(&ctx->h_pinned, want, hipHostMallocDefault));
    ctx->h_pinned_bytes = want;
    return TFREC_OK;
}

extern "C" int tfrec_ctx_create(int device, void *hip_stream, tfrec_ctx **out)
{
    TF_REQUIRE(out != nullptr, TFREC_EINVAL, "out is null");
    int ndev = 0;
    TF_HIP(hipGetDeviceCount(&ndev));
    TF_REQUIRE(device >= 0 && device < ndev, TFREC_EINVAL, "device %d not in [0,%d)", device, ndev);
    TF_HIP(hipSetDevice(device));
    tfrec_ctx *c = new (std::nothrow) tfrec_ctx();
    TF_REQUIRE(c != nullptr, TFREC_ENOMEM, "host allocation failed");
    c->device = device;
    // a failure below releases whatever the partial context already owns
    auto fail = [&](int code, const char *what, hipError_t e) {
        tfrec_set_error("%s: %s", what, hipGetErrorString(e));
        if (c->d_flags) (void)hipFree(c->d_flags);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return code;
    };
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) return fail(TFREC_EHIP, "hipStreamCreate", e);
        c->own_stream = true;
    }
    hipError_t e = hipMalloc((void **)&c->d_flags, 256);
    if (e != hipSuccess) { c->d_flags = nullptr; return fail(TFREC_ENOMEM, "hipMalloc(flags)", e); }
    e = hipMemsetAsync(c->d_flags, 0, 256, c->stream);
    if (e != hipSuccess) return fail(TFREC_EHIP, "hipMemset(flags)", e);
    // (a failure to find the fixed scratch array only disables the no-allocation step, which then reports it)
    (void)tfrec_scratch_bind(device);
    *out = c;
    return TFREC_OK;
}

extern "C" int tfrec_ctx_destroy(tfrec_ctx *ctx)
{
    if (!ctx) return TFREC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    for (hipEvent_t e : ctx->marks)
        if (e) (void)hipEventDestroy(e);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->fork_event) (void)hipEventDestroy(ctx->fork_event);
    for (tfrec_batch_slot &sl : ctx->slots) {
        if (sl.mem) (void)hipFree(sl.mem);
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.released) (void)hipEventDestroy(sl.released);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TFREC_OK;
}

extern "C" int tfrec_ctx_set_stream(tfrec_ctx *ctx, void *hip_stream)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(hip_stream != nullptr, TFREC_EINVAL, "stream is null");
    TF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) TF_HIP(hipStreamDestroy(ctx->stream));
    ctx->own_stream = false;
    ctx->stream = (hipStream_t)hip_stream;
    return TFREC_OK;
}

extern "C" void *tfrec_ctx_stream(tfrec_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int tfrec_ctx_error_flags(tfrec_ctx *ctx, uint32_t *h_flags, int clear)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(h_flags != nullptr, TFREC_EINVAL, "h_flags is null");
    TF_HIP(hipMemcpyAsync(h_flags, ctx->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    TF_HIP(hipStreamSynchronize(ctx->stream));
    if (clear && *h_flags) TF_HIP(hipMemsetAsync(ctx->d_flags, 0, sizeof(uint32_t), ctx->stream));
    return TFREC_OK;
}

extern "C" int tfrec_ctx_synchronize(tfrec_ctx *ctx)
{
    uint32_t flags = 0;
    TF_TRY(tfrec_ctx_error_flags(ctx, &flags, 1));
    if (flags & TFREC_FLAG_BAD_INDEX) {
        tfrec_set_error("a user or item index in a batch was out of range (clamped on the device)");
        return TFREC_EINDEX;
    }
    return TFREC_OK;
}

void tfrec_marks_reset(tfrec_ctx *ctx) { ctx->n_marks = 0; }

void tfrec_mark(tfrec_ctx *ctx, const char *name)
{
#ifdef TF_DEBUG_SYNC
    // developer builds: wait for every phase and say so, to find the one that never ends
    TF_DBG("waiting for %s", name);
    const hipError_t dbg = hipStreamSynchronize(ctx->stream);
    TF_DBG("%s: %s", name, hipGetErrorString(dbg));
#endif
    if (!ctx->profile || ctx->n_marks >= tfrec_ctx::MAX_MARKS) return;
    const int i = ctx->n_marks;
    if (!ctx->marks[i] && hipEventCreate(&ctx->marks[i]) != hipSuccess) return;
    if (hipEventRecord(ctx->marks[i], ctx->stream) != hipSuccess) return;
    ctx->mark_names[i] = name;
    ctx->n_marks = i + 1;
}

extern "C" int tfrec_ctx_set_profiling(tfrec_ctx *ctx, int enable)
{
    TF_REQUIRE(ctx != nullptr, TFREC_EINVAL, "null context");
    ctx->profile = enable != 0;
    ctx->n_marks = 0;
    return TFREC_OK;
}

extern "C" int tfrec_ctx_phase_times(tfrec_ctx *ctx, float *h_ms, int max_phases, int *n_phases)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(h_ms && n_phases, TFREC_EINVAL, "null output");
    int n = ctx->n_marks > 0 ? ctx->n_marks - 1 : 0;
    if (n > max_phases) n = max_phases;
    if (ctx->n_marks > 0) TF_HIP(hipEventSynchronize(ctx->marks[ctx->n_marks - 1]));
    for (int i = 0; i < n; ++i) TF_HIP(hipEventElapsedTime(&h_ms[i], ctx->marks[i], ctx->marks[i + 1]));
    *n_phases = n;
    return TFREC_OK;
}

extern "C" const char *tfrec_ctx_phase_name(tfrec_ctx *ctx, int i)
{
    if (!ctx || i < 0 || i + 1 >= ctx->n_marks) return "";
    return ctx->mark_names[i + 1];
}

size_t tfrec_step_need_bytes(int64_t B, int d);                                        // train.hip
size_t tfrec_eval_scratch_bytes(int64_t n, int k);                                     // eval.hip

extern "C" int tfrec_ctx_reserve(tfrec_ctx *ctx, int64_t max_batch, int d, int64_t max_eval_users, int k)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(max_batch >= 0 && max_eval_users >= 0 && d >= 0 && k >= 0, TFREC_EINVAL, "negative size");
    size_t need = 0;
    if (max_batch > 0 && d > 0) need = tfrec_step_need_bytes(max_batch, d);
    if (max_eval_users > 0 && k > 0) {
        const size_t e = tfrec_eval_scratch_bytes(max_eval_users, k);
        need = e > need ? e : need;
    }
    return tfrec_ws_require(ctx, need);
}

extern "C" int64_t tfrec_ctx_workspace_bytes(tfrec_ctx *ctx) { return ctx ? (int64_t)ctx->ws_bytes : 0; }

extern "C" int tfrec_dev_malloc(tfrec_ctx *ctx, size_t bytes, void **d_ptr)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(d_ptr != nullptr, TFREC_EINVAL, "d_ptr is null");
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) { tfrec_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return TFREC_ENOMEM; }
    return TFREC_OK;
}

extern "C" int tfrec_dev_free(tfrec_ctx *ctx, void *d_ptr)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_HIP(hipStreamSynchronize(ctx->stream));
    if (d_ptr) TF_HIP(hipFree(d_ptr));
    return TFREC_OK;
}

extern "C" int tfrec_memcpy_h2d(tfrec_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    if (!bytes) return TFREC_OK;
    TF_REQUIRE(d_dst && h_src, TFREC_EINVAL, "null pointer");
    TF_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    TF_HIP(hipStreamSynchronize(ctx->stream));
    return TFREC_OK;
}

extern "C" int tfrec_memcpy_d2h(tfrec_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    if (!bytes) return TFREC_OK;
    TF_REQUIRE(h_dst && d_src, TFREC_EINVAL, "null pointer");
    TF_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TF_HIP(hipStreamSynchronize(ctx->stream));
    return TFREC_OK;
}

extern "C" int tfrec_memset_zero(tfrec_ctx *ctx, void *d_dst, size_t bytes)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    if (!bytes) return TFREC_OK;
    TF_REQUIRE(d_dst != nullptr, TFREC_EINVAL, "null pointer");
    TF_HIP(hipMemsetAsync(d_dst, 0, bytes, ctx->stream));
    return TFREC_OK;
}

// ------------------------------------------------------------------------------------------
// parameter initialisation: counter-based (SplitMix64 of seed and element index), so the
// n-th element does not depend on the launch geometry.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

__global__ void fill_uniform_kernel(float *x, int64_t n, float lo, float hi, uint64_t seed, uint64_t first)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix64(seed * 0x100000001b3ull + first + (uint64_t)i);
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0,1)
        x[i] = fmaf(u, hi - lo, lo);
    }
}

__global__ void fill_const_kernel(float *x, int64_t n, float v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = v;
}

static int fill_grid(int64_t n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }

extern "C" int tfrec_fill_uniform(tfrec_ctx *ctx, float *d_x, int64_t n, float lo, float hi, uint64_t seed)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && (d_x || n == 0), TFREC_EINVAL, "bad buffer");
    if (n == 0) return TFREC_OK;
    fill_uniform_kernel<<<fill_grid(n), 256, 0, ctx->stream>>>(d_x, n, lo, hi, seed, 0);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_fill_uniform_at(tfrec_ctx *ctx, float *d_x, int64_t n, float lo, float hi, uint64_t seed, int64_t first_index)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && first_index >= 0 && (d_x || n == 0), TFREC_EINVAL, "bad buffer");
    if (n == 0) return TFREC_OK;
    fill_uniform_kernel<<<fill_grid(n), 256, 0, ctx->stream>>>(d_x, n, lo, hi, seed, (uint64_t)first_index);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_fill_const(tfrec_ctx *ctx, float *d_x, int64_t n, float value)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && (d_x || n == 0), TFREC_EINVAL, "bad buffer");
    if (n == 0) return TFREC_OK;
    fill_const_kernel<<<fill_grid(n), 256, 0, ctx->stream>>>(d_x, n, value);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}
