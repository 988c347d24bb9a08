// Device side of the all-to-all training step on row-sharded tables (no counterpart in the reference, which is
// one process, run_model.py:14-16): de-duplication of the items a batch touches, and the owner-side gather of
// the rows other GPUs ask for.  Everything is enqueued on the context's stream; nothing is read back.
#include "common.hpp"

namespace {

constexpr int DD_ITEMS = 16, DD_TILE = 256 * DD_ITEMS;      // items per workgroup in the rank kernels

__global__ __launch_bounds__(256) void dd_mark_kernel(const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B,
                                                      uint32_t I, uint32_t *__restrict__ mark, uint32_t *__restrict__ flags)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * B) return;
    const int32_t x = t < B ? pos[t] : neg[t - B];
    const bool bad = x < 0 || (uint32_t)x >= I;
    if (bad) atomicOr(flags, TFREC_FLAG_BAD_INDEX);
    mark[bad ? (x < 0 ? 0u : I - 1) : (uint32_t)x] = 1u;     // (the same value from whoever stores it)
}

__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *red /* [4] */)
{
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}

// marked items per tile
__global__ __launch_bounds__(256) void dd_count_kernel(const uint32_t *__restrict__ mark, uint32_t I, uint32_t *__restrict__ tile_sum)
{
    __shared__ uint32_t red[4];
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < DD_ITEMS; ++r) {
        const uint32_t i = blockIdx.x * (uint32_t)DD_TILE + r * 256 + threadIdx.x;
        c += i < I ? mark[i] : 0u;
    }
    const uint32_t s = block_sum(c, red);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s;
}

// exclusive prefix over the tile sums, in place; tile_sum[n_tiles] = total.  One workgroup.
__global__ __launch_bounds__(256) void dd_scan_tiles_kernel(uint32_t *__restrict__ tile_sum, uint32_t n_tiles)
{
    __shared__ uint32_t red[4];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 256) {
        const uint32_t i = t0 + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sum[i] : 0u;
        // inclusive scan over the 256 lanes: within the wave, then across the four waves
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if ((int)(threadIdx.x & 63) >= o) inc += up;
        }
        if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += red[w];
        if (i < n_tiles) tile_sum[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[n_tiles] = carry;
}

// mark (0/1) becomes the exclusive rank of every item among the marked ones; the marked ids are written, ascending,
// to uniq[rank]
__global__ __launch_bounds__(256) void dd_rank_kernel(uint32_t *__restrict__ mark, uint32_t I, const uint32_t *__restrict__ tile_prefix,
                                                      int32_t *__restrict__ uniq)
{
    __shared__ uint32_t wave_tot[4];
    // thread t owns DD_ITEMS consecutive items, so that a thread's running count is a rank
    const uint32_t i0 = blockIdx.x * (uint32_t)DD_TILE + threadIdx.x * DD_ITEMS;
    uint32_t m[DD_ITEMS], c = 0;
#pragma unroll
    for (int r = 0; r < DD_ITEMS; ++r) {
        m[r] = i0 + r < I ? mark[i0 + r] : 0u;
        c += m[r];
    }
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if ((int)(threadIdx.x & 63) >= o) inc += up;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = tile_prefix[blockIdx.x] + inc - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wave_tot[w];
#pragma unroll
    for (int r = 0; r < DD_ITEMS; ++r) {
        if (i0 + r < I) {
            mark[i0 + r] = run;
            if (m[r]) uniq[run] = (int32_t)(i0 + r);
            run += m[r];
        }
    }
}

__global__ __launch_bounds__(256) void dd_slots_kernel(const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B, uint32_t I,
                                                       const uint32_t *__restrict__ rank, int32_t *__restrict__ pos_slot,
                                                       int32_t *__restrict__ neg_slot)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * B) return;
    const int32_t x = t < B ? pos[t] : neg[t - B];
    const uint32_t id = x < 0 ? 0u : ((uint32_t)x >= I ? I - 1 : (uint32_t)x);
    (t < B ? pos_slot : neg_slot)[t < B ? t : t - B] = (int32_t)rank[id];
}

// distinct items inside every owner's range [r * per, (r + 1) * per)
__global__ void dd_counts_kernel(const uint32_t *__restrict__ rank, const uint32_t *__restrict__ total, uint32_t I, int64_t per, int W,
                                 int64_t *__restrict__ counts, int64_t *__restrict__ n_uniq)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    auto below = [&](int64_t x) -> int64_t { return x >= (int64_t)I ? (int64_t)*total : (int64_t)rank[x]; };
    if (r < W) counts[r] = below((int64_t)(r + 1) * per) - below((int64_t)r * per);
    if (r == 0 && n_uniq) *n_uniq = (int64_t)*total;
}

template <int VEC>
__global__ __launch_bounds__(256) void rows_gather_kernel(const float *__restrict__ Q, const float *__restrict__ b, uint32_t I, int d,
                                                          const int32_t *__restrict__ ids, int64_t n, float *__restrict__ out_rows,
                                                          float *__restrict__ out_b, uint32_t *__restrict__ flags)
{
    const int cpr = d / VEC;                                  // chunks per row
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * cpr) return;
    const int64_t r = e / cpr;
    const int c = (int)(e - r * cpr);
    int32_t x = ids[r];
    if (x < 0 || (uint32_t)x >= I) { atomicOr(flags, TFREC_FLAG_BAD_INDEX); x = x < 0 ? 0 : (int32_t)(I - 1); }
    if constexpr (VEC == 4) reinterpret_cast<float4 *>(out_rows)[e] = reinterpret_cast<const float4 *>(Q + (size_t)x * d)[c];
    else out_rows[e] = Q[(size_t)x * d + c];
    if (c == 0) out_b[r] = b[x];
}

}  // namespace

extern "C" int tfrec_items_dedup(tfrec_ctx *ctx, const int32_t *d_pos, const int32_t *d_neg, int64_t B, int64_t I, int64_t items_per_owner,
                                 int n_owners, int32_t *d_uniq, int32_t *d_pos_slot, int32_t *d_neg_slot, int64_t *d_counts,
                                 int64_t *d_n_uniq)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(I > 0 && I < (1ll << 31) && B >= 0 && B < (1ll << 30), TFREC_EINVAL, "bad shape I=%lld B=%lld", (long long)I, (long long)B);
    TF_REQUIRE(n_owners >= 1 && items_per_owner >= 1 && d_counts != nullptr, TFREC_EINVAL, "bad owner layout");
    TF_REQUIRE(B == 0 || (d_pos && d_neg && d_uniq && d_pos_slot && d_neg_slot), TFREC_EINVAL, "null array");
    const uint32_t n_tiles = (uint32_t)((I + DD_TILE - 1) / DD_TILE);
    const size_t mark_bytes = tf_align(sizeof(uint32_t) * (size_t)I), tile_bytes = tf_align(sizeof(uint32_t) * ((size_t)n_tiles + 1));
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + mark_bytes + tile_bytes + 1024));
    uint32_t *mark = (uint32_t *)tfrec_ws_take(ctx, mark_bytes), *tile_sum = (uint32_t *)tfrec_ws_take(ctx, tile_bytes);
    TF_REQUIRE(mark && tile_sum, TFREC_ENOMEM, "internal: workspace carve failed");
    TF_HIP(hipMemsetAsync(mark, 0, sizeof(uint32_t) * (size_t)I, ctx->stream));
    const unsigned pair_blocks = (unsigned)((2 * B + 255) / 256);
    if (B > 0) dd_mark_kernel<<<pair_blocks, 256, 0, ctx->stream>>>(d_pos, d_neg, B, (uint32_t)I, mark, ctx->d_flags);
    dd_count_kernel<<<n_tiles, 256, 0, ctx->stream>>>(mark, (uint32_t)I, tile_sum);
    dd_scan_tiles_kernel<<<1, 256, 0, ctx->stream>>>(tile_sum, n_tiles);
    dd_rank_kernel<<<n_tiles, 256, 0, ctx->stream>>>(mark, (uint32_t)I, tile_sum, d_uniq);
    if (B > 0) dd_slots_kernel<<<pair_blocks, 256, 0, ctx->stream>>>(d_pos, d_neg, B, (uint32_t)I, mark, d_pos_slot, d_neg_slot);
    dd_counts_kernel<<<(n_owners + 63) / 64, 64, 0, ctx->stream>>>(mark, tile_sum + n_tiles, (uint32_t)I, items_per_owner, n_owners, d_counts, d_n_uniq);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_rows_gather(tfrec_ctx *ctx, const float *d_Q, const float *d_b, int64_t I, int d, const int32_t *d_ids, int64_t n,
                                 float *d_out_rows, float *d_out_b)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(I > 0 && I < (1ll << 31) && d > 0 && n >= 0, TFREC_EINVAL, "bad shape");
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_Q && d_b && d_ids && d_out_rows && d_out_b, TFREC_EINVAL, "null array");
    const bool vec = d % 4 == 0 && (((uintptr_t)d_Q | (uintptr_t)d_out_rows) & 15u) == 0;
    const int64_t chunks = n * (vec ? d / 4 : d);
    TF_REQUIRE((chunks + 255) / 256 < (1ll << 31), TFREC_EUNSUP, "too many rows for one launch");
    if (vec) rows_gather_kernel<4><<<(unsigned)((chunks + 255) / 256), 256, 0, ctx->stream>>>(d_Q, d_b, (uint32_t)I, d, d_ids, n, d_out_rows, d_out_b, ctx->d_flags);
    else rows_gather_kernel<1><<<(unsigned)((chunks + 255) / 256), 256, 0, ctx->stream>>>(d_Q, d_b, (uint32_t)I, d, d_ids, n, d_out_rows, d_out_b, ctx->d_flags);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}
