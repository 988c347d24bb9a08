// Device-resident epoch pipeline: negative sampling with exclusion, shuffle, gather.
//
// The reference builds every epoch on the host (model/BPR.py:50-62): per user,
// utils/tools.py:39-49 `random_choice(all_items, size=len(pos), exclusion=pos)` draws one negative
// per positive uniformly (with replacement) from the items the user has NOT interacted with, then
// data/dataiterator.py:52-54 shuffles the triples.  That costs O(I) per user on one core.  Here
// the same distribution is drawn on the GPU, one thread per positive, by rejection against the
// user's sorted CSR row.  numpy's Mersenne-Twister stream cannot be reproduced in parallel, so the
// random numbers are a counter-based hash: draw (seed, stream, index, attempt) is a pure function,
// the result does not depend on launch geometry, and the CPU oracle restates it bit for bit.
#include "common.hpp"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// draw number `attempt` for element `index` of stream (seed, stream)
__device__ __forceinline__ uint64_t draw64(uint64_t seed, uint64_t stream, uint64_t index, uint32_t attempt)
{
    return mix64(mix64(seed ^ (stream * 0xd1342543de82ef95ull)) + index * 0x2545f4914f6cdd1dull + (uint64_t)attempt * 0x9e3779b97f4a7c15ull);
}

__device__ __forceinline__ bool row_contains(const int32_t *__restrict__ idx, int64_t lo, int64_t hi, int32_t id)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = idx[mid];
        if (v == id) return true;
        if (v < id) lo = mid + 1; else hi = mid;
    }
    return false;
}

constexpr int SAMPLE_TRIES = 64;

// One thread per stored interaction t (CSR order): its user, its positive item, and C negatives -- into three arrays,
// or (packed_out, C = 1) as one 16-byte record per interaction, which a shuffle then moves with one read instead of three.
// The user of t is found by a binary search confined to the rows the workgroup's 256 interactions
// span (two whole-array searches per workgroup, a few steps per thread).
__global__ __launch_bounds__(256) void sample_negatives_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                               int64_t U, uint32_t I, int64_t n, int C, uint64_t seed, uint64_t stream,
                                                               uint64_t first, int32_t *__restrict__ users_out, int32_t *__restrict__ pos_out,
                                                               int32_t *__restrict__ neg_out, int4 *__restrict__ packed_out,
                                                               uint32_t *__restrict__ flags)
{
    __shared__ int64_t row_range[2];
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    if (threadIdx.x < 2) {
        const int64_t key = threadIdx.x == 0 ? t0 : (t0 + 255 < n ? t0 + 255 : n - 1);
        int64_t lo = 0, hi = U;                       // last row u with indptr[u] <= key
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (indptr[mid] <= key) lo = mid; else hi = mid - 1;
        }
        row_range[threadIdx.x] = lo;
    }
    __syncthreads();
    const int64_t t = t0 + threadIdx.x;
    if (t >= n) return;
    int64_t lo = row_range[0], hi = row_range[1];
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (indptr[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int64_t u = lo, r0 = indptr[u], r1 = indptr[u + 1];
    if (users_out) users_out[t] = (int32_t)u;
    if (pos_out) pos_out[t] = indices[t];
    const bool full = (uint64_t)(r1 - r0) >= (uint64_t)I;      // no item left to draw: reported, never loops
    if (full) atomicOr(flags, TFREC_FLAG_BAD_INDEX);
    for (int c = 0; c < C; ++c) {
        const uint64_t slot = (uint64_t)t * (uint64_t)C + (uint64_t)c;        // where the draw is written
        const uint64_t index = (first + (uint64_t)t) * (uint64_t)C + (uint64_t)c;  // which draw it is
        uint32_t cand = 0;
        if (!full) {
            bool ok = false;
            for (uint32_t a = 0; a < (uint32_t)SAMPLE_TRIES && !ok; ++a) {
                cand = (uint32_t)__umul64hi(draw64(seed, stream, index, a), (uint64_t)I);
                ok = !row_contains(indices, r0, r1, (int32_t)cand);
            }
            // a user who holds almost every item: walk upwards from the last draw to the next free item
            while (!ok) {
                cand = cand + 1 == I ? 0u : cand + 1;
                ok = !row_contains(indices, r0, r1, (int32_t)cand);
            }
        }
        if (packed_out) packed_out[t] = make_int4((int32_t)u, indices[t], (int32_t)cand, 0);      // (one negative per positive)
        else neg_out[slot] = (int32_t)cand;
    }
}

// The shuffle orders the elements by a 64-bit hash key, equal keys in storage order.  (With 32-bit keys alone about
// n*n / 2^33 pairs would tie -- a million at 10^8 elements -- and keep their storage order; 64-bit keys tie about
// n*n / 2^65 times.)  One stable sort by the high word does almost all of it: what is left are the few runs of equal
// high words, which a second kernel orders by the low word (then by element, the order the stable sort left them in).
__device__ __forceinline__ uint64_t shuffle_key(uint64_t seed, uint64_t stream, uint64_t t)
{
    return draw64(seed, stream ^ 0x73687566666c65ull, t, 0);
}

__global__ __launch_bounds__(256) void shuffle_keys_hi_kernel(int64_t n, uint64_t seed, uint64_t stream, uint32_t *__restrict__ keys,
                                                              uint32_t *__restrict__ vals)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    keys[t] = (uint32_t)(shuffle_key(seed, stream, (uint64_t)t) >> 32);
    vals[t] = (uint32_t)t;
}

// hi: the sorted high words; perm: the elements in that order.  The thread at the first position of a run of equal
// high words orders the run (insertion sort: runs are two or three elements long) by (low word, element).
__global__ __launch_bounds__(256) void shuffle_fix_ties_kernel(int64_t n, uint64_t seed, uint64_t stream, const uint32_t *__restrict__ hi,
                                                               uint32_t *__restrict__ perm)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= n) return;
    const uint32_t k = hi[i];
    if (hi[i + 1] != k || (i > 0 && hi[i - 1] == k)) return;
    int64_t end = i + 2;
    while (end < n && hi[end] == k) ++end;
    for (int64_t a = i + 1; a < end; ++a) {
        const uint32_t e = perm[a];
        const uint32_t lo = (uint32_t)shuffle_key(seed, stream, (uint64_t)e);
        int64_t b = a;
        while (b > i) {
            const uint32_t f = perm[b - 1];
            const uint32_t flo = (uint32_t)shuffle_key(seed, stream, (uint64_t)f);
            if (flo < lo || (flo == lo && f < e)) break;
            perm[b] = f;
            --b;
        }
        perm[b] = e;
    }
}

__global__ __launch_bounds__(256) void gather_rows_i32_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ perm, int64_t n,
                                                              int width, int64_t n_src, int32_t *__restrict__ dst, uint32_t *__restrict__ flags)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * width) return;
    const int64_t r = e / width;
    const int w = (int)(e - r * width);
    int64_t s = perm[r];
    if (s < 0 || s >= n_src) { atomicOr(flags, TFREC_FLAG_BAD_INDEX); s = 0; }
    dst[e] = src[s * width + w];
}

// users, pos, neg [r] = the record of interaction perm[r]
__global__ __launch_bounds__(256) void gather_unpack3_kernel(const int4 *__restrict__ packed, const int32_t *__restrict__ perm, int64_t n,
                                                             int32_t *__restrict__ users_out, int32_t *__restrict__ pos_out,
                                                             int32_t *__restrict__ neg_out)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int4 v = packed[perm[r]];
    users_out[r] = v.x;
    pos_out[r] = v.y;
    neg_out[r] = v.z;
}

size_t shuffle_ws_bytes(int64_t n) { return 3 * tf_align(4 * (size_t)n) + tf_align(tfrec_sort_temp_bytes((size_t)n, 32)) + 1024; }

// (the caller has entered the context and made sure of shuffle_ws_bytes(n) more bytes of workspace)
int shuffle_perm_impl(tfrec_ctx *ctx, int64_t n, uint64_t seed, uint64_t stream, int32_t *d_perm_out)
{
    const size_t sort_bytes = tfrec_sort_temp_bytes((size_t)n, 32);
    uint32_t *keys = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n), *vals = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n);
    uint32_t *keys_sorted = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n);
    void *tmp = tfrec_ws_take(ctx, sort_bytes);
    TF_REQUIRE(tmp != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    shuffle_keys_hi_kernel<<<blocks, 256, 0, ctx->stream>>>(n, seed, stream, keys, vals);
    TF_HIP(hipGetLastError());
    // the sort is stable: equal high words keep their storage order, which is the order the tie kernel starts from
    TF_TRY(tfrec_sort_pairs(tmp, sort_bytes, keys, keys_sorted, vals, reinterpret_cast<uint32_t *>(d_perm_out), (size_t)n, 32, ctx->stream));
    shuffle_fix_ties_kernel<<<blocks, 256, 0, ctx->stream>>>(n, seed, stream, keys_sorted, reinterpret_cast<uint32_t *>(d_perm_out));
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

}  // namespace

extern "C" int tfrec_sample_negatives(tfrec_ctx *ctx, const int64_t *d_indptr, const int32_t *d_indices, int64_t U, int64_t I,
                                      int64_t nnz, int n_neg, uint64_t seed, uint64_t stream, int64_t first_interaction,
                                      int32_t *d_users_out, int32_t *d_pos_out, int32_t *d_neg_out)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(U > 0 && I > 0 && I < (1ll << 31) && U < (1ll << 31), TFREC_EINVAL, "bad shape U=%lld I=%lld", (long long)U, (long long)I);
    TF_REQUIRE(nnz >= 0 && n_neg >= 1 && first_interaction >= 0, TFREC_EINVAL, "bad interaction count, offset or negatives per positive");
    if (nnz == 0) return TFREC_OK;
    TF_REQUIRE(d_indptr && d_indices && d_neg_out, TFREC_EINVAL, "null array");
    sample_negatives_kernel<<<(unsigned)((nnz + 255) / 256), 256, 0, ctx->stream>>>(d_indptr, d_indices, U, (uint32_t)I, nnz, n_neg, seed,
                                                                                  stream, (uint64_t)first_interaction, d_users_out, d_pos_out, d_neg_out, nullptr, ctx->d_flags);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_shuffle_perm(tfrec_ctx *ctx, int64_t n, uint64_t seed, uint64_t stream, int32_t *d_perm_out)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && n < (1ll << 31), TFREC_EINVAL, "element count %lld out of range", (long long)n);
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_perm_out != nullptr, TFREC_EINVAL, "null output");
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + shuffle_ws_bytes(n)));
    return shuffle_perm_impl(ctx, n, seed, stream, d_perm_out);
}

extern "C" int tfrec_epoch_triples(tfrec_ctx *ctx, const int64_t *d_indptr, const int32_t *d_indices, int64_t U, int64_t I, int64_t nnz,
                                   uint64_t seed, uint64_t draw_stream, uint64_t shuffle_stream, int64_t first_interaction,
                                   int32_t *d_users_out, int32_t *d_pos_out, int32_t *d_neg_out)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(U > 0 && I > 0 && I < (1ll << 31) && U < (1ll << 31), TFREC_EINVAL, "bad shape U=%lld I=%lld", (long long)U, (long long)I);
    TF_REQUIRE(nnz >= 0 && nnz < (1ll << 31) && first_interaction >= 0, TFREC_EINVAL, "bad interaction count or offset");
    if (nnz == 0) return TFREC_OK;
    TF_REQUIRE(d_indptr && d_indices && d_users_out && d_pos_out && d_neg_out, TFREC_EINVAL, "null array");
    TF_TRY(tfrec_ws_require(ctx, tf_align(16 * (size_t)nnz) + tf_align(4 * (size_t)nnz) + shuffle_ws_bytes(nnz) + 1024));
    int4 *packed = (int4 *)tfrec_ws_take(ctx, 16 * (size_t)nnz);
    int32_t *perm = (int32_t *)tfrec_ws_take(ctx, 4 * (size_t)nnz);
    TF_REQUIRE(perm != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    const unsigned blocks = (unsigned)((nnz + 255) / 256);
    sample_negatives_kernel<<<blocks, 256, 0, ctx->stream>>>(d_indptr, d_indices, U, (uint32_t)I, nnz, 1, seed, draw_stream,
                                                            (uint64_t)first_interaction, nullptr, nullptr, nullptr, packed, ctx->d_flags);
    TF_HIP(hipGetLastError());
    TF_TRY(shuffle_perm_impl(ctx, nnz, seed, shuffle_stream, perm));
    gather_unpack3_kernel<<<blocks, 256, 0, ctx->stream>>>(packed, perm, nnz, d_users_out, d_pos_out, d_neg_out);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_gather_rows_i32(tfrec_ctx *ctx, const int32_t *d_src, int64_t n_src, const int32_t *d_perm, int64_t n, int width,
                                     int32_t *d_dst)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && n_src >= 0 && width >= 1, TFREC_EINVAL, "bad shape");
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_src && d_perm && d_dst && n_src > 0, TFREC_EINVAL, "null array");
    TF_REQUIRE(d_src != d_dst, TFREC_EINVAL, "the gather is not in place");
    gather_rows_i32_kernel<<<(unsigned)((n * width + 255) / 256), 256, 0, ctx->stream>>>(d_src, d_perm, n, width, n_src, d_dst, ctx->d_flags);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}
