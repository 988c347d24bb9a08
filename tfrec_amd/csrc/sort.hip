// Stable LSD radix sort of (key, value) pairs of 32-bit words on the device.  The duplicate-index
// reduction of the training steps is built on it: occurrences sorted by row id become contiguous
// segments, and stability keeps the occurrences of a row in batch order, so the order in which a
// row's gradient terms are added is fixed.
//
// A batch is a few million pairs, far too few for a chained single-sweep sort to hide its
// tile-to-tile dependency.  Every pass here is three short, fully parallel kernels instead:
//   1. sort_hist_kernel     per-tile digit counts                  -> counts[digit][tile]
//   2. sort_scan_kernel     exclusive prefix along every digit row  (in place) + digit totals
//   3. sort_scatter_kernel  ranks the tile's pairs stably (wave-level digit matching), orders them
//                           in LDS and writes every digit's run to its final place, coalesced
// No kernel needs zeroed memory, so no memsets are enqueued.
#include "common.hpp"

// widest radix digit a pass may use
#ifndef TF_SORT_RB_MAX
#define TF_SORT_RB_MAX 8
#endif


namespace {

constexpr int SORT_WAVES = 4;       // workgroups of 256 lanes

// Exclusive prefix of one value per lane over the 256 lanes of a workgroup; *total = sum of all.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wave_sums /* [SORT_WAVES] in LDS */, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wave_sums[w] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < SORT_WAVES; ++i) {
        const uint32_t s = wave_sums[i];
        if (i < w) before += s;
        all += s;
    }
    __syncthreads();      // wave_sums may be reused by the caller
    *total = all;
    return before + inc - v;
}

// Where a pass reads its pairs from: two arrays, or (first pass of a training batch) the index triples themselves.
// The loads are straight-line code (selects, no branches), so that a lane's ITEMS loads are all in flight together.
struct PlainSrc {
    const uint32_t *__restrict__ k, *__restrict__ v;
    __device__ __forceinline__ uint32_t key(uint32_t i) const { return k[i]; }
    __device__ __forceinline__ void load(uint32_t i, uint32_t &key_, uint32_t &val_, bool &) const { key_ = k[i]; val_ = v[i]; }
    __device__ __forceinline__ void report(bool) const {}
};

// MULTI: several batches sorted together (a batch number above the row id in every key)
template <bool MULTI>
struct TripleSrcT {
    tfrec_triple_src a;
    // WRITE: the scatter pass (which sees every pair exactly once) leaves the checked items in `pair`
    template <bool WRITE>
    __device__ __forceinline__ void get(uint32_t i, uint32_t &key_, uint32_t &val_, bool &bad) const
    {
        const bool is_user = i < a.n;
        const uint32_t e = i - a.n, side = is_user ? 0u : (e & 1u);
        const uint32_t t = is_user ? i : (e >> 1);
        const int32_t *src = is_user ? a.users : (side ? a.neg : a.pos);
        const int32_t x = src[t];
        const uint32_t lim = is_user ? a.U : a.I;
        bad = bad || x < 0 || (uint32_t)x >= lim;
        const uint32_t c = x < 0 ? 0u : ((uint32_t)x >= lim ? lim - 1 : (uint32_t)x);
        uint32_t k = 0, tau = t;
        if constexpr (MULTI) {
            // batch number t / B without a division: multiply by the rounded-up reciprocal, step back if that overshot
            // (batches of one triple: the reciprocal 2^32 does not fit its 32 bits, and there is nothing to divide)
            k = __umulhi(t, a.inv_B);
            k -= (k * a.B > t) ? 1u : 0u;
            k = a.B == 1u ? t : k;
            tau = t - k * a.B;
        }
        key_ = (k << a.kb) | (is_user ? c : a.U + c);
        val_ = is_user ? tau : 2 * tau + side;
        if constexpr (WRITE) if (!is_user) a.pair[e] = c;
    }
    __device__ __forceinline__ uint32_t key(uint32_t i) const { uint32_t k, v; bool bad = false; get<false>(i, k, v, bad); return k; }
    __device__ __forceinline__ void load(uint32_t i, uint32_t &key_, uint32_t &val_, bool &bad) const { get<true>(i, key_, val_, bad); }
    __device__ __forceinline__ void report(bool bad) const { if (bad) atomicOr(a.flags, TFREC_FLAG_BAD_INDEX); }
};

template <int RB, int ITEMS, class Src>
__global__ __launch_bounds__(256) void sort_hist_kernel(Src src, uint32_t n, int shift, uint32_t dmask,
                                                        uint32_t *__restrict__ counts, uint32_t nblk)
{
    constexpr int BINS = 1 << RB, TILE = 256 * ITEMS;
    __shared__ uint32_t h[SORT_WAVES][BINS];
    for (int i = threadIdx.x; i < SORT_WAVES * BINS; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)TILE;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = base + r * 256 + threadIdx.x;
        const uint32_t kk = src.key(i < n ? i : n - 1);
        if (i < n) atomicAdd(&h[w][(kk >> shift) & dmask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += 256) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) s += h[i][b];
        counts[(size_t)b * nblk + blockIdx.x] = s;
    }
}

// One workgroup per digit: counts[digit][0..nblk) becomes its exclusive prefix; digit_total[digit] = row sum.
__global__ __launch_bounds__(256) void sort_scan_kernel(uint32_t *__restrict__ counts, uint32_t nblk, uint32_t *__restrict__ digit_total)
{
    __shared__ uint32_t wave_sums[SORT_WAVES];
    uint32_t *row = counts + (size_t)blockIdx.x * nblk;
    const uint32_t per = (nblk + 255) / 256;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < nblk ? lo + per : nblk;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += row[i];
    uint32_t total;
    uint32_t run = block_excl_scan(s, wave_sums, &total);
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = total;
}

template <int RB, int ITEMS, class Src>
__global__ __launch_bounds__(256) void sort_scatter_kernel(Src src,
                                                           uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, uint32_t n, int shift,
                                                           uint32_t dmask, const uint32_t *__restrict__ prefix,
                                                           const uint32_t *__restrict__ digit_total, uint32_t nblk)
{
    constexpr int BINS = 1 << RB, TILE = 256 * ITEMS, PER_WAVE = 64 * ITEMS;
    constexpr int DPT = BINS >= 256 ? BINS / 256 : 1;       // digits per lane in the prefix steps
    __shared__ uint32_t cnt[SORT_WAVES][BINS];   // per-wave digit counts, then each wave's first slot per digit
    __shared__ uint32_t gofs[BINS];              // slot i of the ordered tile (digit d) goes to gofs[d] + i
    __shared__ uint2 ordered[TILE];
    __shared__ uint32_t wave_sums[SORT_WAVES];

    for (int i = threadIdx.x; i < SORT_WAVES * BINS; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t tile0 = blockIdx.x * (uint32_t)TILE;
    const uint32_t wbase = tile0 + w * PER_WAVE;
    uint32_t k[ITEMS], v[ITEMS], rank[ITEMS];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = wbase + r * 64 + lane;
        src.load(i < n ? i : n - 1, k[r], v[r], bad);      // (a lane past the end repeats the last pair; it is never ranked)
    }
    src.report(bad);
    // Wave w holds PER_WAVE consecutive pairs, 64 consecutive ones per round.  Lanes with the same
    // digit find each other with one ballot per digit bit; their order among themselves is lane
    // order, and the lowest of them advances the wave's counter of that digit.
    uint32_t *my_cnt = &cnt[w][0];
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const bool valid = wbase + r * 64 + lane < n;
        const uint32_t d = (k[r] >> shift) & dmask;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rk = (uint32_t)__popcll(m & below);
        // (one LDS add-and-return per digit and round; a wave's LDS operations complete in issue
        // order, so the rounds pipeline and the counts do not depend on timing)
        uint32_t old = 0;
        if (valid && rk == 0) old = atomicAdd(&my_cnt[d], (uint32_t)__popcll(m));
        const int leader = m ? __ffsll((unsigned long long)m) - 1 : 0;
        rank[r] = __shfl(old, leader, 64) + rk;
    }
    __syncthreads();

    // per digit: where each wave's pairs start in the ordered tile, and where the tile's run goes
    {
        uint32_t c[DPT][SORT_WAVES], tot = 0, gt[DPT], gsum = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = threadIdx.x * DPT + j;
            gt[j] = 0;
#pragma unroll
            for (int i = 0; i < SORT_WAVES; ++i) c[j][i] = 0;
            if (d < BINS) {
#pragma unroll
                for (int i = 0; i < SORT_WAVES; ++i) { c[j][i] = cnt[i][d]; tot += c[j][i]; }
                gt[j] = digit_total[d];
                gsum += gt[j];
            }
        }
        uint32_t unused;
        uint32_t dbase = block_excl_scan(tot, wave_sums, &unused);      // first slot of the lane's first digit in the tile
        uint32_t gbase = block_excl_scan(gsum, wave_sums, &unused);     // first position of that digit in the whole output
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = threadIdx.x * DPT + j;
            if (d < BINS) {
                gofs[d] = gbase + prefix[(size_t)d * nblk + blockIdx.x] - dbase;
                uint32_t run = dbase;
#pragma unroll
                for (int i = 0; i < SORT_WAVES; ++i) { cnt[i][d] = run; run += c[j][i]; }
                dbase = run;
                gbase += gt[j];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if (wbase + r * 64 + lane < n) {
            const uint32_t d = (k[r] >> shift) & dmask;
            ordered[cnt[w][d] + rank[r]] = make_uint2(k[r], v[r]);
        }
    }
    __syncthreads();
    const uint32_t nvalid = n - tile0 < (uint32_t)TILE ? n - tile0 : (uint32_t)TILE;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = j * 256 + threadIdx.x;
        if (i < nvalid) {
            const uint2 kv = ordered[i];
            const uint32_t p = gofs[(kv.x >> shift) & dmask] + i;
            kout[p] = kv.x;
            vout[p] = kv.y;
        }
    }
}

// Small batches (the reference's 512 or 128 triples a step): the radix passes above are nine launches that finish
// in microseconds each, so a step is bound by its launches.  Up to SMALL_MAX pairs are sorted by ONE workgroup
// instead: a bitonic network over 64-bit (key, value) words.  Ordering by the word equals the stable order by key
// because the values of a training batch ascend in input order (occurrence ids).
// A thread keeps E consecutive words in registers.  A stage whose partner distance stays inside the thread is
// register work, one whose partner is in the same wave is a lane exchange, and only the stages that cross waves
// (6 to 10 of the up to 91) go through LDS and its barriers.
constexpr int SMALL_MAX = 8192;

// word i keeps the smaller of (its own, its partner's) when it is the lower index of an ascending pair or the
// upper index of a descending one
__device__ __forceinline__ uint64_t bitonic_keep(uint64_t mine, uint64_t other, uint32_t i, uint32_t j, uint32_t k2)
{
    const bool take_min = ((i & j) == 0) == ((i & k2) == 0);
    const bool other_less = other < mine;
    return (other_less == take_min) ? other : mine;
}

// one stage (partner distance J inside blocks of K2) of the network; x[r] of thread t is word t*E + r
template <int E, uint32_t K2, uint32_t J>
__device__ __forceinline__ void small_sort_stage(uint64_t (&x)[E], uint64_t *e, uint32_t i0, int lane)
{
    if constexpr (J < (uint32_t)E) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            if ((r & J) == 0) {
                const uint64_t a = x[r], b = x[r | J];
                const bool up = ((i0 + r) & K2) == 0;
                const bool sw = (a > b) == up;
                x[r] = sw ? b : a;
                x[r | J] = sw ? a : b;
            }
        }
    } else if constexpr (J < 64u * E) {
        constexpr int M = (int)(J / E);                   // partner lane = lane ^ M, same register
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t lo = xor_lane_u32<M>((uint32_t)x[r], lane);
            const uint32_t hi = xor_lane_u32<M>((uint32_t)(x[r] >> 32), lane);
            x[r] = bitonic_keep(x[r], ((uint64_t)hi << 32) | lo, i0 + r, J, K2);
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) e[i0 + r] = x[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) x[r] = bitonic_keep(x[r], e[(i0 + r) ^ J], i0 + r, J, K2);
        __syncthreads();
    }
}

template <int E, uint32_t NP, uint32_t K2, uint32_t J>
__device__ __forceinline__ void small_sort_stages(uint64_t (&x)[E], uint64_t *e, uint32_t i0, int lane)
{
    small_sort_stage<E, K2, J>(x, e, i0, lane);
    if constexpr (J > 1) small_sort_stages<E, NP, K2, J / 2>(x, e, i0, lane);
    else if constexpr (K2 < NP) small_sort_stages<E, NP, K2 * 2, K2>(x, e, i0, lane);
}

template <class Src, int E, int NT>
__global__ __launch_bounds__(NT) void sort_small_kernel(Src src, uint32_t n, uint32_t *__restrict__ kout, uint32_t *__restrict__ vout)
{
    constexpr uint32_t NP = (uint32_t)E * NT;
    __shared__ uint64_t e[NP];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = (uint32_t)r * NT + threadIdx.x;
        uint32_t k = 0, v = 0;
        src.load(i < n ? i : n - 1, k, v, bad);
        e[i] = i < n ? ((uint64_t)k << 32) | v : ~0ull;
    }
    src.report(bad);
    __syncthreads();
    const uint32_t i0 = threadIdx.x * E;
    uint64_t x[E];
#pragma unroll
    for (int r = 0; r < E; ++r) x[r] = e[i0 + r];
    __syncthreads();
    small_sort_stages<E, NP, 2, 1>(x, e, i0, (int)(threadIdx.x & 63));
#pragma unroll
    for (int r = 0; r < E; ++r) e[i0 + r] = x[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = (uint32_t)r * NT + threadIdx.x;
        if (i < n) {
            const uint64_t w = e[i];
            kout[i] = (uint32_t)(w >> 32);
            vout[i] = (uint32_t)w;
        }
    }
}

template <class Src>
static void sort_small_launch(const Src &src, uint32_t n, uint32_t *kout, uint32_t *vout, hipStream_t stream)
{
    if (n <= 256) sort_small_kernel<Src, 1, 256><<<1, 256, 0, stream>>>(src, n, kout, vout);
    else if (n <= 512) sort_small_kernel<Src, 2, 256><<<1, 256, 0, stream>>>(src, n, kout, vout);
    else if (n <= 1024) sort_small_kernel<Src, 2, 512><<<1, 512, 0, stream>>>(src, n, kout, vout);
    else if (n <= 2048) sort_small_kernel<Src, 4, 512><<<1, 512, 0, stream>>>(src, n, kout, vout);
    else if (n <= 4096) sort_small_kernel<Src, 4, 1024><<<1, 1024, 0, stream>>>(src, n, kout, vout);
    else sort_small_kernel<Src, 8, 1024><<<1, 1024, 0, stream>>>(src, n, kout, vout);
}

constexpr int SORT_RB_MAX = TF_SORT_RB_MAX;      // widest digit

// Pairs per lane: tiles of 4096 pairs move data best; below about a million pairs there are too few
// such tiles to occupy the device, and tiles of 2048 finish sooner.
inline int sort_items_for(size_t n) { return n >= ((size_t)1 << 20) ? 16 : 8; }

struct SortPlan {
    int passes, rb, items;
    uint32_t nblk;
    size_t counts_bytes, total_bytes, tmp_bytes;
};

SortPlan sort_plan(size_t n, int end_bit)
{
    SortPlan p;
    if (end_bit < 1) end_bit = 1;
    p.passes = (end_bit + SORT_RB_MAX - 1) / SORT_RB_MAX;
    p.rb = (end_bit + p.passes - 1) / p.passes;
    if (p.rb < 6) p.rb = 6;
    p.items = sort_items_for(n);
    const size_t tile = (size_t)256 * p.items;
    p.nblk = (uint32_t)((n + tile - 1) / tile);
    if (p.nblk == 0) p.nblk = 1;
    // sized for the smaller tile, so that the byte count does not depend on the tile choice
    const size_t nblk_max = (n + 2047) / 2048 + 1;
    p.counts_bytes = tf_align(sizeof(uint32_t) * ((size_t)1 << p.rb) * nblk_max);
    p.total_bytes = tf_align(sizeof(uint32_t) * ((size_t)1 << p.rb));
    p.tmp_bytes = p.passes > 1 ? tf_align(sizeof(uint32_t) * n) : 0;
    return p;
}

template <int RB, int ITEMS, class Src>
int sort_pass(const Src &src, uint32_t *kout, uint32_t *vout, uint32_t n, int shift, uint32_t dmask,
              uint32_t *counts, uint32_t *totals, uint32_t nblk, hipStream_t stream)
{
    sort_hist_kernel<RB, ITEMS, Src><<<nblk, 256, 0, stream>>>(src, n, shift, dmask, counts, nblk);
    sort_scan_kernel<<<1 << RB, 256, 0, stream>>>(counts, nblk, totals);
    sort_scatter_kernel<RB, ITEMS, Src><<<nblk, 256, 0, stream>>>(src, kout, vout, n, shift, dmask, counts, totals, nblk);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

template <int RB, class Src>
int sort_pass_items(int items, const Src &src, uint32_t *kout, uint32_t *vout, uint32_t n, int shift,
                    uint32_t dmask, uint32_t *counts, uint32_t *totals, uint32_t nblk, hipStream_t stream)
{
    if (items == 16) return sort_pass<RB, 16, Src>(src, kout, vout, n, shift, dmask, counts, totals, nblk, stream);
    return sort_pass<RB, 8, Src>(src, kout, vout, n, shift, dmask, counts, totals, nblk, stream);
}

template <class Src>
int sort_pass_rb(int rb, int items, const Src &src, uint32_t *kout, uint32_t *vout, uint32_t n, int shift, uint32_t dmask,
                 uint32_t *counts, uint32_t *totals, uint32_t nblk, hipStream_t stream)
{
    switch (rb) {
        case 6: return sort_pass_items<6, Src>(items, src, kout, vout, n, shift, dmask, counts, totals, nblk, stream);
        case 7: return sort_pass_items<7, Src>(items, src, kout, vout, n, shift, dmask, counts, totals, nblk, stream);
        case 8: return sort_pass_items<8, Src>(items, src, kout, vout, n, shift, dmask, counts, totals, nblk, stream);
    }
    tfrec_set_error("internal: no sort kernel for %d-bit digits", rb);
    return TFREC_EUNSUP;
}

// first_src: where the first pass reads its pairs (null: keys_in / vals_in)
int sort_run(void *d_temp, size_t temp_bytes, const tfrec_triple_src *first_src, const uint32_t *keys_in, const uint32_t *vals_in,
             uint32_t *keys_out, uint32_t *vals_out, size_t n, int end_bit, hipStream_t stream)
{
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(n < (1ull << 32), TFREC_EUNSUP, "sort of %zu pairs: positions must fit 32 bits", n);
    TF_REQUIRE(end_bit >= 1 && end_bit <= 32, TFREC_EINVAL, "sort key width %d out of range", end_bit);
    if (first_src && n <= (size_t)SMALL_MAX) {
        if (first_src->inv_B) sort_small_launch(TripleSrcT<true>{*first_src}, (uint32_t)n, keys_out, vals_out, stream);
        else sort_small_launch(TripleSrcT<false>{*first_src}, (uint32_t)n, keys_out, vals_out, stream);
        TF_HIP(hipGetLastError());
        return TFREC_OK;
    }
    const SortPlan p = sort_plan(n, end_bit);
    TF_REQUIRE(temp_bytes >= p.counts_bytes + p.total_bytes + 2 * p.tmp_bytes, TFREC_EINVAL, "internal: sort scratch too small");
    char *t = (char *)d_temp;
    uint32_t *counts = (uint32_t *)t; t += p.counts_bytes;
    uint32_t *totals = (uint32_t *)t; t += p.total_bytes;
    uint32_t *ktmp = (uint32_t *)t; t += p.tmp_bytes;
    uint32_t *vtmp = (uint32_t *)t;
    // the last pass writes the output arrays; the passes before it alternate backwards from there
    const uint32_t *kin = keys_in, *vin = vals_in;
    for (int pass = 0; pass < p.passes; ++pass) {
        const bool to_out = ((p.passes - 1 - pass) & 1) == 0;
        uint32_t *ko = to_out ? keys_out : ktmp, *vo = to_out ? vals_out : vtmp;
        const int shift = pass * p.rb;
        const int bits = end_bit - shift < p.rb ? end_bit - shift : p.rb;
        const uint32_t dmask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
        int rc;
        if (pass == 0 && first_src && first_src->inv_B)
            rc = sort_pass_rb<TripleSrcT<true>>(p.rb, p.items, TripleSrcT<true>{*first_src}, ko, vo, (uint32_t)n, shift, dmask, counts, totals, p.nblk, stream);
        else if (pass == 0 && first_src)
            rc = sort_pass_rb<TripleSrcT<false>>(p.rb, p.items, TripleSrcT<false>{*first_src}, ko, vo, (uint32_t)n, shift, dmask, counts, totals, p.nblk, stream);
        else rc = sort_pass_rb<PlainSrc>(p.rb, p.items, PlainSrc{kin, vin}, ko, vo, (uint32_t)n, shift, dmask, counts, totals, p.nblk, stream);
        if (rc != TFREC_OK) return rc;
        kin = ko; vin = vo;
    }
    return TFREC_OK;
}

}  // namespace

size_t tfrec_sort_temp_bytes(size_t n, int end_bit)
{
    const SortPlan p = sort_plan(n, end_bit);
    return p.counts_bytes + p.total_bytes + 2 * p.tmp_bytes + 256;
}

int tfrec_sort_pairs(void *d_temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                     const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit, hipStream_t stream)
{
    return sort_run(d_temp, temp_bytes, nullptr, keys_in, vals_in, keys_out, vals_out, n, end_bit, stream);
}

int tfrec_sort_triples(void *d_temp, size_t temp_bytes, const tfrec_triple_src &src, uint32_t *keys_out, uint32_t *vals_out,
                       int end_bit, hipStream_t stream)
{
    return sort_run(d_temp, temp_bytes, &src, nullptr, nullptr, keys_out, vals_out, (size_t)3 * src.n, end_bit, stream);
}

extern "C" int tfrec_sort_pairs_u32(tfrec_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in, uint32_t *d_keys_out,
                                    uint32_t *d_vals_out, int64_t n, int key_bits)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && n < (1ll << 32), TFREC_EINVAL, "pair count %lld out of range", (long long)n);
    TF_REQUIRE(key_bits >= 1 && key_bits <= 32, TFREC_EINVAL, "key width %d out of range", key_bits);
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_keys_in && d_vals_in && d_keys_out && d_vals_out, TFREC_EINVAL, "null array");
    TF_REQUIRE(d_keys_in != d_keys_out && d_vals_in != d_vals_out, TFREC_EINVAL, "the sort is not in place");
    const size_t bytes = tfrec_sort_temp_bytes((size_t)n, key_bits);
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + tf_align(bytes)));
    void *tmp = tfrec_ws_take(ctx, bytes);
    TF_REQUIRE(tmp != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    return tfrec_sort_pairs(tmp, bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, key_bits, ctx->stream);
}
