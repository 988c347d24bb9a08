// Atomic-free segmented reduction over occurrences sorted by destination row.
//
// Input: `keys` (destination row ids, ascending) and `vals` (occurrence ids) for positions
// [base, base+n).  A *segment* is a maximal run of equal keys: all contributions to one row.
// The positions are cut into fixed tiles of CH occurrences; one aligned group of LPR lanes
// (LPR*VEC*NCH >= d floats: the group holds one whole row, 16 bytes per lane) walks one tile
// in order and keeps the running sum of the current segment in registers.
//
//   * a segment that lies entirely inside a tile is finished on the spot (Pol::apply);
//   * a segment cut by a tile boundary leaves per-tile partial sums (at most one "lead"
//     partial for the segment entering the tile and one "tail" partial for the segment
//     leaving it).  seg_fixup_kernel then adds, for every segment, its tail partial and the
//     lead partials of the following tiles in tile order and applies the result.
//
// Every sum is formed in an order fixed by the sorted input alone, so results are bitwise
// reproducible; no atomics touch the tables.  Tiles have equal work whatever the skew of
// the keys, and a row hit by thousands of occurrences (hot item) is spread over many tiles;
// its long chain of partials is summed cooperatively by a whole workgroup.
//
// A policy `Pol` supplies the arithmetic:
//   struct Pol { enum { LPR, NCH, VEC }; using RowT = Row<LPR,NCH,VEC>;
//     struct Occ;  struct Seg;  struct Loss;
//     uint4 entry(uint32_t key, uint32_t val, int64_t pos)  -> the entry staged in LDS; .x must be the key
//     Occ  load_occ(uint4 entry, int gl)                    -> gathers of one occurrence
//     Seg  load_seg(uint32_t key, int gl)                   -> the destination row(s)
//     void compute(Seg&, const Occ&, uint4 entry, Acc<RowT>&, Loss&, int gl)
//     void apply(uint32_t key, const Seg*, const Acc<RowT>&, int gl)   (Seg==nullptr: load it)
//   };
#pragma once
#include "common.hpp"

template <class RowT>
struct Acc {
    RowT v;
    float s0, s1;
    int cnt;
    __device__ __forceinline__ void clear() { v = RowT::zero(); s0 = 0.f; s1 = 0.f; cnt = 0; }
    __device__ __forceinline__ void add(const Acc &o) { v = v + o.v; s0 += o.s0; s1 += o.s1; cnt += o.cnt; }
};

// occurrences per tile (at least one per lane of the group)
#ifndef TF_TILE_CH
#define TF_TILE_CH 64
#endif
constexpr int tile_ch(int lpr) { return lpr > TF_TILE_CH ? lpr : TF_TILE_CH; }
// Short tiles for small batches: a lane group walks its tile in order, so the tile length is the step's
// latency when there are too few tiles to fill the chip anyway (a policy opts in with `enum { SHORT_TILES = 1 }`).
constexpr int short_tile_ch(int lpr) { return lpr > 16 ? lpr : 16; }
template <class Pol>
constexpr int pol_ch()
{
    if constexpr (requires { Pol::SHORT_TILES; }) return Pol::SHORT_TILES ? short_tile_ch(Pol::LPR) : tile_ch(Pol::LPR);
    else return tile_ch(Pol::LPR);
}

template <int LPR, int CH_>
struct TileGeom {
    static constexpr int CH = CH_;                   // occurrences per tile
    static constexpr int EPL = CH / LPR;             // entries staged per lane
    static constexpr int GPB = 256 / LPR;            // groups (= tiles) per 256-thread workgroup
};

enum : uint32_t { LEAD_NONE = 0, LEAD_END = 1, LEAD_THROUGH = 2, TAIL_BIT = 4 };
static constexpr uint32_t NOKEY = 0xffffffffu;
static constexpr int FIX_SHORT = 8;    // chains up to this many tiles are walked by one group

// Partial records: rows [2*nTiles][ROWF] floats, scalars [2*nTiles] float4 {s0, s1, cnt, -}.
template <class RowT, int LPR, int NCH, int VEC>
__device__ __forceinline__ void part_store(float *rows, float4 *sc, int64_t slot, const Acc<RowT> &a, int gl)
{
    constexpr int ROWF = LPR * NCH * VEC;
    a.v.store(rows + slot * ROWF, gl, LPR * NCH);
    if (gl == 0) sc[slot] = make_float4(a.s0, a.s1, __int_as_float(a.cnt), 0.f);
}
template <class RowT, int LPR, int NCH, int VEC>
__device__ __forceinline__ Acc<RowT> part_load(const float *rows, const float4 *sc, int64_t slot, int gl)
{
    constexpr int ROWF = LPR * NCH * VEC;
    Acc<RowT> a;
    a.v = RowT::load(rows + slot * ROWF, gl, LPR * NCH);
    const float4 s = sc[slot];
    a.s0 = s.x; a.s1 = s.y; a.cnt = __float_as_int(s.z);
    return a;
}

// Sum of a workgroup's per-thread (rank, reg) loss terms, written as two doubles per block.
template <class Loss>
__device__ __forceinline__ void block_loss_store(const Loss &loss, double *loss_part)
{
    __shared__ double red[4][Loss::N];
    double v[Loss::N];
#pragma unroll
    for (int i = 0; i < Loss::N; ++i) {
        v[i] = (double)loss.t[i];
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
    }
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < Loss::N; ++i) red[threadIdx.x >> 6][i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 0; i < Loss::N; ++i)
            loss_part[(size_t)blockIdx.x * Loss::N + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

template <int N_>
struct LossAcc {
    static constexpr int N = N_;
    float t[N_ > 0 ? N_ : 1];
    __device__ __forceinline__ void clear()
    {
        for (int i = 0; i < (N_ > 0 ? N_ : 1); ++i) t[i] = 0.f;
    }
};

template <class Pol>
__global__ __launch_bounds__(256) void seg_tile_kernel(Pol pol, const uint32_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ vals, int64_t base, int64_t n,
                                                       float *__restrict__ part_rows, float4 *__restrict__ part_sc,
                                                       uint32_t *__restrict__ tile_flags, int64_t blk_slot0,
                                                       double *__restrict__ loss_part, const int64_t *__restrict__ range = nullptr)
{
    // range (device memory, optional): the pass covers positions [base + range[0], base + range[1]) only; the grid
    // was sized for n, workgroups past the end of the range find nothing to do
    if (range) {
        base += range[0];
        n = range[1] - range[0];
        if ((int64_t)blockIdx.x * (256 / Pol::LPR) * pol_ch<Pol>() >= n) return;      // the whole workgroup lies past the range
    }
    constexpr int LPR = Pol::LPR, NCH = Pol::NCH, VEC = Pol::VEC;
    using RowT = typename Pol::RowT;
    using G = TileGeom<LPR, pol_ch<Pol>()>;
    constexpr int CH = G::CH, EPL = G::EPL, GPB = G::GPB;
    constexpr int ROWF = LPR * NCH * VEC;
    __shared__ uint4 ent[GPB][CH];
    __shared__ __align__(16) float bred[GPB][ROWF];
    __shared__ float4 bsc[GPB];
    __shared__ uint32_t bthrough[GPB];

    const int gl = threadIdx.x % LPR;
    const int g = threadIdx.x / LPR;
    const int64_t w = (int64_t)blockIdx.x * GPB + g;
    const int64_t a = w * CH;
    const int cnt = a < n ? (int)((n - a) < CH ? (n - a) : CH) : 0;

    // stage this tile's entries {key, val, aux0, aux1}: coalesced index reads, one dependent gather
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int e = s * LPR + gl;
        if (e < cnt) {
            ent[g][e] = pol.entry(keys[base + a + e], vals[base + a + e], base + a + e);
        }
    }
    __syncthreads();

    typename Pol::Loss loss;
    loss.clear();
    Acc<RowT> acc;
    acc.clear();
    uint32_t lead_kind = LEAD_NONE;
    if (cnt > 0) {
        const uint32_t key_prev = a > 0 ? keys[base + a - 1] : NOKEY;
        const uint32_t key_next = a + cnt < n ? keys[base + a + cnt] : NOKEY;
        uint32_t tail = 0;

        // Occurrences are taken DEPTH at a time: all their gathers (and the rows of the segments that
        // start among them) are issued before the first is used, so a lane group keeps DEPTH row
        // gathers in flight instead of one.
        constexpr int DEPTH = Pol::DEPTH;
        static_assert(CH % DEPTH == 0, "a tile holds a whole number of gather groups");
        uint32_t cur_key = ent[g][0].x;
        bool lead_open = (cur_key == key_prev);
        // (a policy whose destination row depends on more than the key defines load_seg_en(entry, gl))
        auto seg_of = [&](const uint4 &en_) {
            if constexpr (requires { pol.load_seg_en(en_, gl); }) return pol.load_seg_en(en_, gl);
            else return pol.load_seg(en_.x, gl);
        };
        typename Pol::Seg seg = seg_of(ent[g][0]);
        uint32_t seg_key = cur_key;          // key of the most recently requested segment row
        // WHOLE: the tile has all CH occurrences (every tile but the last one of a pass), no bounds tests.
        auto run = [&]<bool WHOLE>() {
#pragma unroll 1
            for (int c0 = 0; c0 < (WHOLE ? CH : cnt); c0 += DEPTH) {
                uint4 en[DEPTH];
                typename Pol::Occ oc[DEPTH];
                typename Pol::Seg sg[DEPTH];
                // (slots past the end of a short last tile repeat its last occurrence: loaded, never used)
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) {
                    en[q] = ent[g][(WHOLE || c0 + q < cnt) ? c0 + q : cnt - 1];
                    oc[q] = pol.load_occ(en[q], gl);
                }
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) {
                    if (en[q].x != seg_key) sg[q] = seg_of(en[q]);
                    seg_key = en[q].x;
                }
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) {
                    if (WHOLE || c0 + q < cnt) {
                        if (en[q].x != cur_key) {
                            // the previous segment ended inside this tile
                            if (lead_open) {
                                part_store<RowT, LPR, NCH, VEC>(part_rows, part_sc, 2 * w, acc, gl);
                                lead_kind = LEAD_END;
                            } else {
                                pol.apply(cur_key, &seg, acc, gl);
                            }
                            lead_open = false;
                            cur_key = en[q].x;
                            seg = sg[q];
                            acc.clear();
                        }
                        pol.compute(seg, oc[q], en[q], acc, loss, gl);
                    }
                }
            }
        };
        if (cnt == CH) run.template operator()<true>();
        else run.template operator()<false>();
        // last segment of the tile
        const bool continues = (key_next == cur_key);
        if (lead_open) {
            part_store<RowT, LPR, NCH, VEC>(part_rows, part_sc, 2 * w, acc, gl);
            lead_kind = continues ? LEAD_THROUGH : LEAD_END;
        } else if (continues) {
            part_store<RowT, LPR, NCH, VEC>(part_rows, part_sc, 2 * w + 1, acc, gl);
            tail = TAIL_BIT;
        } else {
            pol.apply(cur_key, &seg, acc, gl);
        }
        if (gl == 0) tile_flags[w] = lead_kind | tail;
    }
    // A hot row fills whole workgroups with tiles that lie entirely inside its segment.  Such a
    // workgroup also leaves ONE pre-summed partial (its tiles added in tile order), so the chain the
    // fix-up kernel has to add for that row is GPB times shorter.
    const bool through = lead_kind == LEAD_THROUGH;
    if (gl == 0) bthrough[g] = through ? 1u : 0u;
    if (through) {
        acc.v.store(&bred[g][0], gl, LPR * NCH);
        if (gl == 0) bsc[g] = make_float4(acc.s0, acc.s1, __int_as_float(acc.cnt), 0.f);
    }
    __syncthreads();
    if (g == 0) {
        bool all = true;
        for (int i = 0; i < GPB; ++i) all = all && bthrough[i] != 0u;
        if (all) {
            Acc<RowT> tot;
            tot.clear();
            for (int i = 0; i < GPB; ++i) {
                Acc<RowT> pi;
                pi.v = RowT::load(&bred[i][0], gl, LPR * NCH);
                const float4 sc4 = bsc[i];
                pi.s0 = sc4.x; pi.s1 = sc4.y; pi.cnt = __float_as_int(sc4.z);
                tot.add(pi);
            }
            part_store<RowT, LPR, NCH, VEC>(part_rows, part_sc, blk_slot0 + blockIdx.x, tot, gl);
        }
    }
    if constexpr (Pol::Loss::N > 0) block_loss_store(loss, loss_part);
}

// Finishes the segments that were cut by tile boundaries.
template <class Pol>
__global__ __launch_bounds__(256) void seg_fixup_kernel(Pol pol, const uint32_t *__restrict__ keys, int64_t base, int64_t n,
                                                        const float *__restrict__ part_rows, const float4 *__restrict__ part_sc,
                                                        const uint32_t *__restrict__ tile_flags, int64_t n_tiles, int64_t blk_slot0,
                                                        const int64_t *__restrict__ range = nullptr)
{
    constexpr int LPR = Pol::LPR, NCH = Pol::NCH, VEC = Pol::VEC;
    using RowT = typename Pol::RowT;
    using G = TileGeom<LPR, pol_ch<Pol>()>;
    constexpr int CH = G::CH, GPB = G::GPB;
    if (range) {
        base += range[0];
        n = range[1] - range[0];
        n_tiles = (n + CH - 1) / CH;
        if ((int64_t)blockIdx.x * GPB >= n_tiles) return;
    }
    constexpr int ROWF = LPR * NCH * VEC;
    __shared__ int64_t long_w[GPB];
    __shared__ int long_n;
    __shared__ int first_end;
    __shared__ __align__(16) float red_rows[GPB][ROWF];
    __shared__ float4 red_sc[GPB];

    const int gl = threadIdx.x % LPR;
    const int g = threadIdx.x / LPR;
    const int64_t w = (int64_t)blockIdx.x * GPB + g;
    if (threadIdx.x == 0) long_n = 0;
    __syncthreads();

    if (w < n_tiles && (tile_flags[w] & TAIL_BIT)) {
        const int64_t last = ((w + 1) * CH < n ? (w + 1) * CH : n) - 1;
        const uint32_t key = keys[base + last];
        // flags of the next FIX_SHORT tiles at once (independent loads), then the chain length among them
        uint32_t fl[FIX_SHORT];
#pragma unroll
        for (int s = 0; s < FIX_SHORT; ++s) fl[s] = w + 1 + s < n_tiles ? (tile_flags[w + 1 + s] & 3u) : LEAD_END;
        int len = 0;
        bool done = false;
#pragma unroll
        for (int s = 0; s < FIX_SHORT; ++s) {
            if (!done) {
                if (fl[s] == LEAD_NONE) { done = true; }              // defensive: nothing to add
                else { len = s + 1; if (fl[s] == LEAD_END) done = true; }
            }
        }
        if (done) {
            Acc<RowT> tot = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, 2 * w + 1, gl);
            Acc<RowT> pr[FIX_SHORT];
#pragma unroll
            for (int s = 0; s < FIX_SHORT; ++s)
                if (s < len) pr[s] = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, 2 * (w + 1 + s), gl);
#pragma unroll
            for (int s = 0; s < FIX_SHORT; ++s)
                if (s < len) tot.add(pr[s]);
            pol.apply(key, nullptr, tot, gl);
        } else if (gl == 0) {
            long_w[atomicAdd(&long_n, 1)] = w;
        }
    }
    __syncthreads();
    const int n_long = long_n;
    // long chains (a row with more than FIX_SHORT*CH occurrences): the whole workgroup sums one
    for (int c = 0; c < n_long; ++c) {
        // chains are taken in ascending tile order so the result does not depend on who registered first
        int64_t wl = -1;
        for (int i = 0; i < n_long; ++i) {
            const int64_t x = long_w[i];
            int rank = 0;
            for (int j = 0; j < n_long; ++j) rank += (long_w[j] < x);
            if (rank == c) wl = x;
        }
        // length of the chain: first following tile whose lead partial ends the segment
        if (threadIdx.x == 0) first_end = INT32_MAX;
        __syncthreads();
        int L = 0;
        for (int64_t r0 = 0;; r0 += 1024) {
            int mine = INT32_MAX;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const int64_t off = r0 + q * 256 + threadIdx.x;
                const int64_t idx = wl + 1 + off;
                if (idx >= n_tiles || (tile_flags[idx] & 3u) != LEAD_THROUGH) mine = (int)off;
            }
            if (mine != INT32_MAX) atomicMin(&first_end, mine);
            __syncthreads();
            const int fe = first_end;
            __syncthreads();
            if (fe != INT32_MAX) { L = fe + 1; break; }
        }
        if (wl + L >= n_tiles) L = (int)(n_tiles - 1 - wl);
        // The chain is the lead partials of tiles wl+1 .. wl+L; all but the last lie wholly inside the row's
        // segment.  Workgroups made only of such tiles were pre-summed by the tile kernel, so the chain
        // becomes head tiles, workgroup partials [a0/GPB, a1/GPB), tail tiles.
        // Group g sums a contiguous share of those items in order, four loads in flight.
        const int64_t t_first = wl + 1, t_last = wl + L;                      // inclusive
        int64_t a0 = (t_first + GPB - 1) / GPB * GPB, a1 = t_last / GPB * GPB;   // tiles [a0, a1) are whole workgroups
        if (a1 < a0) { a0 = t_last + 1; a1 = a0; }
        const int nh = (int)(a0 - t_first), nb = (int)((a1 - a0) / GPB), nt = (int)(t_last + 1 - a1);
        const int n_items = nh + nb + nt;
        auto slot_of = [&](int idx) -> int64_t {
            if (idx < nh) return 2 * (t_first + idx);
            if (idx < nh + nb) return blk_slot0 + a0 / GPB + (idx - nh);
            return 2 * (a1 + (idx - nh - nb));
        };
        const int per = (n_items + GPB - 1) / GPB;
        Acc<RowT> part;
        part.clear();
        {
            const int lo = g * per, hi = (g + 1) * per < n_items ? (g + 1) * per : n_items;
            int t = lo;
            for (; t + 4 <= hi; t += 4) {
                Acc<RowT> q0 = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, slot_of(t), gl);
                Acc<RowT> q1 = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, slot_of(t + 1), gl);
                Acc<RowT> q2 = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, slot_of(t + 2), gl);
                Acc<RowT> q3 = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, slot_of(t + 3), gl);
                part.add(q0); part.add(q1); part.add(q2); part.add(q3);
            }
            for (; t < hi; ++t) part.add(part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, slot_of(t), gl));
        }
        part.v.store(&red_rows[g][0], gl, LPR * NCH);
        if (gl == 0) red_sc[g] = make_float4(part.s0, part.s1, __int_as_float(part.cnt), 0.f);
        __syncthreads();
        if (g == 0) {
            const int64_t last = ((wl + 1) * CH < n ? (wl + 1) * CH : n) - 1;
            const uint32_t key = keys[base + last];
            Acc<RowT> tot = part_load<RowT, LPR, NCH, VEC>(part_rows, part_sc, 2 * wl + 1, gl);
            for (int i = 0; i < GPB; ++i) {
                Acc<RowT> pi;
                pi.v = RowT::load(&red_rows[i][0], gl, LPR * NCH);
                const float4 sc4 = red_sc[i];
                pi.s0 = sc4.x; pi.s1 = sc4.y; pi.cnt = __float_as_int(sc4.z);
                tot.add(pi);
            }
            pol.apply(key, nullptr, tot, gl);
        }
        __syncthreads();
    }
}

// Sums per-block loss partials [n_blocks][N] into out[N]; one workgroup, fixed order.
template <int N>
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double *__restrict__ part, int n_blocks, double *__restrict__ out)
{
    __shared__ double red[256][N];
    double v[N];
    for (int i = 0; i < N; ++i) v[i] = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 256)
        for (int i = 0; i < N; ++i) v[i] += part[(size_t)b * N + i];
    for (int i = 0; i < N; ++i) red[threadIdx.x][i] = v[i];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int i = 0; i < N; ++i) red[threadIdx.x][i] += red[threadIdx.x + s][i];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < N; ++i) out[i] = red[0][i];
}
