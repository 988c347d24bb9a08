// Full-rank evaluation: scoring of every (user, item) pair on the matrix cores with a running
// top-k per user, top-k over dense rows, top-k merge, and the ranking metrics.
//
// The scoring kernels never materialise the [users, items] score matrix that the reference
// builds on the host (model/BPR.py:76-80) and masks in place
// (evaluator/backend/cpp/evaluate.py:25-27).  A wave owns 32 users (score_topk_kernel) or 64
// (score_topk_pp_kernel: bf16, d = 128), their factors resident in registers as the MFMA "B"
// operand, and the workgroup streams item tiles through LDS as the "A" operand:
//     D[item][user] = bias[item] + sum_f Q[item][f] * P[user][f]
//   * TFREC_SCORE_F32: v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain over f ascending that
//     starts at the bias;
//   * TFREC_SCORE_BF16: v_mfma_f32_32x32x16_bf16 on operands rounded to bfloat16.
// In the accumulator layout a lane holds 16 items of ONE user, so one running threshold per
// lane (a lower bound of the user's k-th best score so far) filters the tile: a score above it
// is appended to that user's candidate list.  When a list may overflow, the owning wave compacts
// it: drops the user's train items (binary search in the sorted CSR row), keeps the best k (or a
// few more) by selection, which also tightens the threshold; the k survivors are sorted under the
// total order (score descending, item id ascending) when the scan ends.  The reference's order
// among equal scores is whatever std::partial_sort_copy leaves
// (evaluator/backend/cpp/tools.h:17); this order is the documented replacement and makes
// item-sharded results merge exactly.
#include "common.hpp"

#include <atomic>
#include <cmath>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// keys: (score, id) packed so that a larger 64-bit key means "ranks earlier"
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_orderable(float x)
{
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ uint64_t make_key(float score, int32_t id)
{
    return ((uint64_t)f32_orderable(score) << 32) | (uint32_t)(~(uint32_t)id);
}
__device__ __forceinline__ float key_score(uint64_t k) { return orderable_f32((uint32_t)(k >> 32)); }
__device__ __forceinline__ int32_t key_id(uint64_t k) { return (int32_t)(~(uint32_t)k); }

__device__ __forceinline__ bool csr_row_contains(const int32_t *__restrict__ idx, int64_t lo, int64_t hi, int32_t id)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = idx[mid];
        if (v == id) return true;
        if (v < id) lo = mid + 1; else hi = mid;
    }
    return false;
}

// Wave-wide bitonic sort of CAP 64-bit keys in LDS, descending.  All 64 lanes must call it.
template <int CAP>
__device__ __forceinline__ void wave_sort_desc(volatile uint64_t *buf, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int k2 = 2; k2 <= CAP; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int t = lane; t < CAP / 2; t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const uint64_t a = buf[i], b = buf[p];
                const bool desc = (i & k2) == 0;
                if ((a < b) == desc) { buf[i] = b; buf[p] = a; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Wave-wide bitonic sort, descending, of CAP 64-bit keys held EPL = CAP/64 per lane: key[q] of lane
// l is element q*64 + l.  Partners 64 or more elements apart sit in the same lane (register
// compare-exchange); nearer partners are fetched with cross-lane moves, no LDS round trips.
template <int CAP, int K2, int J>
__device__ __forceinline__ void bitonic_step(uint64_t (&key)[CAP / 64], int lane)
{
    constexpr int EPL = CAP / 64;
    if constexpr (J >= 64) {
        constexpr int JQ = J / 64;
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            if ((q & JQ) == 0) {
                const bool desc = (((q * 64 + lane) & K2) == 0);
                const uint64_t x = key[q], y = key[q | JQ];
                const bool swap = desc ? (x < y) : (x > y);
                key[q] = swap ? y : x;
                key[q | JQ] = swap ? x : y;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const uint64_t x = key[q];
            const uint32_t plo = xor_lane_u32<J>((uint32_t)x, lane), phi = xor_lane_u32<J>((uint32_t)(x >> 32), lane);
            const uint64_t y = ((uint64_t)phi << 32) | plo;
            const bool desc = (((q * 64 + lane) & K2) == 0);
            const bool lower = (lane & J) == 0;
            const bool take_max = (lower == desc);
            key[q] = take_max ? (x > y ? x : y) : (x < y ? x : y);
        }
    }
}
template <int CAP, int K2, int J>
__device__ __forceinline__ void bitonic_merge(uint64_t (&key)[CAP / 64], int lane)
{
    bitonic_step<CAP, K2, J>(key, lane);
    if constexpr (J > 1) bitonic_merge<CAP, K2, J / 2>(key, lane);
}
template <int CAP, int K2>
__device__ __forceinline__ void bitonic_sort(uint64_t (&key)[CAP / 64], int lane)
{
    if constexpr (K2 > 2) bitonic_sort<CAP, K2 / 2>(key, lane);
    bitonic_merge<CAP, K2, K2 / 2>(key, lane);
}

// Compacts one candidate list.  list[0..cnt) holds keys; entries [0,ver) are known not to be
// masked.  Drops masked ids, sorts, keeps the best k.  Returns the new count; *tau_out gets the
// k-th best score (or -inf while fewer than k are held).  Wave-uniform arguments.
// The user's sorted row of masked ids is first copied into the (still unused) sort buffer when it
// fits there, so that the binary searches run in LDS after one coalesced read instead of a chain
// of dependent global loads each.
template <int CAP>
__device__ __forceinline__ int wave_compact(uint64_t *__restrict__ list, int cnt, int ver, int k,
                                            const int32_t *__restrict__ mask_idx, int64_t mlo, int64_t mhi,
                                            volatile uint64_t *sortbuf, int lane, float *tau_out)
{
    constexpr int EPL = CAP / 64;             // entries per lane
    constexpr int ROWCAP = CAP * 2;           // int32 ids that fit the sort buffer
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // this wave's own appends are visible
    const int64_t L = mhi - mlo;
    const bool check = mask_idx != nullptr && L > 0 && ver < cnt;
    const bool staged = check && L <= ROWCAP;
    volatile int32_t *rowbuf = reinterpret_cast<volatile int32_t *>(sortbuf);
    // the list entries and the masked row are requested together (one memory round trip, not two)
    uint64_t keys[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const int e = lane + 64 * q;
        keys[q] = e < cnt ? list[e] : 0;
    }
    if (staged) {
        for (int j = lane; j < (int)L; j += 64) rowbuf[j] = mask_idx[mlo + j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (check) {
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const int e = lane + 64 * q;
            if (e < cnt && e >= ver) {
                const int32_t id = key_id(keys[q]);
                bool hit;
                if (staged) {
                    int lo = 0, hi = (int)L;
                    hit = false;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const int32_t v = rowbuf[mid];
                        if (v == id) { hit = true; break; }
                        if (v < id) lo = mid + 1; else hi = mid;
                    }
                } else {
                    hit = csr_row_contains(mask_idx, mlo, mhi, id);
                }
                if (hit) keys[q] = 0;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();          // every lane is done with the staged row
    int kept = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) kept += __popcll(__ballot(keys[q] != 0));
    bitonic_sort<CAP, CAP>(keys, lane);
#pragma unroll
    for (int q = 0; q < EPL; ++q) sortbuf[lane + 64 * q] = keys[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int newc = kept < k ? kept : k;
    for (int e = lane; e < newc; e += 64) list[e] = sortbuf[e];
    *tau_out = newc >= k ? key_score(sortbuf[k - 1]) : -INFINITY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return newc;
}

// Developer build only: per-wave cycle counts of the scan's parts, summed into tf_prof.
#ifdef TF_EVAL_PROFILE
__device__ unsigned long long tf_trace[8 * 64 * 4];     // [wave][tile][tile start, blocks done, barrier entered, barrier left] of one workgroup
extern "C" int tfrec_debug_trace(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tf_trace), sizeof(unsigned long long) * 8 * 64 * 4) == hipSuccess ? 0 : -1;
}
__device__ unsigned long long tf_prof[16];
#define TF_PROF_T() __builtin_readcyclecounter()
#define TF_PROF_ADD(i, t0) prof_acc[i] += __builtin_readcyclecounter() - (t0)
extern "C" int tfrec_debug_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tf_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(tf_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define TF_PROF_T() 0ull
#define TF_PROF_ADD(i, t0) (void)(t0)
#endif

// Inclusive sum over the lanes 0..lane of a wave, by DPP row shifts and row broadcasts (six vector adds, no LDS trip).
__device__ __forceinline__ int wave_prefix_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

// One level of a histogram selection among the keys a wave holds (EPL per lane).  A key takes part when its score bits from
// `match_shift` up equal `match_value`; its bucket is the `nbits` (<= 6) bits from `shift` up.  Finds the highest bucket L
// such that at least `want` taking-part keys lie in buckets >= L (the caller makes sure that `want` of them take part);
// returns L, *above = keys in buckets > L, *in_bucket = keys in bucket L.  hist: 64 counters in LDS, this wave's own.
// Wave-uniform arguments and results.
template <int EPL>
__device__ __forceinline__ int hist_select_level(const uint64_t (&keys)[EPL], int match_shift, uint32_t match_value, int shift, int nbits, int want,
                                                 volatile uint32_t *hist, int lane, int *above, int *in_bucket)
{
    hist[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t bmask = (1u << nbits) - 1u;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const uint32_t hi = (uint32_t)(keys[q] >> 32);
        if (keys[q] != 0 && (uint32_t)((uint64_t)hi >> match_shift) == match_value) atomicAdd(const_cast<uint32_t *>(hist) + ((hi >> shift) & bmask), 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c = (int)hist[lane];
    const int incl = wave_prefix_sum(c);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int at_or_above = total - incl + c;                     // keys in buckets >= lane
    const uint64_t m = __ballot(at_or_above >= want);
    const int L = m ? 63 - __builtin_clzll(m) : 0;
    *in_bucket = __builtin_amdgcn_readlane(c, L);
    *above = __builtin_amdgcn_readlane(at_or_above, L) - *in_bucket;
    __builtin_amdgcn_wave_barrier();
    return L;
}

// Compaction by selection: like wave_compact, but the k survivors are SELECTED, not sorted -- a list
// only has to be in order when it is written out, and then only its k survivors (wave_finish).  The k-th largest key is found by descending over its
// bits (one wave-uniform count per bit: compares, ballots and scalar population counts; the low
// half -- the id -- only when scores tie at the cut), which costs an order of magnitude fewer
// vector instructions than the sort.  The survivors are written back in lane order.
// A candidate list is two regions of CAP/2 entries, one per lane that holds items of the user (each lane appends to
// its own region without asking the other): region r holds cnt_r entries, the first ver_r of them known not to be
// masked.  INTERLEAVE: survivor j is written to region j & 1, slot j >> 1 (the list itself: both regions get half);
// else to out[j] (the sort buffer of the final pass).
// APPROX (compactions during the scan): the cut is found by two levels of a 64-way histogram over the score bits below the
// highest bit in which the held scores differ, and is the lower edge of a bucket: at least k and in practice k or k + 1 keys
// survive, *tau_out is a lower bound of the k-th best score -- all a filter threshold has to be.  When more than k + 16 keys
// would survive (ties, or scores that differ only in their last bits), the exact descent runs instead.
template <int CAP, bool INTERLEAVE, bool APPROX = false, class Out>
__device__ __forceinline__ int wave_select(const uint64_t *list, int cnt0, int cnt1, int ver0, int ver1, int k,
                                           const int32_t *__restrict__ mask_idx, int64_t mlo, int64_t mhi,
                                           volatile uint64_t *sortbuf, int lane, float *tau_out, Out out,
                                           [[maybe_unused]] unsigned long long *prof_acc = nullptr)
{
    [[maybe_unused]] unsigned long long prof_t = TF_PROF_T();
    constexpr int EPL = CAP / 64;
    constexpr int ROWCAP = CAP * 2;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#ifdef TF_EVAL_PROFILE
    if (prof_acc) { prof_acc[8] += TF_PROF_T() - prof_t; prof_t = TF_PROF_T(); }
#endif
    const int64_t L = mhi - mlo;
    const bool check = mask_idx != nullptr && L > 0 && (ver0 < cnt0 || ver1 < cnt1);
    const bool staged = check && L <= ROWCAP;
    volatile int32_t *rowbuf = reinterpret_cast<volatile int32_t *>(sortbuf);
    constexpr int HQ = EPL / 2;               // register slots per region
    uint64_t keys[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const int e = lane + 64 * (q % HQ);
        keys[q] = e < (q < HQ ? cnt0 : cnt1) ? list[(q / HQ) * (CAP / 2) + e] : 0;
    }
    // short rows are padded to whole groups of four with an id no key carries
    constexpr int LINEAR_MAX = 64;
    const bool linear = staged && L <= LINEAR_MAX;
    if (staged) {
        const int Lp = linear ? (((int)L + 3) & ~3) : (int)L;
        for (int j = lane; j < Lp; j += 64) rowbuf[j] = j < (int)L ? mask_idx[mlo + j] : -2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#ifdef TF_EVAL_PROFILE
    if (prof_acc) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); prof_acc[9] += TF_PROF_T() - prof_t; prof_t = TF_PROF_T(); }
#endif
    if (linear) {
        // every row id against every held id: the row is read four ids at a time (all lanes the same
        // address, one LDS access), independent reads instead of a chain of dependent probes per key
        const int4 *row4 = reinterpret_cast<const int4 *>(const_cast<const int32_t *>(rowbuf));
        bool hit[EPL];
#pragma unroll
        for (int q = 0; q < EPL; ++q) hit[q] = false;
        for (int j = 0; j < (int)L; j += 4) {
            const int4 r = row4[j >> 2];
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                const int32_t id = key_id(keys[q]);
                hit[q] = hit[q] | (id == r.x) | (id == r.y) | (id == r.z) | (id == r.w);
            }
        }
#pragma unroll
        for (int q = 0; q < EPL; ++q)
            if (hit[q]) keys[q] = 0;
    } else if (check) {
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const int e = lane + 64 * (q % HQ);
            if (e < (q < HQ ? cnt0 : cnt1) && e >= (q < HQ ? ver0 : ver1)) {
                const int32_t id = key_id(keys[q]);
                bool hit;
                if (staged) {
                    int lo = 0, hi = (int)L;
                    hit = false;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const int32_t v = rowbuf[mid];
                        if (v == id) { hit = true; break; }
                        if (v < id) lo = mid + 1; else hi = mid;
                    }
                } else {
                    hit = csr_row_contains(mask_idx, mlo, mhi, id);
                }
                if (hit) keys[q] = 0;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();          // every lane is done with the staged row
#ifdef TF_EVAL_PROFILE
    if (prof_acc) { prof_acc[10] += TF_PROF_T() - prof_t; prof_t = TF_PROF_T(); }
#endif
    int kept = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) kept += __popcll(__ballot(keys[q] != 0));
    uint64_t cut = 1;                          // keep every real key while fewer than k are held
    if (kept >= k) {
        // the bits all held scores share need no descent: start below the highest bit in which they differ
        uint32_t all_or = 0, all_and = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const uint32_t hi = (uint32_t)(keys[q] >> 32);
            all_or |= hi;
            all_and &= keys[q] != 0 ? hi : 0xffffffffu;
        }
        all_or |= xor_lane_u32<32>(all_or, lane);  all_and &= xor_lane_u32<32>(all_and, lane);
        all_or |= xor_lane_u32<16>(all_or, lane);  all_and &= xor_lane_u32<16>(all_and, lane);
        all_or |= xor_lane_u32<8>(all_or, lane);   all_and &= xor_lane_u32<8>(all_and, lane);
        all_or |= xor_lane_u32<4>(all_or, lane);   all_and &= xor_lane_u32<4>(all_and, lane);
        all_or |= xor_lane_u32<2>(all_or, lane);   all_and &= xor_lane_u32<2>(all_and, lane);
        all_or |= xor_lane_u32<1>(all_or, lane);   all_and &= xor_lane_u32<1>(all_and, lane);
        const uint32_t differ = __builtin_amdgcn_readfirstlane(all_or ^ all_and);
        bool cut_found = false;
        if constexpr (APPROX) {
            if (differ) {
                volatile uint32_t *hist = reinterpret_cast<volatile uint32_t *>(sortbuf);     // the staged row is no longer needed
                const int top = 31 - __builtin_clz(differ);
                const uint32_t common = __builtin_amdgcn_readfirstlane(all_and);
                // level 1: the six bits from the highest differing one down (every held key shares the bits above them)
                const int shift1 = top >= 5 ? top - 5 : 0;
                int above1, in1;
                const int L1 = hist_select_level<EPL>(keys, shift1 + 6, (uint32_t)((uint64_t)common >> (shift1 + 6)), shift1, 6, k, hist, lane, &above1, &in1);
                uint32_t hi_cut = (uint32_t)(((uint64_t)common >> (shift1 + 6)) << (shift1 + 6)) | ((uint32_t)L1 << shift1);
                int survivors = above1 + in1;
                if (survivors > k && shift1 > 0) {
                    // level 2: the next (up to) six bits, among the keys of the level-1 bucket that holds the k-th
                    const int shift2 = shift1 >= 6 ? shift1 - 6 : 0;
                    int above2, in2;
                    const int L2 = hist_select_level<EPL>(keys, shift1, hi_cut >> shift1, shift2, shift1 - shift2, k - above1, hist, lane, &above2, &in2);
                    hi_cut |= (uint32_t)L2 << shift2;
                    survivors = above1 + above2 + in2;
                }
                if (survivors <= k + 16) {
                    cut = (uint64_t)hi_cut << 32;
                    cut_found = true;
                }
            }
        }
        if (!cut_found) {
        // highest differing bit, moved up to an odd position so that whole pairs of bits remain (a shared
        // bit taken into the descent is simply found again)
        int bit = differ ? ((31 - __builtin_clz(differ)) | 1) : -1;
        uint32_t hi_cut = __builtin_amdgcn_readfirstlane(all_and) & (bit >= 0 ? ~(uint32_t)((2ull << bit) - 1ull) : 0xffffffffu);
        int exact = 0;
        const int may_stop = kept > k ? 1 : 0;        // with exactly k held, all bits are resolved: the cut is then the lowest score
        // two bits per round: three counts, then the largest candidate that still has k keys at or above it
#pragma unroll 1
        for (; bit >= 1; bit -= 2) {
#ifdef TF_EVAL_PROFILE
            if (prof_acc) prof_acc[13] += 1;
#endif
            const uint32_t step = 1u << (bit - 1);
            const uint32_t t1 = hi_cut + step, t2 = hi_cut + 2 * step, t3 = hi_cut + 3 * step;
            // all compares first, each into its own mask, then the scalar counts: one shared condition
            // register would chain every compare to the count before it
            uint64_t m1[EPL], m2[EPL], m3[EPL];
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                const uint32_t hi = (uint32_t)(keys[q] >> 32);
                m1[q] = __ballot(hi >= t1);
                m2[q] = __ballot(hi >= t2);
                m3[q] = __ballot(hi >= t3);
            }
            __builtin_amdgcn_sched_barrier(0);
            int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                c1 += __popcll(m1[q]);
                c2 += __popcll(m2[q]);
                c3 += __popcll(m3[q]);
            }
            // the counts fall from c1 to c3, so the number of candidates that still have k keys is the choice
            const uint32_t sel = (uint32_t)(c1 >= k) + (uint32_t)(c2 >= k) + (uint32_t)(c3 >= k);
            const int cs = sel == 3u ? c3 : (sel == 2u ? c2 : c1);
            hi_cut += sel * step;
            exact = sel != 0u ? (cs == k ? 1 : 0) : exact;
            if (exact & may_stop) break;
        }
        cut = (uint64_t)hi_cut << 32;
        if (!exact) {                          // equal scores may straddle the cut: the id decides
#ifdef TF_EVAL_PROFILE
            if (prof_acc) prof_acc[14] += 1;
#endif
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const uint64_t t2 = cut | (1ull << bit);
                int c2 = 0;
#pragma unroll
                for (int q = 0; q < EPL; ++q) c2 += __popcll(__ballot(keys[q] >= t2));
                if (c2 >= k) {
                    cut = t2;
                    if (c2 == k) break;
                }
            }
        }
        }   // !cut_found
    }
#ifdef TF_EVAL_PROFILE
    if (prof_acc) { prof_acc[11] += TF_PROF_T() - prof_t; prof_t = TF_PROF_T(); }
#endif
    int base = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const bool keep = keys[q] >= cut;
        const uint64_t m = __ballot(keep);
        const int j = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) out[INTERLEAVE ? (j & 1) * (CAP / 2) + (j >> 1) : j] = keys[q];
        base += __popcll(m);
    }
    *tau_out = kept >= k ? key_score(cut) : -INFINITY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef TF_EVAL_PROFILE
    if (prof_acc) { prof_acc[12] += TF_PROF_T() - prof_t; }
#endif
    return base;
}

// End of the scan: select the k survivors into the wave's LDS buffer (the staged row is no longer
// needed once the selection has its keys in registers), sort just those, leave them in sortbuf[0..n).
template <int CAP>
__device__ __forceinline__ int wave_finish(const uint64_t *__restrict__ list, int cnt0, int cnt1, int ver0, int ver1, int k,
                                           const int32_t *__restrict__ mask_idx, int64_t mlo, int64_t mhi,
                                           volatile uint64_t *sortbuf, int lane)
{
    constexpr int SN = (CAP / 2 & (CAP / 2 - 1)) == 0 ? CAP / 2 : 128;     // a power of two (the sort); k <= SN - 16 for every capacity in use
    float tau;
    // a list that fits the sort as it is needs no selection first (asking for more survivors than it can
    // hold makes the selection keep every unmasked entry); the sort then puts the best k in front
    const bool small = cnt0 + cnt1 <= SN;
    const int held = wave_select<CAP, false>(list, cnt0, cnt1, ver0, ver1, small ? CAP : k, mask_idx, mlo, mhi, sortbuf, lane, &tau, sortbuf);
    const int nc = held < k ? held : k;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint64_t keys[SN / 64];
#pragma unroll
    for (int q = 0; q < SN / 64; ++q) {
        const int e = lane + 64 * q;
        keys[q] = e < held ? sortbuf[e] : 0;
    }
    bitonic_sort<SN, SN>(keys, lane);
#pragma unroll
    for (int q = 0; q < SN / 64; ++q) sortbuf[lane + 64 * q] = keys[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return nc;
}

// ------------------------------------------------------------------------------------------
// item table preparation: padded float32 copy or bfloat16 copy of rows [item_lo, item_hi)
// ------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void prep_items_kernel(const float *__restrict__ Q, int64_t item_lo, int64_t n_items, int d, int D,
                                                         void *__restrict__ out)
{
    const int64_t total = n_items * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / D;
        const int f = (int)(e - r * D);
        const float v = f < d ? Q[(item_lo + r) * d + f] : 0.f;
        if constexpr (BF16) reinterpret_cast<__bf16 *>(out)[e] = (__bf16)v;
        else reinterpret_cast<float *>(out)[e] = v;
    }
}

// ------------------------------------------------------------------------------------------
// fused scoring + top-k
// ------------------------------------------------------------------------------------------
struct ScoreArgs {
    const float *P;            // [U, d]
    const void *Qt;            // prepared item rows [n_items, D] (float or bf16), row 0 = item_lo
    const float *b;            // [I] (global ids)
    int d;
    int64_t item_lo, n_items;  // shard
    int64_t split_items;       // items per grid.y slice (multiple of the tile)
    const int32_t *users;      // [n] evaluated users
    int64_t n;
    const int64_t *train_indptr;
    const int32_t *train_indices;
    int k;
    int p_compact;             // 1: row r of P belongs to users[r]; 0: P is the full table, indexed by user id
    int64_t id_offset;         // added to shard-relative item positions to form the reported (global) ids
    uint64_t *lists;           // [gridDim.y, n, CAP]
    int32_t *out_ids;          // [gridDim.y, n, k]
    float *out_scores;         // [gridDim.y, n, k]
};

// UW = sets of 32 users per wave.  Every wave reads the whole item tile from LDS; with two sets the
// same A fragment feeds two matrix instructions, which halves the LDS traffic per flop.  Measured on
// MI355X with bf16: +10% when the users of a wave insert at the same tiles (d=64), -15% when they do
// not (twice the compactions queue up behind one wave), and at d=128 the registers of the second
// set leave one wave per SIMD.  One set is the setting in use.
template <int D, bool BF16, int CAP>
struct ScoreWave { static constexpr int UW = 1, NW = 4; };
// NW = waves per workgroup.  All of them share the staged item tile, so eight waves halve the staging work
// (global loads, LDS writes, barriers) per user.  Measured on MI355X, eight against four waves: bf16 d=128
// +35 % (2M items, k=100), bf16 d=256 +10 %, f32 d=64 +4 %, f32 d=128 unchanged.  Sixteen waves (one
// workgroup per CU) where the kernel fits 128 registers: bf16 d=64 +22 % on a trained model and +4 to +8 % on
// synthetic ones, bf16 d=128 with lists of 256 a further +4 %; f32 d=64 loses 3 % and stays at eight.
template <int CAP>
struct ScoreWave<64, true, CAP> { static constexpr int UW = 1, NW = 16; };   // 125 registers: sixteen waves fit a CU
#ifndef TF_SW128_UW
#define TF_SW128_UW 1
#endif
#ifndef TF_SW128_NW
#define TF_SW128_NW 16
#endif
template <int CAP>
struct ScoreWave<128, true, CAP> { static constexpr int UW = CAP <= 256 ? TF_SW128_UW : 1, NW = CAP <= 256 ? TF_SW128_NW : 8; };
template <int CAP>
struct ScoreWave<256, true, CAP> { static constexpr int UW = 1, NW = 8; };
template <int CAP>
struct ScoreWave<64, false, CAP> { static constexpr int UW = 1, NW = 8; };

// LDS bytes of an instantiation with tiles of `ti` items
constexpr int score_lds_bytes(int ti, int rowb, int nw, int cap) { return 2 * ti * rowb + 2 * ti * 4 + nw * cap * 8 + 1536; }

template <int D, bool BF16, int CAP>
struct ScoreGeom {
    static constexpr int ESZ = BF16 ? 2 : 4;
    static constexpr int ROWB = D * ESZ + 16;                // padded LDS row (bank spread)
    // items per step; a workgroup of sixteen waves is alone on its CU and takes tiles three times as long (a third of
    // the barriers) where the CU's 160 KB hold two of them beside the sort buffers, else twice as long
    static constexpr int TI0 = D * ESZ >= 256 ? 64 : 128;
    static constexpr bool ALONE = ScoreWave<D, BF16, CAP>::NW == 16 || ScoreWave<D, BF16, CAP>::UW > 1;   // one workgroup per CU
    static constexpr int TI = !ALONE ? TI0
                              : (score_lds_bytes(3 * TI0, ROWB, ScoreWave<D, BF16, CAP>::NW, CAP) <= 160 * 1024 &&
                                 (3 * TI0 * (D * ESZ / 16)) % (64 * ScoreWave<D, BF16, CAP>::NW) == 0 ? 3 * TI0 : 2 * TI0);
    static constexpr int CPR = D * ESZ / 16;                 // 16-byte chunks per row
    static constexpr int NW = ScoreWave<D, BF16, CAP>::NW, NT = 64 * NW;   // waves / threads per workgroup
    static constexpr int NLD = TI * CPR / NT;                // chunks per thread per tile
    static_assert(NLD >= 1 && NLD * NT == TI * CPR && NT >= TI, "tile does not divide over the workgroup");
    static constexpr int QBYTES = TI * ROWB;
    static constexpr int LDS_BYTES = 2 * QBYTES + 2 * TI * 4 + NW * CAP * 8 + 1536;
};

// Developer ablations of the scan loop (never defined in the shipped build): 1 = no list filter,
// 2 = also no LDS fragment reads, 3 = also no tile staging, 4 = also no final pass, 5 = also no
// per-tile maximum, 6 = 5 with the matrix work issued as 16x16x32 instructions (bf16; a clock comparison).
#ifndef TF_EVAL_ABLATE
#define TF_EVAL_ABLATE 0
#endif
// compactions during the scan cut by histogram (see wave_select); 0 = by the exact descent
#ifndef TF_EVAL_APPROX_SELECT
#define TF_EVAL_APPROX_SELECT 1
#endif
// a list half is compacted when fewer than this many free entries are left (16 = the most a lane appends per block:
// as late as possible; more = earlier compactions, tighter thresholds)
#ifndef TF_EVAL_TRIGGER_SLACK
#define TF_EVAL_TRIGGER_SLACK 16
#endif
static_assert(TF_EVAL_TRIGGER_SLACK >= 16, "a lane may append all sixteen items of a block before the next check");

template <int D, bool BF16, int CAP>
__global__ __launch_bounds__((64 * ScoreWave<D, BF16, CAP>::NW)) void score_topk_kernel(ScoreArgs a)
{
    using G = ScoreGeom<D, BF16, CAP>;
    constexpr int TI = G::TI, ROWB = G::ROWB, CPR = G::CPR, NLD = G::NLD, ESZ = G::ESZ;
    constexpr int UW = ScoreWave<D, BF16, CAP>::UW, NT = G::NT, UB = 32 * G::NW * UW;   // user sets per wave, threads and users per workgroup
    extern __shared__ __align__(16) char lds[];
    char *qbuf = lds;                                                   // [2][TI*ROWB]
    float *bias = reinterpret_cast<float *>(lds + 2 * G::QBYTES);        // [2][TI]
    uint64_t *sortbuf = reinterpret_cast<uint64_t *>(bias + 2 * TI);     // [waves][CAP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uj = lane & 31, h = lane >> 5;
    [[maybe_unused]] unsigned long long prof_acc[16] = {};
    [[maybe_unused]] const unsigned long long prof_t_start = TF_PROF_T();
    const int64_t wave_user0 = (int64_t)blockIdx.x * UB + wave * 32 * UW;   // first evaluated-user ordinal of the wave
    const int64_t it0 = (int64_t)blockIdx.y * a.split_items;             // first item (shard-relative)
    const int64_t it1 = it0 + a.split_items < a.n_items ? it0 + a.split_items : a.n_items;

    // Per user set: candidate-list state of the lane's user.  The two lanes that hold items of one user (lane and
    // lane^32) each append to their own half of the user's list and keep its count and the count of its entries
    // already checked against the train row; both keep the threshold (the user's k-th best score so far; +inf
    // keeps padding users out).
    uint64_t *mylist[UW];
    int64_t mlo[UW], mhi[UW];
    int mycnt[UW], myver[UW];
    float th[UW];
    float pf[UW][BF16 ? 1 : D / 2];
    bf16x8 pb[UW][BF16 ? D / 16 : 1];
#pragma unroll
    for (int s = 0; s < UW; ++s) {
        const int64_t ur = wave_user0 + s * 32 + uj;                     // evaluated-user ordinal
        const bool uvalid = ur < a.n;
        const int64_t urc = uvalid ? ur : a.n - 1;
        const int32_t user = a.users[urc];
        mylist[s] = a.lists + ((size_t)blockIdx.y * a.n + urc) * CAP + h * (CAP / 2);      // this lane's region of the user's list
        mlo[s] = a.train_indptr ? a.train_indptr[user] : 0;
        mhi[s] = a.train_indptr ? a.train_indptr[user + 1] : 0;
        mycnt[s] = 0; myver[s] = 0;
        th[s] = uvalid ? -INFINITY : INFINITY;
        // user factors -> B-operand fragments, resident for the whole scan
        const float *prow = a.P + (size_t)(a.p_compact ? urc : (int64_t)user) * a.d;
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = 16 * ks + 8 * h + j;
                    v[j] = (__bf16)(f < a.d ? prow[f] : 0.f);
                }
                pb[s][ks] = v;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < D / 2; ++kk) {
                const int f = 2 * kk + h;
                pf[s][kk] = f < a.d ? prow[f] : 0.f;
            }
        }
    }

    // staging: thread -> NLD 16-byte chunks of the tile
    uint4 stg[NLD];
    float bstg = 0.f;
    auto load_tile = [&](int64_t item_base) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int c = tid + NT * m;
            const int row = c / CPR, col = c % CPR;
            const int64_t it = item_base + row;
            stg[m] = it < it1 ? reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.Qt) + (size_t)it * D * ESZ)[col]
                              : make_uint4(0, 0, 0, 0);
        }
        if (tid < TI) {
            const int64_t it = item_base + tid;
            bstg = it < it1 ? a.b[a.item_lo + it] : -INFINITY;    // -inf keeps padded items out of every list
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int c = tid + NT * m;
            const int row = c / CPR, col = c % CPR;
            *reinterpret_cast<uint4 *>(qbuf + buf * G::QBYTES + row * ROWB + col * 16) = stg[m];
        }
        if (tid < TI) bias[buf * TI + tid] = bstg;
    };

    volatile uint64_t *mysort = sortbuf + wave * CAP;
    const int lane_row_off = uj * ROWB, lane_bias_off = 4 * h;     // this lane's place in an LDS tile

    load_tile(it0);
    store_tile(0);
    __syncthreads();
    const int64_t n_steps = (it1 - it0 + TI - 1) / TI;
    for (int64_t step = 0; step < n_steps; ++step) {
        const int buf = (int)(step & 1);
        if (TF_EVAL_ABLATE < 3 && step + 1 < n_steps) load_tile(it0 + (step + 1) * TI);
        const char *qb = qbuf + buf * G::QBYTES;
        const float *bb = bias + buf * TI;
#pragma unroll 1
        for (int sb = 0; sb < TI / 32; ++sb) {
            f32x16 acc[UW];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(bb + lane_bias_off + sb * 32 + 8 * g);
#pragma unroll
                for (int s = 0; s < UW; ++s) {
                    acc[s][4 * g + 0] = bv.x; acc[s][4 * g + 1] = bv.y; acc[s][4 * g + 2] = bv.z; acc[s][4 * g + 3] = bv.w;
                }
            }
            const char *arow = qb + lane_row_off + sb * (32 * ROWB);
            if constexpr (BF16) {
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {
                    const bf16x8 av = TF_EVAL_ABLATE >= 2 ? pb[0][ks] : *reinterpret_cast<const bf16x8 *>(arow + (16 * ks + 8 * h) * 2);
#if TF_EVAL_ABLATE >= 6
                    {   // timing only: the same flops as two 16x16x32 instructions (results meaningless)
                        typedef float f32x4_t __attribute__((ext_vector_type(4)));
                        f32x4_t a0 = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]}, a1 = {acc[0][4], acc[0][5], acc[0][6], acc[0][7]};
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[0][ks], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb[0][ks], a1, 0, 0, 0);
                        acc[0][0] = a0[0]; acc[0][1] = a0[1]; acc[0][2] = a0[2]; acc[0][3] = a0[3];
                        acc[0][4] = a1[0]; acc[0][5] = a1[1]; acc[0][6] = a1[2]; acc[0][7] = a1[3];
                        continue;
                    }
#endif
#pragma unroll
                    for (int s = 0; s < UW; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, pb[s][ks], acc[s], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < D / 4; ++c) {
                    const float4 av = *reinterpret_cast<const float4 *>(arow + c * 16);
#pragma unroll
                    for (int s = 0; s < UW; ++s) {
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? av.y : av.x, pf[s][2 * c], acc[s], 0, 0, 0);
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? av.w : av.z, pf[s][2 * c + 1], acc[s], 0, 0, 0);
                    }
                }
            }
            const int32_t id0 = (int32_t)(a.id_offset + a.item_lo + it0 + step * TI + sb * 32 + 4 * h);
#pragma unroll
            for (int s = 0; s < UW; ++s) {
                // epilogue: filter the lane's 16 items against its user's threshold.  Passing scores are
                // rare per lane but not per wave, so every test is first made wave-uniform (a scalar
                // branch on a ballot): groups of four by their maximum, then the members of a group.
                if (TF_EVAL_ABLATE >= 5) {
                    if (acc[s][0] == 12345.678f) mylist[s][0] = make_key(acc[s][5], id0);
                    continue;
                }
                float gm[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    gm[g] = fmaxf(fmaxf(acc[s][4 * g], acc[s][4 * g + 1]), fmaxf(acc[s][4 * g + 2], acc[s][4 * g + 3]));
                const float mx = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
                if (TF_EVAL_ABLATE >= 1) {
                    if (mx == 12345.678f) mylist[s][0] = make_key(mx, id0);
                    continue;
                }
                [[maybe_unused]] const unsigned long long prof_t_ins = TF_PROF_T();
                if (__ballot(mx > th[s]) != 0) {
                    prof_acc[4] += 1;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (__ballot(gm[g] > th[s]) != 0) {
#pragma unroll
                            for (int r = 4 * g; r < 4 * g + 4; ++r) {
                                if (acc[s][r] > th[s]) {        // per lane: an append asks nobody
                                    mylist[s][mycnt[s]] = make_key(acc[s][r], id0 + (r & 3) + 8 * (r >> 2));
                                    mycnt[s] += 1;
                                }
                            }
                        }
                    }
                } else {
                    continue;                  // nothing appended: no list grew
                }
                TF_PROF_ADD(1, prof_t_ins);
                // a list that could overflow during the next 32 items is compacted now
                // (a lane adds at most its 16 items of a block of 32 to its half)
                uint64_t need = __ballot(mycnt[s] > CAP / 2 - TF_EVAL_TRIGGER_SLACK);
                need = (need | (need >> 32)) & 0xffffffffull;          // by user
                [[maybe_unused]] const unsigned long long prof_t_cmp = TF_PROF_T();
                while (need) {
                    prof_acc[5] += 1;
                    const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(need));
                    need &= need - 1;
                    const int64_t sur = wave_user0 + s * 32 + src;
                    uint64_t *sl = a.lists + ((size_t)blockIdx.y * a.n + sur) * CAP;
                    const int64_t slo = __shfl(mlo[s], src, 64), shi = __shfl(mhi[s], src, 64);
                    float nt;
                    // A user with few masked (train) items is not checked against them during the scan: at most
                    // L of the held entries can be masked, so the best k + L unchecked ones contain the best k
                    // valid ones and their lowest score bounds the k-th valid score from below.  The final pass
                    // checks whatever is left.  (Saves the row fetch and the membership tests of every
                    // mid-scan compaction; the threshold is a few ranks looser.)
                    const int64_t L = shi - slo;
                    const bool defer = a.train_indices != nullptr && a.k + L <= CAP / 2;
                    const int nc = wave_select<CAP, true, TF_EVAL_APPROX_SELECT != 0>(sl, __builtin_amdgcn_readlane(mycnt[s], src), __builtin_amdgcn_readlane(mycnt[s], src + 32),
                                                          __builtin_amdgcn_readlane(myver[s], src), __builtin_amdgcn_readlane(myver[s], src + 32),
                                                          defer ? a.k + (int)L : a.k, defer ? nullptr : a.train_indices, slo, shi, mysort, lane,
                                                          &nt, sl, prof_acc);
                    // the survivors went alternately to the two halves
                    if (uj == src) { mycnt[s] = (nc + 1 - h) >> 1; myver[s] = defer ? 0 : mycnt[s]; th[s] = nt; }
                }
                TF_PROF_ADD(2, prof_t_cmp);
            }
        }
        if (TF_EVAL_ABLATE < 3) {
            [[maybe_unused]] const unsigned long long prof_t_bar = TF_PROF_T();
            if (step + 1 < n_steps) store_tile(buf ^ 1);
            __syncthreads();
            TF_PROF_ADD(3, prof_t_bar);
        }
    }
    [[maybe_unused]] const unsigned long long prof_t_fin = TF_PROF_T();

    // final compaction and output: the wave's users one after the other
    if (TF_EVAL_ABLATE >= 4) return;
#pragma unroll
    for (int s = 0; s < UW; ++s) {
        for (int src = 0; src < 32; ++src) {
            const int64_t sur = wave_user0 + s * 32 + src;
            if (sur >= a.n) break;     // wave-uniform
            uint64_t *sl = a.lists + ((size_t)blockIdx.y * a.n + sur) * CAP;
            const int64_t slo = __shfl(mlo[s], src, 64), shi = __shfl(mhi[s], src, 64);
            const int nc = wave_finish<CAP>(sl, __builtin_amdgcn_readlane(mycnt[s], src), __builtin_amdgcn_readlane(mycnt[s], src + 32),
                                            __builtin_amdgcn_readlane(myver[s], src), __builtin_amdgcn_readlane(myver[s], src + 32), a.k,
                                            a.train_indices, slo, shi, mysort, lane);
            int32_t *oi = a.out_ids + ((size_t)blockIdx.y * a.n + sur) * a.k;
            float *os = a.out_scores + ((size_t)blockIdx.y * a.n + sur) * a.k;
            for (int e = lane; e < a.k; e += 64) {
                const uint64_t key = e < nc ? mysort[e] : 0;
                oi[e] = e < nc ? key_id(key) : -1;
                os[e] = e < nc ? key_score(key) : -INFINITY;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
#ifdef TF_EVAL_PROFILE
    TF_PROF_ADD(6, prof_t_fin);
    TF_PROF_ADD(0, prof_t_start);
    if (lane == 0)
        for (int i = 0; i < 16; ++i) if (i != 7) atomicAdd(&tf_prof[i], prof_acc[i]);
    if (lane == 0) atomicAdd(&tf_prof[7], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------
// bf16 scoring with 64 users per wave (d = 128; d = 64 on request)
//
// In score_topk_kernel a wave's filter work and its LDS reads sit between its matrix instructions, and four waves per
// SIMD are meant to fill each other's gaps; the end of every tile lines them up again.  Here a wave owns 64 users (two
// B-operand sets: every A fragment read from LDS feeds two matrix instructions), a workgroup is eight waves (two per
// SIMD, up to 256 registers each), and the instruction stream of ONE wave keeps the matrix pipe fed by itself:
//     phase A of a block of 32 items: the 8 matrix instructions of user set 0; between them the group maxima of set 1's
//                                     scores of the block before, and one direct-to-LDS load of the next tile;
//     phase B:                        the 8 matrix instructions of set 1; between them the group maxima of set 0's scores
//                                     of this block and the LDS reads of the next block's A fragments (each fragment
//                                     register is refilled right after its last use) and biases.
// Measured on MI355X (tools/mfma_valu_overlap.hip): a SIMD issues about five vector instructions per matrix instruction
// without delaying the matrix pipe, and its arbiter serves the older wave first.  So as long as no score passes its
// user's threshold, the matrix instructions of consecutive blocks follow each other without a gap; only a passing score
// leaves that stream (appends, compactions: the epilogue of score_topk_kernel), and then the other wave of the SIMD has
// the pipe.  Item tiles go from global memory straight into LDS (global_load_lds_dwordx4: no staging registers, no
// store pass at the end of a tile), one kilobyte per wave and block.
// Results are those of score_topk_kernel<128, true, CAP> exactly: the same products in the same order per score, the
// same filter rule, the same total order.
// ------------------------------------------------------------------------------------------
// 0: never; 1: for d = 128; 2: for d = 64 too (measured slower there than score_topk_kernel: a block is then four matrix
// instructions per user set, half the shadow for the same filter work)
#ifndef TF_EVAL_PP
#define TF_EVAL_PP 1
#endif
#ifndef TF_EVAL_PP_TI
#define TF_EVAL_PP_TI 256
#endif
// developer ablations (never defined in the shipped build): 1 = no score passes the filter, 2 = also no fragment reads
// inside a tile, 3 = also no bias reads inside a tile
#ifndef TF_EVAL_PP_ABLATE
#define TF_EVAL_PP_ABLATE 0
#endif
template <int D, int CAP>
struct PpGeom {
    static constexpr int UW = 2, NW = 8, NT = 64 * NW, UB = 32 * NW * UW;   // user sets per wave, waves, threads, users per workgroup
    static constexpr int KS = D / 16;                        // matrix instructions per set and block
    static constexpr int ROWB = D * 2;                       // LDS rows are not padded (see load_rows)
    static constexpr int CPR = ROWB / 16, RPI = 1024 / ROWB; // 16-byte chunks per row; rows per direct-to-LDS instruction of a wave
    // items per tile: as many as fit twice beside the sort buffers, a divisor of the item-split granularity (768)
    static constexpr int TI = D == 128 ? (CAP <= 384 ? TF_EVAL_PP_TI : 192) : 384, NB = TI / 32;   // and blocks per tile
    static constexpr int IPW = TI / (RPI * NW), BPI = NB / IPW;             // load instructions per wave and tile; blocks per instruction
    static_assert(KS == 8 || KS == 4, "the interleaving below is written for eight or four matrix instructions per phase");
    static_assert(NB >= 2 && IPW * RPI * NW == TI && BPI * IPW == NB && NT >= TI && 768 % TI == 0, "tile does not divide over the workgroup");
    static constexpr int QBYTES = TI * ROWB;
    static constexpr int LDS_BYTES = score_lds_bytes(TI, ROWB, NW, CAP);
    // the fragment refills of a tile's last block read one block past the tile (discarded): still inside the allocation
    static_assert(LDS_BYTES <= 160 * 1024 && 32 * ROWB <= 2 * TI * 4 + NW * CAP * 8, "tile buffers and sort buffers exceed the CU's LDS");
};

// a pointer into global memory that keeps its address space through pointer arithmetic
template <class T>
using global_ptr = __attribute__((address_space(1))) T *;

template <int D, int CAP>
__global__ __launch_bounds__(512) void score_topk_pp_kernel(ScoreArgs a)
{
    using G = PpGeom<D, CAP>;
    constexpr int TI = G::TI, ROWB = G::ROWB, NB = G::NB, KS = G::KS, UW = G::UW, CPR = G::CPR, RPI = G::RPI, IPW = G::IPW, BPI = G::BPI;
    extern __shared__ __align__(16) char lds[];
    char *qbuf = lds;                                                   // [2][TI*ROWB]
    float *bias = reinterpret_cast<float *>(lds + 2 * G::QBYTES);        // [2][TI]
    uint64_t *sortbuf = reinterpret_cast<uint64_t *>(bias + 2 * TI);     // [waves][CAP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uj = lane & 31, h = lane >> 5;
    [[maybe_unused]] unsigned long long prof_acc[16] = {};
    [[maybe_unused]] const unsigned long long prof_t_start = TF_PROF_T();
    const int64_t wave_user0 = (int64_t)blockIdx.x * G::UB + wave * 32 * UW;
    const int64_t it0 = (int64_t)blockIdx.y * a.split_items;
    const int64_t it1 = it0 + a.split_items < a.n_items ? it0 + a.split_items : a.n_items;

    // per user set, as in score_topk_kernel: the lane's half of its user's candidate list, its counts, the threshold
    // (the user's train row is looked up when a compaction needs it, not held)
    uint64_t *mylist[UW];
    int32_t myuser[UW];
    int mycnt[UW], myver[UW];
    float th[UW];
    bf16x8 pb[UW][KS];
#pragma unroll
    for (int s = 0; s < UW; ++s) {
        const int64_t ur = wave_user0 + s * 32 + uj;
        const bool uvalid = ur < a.n;
        const int64_t urc = uvalid ? ur : a.n - 1;
        myuser[s] = a.users[urc];
        mylist[s] = a.lists + ((size_t)blockIdx.y * a.n + urc) * CAP + h * (CAP / 2);
        mycnt[s] = 0; myver[s] = 0;
        th[s] = uvalid && TF_EVAL_PP_ABLATE == 0 ? -INFINITY : INFINITY;
        const float *prow = a.P + (size_t)(a.p_compact ? urc : (int64_t)myuser[s]) * a.d;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = 16 * ks + 8 * h + j;
                v[j] = (__bf16)(f < a.d ? prow[f] : 0.f);
            }
            pb[s][ks] = v;
        }
    }

    // Staging of a tile straight into LDS: one global_load_lds_dwordx4 of a wave moves a kilobyte, RPI rows of the tile
    // (lane l: row l / CPR, 16-byte chunk l % CPR of the LDS image); wave w moves rows RPI * (8 * m + w) ... of a tile
    // with its m-th instruction.  The LDS image has no row padding (the instruction writes its kilobyte contiguously), so
    // it is XOR-swizzled against bank conflicts: chunk position c of row r holds the row's chunk c ^ swz(r), swz(r) =
    // r & 15 for rows of 256 bytes and (r >> 1) & 7 for rows of 128 (sixteen consecutive rows then spread every fragment
    // read over all banks); the swizzle is on the source address.  Rows past the end of the item range read the last row
    // (their bias of -inf keeps them out).  The compiler does not see these loads (inline assembly): it neither counts
    // them on the memory counter nor guards LDS reads against them; the waits are the explicit ones at the end of a tile.
    auto swz = [](int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); };
    const int rowlane = RPI * wave + lane / CPR;            // this lane's row among the RPI * 8 rows one instruction round moves
    const global_ptr<const char> qsrc = (global_ptr<const char>)(reinterpret_cast<const char *>(a.Qt)) + (((lane % CPR) ^ swz(rowlane)) * 16);
    const int last_item = (int)(it1 - 1);
    const uint32_t lds_q = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)qbuf;
    auto load_rows = [&](int item_base, int m, int buf) {
        int it = item_base + RPI * 8 * m + rowlane;
        it = it < last_item ? it : last_item;
        const global_ptr<const char> src = qsrc + (uint64_t)(uint32_t)it * ROWB;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_q + (uint32_t)(buf * G::QBYTES + (8 * m + wave) * 1024));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
    };
    float bstg = 0.f;
    auto load_bias = [&](int64_t item_base) {
        if (tid < TI) bstg = item_base + tid < it1 ? (a.b + (a.item_lo + item_base))[tid] : -INFINITY;
    };
    // fragment of k-step ks, half h, of row uj of a block: chunk (2 * ks + h) ^ swz(uj) of the row
    int a_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = uj * ROWB + (((swz(uj) ^ h) * 16) ^ (32 * ks));
    volatile uint64_t *mysort = sortbuf + wave * CAP;
    const int lane_bias_off = 4 * h;

    bf16x8 af[KS];            // A fragments of the block whose matrix instructions come next
    f32x16 bv;                // its biases: the accumulators' starting value
    f32x16 acc[UW];
    auto read_bias = [&](f32x16 &to, const float *bb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(bb + 8 * g);
            to[4 * g + 0] = v.x; to[4 * g + 1] = v.y; to[4 * g + 2] = v.z; to[4 * g + 3] = v.w;
        }
    };
    auto group_max = [&](const f32x16 &x, int g) { return fmaxf(fmaxf(x[4 * g], x[4 * g + 1]), fmaxf(x[4 * g + 2], x[4 * g + 3])); };
    // Scores of block nb of user set S passed their thresholds (gm: the lane's maxima over groups of four): appends, and
    // compaction of every list that could overflow during the next block -- the epilogue of score_topk_kernel.
    auto insert = [&]<int S>(int64_t nb, const float (&gm)[4]) {
        [[maybe_unused]] const unsigned long long prof_t_ins = TF_PROF_T();
        prof_acc[4] += 1;
        const int32_t id0 = (int32_t)(a.id_offset + a.item_lo + it0 + nb * 32 + 4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (__ballot(gm[g] > th[S]) != 0) {
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) {
                    if (acc[S][r] > th[S]) {        // per lane: an append asks nobody
                        mylist[S][mycnt[S]] = make_key(acc[S][r], id0 + (r & 3) + 8 * (r >> 2));
                        mycnt[S] += 1;
                    }
                }
            }
        }
        uint64_t need = __ballot(mycnt[S] > CAP / 2 - TF_EVAL_TRIGGER_SLACK);
        need = (need | (need >> 32)) & 0xffffffffull;          // by user
        TF_PROF_ADD(14, prof_t_ins);
        [[maybe_unused]] const unsigned long long prof_t_cmp = TF_PROF_T();
        while (need) {
            prof_acc[5] += 1;
            const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(need));
            need &= need - 1;
            const int64_t sur = wave_user0 + S * 32 + src;
            uint64_t *sl = a.lists + ((size_t)blockIdx.y * a.n + sur) * CAP;
            const int32_t su = __builtin_amdgcn_readlane(myuser[S], src);
            const int64_t slo = a.train_indptr ? a.train_indptr[su] : 0, shi = a.train_indptr ? a.train_indptr[su + 1] : 0;
            float nt;
            // few train items: not checked during the scan (see score_topk_kernel)
            const int64_t L = shi - slo;
            const bool defer = a.train_indices != nullptr && a.k + L <= CAP / 2;
            const int nc = wave_select<CAP, true, TF_EVAL_APPROX_SELECT != 0>(sl, __builtin_amdgcn_readlane(mycnt[S], src), __builtin_amdgcn_readlane(mycnt[S], src + 32),
                                                  __builtin_amdgcn_readlane(myver[S], src), __builtin_amdgcn_readlane(myver[S], src + 32),
                                                  defer ? a.k + (int)L : a.k, defer ? nullptr : a.train_indices, slo, shi, mysort, lane,
                                                  &nt, sl, prof_acc);
            if (uj == src) { mycnt[S] = (nc + 1 - h) >> 1; myver[S] = defer ? 0 : mycnt[S]; th[S] = nt; }
            // vmcnt(0) lgkmcnt(0): no load of the compaction is left "possibly pending" where the scan resumes (the compiler
            // would otherwise guard the scan's registers with waits that also wait for the next tile's global loads, and wait
            // for all of a block's fragment reads where the first would do)
            __builtin_amdgcn_s_waitcnt(0x0070);
        }
        TF_PROF_ADD(15, prof_t_cmp);
    };

    const int64_t n_tiles = (it1 - it0 + TI - 1) / TI;
#pragma unroll
    for (int m = 0; m < IPW; ++m) load_rows((int)it0, m, 0);
    load_bias(it0);
    if (tid < TI) bias[tid] = bstg;
    if (n_tiles > 1) load_bias(it0 + TI);
    __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[1][j] = -INFINITY;      // "the block before the first": nothing passes
    float gm0[4], gm1[4], mx0 = -INFINITY, mx1 = -INFINITY;
    uint64_t pass0 = 0, pass1 = 0;
#pragma unroll 1
    for (int64_t tile = 0; tile < n_tiles; ++tile) {
        const int buf = (int)(tile & 1);
        {
            // in the order the scan requests them (fragments 0 and 1, biases, the other fragments), so that the first
            // matrix instruction of a block waits for what it needs and no more, wherever the block was entered from
            const char *arow = qbuf + buf * G::QBYTES;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                af[ks] = *reinterpret_cast<const bf16x8 *>(arow + a_off[ks]);
                if (ks == 1) read_bias(bv, bias + buf * TI + lane_bias_off);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        [[maybe_unused]] const unsigned long long prof_t_blk = TF_PROF_T();
#ifdef TF_EVAL_PROFILE
        const bool tracing = blockIdx.x == 7 && blockIdx.y == 0 && tile >= 1000 && tile < 1064 && lane == 0;
        if (tracing) tf_trace[(wave * 64 + (tile - 1000)) * 4 + 0] = prof_t_blk;
#endif
#pragma unroll 1
        for (int sb = 0; sb < NB; ++sb) {
            const int64_t nb = tile * NB + sb;
            const char *arow_next = qbuf + buf * G::QBYTES + (sb + 1) * (32 * ROWB);   // past the tile for its last block
            const int next_base = (int)(it0 + (tile + 1) * TI);
            const float *bias_next = bias + buf * TI + (sb + 1) * 32 + lane_bias_off;
            // ---- phase A: set 0 of block nb; maxima of set 1 of block nb - 1
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], pb[0][ks], ks == 0 ? bv : acc[0], 0, 0, 0);
                // (with four matrix instructions per phase the vector work comes two items to a slot)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (ks == 1 + g * KS / 8) gm1[g] = group_max(acc[1], g);
                if (ks == 1 + KS / 2) mx1 = fmaxf(fmaxf(gm1[0], gm1[1]), fmaxf(gm1[2], gm1[3]));
                if (ks == 2 + KS / 2) pass1 = __ballot(mx1 > th[1]);      // decided before the phase's last instructions: no wait at the branch
                if (ks == (KS == 8 ? 6 : 3)) {
                    // this wave's next kilobyte of the next tile
                    if (BPI == 1 || sb % BPI == 0) load_rows(next_base, sb / BPI, buf ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pass1 != 0) [[unlikely]] insert.template operator()<1>(nb - 1, gm1);
            // ---- phase B: set 1 of block nb; maxima of set 0 of block nb; operands of block nb + 1
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], pb[1][ks], ks == 0 ? bv : acc[1], 0, 0, 0);
                // the biases keep registers of their own up to here (else the accumulator is built in theirs and copied out
                // when they are read again); then the next block's are requested
                if (ks == 0) asm volatile("" ::"v"(bv));
                if (TF_EVAL_PP_ABLATE < 2) af[ks] = *reinterpret_cast<const bf16x8 *>(arow_next + a_off[ks]);
                if (ks == 1 && TF_EVAL_PP_ABLATE < 3) read_bias(bv, bias_next);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (ks == (KS == 8 ? 2 + g : 1 + g / 2)) gm0[g] = group_max(acc[0], g);
                if (ks == (KS == 8 ? 6 : 3)) mx0 = fmaxf(fmaxf(gm0[0], gm0[1]), fmaxf(gm0[2], gm0[3]));
                if (KS == 8 && ks == 7) pass0 = __ballot(mx0 > th[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KS != 8) pass0 = __ballot(mx0 > th[0]);
            if (pass0 != 0) [[unlikely]] insert.template operator()<0>(nb, gm0);
        }
        TF_PROF_ADD(1, prof_t_blk);
#ifdef TF_EVAL_PROFILE
        if (tracing) tf_trace[(wave * 64 + (tile - 1000)) * 4 + 1] = TF_PROF_T();
#endif
        // the next tile: this thread's part into the other buffer, visible to all after the barrier
        [[maybe_unused]] const unsigned long long prof_t_bar = TF_PROF_T();
        // this wave's rows of the next tile have landed, and its biases (requested a tile ago) are here
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
        if (tile + 1 < n_tiles && tid < TI) bias[(buf ^ 1) * TI + tid] = bstg;
        if (tile + 2 < n_tiles) load_bias(it0 + (tile + 2) * TI);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TF_PROF_ADD(2, prof_t_bar);
#ifdef TF_EVAL_PROFILE
        if (tracing) tf_trace[(wave * 64 + (tile - 1000)) * 4 + 2] = TF_PROF_T();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TF_PROF_ADD(3, prof_t_bar);
#ifdef TF_EVAL_PROFILE
        if (tracing) tf_trace[(wave * 64 + (tile - 1000)) * 4 + 3] = TF_PROF_T();
#endif
    }
    // set 1 of the last block
#pragma unroll
    for (int g = 0; g < 4; ++g) gm1[g] = group_max(acc[1], g);
    mx1 = fmaxf(fmaxf(gm1[0], gm1[1]), fmaxf(gm1[2], gm1[3]));
    if (__ballot(mx1 > th[1]) != 0) insert.template operator()<1>(n_tiles * NB - 1, gm1);
    [[maybe_unused]] const unsigned long long prof_t_fin = TF_PROF_T();

    // final compaction and output: the wave's users one after the other
#pragma unroll
    for (int s = 0; s < UW; ++s) {
        for (int src = 0; src < 32; ++src) {
            const int64_t sur = wave_user0 + s * 32 + src;
            if (sur >= a.n) break;     // wave-uniform
            uint64_t *sl = a.lists + ((size_t)blockIdx.y * a.n + sur) * CAP;
            const int32_t su = __builtin_amdgcn_readlane(myuser[s], src);
            const int64_t slo = a.train_indptr ? a.train_indptr[su] : 0, shi = a.train_indptr ? a.train_indptr[su + 1] : 0;
            const int nc = wave_finish<CAP>(sl, __builtin_amdgcn_readlane(mycnt[s], src), __builtin_amdgcn_readlane(mycnt[s], src + 32),
                                            __builtin_amdgcn_readlane(myver[s], src), __builtin_amdgcn_readlane(myver[s], src + 32), a.k,
                                            a.train_indices, slo, shi, mysort, lane);
            int32_t *oi = a.out_ids + ((size_t)blockIdx.y * a.n + sur) * a.k;
            float *os = a.out_scores + ((size_t)blockIdx.y * a.n + sur) * a.k;
            for (int e = lane; e < a.k; e += 64) {
                const uint64_t key = e < nc ? mysort[e] : 0;
                oi[e] = e < nc ? key_id(key) : -1;
                os[e] = e < nc ? key_score(key) : -INFINITY;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
#ifdef TF_EVAL_PROFILE
    TF_PROF_ADD(6, prof_t_fin);
    TF_PROF_ADD(0, prof_t_start);
    if (lane == 0)
        for (int i = 0; i < 16; ++i) if (i != 7) atomicAdd(&tf_prof[i], prof_acc[i]);
    if (lane == 0) atomicAdd(&tf_prof[7], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------
// top-k of dense rows: one wave per row (replaces c_top_k_array_index, tools.h:22-30)
// ------------------------------------------------------------------------------------------
template <int CAP>
__global__ __launch_bounds__(256) void topk_dense_kernel(const float *__restrict__ R, int64_t rows, int64_t cols, int k,
                                                         const int32_t *__restrict__ mask_rows, const int64_t *__restrict__ mptr,
                                                         const int32_t *__restrict__ midx, uint64_t *__restrict__ lists,
                                                         int32_t *__restrict__ out_ids, float *__restrict__ out_scores)
{
    __shared__ uint64_t sortbuf[4][CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float *src = R + (size_t)row * cols;
    uint64_t *list = lists + (size_t)row * CAP;
    int64_t mlo = 0, mhi = 0;
    if (mptr) {
        const int64_t mr = mask_rows ? mask_rows[row] : row;
        mlo = mptr[mr];
        mhi = mptr[mr + 1];
    }
    int cnt = 0, ver = 0;
    float tau = -INFINITY;
    for (int64_t c0 = 0; c0 < cols; c0 += 256) {
        if (cnt > CAP - 256) {
            cnt = wave_compact<CAP>(list, cnt, ver, k, midx, mlo, mhi, sortbuf[wave], lane, &tau);
            ver = cnt;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t c = c0 + q * 64 + lane;
            const float v = c < cols ? src[c] : -INFINITY;
            const bool pass = v > tau;
            const uint64_t m = __ballot(pass);
            if (pass) list[cnt + __popcll(m & ((1ull << lane) - 1))] = make_key(v, (int32_t)c);
            cnt += __popcll(m);
        }
    }
    cnt = wave_compact<CAP>(list, cnt, ver, k, midx, mlo, mhi, sortbuf[wave], lane, &tau);
    for (int e = lane; e < k; e += 64) {
        const uint64_t key = e < cnt ? sortbuf[wave][e] : 0;
        out_ids[(size_t)row * k + e] = e < cnt ? key_id(key) : -1;
        if (out_scores) out_scores[(size_t)row * k + e] = e < cnt ? key_score(key) : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------
// merge of G top-k lists per user (item-sharded evaluation, or item splits of one GPU)
// ------------------------------------------------------------------------------------------
template <int CAP>
__global__ __launch_bounds__(256) void topk_merge_kernel(const int32_t *__restrict__ ids, const float *__restrict__ scores, int G,
                                                         int64_t n, int k, int32_t *__restrict__ out_ids,
                                                         float *__restrict__ out_scores)
{
    __shared__ uint64_t sortbuf[4][CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= n) return;
    volatile uint64_t *sb = sortbuf[wave];
    const int total = G * k;
    int kept = 0;
    for (int e = lane; e < CAP; e += 64) {
        uint64_t key = 0;
        if (e < total) {
            const int g = e / k, r = e - g * k;
            const int32_t id = ids[((size_t)g * n + u) * k + r];
            if (id >= 0) key = make_key(scores[((size_t)g * n + u) * k + r], id);
        }
        sb[e] = key;
        kept += __popcll(__ballot(key != 0));
    }
    wave_sort_desc<CAP>(sb, lane);
    const int nc = kept < k ? kept : k;
    for (int e = lane; e < k; e += 64) {
        const uint64_t key = e < nc ? sb[e] : 0;
        out_ids[(size_t)u * k + e] = e < nc ? key_id(key) : -1;
        out_scores[(size_t)u * k + e] = e < nc ? key_score(key) : -INFINITY;
    }
}

// ids still -1 are filled with the user's train items in ascending order (the reference's
// ranking runs into its -inf masked entries when fewer than k items are rankable)
__global__ __launch_bounds__(256) void fill_masked_kernel(int32_t *__restrict__ ids, const int32_t *__restrict__ users, int64_t n,
                                                          int k, const int64_t *__restrict__ tptr, const int32_t *__restrict__ tidx)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    int32_t *row = ids + (size_t)r * k;
    int c = 0;
    while (c < k && row[c] >= 0) ++c;
    const int32_t u = users[r];
    int64_t t = tptr[u];
    const int64_t te = tptr[u + 1];
    for (; c < k && t < te; ++c, ++t) row[c] = tidx[t];
}

// ------------------------------------------------------------------------------------------
// metrics: one thread per user, arithmetic exactly as evaluator/backend/cpp/metrics.h:11-100
// (double inner arithmetic, float running sums and stores)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void metrics_user_kernel(const int32_t *__restrict__ ids, const int32_t *__restrict__ users,
                                                           int64_t n, int k, const int64_t *__restrict__ tptr,
                                                           const int32_t *__restrict__ tidx, const double *__restrict__ inv_log2,
                                                           float *__restrict__ per_user)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int32_t u = users[r];
    const int64_t lo = tptr[u], hi = tptr[u + 1];
    const int n_truth = (int)(hi - lo);
    const int32_t *rank = ids + (size_t)r * k;
    float *pre = per_user + (size_t)r * 5 * k, *rec = pre + k, *apv = pre + 2 * k, *nd = pre + 3 * k, *mr = pre + 4 * k;
    int hits = 0, first = -1;
    float sum_pre = 0.f, dcg = 0.f, idcg = 0.f;
    for (int i = 0; i < k; ++i) {
        const int32_t id = rank[i];
        const bool hit = id >= 0 && csr_row_contains(tidx, lo, hi, id);
        hits += hit;
        const float p = (float)(1.0 * hits / (i + 1));
        pre[i] = p;
        rec[i] = (float)(1.0 * hits / n_truth);
        if (hit) sum_pre += p;
        apv[i] = (float)(1.0 * (double)sum_pre / (n_truth < i + 1 ? n_truth : i + 1));
        if (hit) dcg = (float)((double)dcg + inv_log2[i]);
        idcg = (float)((double)idcg + inv_log2[i]);
        nd[i] = dcg / idcg;
        if (hit && first < 0) first = i;
        mr[i] = first >= 0 ? (float)(1.0 / (first + 1)) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// metric SUMS over users without the per-user rows (what an evaluation needs when nobody asked for
// the [n, 5, k] curves): one lane per user walks its rank list once; for every rank position the
// wave adds up its 64 users -- hits and first hits as ballot counts, the recall and average-precision
// terms (whose weights differ per user) as float64 butterfly sums -- and the four per-position arrays
// of a wave are combined per workgroup, then over workgroups, in a fixed order.  metrics_final_kernel
// turns them into the five curves' sums:
//   precision@t = H_t/(t+1), ndcg@t = DCG_t/IDCG_t and mrr@t are prefix sums of the hit / first-hit
//   counts (evaluator/backend/cpp/metrics.h:11-25, 59-100); recall@t and map@t are summed per user as
//   metrics.h:27-57 rounds them (float per user and position), in float64 over users.
// ------------------------------------------------------------------------------------------
constexpr int METRIC_KMAX = 240;

__global__ __launch_bounds__(256) void metrics_sums_kernel(const int32_t *__restrict__ ids, const int32_t *__restrict__ users,
                                                           int64_t n, int k, const int64_t *__restrict__ tptr,
                                                           const int32_t *__restrict__ tidx, double *__restrict__ block_part)
{
    // per wave and position: [0] hits, [1] first hits, [2] sum of recall terms, [3] sum of average-precision terms
    __shared__ double part[4][4][METRIC_KMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = r < n;
    int64_t lo = 0, hi = 0;
    if (valid) {
        const int32_t u = users[r];
        lo = tptr[u];
        hi = tptr[u + 1];
    }
    const int n_truth = (int)(hi - lo);
    const int32_t *rank = ids + (size_t)(valid ? r : 0) * k;
    int hits = 0;
    bool had = false;
    float sum_pre = 0.f;
    for (int i = 0; i < k; ++i) {
        const int32_t id = valid ? rank[i] : -1;
        const bool hit = id >= 0 && csr_row_contains(tidx, lo, hi, id);
        hits += hit;
        const float p = (float)(1.0 * hits / (i + 1));
        if (hit) sum_pre += p;
        double rec = valid ? (double)(float)(1.0 * hits / n_truth) : 0.0;
        double ap = valid ? (double)(float)(1.0 * (double)sum_pre / (n_truth < i + 1 ? n_truth : i + 1)) : 0.0;
        const int c_hit = __popcll(__ballot(hit));
        const int c_first = __popcll(__ballot(hit && !had));
        had = had || hit;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            rec += __shfl_xor(rec, m, 64);
            ap += __shfl_xor(ap, m, 64);
        }
        if (lane == 0) {
            part[wave][0][i] = (double)c_hit;
            part[wave][1][i] = (double)c_first;
            part[wave][2][i] = rec;
            part[wave][3][i] = ap;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * k; e += 256) {
        const int q = e / k, i = e - q * k;
        block_part[(size_t)blockIdx.x * 4 * k + e] = (part[0][q][i] + part[1][q][i]) + (part[2][q][i] + part[3][q][i]);
    }
}

// block_part [n_blocks][4k] -> sums [5k] (accumulated onto the existing sums when `accumulate`); one workgroup.
__global__ __launch_bounds__(256) void metrics_final_kernel(const double *__restrict__ block_part, int n_blocks, int k,
                                                            const double *__restrict__ inv_log2, double *__restrict__ sums, int accumulate)
{
    __shared__ double col[4][METRIC_KMAX];
    // column sums over workgroups: four partial sums per column (one per wave), combined in a fixed order
    __shared__ double quarter[4][4 * METRIC_KMAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (n_blocks + 3) / 4, b0 = wave * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    for (int e = lane; e < 4 * k; e += 64) {
        double v = 0.0;
        for (int b = b0; b < b1; ++b) v += block_part[(size_t)b * 4 * k + e];
        quarter[wave][e] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * k; e += 256)
        col[e / k][e % k] = (quarter[0][e] + quarter[1][e]) + (quarter[2][e] + quarter[3][e]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double H = 0.0, dcg = 0.0, idcg = 0.0, mrr = 0.0;
        for (int i = 0; i < k; ++i) {
            H += col[0][i];
            dcg += col[0][i] * inv_log2[i];
            idcg += inv_log2[i];
            mrr += col[1][i] / (double)(i + 1);
            const double v[5] = {H / (double)(i + 1), col[2][i], col[3][i], dcg / idcg, mrr};
            for (int m = 0; m < 5; ++m) sums[m * k + i] = (accumulate ? sums[m * k + i] : 0.0) + v[m];
        }
    }
}

// column sums of per_user [n, C] in a fixed order: grid (column chunk, user chunk)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ per_user, int64_t n, int C, int64_t users_per_chunk,
                                                     double *__restrict__ partial)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t u0 = (int64_t)blockIdx.y * users_per_chunk;
    const int64_t u1 = u0 + users_per_chunk < n ? u0 + users_per_chunk : n;
    double s = 0.0;
    for (int64_t u = u0; u < u1; ++u) s += (double)per_user[(size_t)u * C + c];
    partial[(size_t)blockIdx.y * C + c] = s;
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const double *__restrict__ partial, int n_chunks, int C, double *__restrict__ sums,
                                                           int accumulate)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = accumulate ? sums[c] : 0.0;
    for (int j = 0; j < n_chunks; ++j) s += partial[(size_t)j * C + c];
    sums[c] = s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
#ifndef TF_EVAL_CAP256_MAXK
#define TF_EVAL_CAP256_MAXK 112
#endif
static int cap_for_k(int k) { return k <= TF_EVAL_CAP256_MAXK ? 256 : (k <= 240 ? 512 : 0); }
// The second kernel (bf16, padded d = 128) keeps lists of 384 entries for k between 65 and 112: with a hundred survivors a
// list of 256 has room for some fifty appends per half before the next compaction, a list of 384 for more than a hundred
// (half as many compactions for slightly looser thresholds; measured -1 % / -3 % at 2M items, k = 100; its sort buffers
// still fit beside two tiles of 256 items, which lists of 512 do not).  0 = lists of 256 there too.
#ifndef TF_EVAL_CAP384
#define TF_EVAL_CAP384 1
#endif
static int cap_for(int k, bool bf16, int D)
{
    if (TF_EVAL_CAP384 && TF_EVAL_PP && bf16 && D == 128 && k > 64 && k <= 112) return 384;
    return cap_for_k(k);
}

// the dynamic-LDS limit is a per-device attribute of a kernel: set once on every device an instantiation runs on
// (bit = device ordinal; devices beyond 63 set it on every launch)
static int raise_lds_limit(tfrec_ctx *ctx, const void *kernel, int bytes, std::atomic<uint64_t> &done)
{
    const uint64_t dev_bit = ctx->device < 64 ? (1ull << ctx->device) : 0ull;
    if (!(done.load(std::memory_order_acquire) & dev_bit)) {
        TF_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done.fetch_or(dev_bit, std::memory_order_release);
    }
    return TFREC_OK;
}

template <int D, int CAP>
static int launch_score_pp(tfrec_ctx *ctx, const ScoreArgs &a, int splits)
{
    using G = PpGeom<D, CAP>;
    static std::atomic<uint64_t> attr_devices{0};
    TF_TRY(raise_lds_limit(ctx, reinterpret_cast<const void *>(&score_topk_pp_kernel<D, CAP>), G::LDS_BYTES, attr_devices));
    const int user_blocks = (int)((a.n + G::UB - 1) / G::UB);
    score_topk_pp_kernel<D, CAP><<<dim3(user_blocks, splits), G::NT, G::LDS_BYTES, ctx->stream>>>(a);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

template <int D, bool BF16, int CAP>
static int launch_score(tfrec_ctx *ctx, const ScoreArgs &a, int user_blocks, int splits)
{
    if constexpr (TF_EVAL_PP && BF16 && (D == 128 || (D == 64 && TF_EVAL_PP >= 2))) {
        // developer switch, read at every launch: TFREC_AMD_SCORE_KERNEL=1 keeps score_topk_kernel for this shape too
        // (the two kernels give identical lists; tests/test_gpu_eval.py compares them)
        const char *force = std::getenv("TFREC_AMD_SCORE_KERNEL");
        if (!(force && force[0] == '1')) return launch_score_pp<D, CAP>(ctx, a, splits);
    }
    using G = ScoreGeom<D, BF16, CAP>;
    static std::atomic<uint64_t> attr_devices{0};
    TF_TRY(raise_lds_limit(ctx, reinterpret_cast<const void *>(&score_topk_kernel<D, BF16, CAP>), G::LDS_BYTES, attr_devices));
    // the caller counts workgroups of 128 users; this instantiation may hold more
    constexpr int UB = 32 * G::NW * ScoreWave<D, BF16, CAP>::UW;
    user_blocks = (int)((a.n + UB - 1) / UB);
    score_topk_kernel<D, BF16, CAP><<<dim3(user_blocks, splits), G::NT, G::LDS_BYTES, ctx->stream>>>(a);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

template <int CAP>
static int launch_score_cap(tfrec_ctx *ctx, const ScoreArgs &a, int D, int mode, int user_blocks, int splits)
{
    if (mode == TFREC_SCORE_BF16) {
        switch (D) {
            case 32: return launch_score<32, true, CAP>(ctx, a, user_blocks, splits);
            case 64: return launch_score<64, true, CAP>(ctx, a, user_blocks, splits);
            case 128: return launch_score<128, true, CAP>(ctx, a, user_blocks, splits);
            case 256: return launch_score<256, true, CAP>(ctx, a, user_blocks, splits);
        }
    } else {
        switch (D) {
            case 32: return launch_score<32, false, CAP>(ctx, a, user_blocks, splits);
            case 64: return launch_score<64, false, CAP>(ctx, a, user_blocks, splits);
            case 128: return launch_score<128, false, CAP>(ctx, a, user_blocks, splits);
        }
    }
    tfrec_set_error("internal: no scoring kernel for D=%d", D);
    return TFREC_EUNSUP;
}

static int merge_lists(tfrec_ctx *ctx, const int32_t *ids, const float *scores, int G, int64_t n, int k, int32_t *out_ids,
                       float *out_scores)
{
    const int blocks = (int)((n + 3) / 4);
    const int total = G * k;
    if (total <= 256) topk_merge_kernel<256><<<blocks, 256, 0, ctx->stream>>>(ids, scores, G, n, k, out_ids, out_scores);
    else if (total <= 512) topk_merge_kernel<512><<<blocks, 256, 0, ctx->stream>>>(ids, scores, G, n, k, out_ids, out_scores);
    else if (total <= 1024) topk_merge_kernel<1024><<<blocks, 256, 0, ctx->stream>>>(ids, scores, G, n, k, out_ids, out_scores);
    else if (total <= 2048) topk_merge_kernel<2048><<<blocks, 256, 0, ctx->stream>>>(ids, scores, G, n, k, out_ids, out_scores);
    else { tfrec_set_error("merge of %d lists of %d exceeds 2048 entries", G, k); return TFREC_EUNSUP; }
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_topk_merge(tfrec_ctx *ctx, const int32_t *d_ids, const float *d_scores, int G, int64_t n, int k,
                                int32_t *d_out_ids, float *d_out_scores)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(G >= 1 && n >= 0 && k >= 1, TFREC_EINVAL, "bad merge shape G=%d n=%lld k=%d", G, (long long)n, k);
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_ids && d_scores && d_out_ids && d_out_scores, TFREC_EINVAL, "null pointer");
    return merge_lists(ctx, d_ids, d_scores, G, n, k, d_out_ids, d_out_scores);
}

static int pad_dim(int d, int mode)
{
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    if (d <= 256 && mode == TFREC_SCORE_BF16) return 256;
    return 0;
}

struct ScorePlan {
    int CAP, D, TI, splits;
    bool bf16, direct;
    int64_t n_items, split_items, chunk;
    size_t q_bytes, list_bytes, part_bytes, part2_bytes, total;
};

// lists of k one merge launch takes per user (its sort buffer holds 2048 candidates)
static inline int merge_group(int k) { return 2048 / k; }

static int plan_score(const float *d_Q, int d, int64_t item_lo, int64_t item_hi, int64_t n, int k, int score_mode, ScorePlan *p)
{
    TF_REQUIRE(cap_for_k(k) != 0, TFREC_EUNSUP, "top-k above 240 is not supported (k=%d)", k);
    p->D = pad_dim(d, score_mode);
    TF_REQUIRE(p->D != 0, TFREC_EUNSUP, "factor size d=%d is not supported for this score mode", d);
    p->n_items = item_hi - item_lo;
    p->bf16 = score_mode == TFREC_SCORE_BF16;
    p->CAP = cap_for(k, p->bf16, p->D);
    p->direct = !p->bf16 && d == p->D && (((uintptr_t)d_Q) & 15u) == 0;
    p->TI = 768;      // a multiple of every ScoreGeom::TI
    p->chunk = 1 << 20;          // users per launch: bounds the candidate-list workspace
    // waves per workgroup of the instantiation that will run (ScoreWave<D, BF16, CAP>::NW)
    const int nw = p->bf16 ? (p->D == 64 ? 16 : (p->D == 128 ? (TF_EVAL_PP || p->CAP <= 256 ? 16 : 8) : (p->D == 256 ? 8 : 4))) : (p->D == 64 ? 8 : 4);
    const int64_t ub = 32 * nw;                    // users per workgroup
    const int64_t user_blocks_all = (n + ub - 1) / ub;
    const int64_t want_blocks = 512 * 4 / nw > 256 ? 512 * 4 / nw : 256;     // 2048 waves, and no CU left without a workgroup
    int splits = 1;
    if (user_blocks_all < want_blocks && p->n_items > 0) {
        // few users: also split the item range over workgroups, then merge the partial lists
        const int64_t want = (want_blocks + user_blocks_all - 1) / user_blocks_all;
        const int64_t maxs = (p->n_items + 4095) / 4096;
        splits = (int)(want < maxs ? want : maxs);
        if (splits < 1) splits = 1;
        // beyond 16 the merge of the partial lists costs more than the scan gains -- except for a handful of users
        // (a few workgroups' worth: a serving call), where the scan of one workgroup IS the latency of the call: there
        // the item range goes to up to 128 workgroups and the partial lists are merged in two levels
        const bool handful = user_blocks_all <= 4;
        const int cap = handful ? (int)(128 / user_blocks_all) : 16;
        if (splits > cap) splits = cap;
        const int g1 = merge_group(k);
        if ((int64_t)splits * k > 2048) splits = handful ? (splits < g1 * g1 ? splits : g1 * g1) : g1;
    }
    p->split_items = p->n_items > 0 ? ((p->n_items + splits - 1) / splits + p->TI - 1) / p->TI * p->TI : p->TI;
    p->splits = p->n_items > 0 ? (int)((p->n_items + p->split_items - 1) / p->split_items) : 1;
    const int64_t nc = n < p->chunk ? n : p->chunk;
    p->q_bytes = p->direct ? 0 : tf_align((size_t)p->n_items * p->D * (p->bf16 ? 2 : 4));
    p->list_bytes = tf_align(sizeof(uint64_t) * (size_t)p->splits * nc * p->CAP);
    p->part_bytes = p->splits > 1 ? tf_align((size_t)p->splits * nc * k * 4) : 0;
    const int64_t groups = (p->splits + merge_group(k) - 1) / merge_group(k);
    p->part2_bytes = (int64_t)p->splits * k > 2048 ? tf_align((size_t)groups * nc * k * 4) : 0;
    p->total = p->q_bytes + p->list_bytes + 2 * p->part_bytes + 2 * p->part2_bytes + 4096;
    return TFREC_OK;
}

static int score_topk_impl(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, int64_t I, int d,
                           int64_t item_lo, int64_t item_hi, const int32_t *d_eval_users, int64_t n,
                           const int64_t *d_train_indptr, const int32_t *d_train_indices, int k, int score_mode,
                           int32_t *d_out_ids, float *d_out_scores, int p_compact = 0, int64_t id_offset = 0)
{
    TF_REQUIRE(d_P && d_Q && d_b, TFREC_EINVAL, "null table pointer");
    TF_REQUIRE(0 <= item_lo && item_lo <= item_hi && item_hi <= I, TFREC_EINVAL, "bad item range [%lld,%lld) of %lld",
               (long long)item_lo, (long long)item_hi, (long long)I);
    TF_REQUIRE(I + id_offset < (1ll << 31), TFREC_EUNSUP, "item ids must fit in int32");
    TF_REQUIRE(n >= 0 && k >= 1, TFREC_EINVAL, "bad n or k");
    TF_REQUIRE(score_mode == TFREC_SCORE_F32 || score_mode == TFREC_SCORE_BF16, TFREC_EINVAL, "unknown score mode %d", score_mode);
    TF_REQUIRE((d_train_indptr == nullptr) == (d_train_indices == nullptr), TFREC_EINVAL, "train CSR needs both arrays");
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_eval_users && d_out_ids && d_out_scores, TFREC_EINVAL, "null pointer");
    ScorePlan pl;
    TF_TRY(plan_score(d_Q, d, item_lo, item_hi, n, k, score_mode, &pl));
    const int CAP = pl.CAP, D = pl.D, splits = pl.splits;
    const bool bf16 = pl.bf16, direct = pl.direct;
    const int64_t n_items = pl.n_items, split_items = pl.split_items, chunk_max = pl.chunk;
    if (n_items == 0) {
        // an empty shard: every slot is "no candidate" (id -1 = 0xffffffff, score -inf = 0xff800000)
        TF_HIP(hipMemsetAsync(d_out_ids, 0xff, sizeof(int32_t) * n * k, ctx->stream));
        TF_HIP(hipMemsetD32Async((hipDeviceptr_t)d_out_scores, (int)0xff800000u, (size_t)n * k, ctx->stream));
        return TFREC_OK;
    }
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + pl.total));
    void *qt = direct ? nullptr : tfrec_ws_take(ctx, pl.q_bytes);
    uint64_t *lists = (uint64_t *)tfrec_ws_take(ctx, pl.list_bytes);
    int32_t *part_ids = splits > 1 ? (int32_t *)tfrec_ws_take(ctx, pl.part_bytes) : nullptr;
    float *part_sc = splits > 1 ? (float *)tfrec_ws_take(ctx, pl.part_bytes) : nullptr;
    int32_t *part2_ids = pl.part2_bytes ? (int32_t *)tfrec_ws_take(ctx, pl.part2_bytes) : nullptr;
    float *part2_sc = pl.part2_bytes ? (float *)tfrec_ws_take(ctx, pl.part2_bytes) : nullptr;
    TF_REQUIRE(lists != nullptr && (!pl.part2_bytes || part2_sc != nullptr), TFREC_ENOMEM, "internal: workspace carve failed");

    if (!direct) {
        const int64_t total = n_items * D;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (bf16) prep_items_kernel<true><<<blocks, 256, 0, ctx->stream>>>(d_Q, item_lo, n_items, d, D, qt);
        else prep_items_kernel<false><<<blocks, 256, 0, ctx->stream>>>(d_Q, item_lo, n_items, d, D, qt);
        TF_HIP(hipGetLastError());
    }

    for (int64_t u0 = 0; u0 < n; u0 += chunk_max) {
        const int64_t nn = n - u0 < chunk_max ? n - u0 : chunk_max;
        ScoreArgs a{};
        a.P = d_P;
        a.Qt = direct ? (const void *)(d_Q + (size_t)item_lo * d) : qt;
        a.b = d_b;
        a.d = d;
        a.item_lo = item_lo;
        a.n_items = n_items;
        a.split_items = split_items;
        a.P = p_compact ? d_P + (size_t)u0 * d : d_P;
        a.p_compact = p_compact;
        a.id_offset = id_offset;
        a.users = d_eval_users + u0;
        a.n = nn;
        a.train_indptr = d_train_indptr;
        a.train_indices = d_train_indices;
        a.k = k;
        a.lists = lists;
        a.out_ids = splits > 1 ? part_ids : d_out_ids + (size_t)u0 * k;
        a.out_scores = splits > 1 ? part_sc : d_out_scores + (size_t)u0 * k;
        const int64_t ub = 128;
        const int user_blocks = (int)((nn + ub - 1) / ub);
        if (CAP == 256) TF_TRY(launch_score_cap<256>(ctx, a, D, score_mode, user_blocks, splits));
#if TF_EVAL_CAP384 && TF_EVAL_PP
        else if (CAP == 384) TF_TRY((launch_score<128, true, 384>(ctx, a, user_blocks, splits)));      // (cap_for: bf16, D = 128 only)
#endif
        else TF_TRY(launch_score_cap<512>(ctx, a, D, score_mode, user_blocks, splits));
        if (splits > 1 && !pl.part2_bytes) {
            TF_TRY(merge_lists(ctx, part_ids, part_sc, splits, nn, k, d_out_ids + (size_t)u0 * k, d_out_scores + (size_t)u0 * k));
        } else if (splits > 1) {
            // two levels: groups of lists first (the total order makes the merge of merged lists the merge of all)
            const int g1 = merge_group(k), groups = (splits + g1 - 1) / g1;
            for (int g = 0; g < groups; ++g) {
                const int cnt = splits - g * g1 < g1 ? splits - g * g1 : g1;
                const size_t src = (size_t)g * g1 * nn * k, dst = (size_t)g * nn * k;
                TF_TRY(merge_lists(ctx, part_ids + src, part_sc + src, cnt, nn, k, part2_ids + dst, part2_sc + dst));
            }
            TF_TRY(merge_lists(ctx, part2_ids, part2_sc, groups, nn, k, d_out_ids + (size_t)u0 * k, d_out_scores + (size_t)u0 * k));
        }
    }
    return TFREC_OK;
}

extern "C" int tfrec_score_topk(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, int64_t I, int d,
                                int64_t item_lo, int64_t item_hi, const int32_t *d_eval_users, int64_t n,
                                const int64_t *d_train_indptr, const int32_t *d_train_indices, int k, int score_mode,
                                int32_t *d_out_ids, float *d_out_scores)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    return score_topk_impl(ctx, d_P, d_Q, d_b, I, d, item_lo, item_hi, d_eval_users, n, d_train_indptr, d_train_indices, k,
                           score_mode, d_out_ids, d_out_scores);
}

extern "C" int tfrec_score_topk_shard(tfrec_ctx *ctx, const float *d_P_rows, const float *d_Q_shard, const float *d_b_shard,
                                      int64_t n_shard_items, int64_t item_id_lo, int d, const int32_t *d_eval_users, int64_t n,
                                      const int64_t *d_train_indptr, const int32_t *d_train_indices, int k, int score_mode,
                                      int32_t *d_out_ids, float *d_out_scores)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(item_id_lo >= 0, TFREC_EINVAL, "negative id offset");
    return score_topk_impl(ctx, d_P_rows, d_Q_shard, d_b_shard, n_shard_items, d, 0, n_shard_items, d_eval_users, n, d_train_indptr,
                           d_train_indices, k, score_mode, d_out_ids, d_out_scores, 1, item_id_lo);
}

extern "C" int tfrec_topk_fill_masked(tfrec_ctx *ctx, int32_t *d_ids, const int32_t *d_eval_users, int64_t n, int k,
                                      const int64_t *d_train_indptr, const int32_t *d_train_indices)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    if (n == 0 || !d_train_indptr) return TFREC_OK;
    TF_REQUIRE(d_ids && d_eval_users && d_train_indices, TFREC_EINVAL, "null pointer");
    fill_masked_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(d_ids, d_eval_users, n, k, d_train_indptr, d_train_indices);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_topk_dense(tfrec_ctx *ctx, const float *d_ratings, int64_t rows, int64_t cols, int k,
                                const int32_t *d_mask_rows, const int64_t *d_mask_indptr, const int32_t *d_mask_indices,
                                int32_t *d_out_ids, float *d_out_scores)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(rows >= 0 && cols >= 1 && k >= 1, TFREC_EINVAL, "bad shape rows=%lld cols=%lld k=%d", (long long)rows, (long long)cols, k);
    TF_REQUIRE(cols < (1ll << 31), TFREC_EUNSUP, "column ids must fit in int32");
    TF_REQUIRE(k <= 240, TFREC_EUNSUP, "top-k above 240 is not supported (k=%d)", k);
    TF_REQUIRE((d_mask_indptr == nullptr) == (d_mask_indices == nullptr), TFREC_EINVAL, "mask CSR needs both arrays");
    if (rows == 0) return TFREC_OK;
    TF_REQUIRE(d_ratings && d_out_ids, TFREC_EINVAL, "null pointer");
    constexpr int CAP = 512;
    const size_t list_bytes = tf_align(sizeof(uint64_t) * (size_t)rows * CAP);
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + list_bytes + 4096));
    uint64_t *lists = (uint64_t *)tfrec_ws_take(ctx, list_bytes);
    TF_REQUIRE(lists != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    topk_dense_kernel<CAP><<<(int)((rows + 3) / 4), 256, 0, ctx->stream>>>(d_ratings, rows, cols, k, d_mask_rows, d_mask_indptr,
                                                                          d_mask_indices, lists, d_out_ids, d_out_scores);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

// Leaves [5k] double sums over the users in d_sums: from the per-user curves when the caller wants those rows
// (d_per_user), else straight from the rank lists.  Scratch is carved above whatever the caller already holds
// in the workspace.
static size_t metrics_need(int64_t n, int k, bool have_rows)
{
    if (!have_rows) return tf_align(sizeof(double) * (size_t)((n + 255) / 256) * 4 * k) + tf_align(sizeof(double) * k) + 4096;
    const int C = 5 * k;
    const int64_t max_chunks = (n + 1023) / 1024;
    return tf_align(sizeof(double) * (size_t)max_chunks * C) + tf_align(sizeof(double) * k) + 4096;
}

static int metrics_sums(tfrec_ctx *ctx, const int32_t *d_ids, const int32_t *d_users, int64_t n, int k, const int64_t *tptr,
                        const int32_t *tidx, float *d_per_user, double *d_sums)
{
    const int C = 5 * k;
    TF_REQUIRE(k <= METRIC_KMAX, TFREC_EUNSUP, "metrics beyond top-%d are not supported (k=%d)", METRIC_KMAX, k);
    const int64_t chunk = n;
    const int64_t upc = 1024;                                                  // users per column-sum block
    const int64_t max_chunks = (chunk + upc - 1) / upc;
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + metrics_need(n, k, d_per_user != nullptr)));
    float *rows = d_per_user;
    const int64_t n_blocks = (n + 255) / 256;
    double *partial = (double *)tfrec_ws_take(ctx, d_per_user ? sizeof(double) * (size_t)max_chunks * C : sizeof(double) * (size_t)n_blocks * 4 * k);
    double *tab = (double *)tfrec_ws_take(ctx, sizeof(double) * k);
    TF_REQUIRE(partial && tab, TFREC_ENOMEM, "internal: workspace carve failed");
    // 1/log2(i+2) from the host's libm, so the device adds exactly the doubles metrics.h:69-71 adds
    TF_TRY(tfrec_pinned_require(ctx, sizeof(double) * (k + C)));
    double *h_tab = (double *)ctx->h_pinned;
    if (ctx->pinned_tab_k != k) {
        TF_HIP(hipStreamSynchronize(ctx->stream));     // an earlier upload may still be reading the table
        for (int i = 0; i < k; ++i) h_tab[i] = 1.0 / std::log2((double)(i + 2));
        ctx->pinned_tab_k = k;
    }
    TF_HIP(hipMemcpyAsync(tab, h_tab, sizeof(double) * k, hipMemcpyHostToDevice, ctx->stream));
    if (!d_per_user) {
        // nobody wants the per-user curves: sums only, no [n, 5, k] table
        TF_REQUIRE(n_blocks < (1ll << 31), TFREC_EUNSUP, "too many users for one metrics launch");
        metrics_sums_kernel<<<(unsigned)n_blocks, 256, 0, ctx->stream>>>(d_ids, d_users, n, k, tptr, tidx, partial);
        TF_HIP(hipGetLastError());
        metrics_final_kernel<<<1, 256, 0, ctx->stream>>>(partial, (int)n_blocks, k, tab, d_sums, 0);
        TF_HIP(hipGetLastError());
        return TFREC_OK;
    }
    for (int64_t u0 = 0; u0 < n; u0 += chunk) {
        const int64_t nn = n - u0 < chunk ? n - u0 : chunk;
        float *rp = d_per_user ? d_per_user + (size_t)u0 * C : rows;
        metrics_user_kernel<<<(int)((nn + 255) / 256), 256, 0, ctx->stream>>>(d_ids + (size_t)u0 * k, d_users + u0, nn, k, tptr, tidx,
                                                                              tab, rp);
        TF_HIP(hipGetLastError());
        const int nchunks = (int)((nn + upc - 1) / upc);
        colsum_kernel<<<dim3((C + 255) / 256, nchunks), 256, 0, ctx->stream>>>(rp, nn, C, upc, partial);
        TF_HIP(hipGetLastError());
        colsum_final_kernel<<<(C + 255) / 256, 256, 0, ctx->stream>>>(partial, nchunks, C, d_sums, u0 > 0);
        TF_HIP(hipGetLastError());
    }
    return TFREC_OK;
}

extern "C" int tfrec_rank_metrics_sums(tfrec_ctx *ctx, const int32_t *d_ids, const int32_t *d_eval_users, int64_t n, int k,
                                       const int64_t *d_test_indptr, const int32_t *d_test_indices, float *d_per_user,
                                       double *d_sums)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0 && k >= 1, TFREC_EINVAL, "bad n or k");
    TF_REQUIRE(d_sums != nullptr, TFREC_EINVAL, "null output");
    if (n == 0) { TF_HIP(hipMemsetAsync(d_sums, 0, sizeof(double) * 5 * k, ctx->stream)); return TFREC_OK; }
    TF_REQUIRE(d_ids && d_eval_users && d_test_indptr && d_test_indices, TFREC_EINVAL, "null pointer");
    return metrics_sums(ctx, d_ids, d_eval_users, n, k, d_test_indptr, d_test_indices, d_per_user, d_sums);
}

static size_t metrics_host_need(int64_t n, int k, int mean_mode, bool have_rows)
{
    const int C = 5 * k;
    size_t s = tf_align(sizeof(double) * C) + 4096;
    if (mean_mode == 1 && !have_rows) s += tf_align(sizeof(float) * (size_t)n * C);
    return s + metrics_need(n, k, have_rows || mean_mode == 1);
}

static int metrics_to_host(tfrec_ctx *ctx, const int32_t *d_ids, const int32_t *d_users, int64_t n, int k, const int64_t *tptr,
                           const int32_t *tidx, int mean_mode, float *d_per_user, float *h_mean)
{
    const int C = 5 * k;
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + metrics_host_need(n, k, mean_mode, d_per_user != nullptr)));
    double *d_sums = (double *)tfrec_ws_take(ctx, sizeof(double) * C);
    float *rows = d_per_user;
    if (mean_mode == 1 && !rows) rows = (float *)tfrec_ws_take(ctx, sizeof(float) * (size_t)n * C);
    TF_REQUIRE(d_sums && (mean_mode == 0 || rows), TFREC_ENOMEM, "internal: workspace carve failed");
    TF_TRY(metrics_sums(ctx, d_ids, d_users, n, k, tptr, tidx, rows, d_sums));
    if (mean_mode == 1) {
        // per-user rows to the host, summed sequentially in float32 in user order (evaluate.h:40-92)
        // (malloc, not a container: an allocation failure is reported, nothing can throw across the C boundary)
        float *host = (float *)std::malloc(sizeof(float) * (size_t)n * C);
        TF_REQUIRE(host != nullptr, TFREC_ENOMEM, "host allocation of %zu bytes for the per-user metric rows failed", sizeof(float) * (size_t)n * C);
        hipError_t e = hipMemcpyAsync(host, rows, sizeof(float) * (size_t)n * C, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            std::free(host);
            tfrec_set_error("copy of the per-user metric rows failed: %s", hipGetErrorString(e));
            return TFREC_EHIP;
        }
        for (int c = 0; c < C; ++c) h_mean[c] = 0.f;
        for (int64_t u = 0; u < n; ++u)
            for (int c = 0; c < C; ++c) h_mean[c] += host[(size_t)u * C + c];
        for (int c = 0; c < C; ++c) h_mean[c] /= (float)n;
        std::free(host);
        return TFREC_OK;
    }
    double *h = (double *)ctx->h_pinned + k;
    TF_HIP(hipMemcpyAsync(h, d_sums, sizeof(double) * C, hipMemcpyDeviceToHost, ctx->stream));
    TF_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < C; ++c) h_mean[c] = (float)(h[c] / (double)n);
    return TFREC_OK;
}

extern "C" int tfrec_rank_metrics(tfrec_ctx *ctx, const int32_t *d_ids, const int32_t *d_eval_users, int64_t n, int k,
                                  const int64_t *d_test_indptr, const int32_t *d_test_indices, int mean_mode,
                                  float *d_per_user, float *h_mean)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 1 && k >= 1, TFREC_EINVAL, "bad n or k");
    TF_REQUIRE(d_ids && d_eval_users && d_test_indptr && d_test_indices && h_mean, TFREC_EINVAL, "null pointer");
    TF_REQUIRE(mean_mode == 0 || mean_mode == 1, TFREC_EINVAL, "unknown mean mode %d", mean_mode);
    return metrics_to_host(ctx, d_ids, d_eval_users, n, k, d_test_indptr, d_test_indices, mean_mode, d_per_user, h_mean);
}

// evaluation scratch that depends only on the number of users and k (candidate lists, id and
// score buffers, metric rows); the prepared item table comes on top for bf16 / padded scoring
size_t tfrec_eval_scratch_bytes(int64_t n, int k)
{
    int CAP = cap_for_k(k);
    if (!CAP) return 0;
    if (cap_for(k, true, 128) > CAP) CAP = cap_for(k, true, 128);      // whatever score mode and factor size the call will have
    const int64_t nc = n < (1 << 20) ? n : (1 << 20);
    return 2 * tf_align(4 * (size_t)n * k) + tf_align(sizeof(uint64_t) * (size_t)nc * CAP) + metrics_host_need(n, k, 0, false) + 8192;
}

// workspace bytes one full evaluation needs
size_t tfrec_eval_need_bytes(const float *d_Q, int64_t I, int d, int64_t n, int k, int score_mode, int mean_mode)
{
    ScorePlan pl;
    if (plan_score(d_Q, d, 0, I, n, k, score_mode, &pl) != TFREC_OK) return 0;
    return 2 * tf_align(4 * (size_t)n * k) + pl.total + metrics_host_need(n, k, mean_mode, false) + 4096;
}

extern "C" int tfrec_eval_fullrank(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, int64_t I, int d,
                                   const int32_t *d_eval_users, int64_t n, const int64_t *d_train_indptr,
                                   const int32_t *d_train_indices, const int64_t *d_test_indptr, const int32_t *d_test_indices,
                                   int k, int score_mode, int mean_mode, int32_t *d_out_ids, float *h_mean)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 1 && k >= 1 && h_mean, TFREC_EINVAL, "bad n, k or output");
    TF_REQUIRE(d_test_indptr && d_test_indices && d_eval_users, TFREC_EINVAL, "null test CSR or user list");
    TF_REQUIRE(mean_mode == 0 || mean_mode == 1, TFREC_EINVAL, "unknown mean mode %d", mean_mode);
    // size the workspace once for all stages: growing it later would move the buffers carved here
    const size_t need = tfrec_eval_need_bytes(d_Q, I, d, n, k, score_mode, mean_mode);
    TF_REQUIRE(need != 0, TFREC_EUNSUP, "unsupported k=%d or d=%d for this score mode", k, d);
    TF_TRY(tfrec_ws_require(ctx, need));
    int32_t *ids = d_out_ids ? d_out_ids : (int32_t *)tfrec_ws_take(ctx, sizeof(int32_t) * (size_t)n * k);
    float *scores = (float *)tfrec_ws_take(ctx, sizeof(float) * (size_t)n * k);
    TF_REQUIRE(ids && scores, TFREC_ENOMEM, "internal: workspace carve failed");
    const size_t floor = ctx->ws_used;
    TF_TRY(score_topk_impl(ctx, d_P, d_Q, d_b, I, d, 0, I, d_eval_users, n, d_train_indptr, d_train_indices, k, score_mode, ids,
                           scores));
    if (d_train_indptr) {
        fill_masked_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(ids, d_eval_users, n, k, d_train_indptr, d_train_indices);
        TF_HIP(hipGetLastError());
    }
    ctx->ws_used = floor;    // the scoring scratch is dead (stream order), the metrics reuse it
    return metrics_to_host(ctx, ids, d_eval_users, n, k, d_test_indptr, d_test_indices, mean_mode, nullptr, h_mean);
}
