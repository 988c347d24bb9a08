// Shared host/device helpers of libtfrec_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/tfrec_amd.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
void tfrec_set_error(const char *fmt, ...);

#define TF_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            tfrec_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TFREC_EHIP;                                                               \
        }                                                                                    \
    } while (0)

#define TF_REQUIRE(cond, code, ...)         \
    do {                                    \
        if (!(cond)) {                      \
            tfrec_set_error(__VA_ARGS__);   \
            return (code);                  \
        }                                   \
    } while (0)

#define TF_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != TFREC_OK) return _rc; \
    } while (0)

// ------------------------------------------------------------------------------------------
// context: a device, a stream, and one growable workspace carved by a bump allocator
// ------------------------------------------------------------------------------------------
// Sorted index arrays of one prepared batch (train.hip)
struct tfrec_batch_slot {
    char *mem = nullptr;
    size_t bytes = 0;
    int64_t B = -1, U = 0, I = 0;            // B < 0: nothing prepared
    int K = 1, next_k = 0;                   // batches sorted together / next one to consume
    uint32_t kmask = 0xffffffffu;            // key bits below the batch number
    uint32_t *keys = nullptr, *vals = nullptr;   // [3B] sorted: user side [0,B), item side [B,3B)
    uint2 *pair = nullptr;                   // [K*B] (i, j) of every triple
    const int32_t *users_raw = nullptr, *pos_raw = nullptr, *neg_raw = nullptr;   // the caller's index arrays [K*B]
    hipEvent_t ready = nullptr;              // recorded on the auxiliary stream after preparation
    hipEvent_t released = nullptr;           // recorded on the main stream after the step that consumed the slot
    bool has_released = false;
    int apr_pending = 0;                     // 1 between the two calls of a replicated-item APR step
};

struct tfrec_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char *ws = nullptr;          // device workspace
    size_t ws_bytes = 0;
    size_t ws_used = 0;          // bump pointer, reset at the start of every entry point
    bool ws_fixed = false;       // the workspace is a borrowed block of fixed size: never grown, never freed
    uint32_t *d_flags = nullptr; // sticky device-side error flags (4 bytes, separate allocation)
    void *h_pinned = nullptr;    // small pinned staging buffer for host outputs
    size_t h_pinned_bytes = 0;
    int pinned_tab_k = 0;        // k for which the pinned 1/log2 table is filled
    tfrec_batch_slot slots[3];      // 0,1: pipelined batches; 2: private to the unpipelined steps
    hipStream_t aux_stream = nullptr;   // index preparation runs here, beside the row passes
    hipEvent_t fork_event = nullptr;
    // optional phase timing: events recorded on the context's stream between the kernels of a call
    bool profile = false;
    static constexpr int MAX_MARKS = 16;
    hipEvent_t marks[MAX_MARKS] = {};
    const char *mark_names[MAX_MARKS] = {};
    int n_marks = 0;
};

// Records a named timing mark on the stream when profiling is on (no-op otherwise).
void tfrec_mark(tfrec_ctx *ctx, const char *name);
void tfrec_marks_reset(tfrec_ctx *ctx);

// Looks up this device's copy of the file-scope scratch array of the no-allocation step (train.hip); called by
// tfrec_ctx_create with the device current.
int tfrec_scratch_bind(int device);
// Makes ctx's device current for the calling thread.
int tfrec_ctx_enter(tfrec_ctx *ctx);
// Ensures the workspace holds at least `bytes` (may reallocate: previous carve-outs die).
int tfrec_ws_require(tfrec_ctx *ctx, size_t bytes);
// Carves `bytes` (256-byte aligned) out of the workspace; tfrec_ws_require must have covered it.
void *tfrec_ws_take(tfrec_ctx *ctx, size_t bytes);
int tfrec_pinned_require(tfrec_ctx *ctx, size_t bytes);

static inline size_t tf_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// stable LSD radix sort of (key,value) pairs (sort.hip)
size_t tfrec_sort_temp_bytes(size_t n, int end_bit);
int tfrec_sort_pairs(void *d_temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                     const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit, hipStream_t stream);

// The 3n (key, occurrence) pairs of n index triples, never materialised: the first pass of the sort reads them
// straight from the caller's arrays (range check and clamping included) and leaves the checked (i, j) of every
// triple in `pair`.  Pair e < n is the user side of triple e; pair n + 2t + side is item side `side` of triple t.
// K consecutive batches of B triples: the batch number t / B goes above bit `kb` of every key, occurrence ids are
// local to the batch (user side: tau, item sides: 2 tau and 2 tau + 1).
struct tfrec_triple_src {
    const int32_t *users, *pos, *neg;
    uint32_t n, B, U, I;
    uint32_t inv_B;       // ceil(2^32 / B), any non-zero value when B = 1 (0 when n <= B: one batch, no batch numbers)
    int kb;
    uint32_t *pair;       // [2n] = uint2[n]: (i, j) range-checked
    uint32_t *flags;      // sticky error flags
};
int tfrec_sort_triples(void *d_temp, size_t temp_bytes, const tfrec_triple_src &src, uint32_t *keys_out, uint32_t *vals_out,
                       int end_bit, hipStream_t stream);

// developer builds (-DTF_DEBUG_SYNC): a line per step of the calls under investigation, appended to a file
#ifdef TF_DEBUG_SYNC
#include <cstdio>
#define TF_DBG(...) do { FILE *f_ = fopen("gpurun_out/tfrec_dbg.log", "a"); if (f_) { fprintf(f_, __VA_ARGS__); fputc('\n', f_); fclose(f_); } } while (0)
#else
#define TF_DBG(...) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// The value the lane 32 places away holds (v_permlane32_swap: a vector-ALU move, no LDS trip).
__device__ __forceinline__ uint32_t swap32(uint32_t x, int lane)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane & 32) ? r[0] : r[1];
}

// Lane exchange of a 32-bit word with the lane whose number differs in bit J (J = 1..32).
template <int J>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t v, int lane)
{
    if constexpr (J == 32) return swap32(v, lane);
    else if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (J << 10) | 0x1f);                           // bit-mask mode, xor J
}

// Butterfly exchange inside an aligned group of up to 64 lanes.  ds_swizzle's bit-mask mode
// covers XOR masks below 32 without an address register; mask 32 goes through ds_bpermute.
template <int MASK>
__device__ __forceinline__ float lane_xor(float v)
{
    if constexpr (MASK < 32) {
        // pattern: and_mask=0x1f, or_mask=0, xor_mask=MASK  (bit 15 clear = bit-mask mode)
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (MASK << 10) | 0x1f));
    } else {
        return __shfl_xor(v, MASK, 64);
    }
}

// Lane permutations inside a row of 16 lanes done by the data-parallel-primitive path of the vector
// ALU (no LDS round trip).  CTRL: 0xB1 swaps neighbours, 0x4E swaps pairs, 0x141 mirrors each half
// row of 8 lanes, 0x140 mirrors the row of 16.
template <int CTRL>
__device__ __forceinline__ float lane_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Sum over an aligned group of LPR lanes; every lane of the group gets the total.  The
// combination order is fixed, so the result does not depend on scheduling.  After the first two
// steps all lanes of a quad hold the same value, so the mirror permutations pair every lane with a
// lane that holds what its XOR partner holds: the sums equal those of an XOR butterfly, bit for bit.
template <int LPR>
__device__ __forceinline__ float group_sum(float v)
{
    if constexpr (LPR > 1) v += lane_dpp<0xB1>(v);
    if constexpr (LPR > 2) v += lane_dpp<0x4E>(v);
    if constexpr (LPR > 4) v += lane_dpp<0x141>(v);
    if constexpr (LPR > 8) v += lane_dpp<0x140>(v);
    if constexpr (LPR > 16) v += lane_xor<16>(v);
    if constexpr (LPR > 32) v += lane_xor<32>(v);
    return v;
}

// Row geometry of a kernel configuration.  FULL: the factor size is exactly LPR*NCH*VEC, so row
// strides and chunk counts are compile-time constants (shifts instead of 64-bit multiplies, no
// per-lane bounds tests); otherwise the run-time values are used.
template <int LPR, int NCH, int VEC, bool FULL>
struct RowDims {
    __device__ __forceinline__ static int d(int rt) { return FULL ? LPR * NCH * VEC : rt; }
    __device__ __forceinline__ static int nchunk(int rt) { return FULL ? LPR * NCH : rt; }
};

// Read lane `src` (0..LPR-1, relative to the group) of every aligned LPR-lane group.
template <int LPR>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v, int src)
{
    return (uint32_t)__shfl((int)v, src, LPR);
}
template <int LPR>
__device__ __forceinline__ float group_bcastf(float v, int src)
{
    return __shfl(v, src, LPR);
}

// A row chunk: VEC floats (4 -> one 16-byte access per lane, 1 -> scalar fallback).
template <int VEC>
struct Chunk;
template <>
struct Chunk<4> {
    float4 v;
    __device__ __forceinline__ static Chunk zero() { return {make_float4(0.f, 0.f, 0.f, 0.f)}; }
    __device__ __forceinline__ static Chunk load(const float *row, int c) { return {reinterpret_cast<const float4 *>(row)[c]}; }
    __device__ __forceinline__ void store(float *row, int c) const { reinterpret_cast<float4 *>(row)[c] = v; }
    __device__ __forceinline__ float dot(const Chunk &o) const { return v.x * o.v.x + v.y * o.v.y + v.z * o.v.z + v.w * o.v.w; }
    __device__ __forceinline__ Chunk operator-(const Chunk &o) const { return {make_float4(v.x - o.v.x, v.y - o.v.y, v.z - o.v.z, v.w - o.v.w)}; }
    __device__ __forceinline__ Chunk operator+(const Chunk &o) const { return {make_float4(v.x + o.v.x, v.y + o.v.y, v.z + o.v.z, v.w + o.v.w)}; }
    __device__ __forceinline__ Chunk operator*(const Chunk &o) const { return {make_float4(v.x * o.v.x, v.y * o.v.y, v.z * o.v.z, v.w * o.v.w)}; }
    __device__ __forceinline__ Chunk scaled(float s) const { return {make_float4(v.x * s, v.y * s, v.z * s, v.w * s)}; }
    // this += s * o
    __device__ __forceinline__ void fma(float s, const Chunk &o)
    {
        v.x = fmaf(s, o.v.x, v.x); v.y = fmaf(s, o.v.y, v.y); v.z = fmaf(s, o.v.z, v.z); v.w = fmaf(s, o.v.w, v.w);
    }
    template <class F>
    __device__ __forceinline__ Chunk map2(const Chunk &o, F f) const { return {make_float4(f(v.x, o.v.x), f(v.y, o.v.y), f(v.z, o.v.z), f(v.w, o.v.w))}; }
};
template <>
struct Chunk<1> {
    float v;
    __device__ __forceinline__ static Chunk zero() { return {0.f}; }
    __device__ __forceinline__ static Chunk load(const float *row, int c) { return {row[c]}; }
    __device__ __forceinline__ void store(float *row, int c) const { row[c] = v; }
    __device__ __forceinline__ float dot(const Chunk &o) const { return v * o.v; }
    __device__ __forceinline__ Chunk operator-(const Chunk &o) const { return {v - o.v}; }
    __device__ __forceinline__ Chunk operator+(const Chunk &o) const { return {v + o.v}; }
    __device__ __forceinline__ Chunk operator*(const Chunk &o) const { return {v * o.v}; }
    __device__ __forceinline__ Chunk scaled(float s) const { return {v * s}; }
    __device__ __forceinline__ void fma(float s, const Chunk &o) { v = fmaf(s, o.v, v); }
    template <class F>
    __device__ __forceinline__ Chunk map2(const Chunk &o, F f) const { return {f(v, o.v)}; }
};

// A whole row spread over an LPR-lane group: lane l holds chunks l, l+LPR, ... (NCH of them).
// `nchunk` = chunks per row (d / VEC); lanes/chunks beyond it hold zeros and never touch memory.
template <int LPR, int NCH, int VEC>
struct Row {
    Chunk<VEC> c[NCH];
    __device__ __forceinline__ static Row zero()
    {
        Row r;
#pragma unroll
        for (int g = 0; g < NCH; ++g) r.c[g] = Chunk<VEC>::zero();
        return r;
    }
    __device__ __forceinline__ static Row load(const float *row, int gl, int nchunk)
    {
        Row r;
#pragma unroll
        for (int g = 0; g < NCH; ++g) {
            const int ci = gl + g * LPR;
            r.c[g] = ci < nchunk ? Chunk<VEC>::load(row, ci) : Chunk<VEC>::zero();
        }
        return r;
    }
    __device__ __forceinline__ void store(float *row, int gl, int nchunk) const
    {
#pragma unroll
        for (int g = 0; g < NCH; ++g) {
            const int ci = gl + g * LPR;
            if (ci < nchunk) c[g].store(row, ci);
        }
    }
    // per-lane partial of <this, o>; finish with group_sum<LPR>
    __device__ __forceinline__ float dot_partial(const Row &o) const
    {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < NCH; ++g) s += c[g].dot(o.c[g]);
        return s;
    }
    __device__ __forceinline__ Row operator-(const Row &o) const
    {
        Row r;
#pragma unroll
        for (int g = 0; g < NCH; ++g) r.c[g] = c[g] - o.c[g];
        return r;
    }
    __device__ __forceinline__ Row operator+(const Row &o) const
    {
        Row r;
#pragma unroll
        for (int g = 0; g < NCH; ++g) r.c[g] = c[g] + o.c[g];
        return r;
    }
    __device__ __forceinline__ void fma(float s, const Row &o)
    {
#pragma unroll
        for (int g = 0; g < NCH; ++g) c[g].fma(s, o.c[g]);
    }
    template <class F>
    __device__ __forceinline__ Row map2(const Row &o, F f) const
    {
        Row r;
#pragma unroll
        for (int g = 0; g < NCH; ++g) r.c[g] = c[g].map2(o.c[g], f);
        return r;
    }
};

// numerically stable pieces of the pairwise losses (float32)
__device__ __forceinline__ float sigmoid_neg(float z)   // sigma(-z)
{
    const float e = __expf(-fabsf(z));
    const float r = __builtin_amdgcn_rcpf(1.f + e);   // 1 ulp; 1 + e lies in [1, 2]
    return z >= 0.f ? e * r : r;
}
__device__ __forceinline__ float softplus_neg(float z)  // log(1 + exp(-z))
{
    return fmaxf(-z, 0.f) + log1pf(__expf(-fabsf(z)));
}

#endif  // __HIPCC__
