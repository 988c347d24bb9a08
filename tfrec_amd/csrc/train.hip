// Training steps of the matrix-factorization ranking path: BPR-SGD, APR-Adagrad, DNS candidate
// selection and plain prediction.  The duplicate-index reduction lives in seg_pass.hpp; this
// file holds the arithmetic (policies) and the C entry points.
//
// One step, for a batch of B triples (u, i, j):
//   1. index (can run ahead on a second stream, see tfrec_batch_prepare)
//        3B (key, occurrence) pairs: key u for the user side, U+item for both item sides,
//        range-checked (clamped + sticky flag), in ONE stable radix sort by key whose first pass
//        reads the index triples themselves: user occurrences land in [0,B), item occurrences
//        in [B,3B)
//   2. user pass   gather P[u], Q[i], Q[j] -> s = sigma(-x); sum the user gradient per distinct
//                  u and update P[u]; leave s of every triple for the item pass
//   3. item pass   per item occurrence: gather s and the PRE-step P[u] -> sum the item gradient
//                  per distinct item and update Q, b in place
// Every gradient is taken at the pre-step parameters although both tables are updated.  Plain user
// table: step 2 updates P in place and stages a copy of the pre-step P[u] of every triple, which
// is all that step 3 reads.  Versioned user table (tfrec_utable): step 2 writes the updated row
// into the row's other buffer, step 3 reads the pre-step row where it still is, and the version
// byte of a user is flipped as its new row is written: no copies.
#include "seg_pass.hpp"

#include <atomic>

// ------------------------------------------------------------------------------------------
// keys
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t clamp_index(int32_t v, uint32_t n, bool *bad)
{
    if (v < 0) { *bad = true; return 0u; }
    if ((uint32_t)v >= n) { *bad = true; return n - 1; }
    return (uint32_t)v;
}

// ------------------------------------------------------------------------------------------
// BPR-SGD policies.  Reference math: model/base/MatrixFactorization.py:26-32 (score),
// model/losses.py:4-12 (-log_sigmoid(x + 1e-7)), model/BPR.py:42-48 (loss, l2 on gathered rows, SGD)
// ------------------------------------------------------------------------------------------
// occurrences whose gathers a lane group keeps in flight (tile sizes are multiples of these)
#ifndef TF_DEPTH_USER
#define TF_DEPTH_USER 4
#endif
#ifndef TF_DEPTH_ITEM
#define TF_DEPTH_ITEM 4
#endif
// developer comparison: 1 = flip the version bytes in a kernel of their own after both passes instead of in the user pass
#ifndef TF_FLIP_KERNEL
#define TF_FLIP_KERNEL 0
#endif
#ifndef TF_SHORT_TILES_MAX_B
#define TF_SHORT_TILES_MAX_B 131072
#endif

struct BprParams {
    float *P, *Q, *b;
    // Versioned user table (tfrec_utable): row u lives in P2 when ver[u] is set, else in P.  A step reads the
    // current copy of a row and writes the updated row into the other buffer; the item pass, which needs the
    // pre-step rows, finds them through the record every triple's user pass leaves (it carries the pre-step location);
    // the user pass flips a user's version byte as it writes the user's new row.  Null in the plain layout.
    float *P2;
    uint8_t *ver;
    float *gQ, *gb;      // where the item pass leaves gradient sums when it does not apply them (may alias Q, b)
    float *stage;        // [B, d] pre-step P[u] of every triple
    float *sbuf;         // [B]    s = sigma(-(x + 1e-7)) of every triple
    const uint2 *pair;   // [B]    (i, j) of every triple, range-checked
    const int32_t *users;   // [B] the caller's user ids (clamped on use)
    uint32_t U;
    int d, nchunk;
    float lr, reg;
    uint32_t kmask;      // sorted keys carry a batch number above these bits when several batches were sorted together
};

// VER: the user table is versioned (see BprParams): no pre-step rows are staged for the item pass.
template <int LPR_, int NCH_, int VEC_, bool FULL_, bool LOSS, bool VER = false, bool SHORT = false>
struct BprUserPol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = TF_DEPTH_USER, SHORT_TILES = SHORT };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<LOSS ? 2 : 0>;
    BprParams a;
    // The user's row is read once per run of the user's occurrences (they are adjacent in the sorted order), not once
    // per occurrence: it rides in the segment state, the occurrence brings only the two item rows.
    struct Occ { RowT qi, qj; float bi, bj; };
    struct Seg { RowT p; };

    // .y = occurrence id; its top bit carries the row's version (occurrence ids stay below 2^30)
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const
    {
        const uint2 ij = a.pair[val];
        if constexpr (VER) val |= (uint32_t)a.ver[key & a.kmask] << 31;
        return make_uint4(key, val, ij.x, ij.y);
    }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        Occ o;
        o.qi = RowT::load(a.Q + (size_t)en.z * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.qj = RowT::load(a.Q + (size_t)en.w * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.bi = a.b[en.z];
        o.bj = a.b[en.w];
        return o;
    }
    __device__ __forceinline__ Seg load_seg_en(uint4 en, int gl) const
    {
        const float *src = (VER && (en.y >> 31)) ? a.P2 : a.P;
        return Seg{RowT::load(src + (size_t)(en.x & a.kmask) * DM::d(a.d), gl, DM::nchunk(a.nchunk))};
    }
    __device__ __forceinline__ void compute(Seg &seg, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &loss, int gl) const
    {
        const RowT diff = o.qi - o.qj;
        const float x = group_sum<LPR_>(seg.p.dot_partial(diff)) + (o.bi - o.bj);
        const float z = x + 1e-7f;
        const float s = sigmoid_neg(z);
        acc.v.fma(s, diff);          // user gradient is  -sum s*(q_i - q_j) + n*reg*p
        acc.cnt += 1;
        const uint32_t t = VER ? (en.y & 0x7fffffffu) : en.y;
        if constexpr (VER) {
            // what the item pass needs from this triple in one 8-byte record: s, and where the pre-step user row is
            if (gl == 0) reinterpret_cast<uint2 *>(a.sbuf)[t] = make_uint2(__float_as_uint(s), (en.x & a.kmask) | (en.y & 0x80000000u));
        } else {
            seg.p.store(a.stage + (size_t)t * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            if (gl == 0) a.sbuf[t] = s;
        }
        if constexpr (LOSS) {
            float r = seg.p.dot_partial(seg.p) + o.qi.dot_partial(o.qi) + o.qj.dot_partial(o.qj);
            if (gl == 0) {
                loss.t[0] += softplus_neg(z);
                r += o.bi * o.bi + o.bj * o.bj;
            }
            loss.t[1] += 0.5f * r;
        }
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        const size_t off = (size_t)(key & a.kmask) * DM::d(a.d);
        const bool v = VER && a.ver[key & a.kmask] != 0;        // where the current row is; the new one goes to the other buffer
        const float *src = v ? a.P2 : a.P;
        float *dst = VER ? (v ? a.P : a.P2) : a.P;
        const RowT p = seg ? seg->p : RowT::load(src + off, gl, DM::nchunk(a.nchunk));
        const float nreg = (float)acc.cnt * a.reg, lr = a.lr;
        const RowT out = p.map2(acc.v, [=](float pv, float av) { return fmaf(-lr, fmaf(nreg, pv, -av), pv); });
        out.store(dst + off, gl, DM::nchunk(a.nchunk));
        // The row now lives in the other buffer.  apply() runs once per distinct user of the batch (in the tile kernel
        // for a run that lies inside one tile, in the fix-up kernel for a run cut by tile boundaries), after every
        // entry() that looks this user's byte up has been staged: entries are staged by the tile that holds them,
        // and a run that is applied in the tile kernel lies in that one tile.  The item pass does not read the
        // bytes (the triples' records carry the pre-step location).
        if constexpr (VER && !TF_FLIP_KERNEL) if (gl == 0) a.ver[key & a.kmask] = v ? 0 : 1;
    }
};

// GRAD = false: apply the SGD update to Q, b.  GRAD = true: leave the summed gradient of every
// distinct item in row `it` of gQ / entry `it` of gb instead.  Those are either Q, b themselves
// (the tables then hold fetched copies of rows owned by other GPUs, and the sums travel back to
// the owners) or a zeroed table of the shape of Q, b that is then summed over the GPUs.
template <int LPR_, int NCH_, int VEC_, bool FULL_, bool GRAD = false, bool VER = false, bool SHORT = false>
struct BprItemPol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = TF_DEPTH_ITEM, SHORT_TILES = SHORT };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<0>;
    BprParams a;
    struct Occ { RowT p; float s; };
    struct Seg { RowT q; float bk; };

    // versioned user table: the triple's record from the user pass rides in the entry: .z = user | version << 31, .w = s
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const
    {
        if constexpr (VER) {
            const uint2 su = reinterpret_cast<const uint2 *>(a.sbuf)[val >> 1];
            return make_uint4(key, val, su.y, su.x);
        }
        return make_uint4(key, val, 0, 0);
    }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        const uint32_t t = en.y >> 1;
        Occ o;
        if constexpr (VER) {
            o.p = RowT::load(((en.z >> 31) ? a.P2 : a.P) + (size_t)(en.z & 0x7fffffffu) * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            o.s = __uint_as_float(en.w);
        } else {
            o.p = RowT::load(a.stage + (size_t)t * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            o.s = a.sbuf[t];
        }
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        const uint32_t it = (key & a.kmask) - a.U;
        return Seg{RowT::load(a.Q + (size_t)it * DM::d(a.d), gl, DM::nchunk(a.nchunk)), a.b[it]};
    }
    __device__ __forceinline__ void compute(Seg &, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &, int) const
    {
        const float ss = (en.y & 1u) ? -o.s : o.s;   // positive side: +s*p, negative side: -s*p
        acc.v.fma(ss, o.p);
        acc.s0 += ss;
        acc.cnt += 1;
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        const uint32_t it = (key & a.kmask) - a.U;
        float *row = a.Q + (size_t)it * DM::d(a.d);
        const RowT q = seg ? seg->q : RowT::load(row, gl, DM::nchunk(a.nchunk));
        const float bk = seg ? seg->bk : a.b[it];
        const float nreg = (float)acc.cnt * a.reg, lr = a.lr;
        if constexpr (GRAD) {
            const RowT out = q.map2(acc.v, [=](float qv, float av) { return fmaf(nreg, qv, -av); });
            out.store(a.gQ + (size_t)it * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            if (gl == 0) a.gb[it] = fmaf(nreg, bk, -acc.s0);
        } else {
            const RowT out = q.map2(acc.v, [=](float qv, float av) { return fmaf(-lr, fmaf(nreg, qv, -av), qv); });
            out.store(row, gl, DM::nchunk(a.nchunk));
            if (gl == 0) a.b[it] = fmaf(-lr, fmaf(nreg, bk, -acc.s0), bk);
        }
    }
};

// Q[id] -= lr * (sum of the gradient rows sent for id), b likewise: the owner side of a sharded step
struct RowsApplyParams {
    float *Q, *b;
    const float *grad_rows, *grad_b;
    int d, nchunk;
    float lr;
};

template <int LPR_, int NCH_, int VEC_, bool FULL_>
struct RowsApplyPol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = 8 };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<0>;
    RowsApplyParams a;
    struct Occ { RowT g; float gb; };
    struct Seg { RowT q; float bk; };
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const { return make_uint4(key, val, 0, 0); }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        return Occ{RowT::load(a.grad_rows + (size_t)en.y * DM::d(a.d), gl, DM::nchunk(a.nchunk)), a.grad_b[en.y]};
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        return Seg{RowT::load(a.Q + (size_t)key * DM::d(a.d), gl, DM::nchunk(a.nchunk)), a.b[key]};
    }
    __device__ __forceinline__ void compute(Seg &, const Occ &o, uint4, Acc<RowT> &acc, Loss &, int) const
    {
        acc.v = acc.v + o.g;
        acc.s0 += o.gb;
        acc.cnt += 1;
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        float *row = a.Q + (size_t)key * DM::d(a.d);
        const RowT q = seg ? seg->q : RowT::load(row, gl, DM::nchunk(a.nchunk));
        const float bk = seg ? seg->bk : a.b[key];
        const float lr = a.lr;
        q.map2(acc.v, [=](float qv, float gv) { return fmaf(-lr, gv, qv); }).store(row, gl, DM::nchunk(a.nchunk));
        if (gl == 0) a.b[key] = fmaf(-lr, acc.s0, bk);
    }
};

__global__ __launch_bounds__(256) void iota_keys_kernel(const int32_t *__restrict__ ids, int64_t n, uint32_t I,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        uint32_t *__restrict__ flags)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    bool bad = false;
    keys[t] = clamp_index(ids[t], I, &bad);
    vals[t] = (uint32_t)t;
    if (bad) atomicOr(flags, TFREC_FLAG_BAD_INDEX);
}

// ------------------------------------------------------------------------------------------
// APR policies.  model/APR.py:64-136.
//   phase 1 (delta):  g = d L / d Theta with L = sum softplus(-clip(y_i - y_j, -80, 1e8)),
//                     delta_row = eps * g_row / sqrt(max(|g_row|^2, 1e-12)); the bias gradient is
//                     normalised as ONE vector, so its squared norm is a sum over all items.
//   phase 2 (Adagrad on L + reg_adv*L(Theta+delta) + 2*reg*mean-square).
// ------------------------------------------------------------------------------------------
struct AprParams {
    float *P, *Q, *b, *accP, *accQ, *accb, *dP, *dQ, *db;
    float *gQ, *gb;       // gradient-only item passes: [I, d] and [I] tables that receive the per-item sums
    float *stage;         // [B, d] phase 2: c_t = s*p + sa*(p+dp)
    float *sbuf;          // [B]    phase 1: s ; phase 2: s + sa
    const uint2 *pair;    // [B]    (i, j), range-checked
    const int32_t *users; // [B]    caller's user ids (clamped on use)
    uint32_t U;
    int d, nchunk;
    float lr, reg_coef /* 4*reg/(B*d) */, reg_adv, eps;
};

__device__ __forceinline__ float clip_sigmoid_neg(float r)   // -dL/dr of softplus(-clip(r))
{
    return (r >= -80.f && r <= 1e8f) ? sigmoid_neg(r) : 0.f;
}
__device__ __forceinline__ float clip_softplus_neg(float r) { return softplus_neg(fminf(fmaxf(r, -80.f), 1e8f)); }

// phase 1, user side: s_t and delta_P rows (P is not modified in this phase, so the item side
// gathers P[u] straight from the table)
template <int LPR_, int NCH_, int VEC_, bool FULL_, bool SHORT = false>
struct AprUser1Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = 4, SHORT_TILES = SHORT };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<0>;
    AprParams a;
    // (the user's row is read once per run of the user's occurrences: it rides in the segment state)
    struct Occ { RowT qi, qj; float bi, bj; };
    struct Seg { RowT p; };
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const
    {
        const uint2 ij = a.pair[val];
        return make_uint4(key, val, ij.x, ij.y);
    }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        Occ o;
        o.qi = RowT::load(a.Q + (size_t)en.z * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.qj = RowT::load(a.Q + (size_t)en.w * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.bi = a.b[en.z];
        o.bj = a.b[en.w];
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        return Seg{RowT::load(a.P + (size_t)key * DM::d(a.d), gl, DM::nchunk(a.nchunk))};
    }
    __device__ __forceinline__ void compute(Seg &seg, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &, int gl) const
    {
        const RowT diff = o.qi - o.qj;
        const float r = group_sum<LPR_>(seg.p.dot_partial(diff)) + (o.bi - o.bj);
        const float s = clip_sigmoid_neg(r);
        acc.v.fma(-s, diff);         // g_P = -sum s*(q_i - q_j)
        acc.cnt += 1;
        if (gl == 0) a.sbuf[en.y] = s;
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *, const Acc<RowT> &acc, int gl) const
    {
        const float n2 = group_sum<LPR_>(acc.v.dot_partial(acc.v));
        const float sc = a.eps * rsqrtf(fmaxf(n2, 1e-12f));
        RowT out = RowT::zero();
        out.fma(sc, acc.v);
        out.store(a.dP + (size_t)key * DM::d(a.d), gl, DM::nchunk(a.nchunk));
    }
};

// phase 1, item side: delta_Q rows, raw bias gradient into db, and sum of its squares.
// GRAD = true leaves the raw gradient sums in gQ / gb instead (they are then summed over the GPUs
// before they are normalised).
template <int LPR_, int NCH_, int VEC_, bool FULL_, bool GRAD = false, bool SHORT = false>
struct AprItem1Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = 8, SHORT_TILES = SHORT };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<0>;
    AprParams a;
    struct Occ { RowT p; float s; };
    struct Seg { int unused; };
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const { return make_uint4(key, val, 0, 0); }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        const uint32_t t = en.y >> 1;
        const int32_t ur = a.users[t];
        const uint32_t u = ur < 0 ? 0u : ((uint32_t)ur >= a.U ? a.U - 1 : (uint32_t)ur);
        return Occ{RowT::load(a.P + (size_t)u * DM::d(a.d), gl, DM::nchunk(a.nchunk)), a.sbuf[t]};
    }
    __device__ __forceinline__ Seg load_seg(uint32_t, int) const { return Seg{0}; }
    __device__ __forceinline__ void compute(Seg &, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &, int) const
    {
        const float ss = (en.y & 1u) ? o.s : -o.s;   // g_Q[i] = -s*p (positive side), +s*p (negative side)
        acc.v.fma(ss, o.p);
        acc.s0 += ss;
        acc.cnt += 1;
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *, const Acc<RowT> &acc, int gl) const
    {
        const uint32_t it = key - a.U;
        if constexpr (GRAD) {
            acc.v.store(a.gQ + (size_t)it * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            if (gl == 0) a.gb[it] = acc.s0;
        } else {
            const float n2 = group_sum<LPR_>(acc.v.dot_partial(acc.v));
            const float sc = a.eps * rsqrtf(fmaxf(n2, 1e-12f));
            RowT out = RowT::zero();
            out.fma(sc, acc.v);
            out.store(a.dQ + (size_t)it * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            if (gl == 0) a.db[it] = acc.s0;     // scaled once the global norm is known
        }
    }
};

// phase 2, user side
template <int LPR_, int NCH_, int VEC_, bool FULL_, bool LOSS, bool SHORT = false>
struct AprUser2Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = 2, SHORT_TILES = SHORT };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<LOSS ? 3 : 0>;
    AprParams a;
    // (the user's row and its delta row are read once per run of the user's occurrences)
    struct Occ { RowT qi, qj, dqi, dqj; float bi, bj, dbi, dbj; };
    struct Seg { RowT p, dp; };
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const
    {
        const uint2 ij = a.pair[val];
        return make_uint4(key, val, ij.x, ij.y);
    }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        Occ o;
        o.qi = RowT::load(a.Q + (size_t)en.z * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.qj = RowT::load(a.Q + (size_t)en.w * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.dqi = RowT::load(a.dQ + (size_t)en.z * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.dqj = RowT::load(a.dQ + (size_t)en.w * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        o.bi = a.b[en.z]; o.bj = a.b[en.w];
        o.dbi = a.db[en.z]; o.dbj = a.db[en.w];
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        return Seg{RowT::load(a.P + (size_t)key * DM::d(a.d), gl, DM::nchunk(a.nchunk)),
                   RowT::load(a.dP + (size_t)key * DM::d(a.d), gl, DM::nchunk(a.nchunk))};
    }
    __device__ __forceinline__ void compute(Seg &seg, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &loss, int gl) const
    {
        const RowT diff = o.qi - o.qj;
        const RowT pa = seg.p + seg.dp;
        const RowT diffa = (o.qi + o.dqi) - (o.qj + o.dqj);
        const float r = group_sum<LPR_>(seg.p.dot_partial(diff)) + (o.bi - o.bj);
        const float ra = group_sum<LPR_>(pa.dot_partial(diffa)) + ((o.bi + o.dbi) - (o.bj + o.dbj));
        const float s = clip_sigmoid_neg(r);
        const float sa = a.reg_adv * clip_sigmoid_neg(ra);
        // G_P += -s*(q_i-q_j) - sa*(q_i+dq_i - q_j-dq_j) + creg*p
        acc.v.fma(-s, diff);
        acc.v.fma(-sa, diffa);
        acc.v.fma(a.reg_coef, seg.p);
        acc.cnt += 1;
        // what the item side needs from this triple: c = s*p + sa*(p+dp)  and  s+sa
        RowT c = RowT::zero();
        c.fma(s, seg.p);
        c.fma(sa, pa);
        c.store(a.stage + (size_t)en.y * DM::d(a.d), gl, DM::nchunk(a.nchunk));
        if (gl == 0) a.sbuf[en.y] = s + sa;
        if constexpr (LOSS) {
            float m = seg.p.dot_partial(seg.p) + o.qi.dot_partial(o.qi) + o.qj.dot_partial(o.qj);
            if (gl == 0) {
                loss.t[0] += clip_softplus_neg(r);
                loss.t[1] += clip_softplus_neg(ra);
                m += (float)DM::d(a.d) * (o.bi * o.bi + o.bj * o.bj);
            }
            loss.t[2] += m;
        }
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        float *row = a.P + (size_t)key * DM::d(a.d), *arow = a.accP + (size_t)key * DM::d(a.d);
        const RowT p = seg ? seg->p : RowT::load(row, gl, DM::nchunk(a.nchunk));
        const RowT ac = RowT::load(arow, gl, DM::nchunk(a.nchunk));
        const RowT an = ac.map2(acc.v, [](float av, float g) { return fmaf(g, g, av); });
        an.store(arow, gl, DM::nchunk(a.nchunk));
        const float lr = a.lr;
        // w -= lr * g / sqrt(acc_new)
        RowT step = acc.v.map2(an, [=](float g, float av) { return lr * g / sqrtf(av); });
        (p - step).store(row, gl, DM::nchunk(a.nchunk));
    }
};

// phase 2, item side.  GRAD = true leaves the gradient sums (mean-square term included) in gQ / gb
// and touches neither the tables nor the accumulators.
template <int LPR_, int NCH_, int VEC_, bool FULL_, bool GRAD = false, bool SHORT = false>
struct AprItem2Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_, DEPTH = 8, SHORT_TILES = SHORT };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using DM = RowDims<LPR_, NCH_, VEC_, FULL_>;
    using Loss = LossAcc<0>;
    AprParams a;
    struct Occ { RowT c; float s; };
    struct Seg { RowT q; float bk; };
    __device__ __forceinline__ uint4 entry(uint32_t key, uint32_t val, int64_t) const { return make_uint4(key, val, 0, 0); }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        const uint32_t t = en.y >> 1;
        return Occ{RowT::load(a.stage + (size_t)t * DM::d(a.d), gl, DM::nchunk(a.nchunk)), a.sbuf[t]};
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        const uint32_t it = key - a.U;
        return Seg{RowT::load(a.Q + (size_t)it * DM::d(a.d), gl, DM::nchunk(a.nchunk)), a.b[it]};
    }
    __device__ __forceinline__ void compute(Seg &, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &, int) const
    {
        const float sg = (en.y & 1u) ? 1.f : -1.f;    // G_Q[i] = -c (positive side), +c (negative side)
        acc.v.fma(sg, o.c);
        acc.s0 += sg * o.s;
        acc.cnt += 1;
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        const uint32_t it = key - a.U;
        float *row = a.Q + (size_t)it * DM::d(a.d), *arow = a.accQ + (size_t)it * DM::d(a.d);
        const RowT q = seg ? seg->q : RowT::load(row, gl, DM::nchunk(a.nchunk));
        const float bk = seg ? seg->bk : a.b[it];
        const float n = (float)acc.cnt, creg = a.reg_coef, lr = a.lr;
        // every occurrence adds creg*q (rows) and creg*d*b (bias) from the mean-square term
        const RowT g = acc.v.map2(q, [=](float gv, float qv) { return fmaf(n * creg, qv, gv); });
        if constexpr (GRAD) {
            g.store(a.gQ + (size_t)it * DM::d(a.d), gl, DM::nchunk(a.nchunk));
            if (gl == 0) a.gb[it] = fmaf(n * creg * (float)DM::d(a.d), bk, acc.s0);
            return;
        }
        const RowT ac = RowT::load(arow, gl, DM::nchunk(a.nchunk));
        const RowT an = ac.map2(g, [](float av, float gv) { return fmaf(gv, gv, av); });
        an.store(arow, gl, DM::nchunk(a.nchunk));
        RowT step = g.map2(an, [=](float gv, float av) { return lr * gv / sqrtf(av); });
        (q - step).store(row, gl, DM::nchunk(a.nchunk));
        if (gl == 0) {
            const float gb = fmaf(n * creg * (float)DM::d(a.d), bk, acc.s0);
            const float abn = fmaf(gb, gb, a.accb[it]);
            a.accb[it] = abn;
            a.b[it] = bk - lr * gb / sqrtf(abn);
        }
    }
};

// Bias delta: db holds the raw per-item gradient for the batch's distinct items.  The squared
// norm is summed over the distinct item segments in sorted order (one workgroup, fixed order),
// then every distinct item's db is scaled in place.
__global__ __launch_bounds__(256) void apr_bias_norm_kernel(const uint32_t *__restrict__ keys, int64_t base, int64_t n, uint32_t U,
                                                            const float *__restrict__ db, double *__restrict__ out)
{
    __shared__ double red[256];
    double v = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const uint32_t k = keys[base + t];
        if (t == 0 || keys[base + t - 1] != k) {
            const double g = (double)db[k - U];
            v += g * g;
        }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

__global__ void apr_loss_out_kernel(const double *in, double *out, double mean_scale)
{
    out[0] = in[0];
    out[1] = in[1];
    out[2] = in[2] * mean_scale;
}

__global__ __launch_bounds__(256) void apr_bias_scale_kernel(const uint32_t *__restrict__ keys, int64_t base, int64_t n, uint32_t U,
                                                             float *__restrict__ db, const double *__restrict__ norm2, float eps)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t k = keys[base + t];
    if (t == 0 || keys[base + t - 1] != k) {
        const float sc = eps * rsqrtf(fmaxf((float)norm2[0], 1e-12f));
        db[k - U] *= sc;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct RowCfg { int lpr, nch, vec; bool full; };   // full: d == lpr * nch * vec, row geometry known at compile time

static bool pick_cfg(int d, bool aligned16, RowCfg *c)
{
    if (d <= 0) return false;
    if (d % 4 == 0 && aligned16) {
        const int n4 = d / 4;
        c->vec = 4; c->nch = 1;
        if (n4 <= 8) c->lpr = 8;
        else if (n4 <= 16) c->lpr = 16;
        else if (n4 <= 32) c->lpr = 32;
        else if (n4 <= 64) c->lpr = 64;
        else return false;
        c->full = (n4 == c->lpr);
        return true;
    }
    c->vec = 1; c->lpr = 64; c->full = false;
    if (d <= 64) c->nch = 1;
    else if (d <= 128) c->nch = 2;
    else if (d <= 256) c->nch = 4;
    else return false;
    return true;
}

// Calls f.template operator()<LPR,NCH,VEC,FULL>() for the chosen configuration.  WITH_FULL = false
// never picks the compile-time-geometry variants (for kernels that gain nothing from them).
template <bool WITH_FULL = true, class F>
static int dispatch_cfg(const RowCfg &c, F &&f)
{
    if (c.vec == 4) {
        if constexpr (WITH_FULL) {
            if (c.full) {
                switch (c.lpr) {
                    case 8: return f.template operator()<8, 1, 4, true>();
                    case 16: return f.template operator()<16, 1, 4, true>();
                    case 32: return f.template operator()<32, 1, 4, true>();
                    case 64: return f.template operator()<64, 1, 4, true>();
                }
            }
        }
        switch (c.lpr) {
            case 8: return f.template operator()<8, 1, 4, false>();
            case 16: return f.template operator()<16, 1, 4, false>();
            case 32: return f.template operator()<32, 1, 4, false>();
            case 64: return f.template operator()<64, 1, 4, false>();
        }
    } else {
        switch (c.nch) {
            case 1: return f.template operator()<64, 1, 1, false>();
            case 2: return f.template operator()<64, 2, 1, false>();
            case 4: return f.template operator()<64, 4, 1, false>();
        }
    }
    tfrec_set_error("internal: no kernel configuration");
    return TFREC_EUNSUP;
}

static int bits_for(uint64_t n)   // bits needed to represent values < n
{
    int b = 1;
    while (b < 32 && (1ull << b) < n) ++b;
    return b;
}

// The sorted index arrays of a batch live in a context-owned slot (not in the per-call
// workspace) so that a batch can be prepared on the auxiliary stream while the previous step
// still runs.
static size_t slot_bytes(int64_t n, size_t *sort_bytes)
{
    *sort_bytes = tfrec_sort_temp_bytes((size_t)3 * n, 32);
    return 2 * tf_align(12 * (size_t)n) + tf_align(8 * (size_t)n) + tf_align(*sort_bytes) + 4096;
}

// Sorts K consecutive batches of B triples (K*B index triples) into slot `sl`.
static int slot_prepare(tfrec_ctx *ctx, tfrec_batch_slot &sl, hipStream_t stream, const int32_t *users, const int32_t *pos,
                        const int32_t *neg, int64_t B, int64_t U, int64_t I, int K = 1)
{
    const int64_t n = B * K;
    const int kb = bits_for((uint64_t)U + (uint64_t)I);
    const int total_bits = K > 1 ? kb + bits_for((uint64_t)K) : kb;
    TF_REQUIRE(total_bits <= 32 && n < (1ll << 30), TFREC_EUNSUP, "%d batches of %lld triples do not fit the 32-bit sort keys", K, (long long)B);
    size_t sort_bytes = 0;
    const size_t need = slot_bytes(n, &sort_bytes);
    if (need > sl.bytes) {
        // growing: nothing may still be reading the old block
        TF_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->aux_stream) TF_HIP(hipStreamSynchronize(ctx->aux_stream));
        if (sl.mem) TF_HIP(hipFree(sl.mem));
        sl.mem = nullptr; sl.bytes = 0;
        hipError_t e = hipMalloc((void **)&sl.mem, need + need / 8);
        if (e != hipSuccess) { tfrec_set_error("batch slot allocation of %zu bytes failed: %s", need, hipGetErrorString(e)); return TFREC_ENOMEM; }
        sl.bytes = need + need / 8;
    }
    char *p = sl.mem;
    auto take = [&](size_t bytes) { char *r = p; p += tf_align(bytes); return r; };
    sl.keys = (uint32_t *)take(12 * (size_t)n);
    sl.vals = (uint32_t *)take(12 * (size_t)n);
    sl.pair = (uint2 *)take(8 * (size_t)n);
    void *sort_tmp = take(sort_bytes);
    // the first sort pass reads the index triples themselves (range check included) and leaves the checked items in pair
    // ceil(2^32 / B), non-zero exactly when batch numbers are wanted (B = 1: the quotient does not fit, and is not used)
    const uint32_t inv_B = K > 1 ? (B == 1 ? 0xffffffffu : (uint32_t)(((1ull << 32) + (uint64_t)B - 1) / (uint64_t)B)) : 0u;
    const tfrec_triple_src src{users, pos, neg, (uint32_t)n, (uint32_t)B, (uint32_t)U, (uint32_t)I, inv_B, kb,
                               reinterpret_cast<uint32_t *>(sl.pair), ctx->d_flags};
    TF_TRY(tfrec_sort_triples(sort_tmp, sort_bytes, src, sl.keys, sl.vals, total_bits, stream));
    sl.B = B; sl.U = U; sl.I = I; sl.K = K; sl.next_k = 0; sl.apr_pending = 0;
    sl.users_raw = users; sl.pos_raw = pos; sl.neg_raw = neg;
    sl.kmask = kb >= 32 ? 0xffffffffu : ((1u << kb) - 1u);
    return TFREC_OK;
}

struct StepWs {
    float *stage, *sbuf, *part_rows;
    float4 *part_sc;
    uint32_t *tile_flags;
    int64_t blk_slot0;       // partial-record slot of workgroup 0's pre-summed partial
    double *loss_part, *scalars;
    int64_t *bounds;         // [64] positions where the item half changes item range (item pass in pieces)
    int64_t n_tiles_u, n_tiles_i;
    int blocks_u, blocks_i;
};

// batches up to this many triples run the BPR row passes on short tiles
static constexpr int64_t SHORT_TILES_MAX_B = TF_SHORT_TILES_MAX_B;

// dispatch_cfg plus the choice between long and short tiles: f is called with <LPR, NCH, VEC, FULL, SHORT>
template <class F>
static int dispatch_cfg_tiles(const RowCfg &cfg, bool short_tiles, F &&f)
{
    return dispatch_cfg(cfg, [&]<int LPR, int NCH, int VEC, bool FULL>() -> int {
        // (a lane group of 64 has no shorter tile than the regular one)
        if constexpr (short_tile_ch(LPR) < tile_ch(LPR))
            if (short_tiles) return f.template operator()<LPR, NCH, VEC, FULL, true>();
        return f.template operator()<LPR, NCH, VEC, FULL, false>();
    });
}

// (sized for the short tiles of small batches where those may be used: more tiles, more partial records)
static size_t step_ws_bytes(int64_t B, int d, const RowCfg &c)
{
    const int CH = B <= SHORT_TILES_MAX_B ? short_tile_ch(c.lpr) : tile_ch(c.lpr);
    const int64_t tiles = (2 * B + CH - 1) / CH + 1;      // the larger (item) side
    const size_t rowf = (size_t)c.lpr * c.nch * c.vec;
    size_t s = 0;
    s += tf_align(sizeof(float) * (size_t)B * d);          // stage
    s += tf_align(sizeof(float) * 2 * B);                  // sbuf (8-byte records on a versioned user table)
    const int64_t slots = 2 * tiles + tiles / (256 / c.lpr) + 2;   // two per tile + one per workgroup
    s += tf_align(sizeof(float) * slots * rowf);           // partial rows
    s += tf_align(sizeof(float4) * slots);
    s += tf_align(sizeof(uint32_t) * tiles);
    s += tf_align(sizeof(double) * 4 * (tiles + 8));       // loss partials (<= one per block)
    s += tf_align(sizeof(double) * 16);
    s += tf_align(sizeof(int64_t) * 64);
    return s + 4096;
}

// upper bound over all row configurations, for tfrec_ctx_reserve
size_t tfrec_step_need_bytes(int64_t B, int d)
{
    RowCfg c;
    if (!pick_cfg(d, true, &c)) return 0;
    size_t best = step_ws_bytes(B, d, c);
    RowCfg c1;
    if (pick_cfg(d | 1, false, &c1)) {      // the scalar fallback used for unaligned tables
        c1.nch = d <= 64 ? 1 : (d <= 128 ? 2 : 4);
        const size_t alt = step_ws_bytes(B, d, c1);
        best = alt > best ? alt : best;
    }
    return best;
}

static int step_ws_carve(tfrec_ctx *ctx, int64_t B, int d, const RowCfg &c, StepWs *w, bool with_stage = true, bool short_tiles = false)
{
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + step_ws_bytes(B, d, c)));
    const int CH = short_tiles ? short_tile_ch(c.lpr) : tile_ch(c.lpr), GPB = 256 / c.lpr;
    const int64_t tiles = (2 * B + CH - 1) / CH + 1;
    const size_t rowf = (size_t)c.lpr * c.nch * c.vec;
    w->stage = with_stage ? (float *)tfrec_ws_take(ctx, sizeof(float) * (size_t)B * d) : nullptr;
    w->sbuf = (float *)tfrec_ws_take(ctx, sizeof(float) * 2 * B);
    const int64_t slots = 2 * tiles + tiles / GPB + 2;
    w->part_rows = (float *)tfrec_ws_take(ctx, sizeof(float) * slots * rowf);
    w->part_sc = (float4 *)tfrec_ws_take(ctx, sizeof(float4) * slots);
    w->tile_flags = (uint32_t *)tfrec_ws_take(ctx, sizeof(uint32_t) * tiles);
    w->blk_slot0 = 2 * tiles;
    w->loss_part = (double *)tfrec_ws_take(ctx, sizeof(double) * 4 * (tiles + 8));
    w->scalars = (double *)tfrec_ws_take(ctx, sizeof(double) * 16);
    w->bounds = (int64_t *)tfrec_ws_take(ctx, sizeof(int64_t) * 64);
    TF_REQUIRE(w->bounds != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    w->n_tiles_u = (B + CH - 1) / CH;
    w->n_tiles_i = (2 * B + CH - 1) / CH;
    w->blocks_u = (int)((w->n_tiles_u + GPB - 1) / GPB);
    w->blocks_i = (int)((w->n_tiles_i + GPB - 1) / GPB);
    return TFREC_OK;
}

static int check_step_args(const void *P, const void *Q, const void *b, int64_t U, int64_t I, int d,
                           const void *u, const void *i, const void *j, int64_t B)
{
    TF_REQUIRE(P && Q && b, TFREC_EINVAL, "null table pointer");
    TF_REQUIRE(U > 0 && I > 0 && d > 0, TFREC_EINVAL, "bad table shape U=%lld I=%lld d=%d", (long long)U, (long long)I, d);
    TF_REQUIRE((uint64_t)U + (uint64_t)I < (1ull << 32) && U < (1ll << 31) && I < (1ll << 31), TFREC_EUNSUP, "U+I must be below 2^32");
    TF_REQUIRE(B >= 0 && B < (1ll << 30), TFREC_EINVAL, "batch size %lld out of range", (long long)B);
    TF_REQUIRE(B == 0 || (u && i && j), TFREC_EINVAL, "null index pointer");
    return TFREC_OK;
}

static bool all_aligned16(std::initializer_list<const void *> ps)
{
    for (const void *p : ps)
        if (p && ((uintptr_t)p & 15u)) return false;
    return true;
}

template <class Pol>
static int run_pass(tfrec_ctx *ctx, const Pol &pol, const StepWs &w, const uint32_t *keys, const uint32_t *vals, int64_t base,
                    int64_t n, int64_t n_tiles, int blocks, bool with_loss_part, const char *tile_name, const char *fix_name,
                    const int64_t *d_range = nullptr)
{
    if (n <= 0) return TFREC_OK;
    seg_tile_kernel<Pol><<<blocks, 256, 0, ctx->stream>>>(pol, keys, vals, base, n, w.part_rows, w.part_sc, w.tile_flags, w.blk_slot0,
                                                         with_loss_part ? w.loss_part : nullptr, d_range);
    TF_HIP(hipGetLastError());
    tfrec_mark(ctx, tile_name);
    seg_fixup_kernel<Pol><<<blocks, 256, 0, ctx->stream>>>(pol, keys, base, n, w.part_rows, w.part_sc, w.tile_flags, n_tiles, w.blk_slot0,
                                                          d_range);
    TF_HIP(hipGetLastError());
    tfrec_mark(ctx, fix_name);
    return TFREC_OK;
}

// ------------------------------------------------------------------------------------------
// versioned user table
// ------------------------------------------------------------------------------------------
struct tfrec_utable {
    float *P = nullptr;       // the caller's table [U, d]: the rows whose version byte is 0
    float *P2 = nullptr;      // second row buffer [U, d] (owned): the rows whose version byte is 1
    uint8_t *ver = nullptr;   // [U] (owned)
    int64_t U = 0;
    int d = 0, device = 0;
};

#if TF_FLIP_KERNEL
// the batch's distinct users (heads of the sorted user keys) now live in the other buffer
__global__ __launch_bounds__(256) void utable_flip_kernel(const uint32_t *__restrict__ keys, int64_t n, uint32_t kmask,
                                                          uint8_t *__restrict__ ver)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t k = keys[t];
    if (t == 0 || keys[t - 1] != k) ver[k & kmask] ^= 1;
}
#endif

// one wave per row: rows that live in the second buffer are copied back, their version byte is cleared
__global__ __launch_bounds__(256) void utable_flatten_kernel(float *__restrict__ P, const float *__restrict__ P2, uint8_t *__restrict__ ver,
                                                             int64_t U, int d)
{
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < U; r += stride) {
        if (ver[r] == 0) continue;                      // wave-uniform
        for (int f = lane; f < d; f += 64) P[(size_t)r * d + f] = P2[(size_t)r * d + f];
        if (lane == 0) ver[r] = 0;                      // (every lane of the wave has read it by now: one instruction stream)
    }
}

// Where the item half of a sorted batch changes from one range of item ids to the next: pos[c] = first position
// (relative to the item half) whose item is >= c * items_per_chunk; pos[n_chunks] = n.  One lane per boundary.
__global__ void item_chunk_bounds_kernel(const uint32_t *__restrict__ item_keys, int64_t n, uint32_t kmask, uint32_t U,
                                         uint32_t items_per_chunk, int n_chunks, int64_t *__restrict__ pos)
{
    const int c = threadIdx.x;
    if (c > n_chunks) return;
    if (c == n_chunks) { pos[c] = n; return; }
    const uint64_t want = (uint64_t)U + (uint64_t)c * items_per_chunk;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)(item_keys[mid] & kmask) < want) lo = mid + 1; else hi = mid;
    }
    pos[c] = lo;
}

// Item pass in pieces: what a caller passes to have the item half of a step run one range of item ids after the
// other, with an event recorded on the stream when a range's gradient sums are complete.
struct ItemChunks {
    int n = 1;                         // ranges of ceil(I / n) items
    int64_t I = 0;
    hipEvent_t *done = nullptr;        // [n]
};

// the row passes of one BPR step over an already sorted batch; ut: the versioned user table, or null for the
// plain layout (the pre-step user rows are then staged for the item pass)
static int bpr_passes(tfrec_ctx *ctx, const tfrec_batch_slot &sl, float *d_P, float *d_Q, float *d_b, int64_t U, int d,
                      float lr, float reg, double *d_loss, bool item_grad_only = false, float *d_gQ = nullptr, float *d_gb = nullptr,
                      const tfrec_utable *ut = nullptr, const ItemChunks *chunks = nullptr)
{
    const int64_t B = sl.B;
    const int64_t k0 = sl.next_k;                 // which of the slot's batches this step consumes
    const uint32_t *keys = sl.keys + 3 * B * k0, *vals = sl.vals + 3 * B * k0;
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_P, d_Q, ut ? ut->P2 : nullptr}), &cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    StepWs w;
    const bool short_tiles = B <= SHORT_TILES_MAX_B;
    TF_TRY(step_ws_carve(ctx, B, d, cfg, &w, ut == nullptr, short_tiles));
    BprParams a{d_P, d_Q, d_b, ut ? ut->P2 : nullptr, ut ? ut->ver : nullptr, d_gQ ? d_gQ : d_Q, d_gb ? d_gb : d_b, w.stage, w.sbuf,
                sl.pair + B * k0, sl.users_raw + B * k0, (uint32_t)U, d, d / cfg.vec, lr, reg, sl.kmask};
    const bool want_loss = d_loss != nullptr;
    int64_t *d_bounds = nullptr;
    if (chunks && chunks->n > 1) {
        TF_REQUIRE(item_grad_only && chunks->n <= 32 && chunks->done, TFREC_EINVAL, "internal: bad item chunking");
        d_bounds = w.bounds;
        const uint32_t ipc = (uint32_t)((chunks->I + chunks->n - 1) / chunks->n);
        item_chunk_bounds_kernel<<<1, 64, 0, ctx->stream>>>(keys + B, 2 * B, sl.kmask, (uint32_t)U, ipc, chunks->n, d_bounds);
        TF_HIP(hipGetLastError());
    }
    auto passes = [&]<int LPR, int NCH, int VEC, bool FULL, bool VER, bool SHORT>() -> int {
        if (want_loss) {
            TF_TRY(run_pass(ctx, BprUserPol<LPR, NCH, VEC, FULL, true, VER, SHORT>{a}, w, keys, vals, 0, B, w.n_tiles_u, w.blocks_u, true,
                            "user_tile", "user_fixup"));
            loss_finalize_kernel<2><<<1, 256, 0, ctx->stream>>>(w.loss_part, w.blocks_u, d_loss);
            TF_HIP(hipGetLastError());
        } else {
            TF_TRY(run_pass(ctx, BprUserPol<LPR, NCH, VEC, FULL, false, VER, SHORT>{a}, w, keys, vals, 0, B, w.n_tiles_u, w.blocks_u, false,
                            "user_tile", "user_fixup"));
        }
        if (item_grad_only && chunks && chunks->n > 1) {
            // one launch pair per range of item ids; the ranges' positions are found on the device, every launch is
            // sized for the whole item half
            for (int c = 0; c < chunks->n; ++c) {
                TF_TRY(run_pass(ctx, BprItemPol<LPR, NCH, VEC, FULL, true, VER, SHORT>{a}, w, keys, vals, B, 2 * B, w.n_tiles_i, w.blocks_i,
                                false, "item_tile", "item_fixup", d_bounds + c));
                TF_HIP(hipEventRecord(chunks->done[c], ctx->stream));
            }
            return TFREC_OK;
        }
        if (item_grad_only)
            return run_pass(ctx, BprItemPol<LPR, NCH, VEC, FULL, true, VER, SHORT>{a}, w, keys, vals, B, 2 * B, w.n_tiles_i, w.blocks_i, false,
                            "item_tile", "item_fixup");
        return run_pass(ctx, BprItemPol<LPR, NCH, VEC, FULL, false, VER, SHORT>{a}, w, keys, vals, B, 2 * B, w.n_tiles_i, w.blocks_i, false,
                        "item_tile", "item_fixup");
    };
    auto layout = [&]<bool VER>() -> int {
        return dispatch_cfg_tiles(cfg, short_tiles, [&]<int LPR, int NCH, int VEC, bool FULL, bool SHORT>() -> int {
            return passes.template operator()<LPR, NCH, VEC, FULL, VER, SHORT>();
        });
    };
    if (!ut) return layout.template operator()<false>();
#if TF_FLIP_KERNEL
    TF_TRY(layout.template operator()<true>());
    utable_flip_kernel<<<(unsigned)((B + 255) / 256), 256, 0, ctx->stream>>>(keys, B, sl.kmask, ut->ver);
    TF_HIP(hipGetLastError());
    tfrec_mark(ctx, "version_flip");
    return TFREC_OK;
#else
    return layout.template operator()<true>();      // (the user pass flips the version bytes of the batch's users as it goes)
#endif
}

static int check_utable(tfrec_ctx *ctx, const tfrec_utable *t)
{
    TF_REQUIRE(t != nullptr && t->P && t->P2 && t->ver, TFREC_EINVAL, "null or destroyed user table");
    TF_REQUIRE(t->device == ctx->device, TFREC_EINVAL, "the user table lives on device %d, the context on device %d", t->device, ctx->device);
    return TFREC_OK;
}

extern "C" int tfrec_utable_create(tfrec_ctx *ctx, float *d_P, int64_t U, int d, tfrec_utable **out)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(out != nullptr && d_P != nullptr, TFREC_EINVAL, "null pointer");
    TF_REQUIRE(U > 0 && U < (1ll << 31) && d > 0 && d <= 256, TFREC_EINVAL, "bad table shape U=%lld d=%d", (long long)U, d);
    tfrec_utable *t = new (std::nothrow) tfrec_utable();
    TF_REQUIRE(t != nullptr, TFREC_ENOMEM, "host allocation failed");
    t->P = d_P; t->U = U; t->d = d; t->device = ctx->device;
    hipError_t e = hipMalloc((void **)&t->P2, sizeof(float) * (size_t)U * d);
    if (e == hipSuccess) e = hipMalloc((void **)&t->ver, (size_t)U);
    if (e == hipSuccess) e = hipMemsetAsync(t->ver, 0, (size_t)U, ctx->stream);
    if (e != hipSuccess) {
        tfrec_set_error("allocation of the second row buffer (%zu bytes) failed: %s", sizeof(float) * (size_t)U * d, hipGetErrorString(e));
        if (t->P2) (void)hipFree(t->P2);
        if (t->ver) (void)hipFree(t->ver);
        delete t;
        return TFREC_ENOMEM;
    }
    *out = t;
    return TFREC_OK;
}

extern "C" int tfrec_utable_flatten(tfrec_ctx *ctx, tfrec_utable *t)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_utable(ctx, t));
    const int64_t blocks = (t->U + 3) / 4;
    utable_flatten_kernel<<<(unsigned)(blocks < 65536 ? blocks : 65536), 256, 0, ctx->stream>>>(t->P, t->P2, t->ver, t->U, t->d);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_utable_destroy(tfrec_ctx *ctx, tfrec_utable *t)
{
    if (!t) return TFREC_OK;
    // the buffers are released whatever the flatten says (its error, if any, is what the call returns)
    const int rc = tfrec_utable_flatten(ctx, t);
    if (ctx) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    } else {
        (void)hipDeviceSynchronize();
    }
    if (t->P2) (void)hipFree(t->P2);
    if (t->ver) (void)hipFree(t->ver);
    t->P2 = nullptr; t->ver = nullptr; t->P = nullptr;
    delete t;
    return rc;
}

extern "C" int tfrec_bpr_grad_step(tfrec_ctx *ctx, float *d_P, float *d_Qc, float *d_bc, int64_t U, int64_t Ic, int d,
                                   const int32_t *d_users, const int32_t *d_pos_slot, const int32_t *d_neg_slot, int64_t B,
                                   float lr, float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Qc, d_bc, U, Ic, d, d_users, d_pos_slot, d_neg_slot, B));
    if (B == 0) {
        if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 2 * sizeof(double), ctx->stream));
        return TFREC_OK;
    }
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    tfrec_batch_slot &sl = ctx->slots[2];
    TF_TRY(slot_prepare(ctx, sl, ctx->stream, d_users, d_pos_slot, d_neg_slot, B, U, Ic));
    tfrec_mark(ctx, "index");
    return bpr_passes(ctx, sl, d_P, d_Qc, d_bc, U, d, lr, reg, d_loss, true);
}

extern "C" int tfrec_rows_sgd_apply(tfrec_ctx *ctx, float *d_Q, float *d_b, int64_t I, int d, const int32_t *d_ids,
                                    const float *d_grad_rows, const float *d_grad_b, int64_t n, float lr)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(d_Q && d_b && I > 0 && I < (1ll << 31) && d > 0, TFREC_EINVAL, "bad table");
    TF_REQUIRE(n >= 0 && n < (1ll << 31), TFREC_EINVAL, "bad row count");
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_ids && d_grad_rows && d_grad_b, TFREC_EINVAL, "null pointer");
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_Q, d_grad_rows}), &cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    const int CH = tile_ch(cfg.lpr), GPB = 256 / cfg.lpr;
    const int64_t tiles = (n + CH - 1) / CH + 1;
    const size_t rowf = (size_t)cfg.lpr * cfg.nch * cfg.vec;
    const int end_bit = bits_for((uint64_t)I);
    const size_t sort_bytes = tfrec_sort_temp_bytes((size_t)n, end_bit);
    const int64_t slots = 2 * tiles + tiles / GPB + 2;
    const size_t need = 4 * tf_align(4 * (size_t)n) + tf_align(sort_bytes) + tf_align(sizeof(float) * slots * rowf) +
                        tf_align(sizeof(float4) * slots) + tf_align(4 * tiles) + 8192;
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + need));
    uint32_t *keys_in = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n), *vals_in = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n);
    uint32_t *keys = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n), *vals = (uint32_t *)tfrec_ws_take(ctx, 4 * (size_t)n);
    void *sort_tmp = tfrec_ws_take(ctx, sort_bytes);
    StepWs w{};
    w.part_rows = (float *)tfrec_ws_take(ctx, sizeof(float) * slots * rowf);
    w.part_sc = (float4 *)tfrec_ws_take(ctx, sizeof(float4) * slots);
    w.tile_flags = (uint32_t *)tfrec_ws_take(ctx, 4 * tiles);
    w.blk_slot0 = 2 * tiles;
    TF_REQUIRE(w.tile_flags != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    const int64_t n_tiles = (n + CH - 1) / CH;
    const int blocks = (int)((n_tiles + GPB - 1) / GPB);
    iota_keys_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(d_ids, n, (uint32_t)I, keys_in, vals_in, ctx->d_flags);
    TF_HIP(hipGetLastError());
    TF_TRY(tfrec_sort_pairs(sort_tmp, sort_bytes, keys_in, keys, vals_in, vals, (size_t)n, end_bit, ctx->stream));
    RowsApplyParams a{d_Q, d_b, d_grad_rows, d_grad_b, d, d / cfg.vec, lr};
    return dispatch_cfg(cfg, [&]<int LPR, int NCH, int VEC, bool FULL>() -> int {
        return run_pass(ctx, RowsApplyPol<LPR, NCH, VEC, FULL>{a}, w, keys, vals, 0, n, n_tiles, blocks, false, "apply_tile", "apply_fixup");
    });
}

static int sgd_step_impl(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b, int64_t U, int64_t I, int d, const int32_t *d_users,
                         const int32_t *d_pos, const int32_t *d_neg, int64_t B, float lr, float reg, double *d_loss,
                         const tfrec_utable *ut)
{
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B));
    if (B == 0) {
        if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 2 * sizeof(double), ctx->stream));
        return TFREC_OK;
    }
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    tfrec_batch_slot &sl = ctx->slots[2];      // private slot of the unpipelined entry points, main stream only
    TF_TRY(slot_prepare(ctx, sl, ctx->stream, d_users, d_pos, d_neg, B, U, I));
    tfrec_mark(ctx, "index");
    return bpr_passes(ctx, sl, d_P, d_Q, d_b, U, d, lr, reg, d_loss, false, nullptr, nullptr, ut);
}

extern "C" int tfrec_bpr_sgd_step(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b, int64_t U, int64_t I, int d,
                                  const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B, float lr,
                                  float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    return sgd_step_impl(ctx, d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B, lr, reg, d_loss, nullptr);
}

extern "C" int tfrec_bpr_sgd_step_v(tfrec_ctx *ctx, tfrec_utable *t, float *d_Q, float *d_b, int64_t I, const int32_t *d_users,
                                    const int32_t *d_pos, const int32_t *d_neg, int64_t B, float lr, float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_utable(ctx, t));
    return sgd_step_impl(ctx, t->P, d_Q, d_b, t->U, I, t->d, d_users, d_pos, d_neg, B, lr, reg, d_loss, t);
}

// ------------------------------------------------------------------------------------------
// The step for callers that cannot allocate (code under HIP graph capture, tight serving or
// training loops, harnesses that forbid hipMalloc): every byte of device scratch -- sorted index
// arrays, sort counters, staged rows, partial sums -- lives in a file-scope device array instead of
// the context's growable workspace.  Each GPU holds its own copy of the array at its own address;
// host code reaches it only through the address hipGetSymbolAddress returned for that device.
// ------------------------------------------------------------------------------------------
static constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)16 << 20;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static constexpr int NOALLOC_MAX_DEVICES = 64;
static std::atomic<void *> g_scratch_ptr[NOALLOC_MAX_DEVICES];     // per device: where `scratch` lives on it

int tfrec_scratch_bind(int device)
{
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return TFREC_OK;     // the no-allocation step is not offered there
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return TFREC_OK;
    void *p = nullptr;
    TF_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    TF_REQUIRE(p != nullptr, TFREC_EHIP, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[device].store(p, std::memory_order_release);
    return TFREC_OK;
}

extern "C" int tfrec_bpr_sgd_step_noalloc(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b, int64_t U, int64_t I, int d,
                                          const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B, float lr,
                                          float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B));
    if (B == 0) {
        if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 2 * sizeof(double), ctx->stream));
        return TFREC_OK;
    }
    TF_REQUIRE(ctx->device >= 0 && ctx->device < NOALLOC_MAX_DEVICES, TFREC_EUNSUP, "the no-allocation step serves devices 0..%d",
               NOALLOC_MAX_DEVICES - 1);
    char *base = (char *)g_scratch_ptr[ctx->device].load(std::memory_order_acquire);
    TF_REQUIRE(base != nullptr, TFREC_EHIP, "the scratch array of device %d was never looked up", ctx->device);
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_P, d_Q}), &cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    size_t sort_bytes = 0;
    const size_t slot_need = tf_align(slot_bytes(B, &sort_bytes)), ws_need = step_ws_bytes(B, d, cfg);
    TF_REQUIRE(slot_need + ws_need <= NOALLOC_SCRATCH_BYTES, TFREC_EUNSUP,
               "a batch of %lld triples at d=%d needs %zu bytes of scratch; the no-allocation step holds %zu (use tfrec_bpr_sgd_step)",
               (long long)B, d, slot_need + ws_need, NOALLOC_SCRATCH_BYTES);
    // the index arrays at the front of the array, the row passes' workspace behind them; nothing here has to start at
    // zero (every word a kernel reads was written earlier in the same call)
    tfrec_batch_slot sl;
    sl.mem = base;
    sl.bytes = slot_need;
    char *const saved_ws = ctx->ws;
    const size_t saved_bytes = ctx->ws_bytes;
    const bool saved_fixed = ctx->ws_fixed, saved_profile = ctx->profile;
    ctx->ws = base + slot_need;
    ctx->ws_bytes = NOALLOC_SCRATCH_BYTES - slot_need;
    ctx->ws_used = 0;
    ctx->ws_fixed = true;
    ctx->profile = false;              // phase marks create events on first use
    int rc = slot_prepare(ctx, sl, ctx->stream, d_users, d_pos, d_neg, B, U, I);
    if (rc == TFREC_OK) rc = bpr_passes(ctx, sl, d_P, d_Q, d_b, U, d, lr, reg, d_loss);
    ctx->ws = saved_ws;
    ctx->ws_bytes = saved_bytes;
    ctx->ws_used = 0;
    ctx->ws_fixed = saved_fixed;
    ctx->profile = saved_profile;
    return rc;
}

extern "C" int tfrec_batch_prepare(tfrec_ctx *ctx, int slot, int64_t U, int64_t I, const int32_t *d_users, const int32_t *d_pos,
                                   const int32_t *d_neg, int64_t B)
{
    return tfrec_batch_prepare_many(ctx, slot, U, I, d_users, d_pos, d_neg, B, 1);
}

extern "C" int tfrec_batch_prepare_many(tfrec_ctx *ctx, int slot, int64_t U, int64_t I, const int32_t *d_users,
                                        const int32_t *d_pos, const int32_t *d_neg, int64_t B, int K)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(slot == 0 || slot == 1, TFREC_EINVAL, "slot must be 0 or 1");
    TF_REQUIRE(K >= 1 && K <= 64, TFREC_EINVAL, "between 1 and 64 batches per preparation");
    TF_REQUIRE(U > 0 && I > 0 && U < (1ll << 31) && I < (1ll << 31), TFREC_EINVAL, "bad table shape");
    TF_REQUIRE(B >= 1 && B < (1ll << 30) && d_users && d_pos && d_neg, TFREC_EINVAL, "bad batch");
    if (!ctx->aux_stream) {
        // An ordinary stream: a high priority bought no throughput when it was measured.  (A hang was once seen in a
        // run that had this stream at high priority; a stand-alone replay of the event protocol below,
        // tools/stream_probe.hip, does not stall at either priority, so the priority is not known to be its cause.)
        TF_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    }
    tfrec_batch_slot &sl = ctx->slots[slot];
    if (!sl.ready) TF_HIP(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    if (!sl.released) TF_HIP(hipEventCreateWithFlags(&sl.released, hipEventDisableTiming));
    // the index arrays of this slot may still be read by the step that used it last
    if (sl.has_released) TF_HIP(hipStreamWaitEvent(ctx->aux_stream, sl.released, 0));
    // the caller's index arrays were produced on the main stream
    if (!ctx->fork_event) TF_HIP(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
    TF_HIP(hipEventRecord(ctx->fork_event, ctx->stream));
    TF_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->fork_event, 0));
    sl.B = -1;
    TF_TRY(slot_prepare(ctx, sl, ctx->aux_stream, d_users, d_pos, d_neg, B, U, I, K));
    TF_HIP(hipEventRecord(sl.ready, ctx->aux_stream));
    return TFREC_OK;
}

static int step_prepared_impl(tfrec_ctx *ctx, int slot, float *d_P, float *d_Q, float *d_b, int64_t U, int64_t I, int d, float lr,
                              float reg, double *d_loss, const tfrec_utable *ut, bool item_grad_only = false)
{
    TF_REQUIRE(slot == 0 || slot == 1, TFREC_EINVAL, "slot must be 0 or 1");
    tfrec_batch_slot &sl = ctx->slots[slot];
    TF_REQUIRE(sl.B >= 1, TFREC_EINVAL, "slot %d holds no prepared batch", slot);
    TF_REQUIRE(sl.U == U && sl.I == I, TFREC_EINVAL, "slot %d was prepared for a %lldx%lld model", slot, (long long)sl.U, (long long)sl.I);
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_P, d_P, d_P, sl.B));
    TF_HIP(hipStreamWaitEvent(ctx->stream, sl.ready, 0));
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    const int rc = bpr_passes(ctx, sl, d_P, d_Q, d_b, U, d, lr, reg, d_loss, item_grad_only, nullptr, nullptr, ut);
    if (++sl.next_k >= sl.K) {      // the last batch of the slot: its arrays may be overwritten once this step is done
        TF_HIP(hipEventRecord(sl.released, ctx->stream));
        sl.has_released = true;
        sl.B = -1;
    }
    return rc;
}

extern "C" int tfrec_bpr_sgd_step_prepared(tfrec_ctx *ctx, int slot, float *d_P, float *d_Q, float *d_b, int64_t U, int64_t I,
                                           int d, float lr, float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    return step_prepared_impl(ctx, slot, d_P, d_Q, d_b, U, I, d, lr, reg, d_loss, nullptr);
}

extern "C" int tfrec_bpr_grad_step_prepared(tfrec_ctx *ctx, int slot, float *d_P, float *d_Qc, float *d_bc, int64_t U, int64_t Ic, int d,
                                            float lr, float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    return step_prepared_impl(ctx, slot, d_P, d_Qc, d_bc, U, Ic, d, lr, reg, d_loss, nullptr, true);
}

extern "C" int tfrec_bpr_grad_step_prepared_v(tfrec_ctx *ctx, int slot, tfrec_utable *t, float *d_Qc, float *d_bc, int64_t Ic, float lr,
                                              float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_utable(ctx, t));
    return step_prepared_impl(ctx, slot, t->P, d_Qc, d_bc, t->U, Ic, t->d, lr, reg, d_loss, t, true);
}

extern "C" int tfrec_bpr_sgd_step_prepared_v(tfrec_ctx *ctx, int slot, tfrec_utable *t, float *d_Q, float *d_b, int64_t I, float lr,
                                             float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_utable(ctx, t));
    return step_prepared_impl(ctx, slot, t->P, d_Q, d_b, t->U, I, t->d, lr, reg, d_loss, t);
}

// The step for a GPU that holds a shard of P and a full copy of Q, b: P is updated in place; the
// item side leaves, in the zeroed table G = [I*d gradient rows | I bias gradients], the gradient sum
// (regularisation term included) of every item this batch touches.  Summing G over the GPUs and
// applying Q -= lr*G on every copy is the unsharded step.
static int dense_grad_step_impl(tfrec_ctx *ctx, int slot, float *d_P, const float *d_Q, const float *d_b, float *d_G, int64_t U,
                                int64_t I, int d, const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                                float lr, float reg, double *d_loss, const tfrec_utable *ut, const ItemChunks *chunks = nullptr)
{
    TF_REQUIRE(slot >= -1 && slot <= 1, TFREC_EINVAL, "slot must be -1 (sort now), 0 or 1");
    TF_REQUIRE(d_G != nullptr && ((uintptr_t)d_G & 15u) == 0, TFREC_EINVAL, "gradient table missing or not 16-byte aligned");
    if (slot < 0) TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B));
    else TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_P, d_P, d_P, 1));
    TF_HIP(hipMemsetAsync(d_G, 0, sizeof(float) * ((size_t)I * d + (size_t)I), ctx->stream));
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    float *gQ = d_G, *gb = d_G + (size_t)I * d;
    if (slot < 0) {
        if (B == 0) {
            if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 2 * sizeof(double), ctx->stream));
            if (chunks)          // nothing to add: the zeroed table is complete
                for (int c = 0; c < chunks->n; ++c) TF_HIP(hipEventRecord(chunks->done[c], ctx->stream));
            return TFREC_OK;
        }
        tfrec_batch_slot &sl = ctx->slots[2];
        TF_TRY(slot_prepare(ctx, sl, ctx->stream, d_users, d_pos, d_neg, B, U, I));
        tfrec_mark(ctx, "index");
        return bpr_passes(ctx, sl, d_P, const_cast<float *>(d_Q), const_cast<float *>(d_b), U, d, lr, reg, d_loss, true, gQ, gb, ut, chunks);
    }
    tfrec_batch_slot &sl = ctx->slots[slot];
    TF_REQUIRE(sl.B >= 1, TFREC_EINVAL, "slot %d holds no prepared batch", slot);
    TF_REQUIRE(sl.U == U && sl.I == I, TFREC_EINVAL, "slot %d was prepared for a %lldx%lld model", slot, (long long)sl.U, (long long)sl.I);
    TF_HIP(hipStreamWaitEvent(ctx->stream, sl.ready, 0));
    const int rc = bpr_passes(ctx, sl, d_P, const_cast<float *>(d_Q), const_cast<float *>(d_b), U, d, lr, reg, d_loss, true, gQ, gb, ut, chunks);
    if (++sl.next_k >= sl.K) {
        TF_HIP(hipEventRecord(sl.released, ctx->stream));
        sl.has_released = true;
        sl.B = -1;
    }
    return rc;
}

extern "C" int tfrec_bpr_dense_grad_step(tfrec_ctx *ctx, int slot, float *d_P, const float *d_Q, const float *d_b, float *d_G,
                                         int64_t U, int64_t I, int d, const int32_t *d_users, const int32_t *d_pos,
                                         const int32_t *d_neg, int64_t B, float lr, float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    return dense_grad_step_impl(ctx, slot, d_P, d_Q, d_b, d_G, U, I, d, d_users, d_pos, d_neg, B, lr, reg, d_loss, nullptr);
}

extern "C" int tfrec_bpr_dense_grad_step_v(tfrec_ctx *ctx, int slot, tfrec_utable *t, const float *d_Q, const float *d_b, float *d_G,
                                           int64_t I, const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B,
                                           float lr, float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_utable(ctx, t));
    return dense_grad_step_impl(ctx, slot, t->P, d_Q, d_b, d_G, t->U, I, t->d, d_users, d_pos, d_neg, B, lr, reg, d_loss, t);
}

extern "C" int tfrec_bpr_dense_grad_step_chunked(tfrec_ctx *ctx, int slot, tfrec_utable *t, const float *d_Q, const float *d_b, float *d_G,
                                                 int64_t I, const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg,
                                                 int64_t B, float lr, float reg, double *d_loss, int n_chunks, void **chunk_done)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_utable(ctx, t));
    TF_REQUIRE(n_chunks >= 1 && n_chunks <= 32, TFREC_EINVAL, "n_chunks must be 1..32, got %d", n_chunks);
    TF_REQUIRE(chunk_done != nullptr, TFREC_EINVAL, "null event array");
    for (int c = 0; c < n_chunks; ++c) TF_REQUIRE(chunk_done[c] != nullptr, TFREC_EINVAL, "event %d is null", c);
    ItemChunks ch;
    ch.n = n_chunks; ch.I = I; ch.done = reinterpret_cast<hipEvent_t *>(chunk_done);
    if (n_chunks == 1) {
        TF_TRY(dense_grad_step_impl(ctx, slot, t->P, d_Q, d_b, d_G, t->U, I, t->d, d_users, d_pos, d_neg, B, lr, reg, d_loss, t));
        TF_HIP(hipEventRecord(ch.done[0], ctx->stream));
        return TFREC_OK;
    }
    return dense_grad_step_impl(ctx, slot, t->P, d_Q, d_b, d_G, t->U, I, t->d, d_users, d_pos, d_neg, B, lr, reg, d_loss, t, &ch);
}

__global__ __launch_bounds__(256) void axpy_kernel(float *__restrict__ y, const float *__restrict__ x, float alpha, int64_t n4, int64_t n)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n4) {
        float4 a = reinterpret_cast<float4 *>(y)[t];
        const float4 g = reinterpret_cast<const float4 *>(x)[t];
        a.x = fmaf(alpha, g.x, a.x); a.y = fmaf(alpha, g.y, a.y); a.z = fmaf(alpha, g.z, a.z); a.w = fmaf(alpha, g.w, a.w);
        reinterpret_cast<float4 *>(y)[t] = a;
    } else {
        const int64_t i = n4 * 4 + (t - n4);
        if (i < n) y[i] = fmaf(alpha, x[i], y[i]);
    }
}

// y[i] = fma(alpha, x[i], y[i]): with alpha = -lr and x the summed gradient table this is the SGD update
// of model/BPR.py:48 on a replicated table.
extern "C" int tfrec_axpy(tfrec_ctx *ctx, float *d_y, const float *d_x, float alpha, int64_t n)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0, TFREC_EINVAL, "negative length");
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_y && d_x, TFREC_EINVAL, "null array");
    const bool vec = (((uintptr_t)d_y | (uintptr_t)d_x) & 15u) == 0;
    const int64_t n4 = vec ? n / 4 : 0, threads = n4 + (n - 4 * n4);
    axpy_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, ctx->stream>>>(d_y, d_x, alpha, n4, n);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_apr_adagrad_step(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b, float *d_accP, float *d_accQ,
                                      float *d_accb, float *d_dP, float *d_dQ, float *d_db, int64_t U, int64_t I, int d,
                                      const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B, float lr,
                                      float reg, float reg_adv, float eps, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B));
    TF_REQUIRE(d_accP && d_accQ && d_accb && d_dP && d_dQ && d_db, TFREC_EINVAL, "null accumulator or delta table");
    if (B == 0) {
        if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 3 * sizeof(double), ctx->stream));
        return TFREC_OK;
    }
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_P, d_Q, d_accP, d_accQ, d_dP, d_dQ}), &cfg), TFREC_EUNSUP,
               "factor size d=%d is not supported (max 256)", d);
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    tfrec_batch_slot &sl = ctx->slots[2];
    TF_TRY(slot_prepare(ctx, sl, ctx->stream, d_users, d_pos, d_neg, B, U, I));
    tfrec_mark(ctx, "index");
    StepWs w;
    const bool short_tiles = B <= SHORT_TILES_MAX_B;
    TF_TRY(step_ws_carve(ctx, B, d, cfg, &w, true, short_tiles));

    AprParams a{};
    a.P = d_P; a.Q = d_Q; a.b = d_b; a.accP = d_accP; a.accQ = d_accQ; a.accb = d_accb;
    a.dP = d_dP; a.dQ = d_dQ; a.db = d_db;
    a.stage = w.stage; a.sbuf = w.sbuf; a.pair = sl.pair; a.users = d_users;
    a.U = (uint32_t)U; a.d = d; a.nchunk = d / cfg.vec;
    a.lr = lr; a.reg_coef = (float)(4.0 * (double)reg / ((double)B * (double)d)); a.reg_adv = reg_adv; a.eps = eps;
    const bool want_loss = d_loss != nullptr;
    return dispatch_cfg_tiles(cfg, short_tiles, [&]<int LPR, int NCH, int VEC, bool FULL, bool SHORT>() -> int {
        // phase 1: deltas
        TF_TRY(run_pass(ctx, AprUser1Pol<LPR, NCH, VEC, FULL, SHORT>{a}, w, sl.keys, sl.vals, 0, B, w.n_tiles_u, w.blocks_u, false,
                        "delta_user_tile", "delta_user_fixup"));
        TF_TRY(run_pass(ctx, AprItem1Pol<LPR, NCH, VEC, FULL, false, SHORT>{a}, w, sl.keys, sl.vals, B, 2 * B, w.n_tiles_i, w.blocks_i, false,
                        "delta_item_tile", "delta_item_fixup"));
        const int nb = (int)((2 * B + 255) / 256 < 512 ? (2 * B + 255) / 256 : 512);
        apr_bias_norm_kernel<<<nb, 256, 0, ctx->stream>>>(sl.keys, B, 2 * B, (uint32_t)U, d_db, w.loss_part);
        TF_HIP(hipGetLastError());
        loss_finalize_kernel<1><<<1, 256, 0, ctx->stream>>>(w.loss_part, nb, w.scalars);
        TF_HIP(hipGetLastError());
        apr_bias_scale_kernel<<<(int)((2 * B + 255) / 256), 256, 0, ctx->stream>>>(sl.keys, B, 2 * B, (uint32_t)U, d_db, w.scalars, eps);
        TF_HIP(hipGetLastError());
        tfrec_mark(ctx, "delta_bias_norm");
        // phase 2: Adagrad step at (Theta, delta)
        if (want_loss) {
            TF_TRY(run_pass(ctx, AprUser2Pol<LPR, NCH, VEC, FULL, true, SHORT>{a}, w, sl.keys, sl.vals, 0, B, w.n_tiles_u, w.blocks_u, true,
                            "user_tile", "user_fixup"));
            loss_finalize_kernel<3><<<1, 256, 0, ctx->stream>>>(w.loss_part, w.blocks_u, w.scalars + 4);
            TF_HIP(hipGetLastError());
        } else {
            TF_TRY(run_pass(ctx, AprUser2Pol<LPR, NCH, VEC, FULL, false, SHORT>{a}, w, sl.keys, sl.vals, 0, B, w.n_tiles_u, w.blocks_u, false,
                            "user_tile", "user_fixup"));
        }
        TF_TRY(run_pass(ctx, AprItem2Pol<LPR, NCH, VEC, FULL, false, SHORT>{a}, w, sl.keys, sl.vals, B, 2 * B, w.n_tiles_i, w.blocks_i, false, "item_tile",
                        "item_fixup"));
        if (want_loss) {
            // the third number is a mean over B*d entries
            apr_loss_out_kernel<<<1, 1, 0, ctx->stream>>>(w.scalars + 4, d_loss, 1.0 / ((double)B * (double)d));
            TF_HIP(hipGetLastError());
        }
        return TFREC_OK;
    });
}

// ------------------------------------------------------------------------------------------
// APR with a replicated item table: each GPU holds a shard of P (+ accumulator, + delta) and full
// copies of Q, b, their accumulators and deltas.  The user side of both phases is local; the item
// side leaves gradient sums in a zeroed table G = [I*d rows | I biases] that the caller sums over
// the GPUs; the normalisation (phase 1) and the Adagrad update (phase 2) then run densely on every
// copy.  Rows no triple touches have a zero gradient: their delta is zero (model/APR.py:124-132
// rewrites the whole delta tables every step) and the Adagrad update leaves them unchanged.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void apr_delta_rows_kernel(const float *__restrict__ g, float *__restrict__ dq, int64_t rows, int d,
                                                             float eps)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (r >= rows) return;
    const float *gr = g + (size_t)r * d;
    float n2 = 0.f;
    for (int f = l; f < d; f += 64) n2 = fmaf(gr[f], gr[f], n2);
    for (int m = 32; m > 0; m >>= 1) n2 += __shfl_xor(n2, m);
    const float sc = eps * rsqrtf(fmaxf(n2, 1e-12f));
    for (int f = l; f < d; f += 64) dq[(size_t)r * d + f] = sc * gr[f];
}

__global__ __launch_bounds__(256) void sum_squares_kernel(const float *__restrict__ x, int64_t n, double *__restrict__ out)
{
    __shared__ double red[256];
    double v = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) v += (double)x[t] * (double)x[t];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void apr_bias_delta_kernel(const float *__restrict__ gb, float *__restrict__ db, int64_t n,
                                                             const double *__restrict__ norm2, float eps)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) db[t] = gb[t] * (eps * rsqrtf(fmaxf((float)norm2[0], 1e-12f)));
}

// acc += g*g ; w -= lr*g/sqrt(acc) where g != 0 (sparse Adagrad never visits the other entries)
__global__ __launch_bounds__(256) void adagrad_apply_kernel(float *__restrict__ w, float *__restrict__ acc, const float *__restrict__ g,
                                                            float lr, int64_t n)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float gv = g[t];
    if (gv == 0.f) return;
    const float an = fmaf(gv, gv, acc[t]);
    acc[t] = an;
    w[t] = w[t] - lr * gv / sqrtf(an);
}

static int apr_common(tfrec_ctx *ctx, AprParams *a, RowCfg *cfg, StepWs *w, const tfrec_batch_slot &sl, int64_t B, int d,
                      std::initializer_list<const void *> tables)
{
    TF_REQUIRE(pick_cfg(d, all_aligned16(tables), cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    TF_TRY(step_ws_carve(ctx, B, d, *cfg, w, true, B <= SHORT_TILES_MAX_B));
    a->stage = w->stage; a->sbuf = w->sbuf; a->pair = sl.pair;
    a->d = d; a->nchunk = d / cfg->vec;
    return TFREC_OK;
}

extern "C" int tfrec_apr_dense_delta_grads(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, float *d_dP,
                                           float *d_G, int64_t U, int64_t I, int d, const int32_t *d_users, const int32_t *d_pos,
                                           const int32_t *d_neg, int64_t B, float eps)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B));
    TF_REQUIRE(d_dP != nullptr, TFREC_EINVAL, "null delta table");
    TF_REQUIRE(d_G != nullptr && ((uintptr_t)d_G & 15u) == 0, TFREC_EINVAL, "gradient table missing or not 16-byte aligned");
    TF_HIP(hipMemsetAsync(d_G, 0, sizeof(float) * ((size_t)I * d + (size_t)I), ctx->stream));
    tfrec_batch_slot &sl = ctx->slots[2];
    sl.apr_pending = 0;
    if (B == 0) { sl.B = 0; sl.U = U; sl.I = I; sl.apr_pending = 1; return TFREC_OK; }
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    TF_TRY(slot_prepare(ctx, sl, ctx->stream, d_users, d_pos, d_neg, B, U, I));
    tfrec_mark(ctx, "index");
    AprParams a{};
    RowCfg cfg;
    StepWs w;
    TF_TRY(apr_common(ctx, &a, &cfg, &w, sl, B, d, {d_P, d_Q, d_dP}));
    a.P = const_cast<float *>(d_P); a.Q = const_cast<float *>(d_Q); a.b = const_cast<float *>(d_b);
    a.dP = d_dP; a.gQ = d_G; a.gb = d_G + (size_t)I * d;
    a.users = d_users; a.U = (uint32_t)U; a.eps = eps;
    TF_TRY(dispatch_cfg_tiles(cfg, B <= SHORT_TILES_MAX_B, [&]<int LPR, int NCH, int VEC, bool FULL, bool SHORT>() -> int {
        TF_TRY(run_pass(ctx, AprUser1Pol<LPR, NCH, VEC, FULL, SHORT>{a}, w, sl.keys, sl.vals, 0, B, w.n_tiles_u, w.blocks_u, false,
                        "delta_user_tile", "delta_user_fixup"));
        return run_pass(ctx, AprItem1Pol<LPR, NCH, VEC, FULL, true, SHORT>{a}, w, sl.keys, sl.vals, B, 2 * B, w.n_tiles_i, w.blocks_i, false,
                        "delta_item_tile", "delta_item_fixup");
    }));
    sl.apr_pending = 1;
    return TFREC_OK;
}

extern "C" int tfrec_apr_delta_from_grads(tfrec_ctx *ctx, const float *d_G, float *d_dQ, float *d_db, int64_t I, int d, float eps)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(d_G && d_dQ && d_db && I > 0 && I < (1ll << 31) && d > 0, TFREC_EINVAL, "bad gradient or delta table");
    TF_TRY(tfrec_ws_require(ctx, ctx->ws_used + 8192));
    double *part = (double *)tfrec_ws_take(ctx, sizeof(double) * 520);
    TF_REQUIRE(part != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
    apr_delta_rows_kernel<<<(unsigned)((I + 3) / 4), 256, 0, ctx->stream>>>(d_G, d_dQ, I, d, eps);
    TF_HIP(hipGetLastError());
    const int nb = (int)((I + 255) / 256 < 512 ? (I + 255) / 256 : 512);
    sum_squares_kernel<<<nb, 256, 0, ctx->stream>>>(d_G + (size_t)I * d, I, part);
    TF_HIP(hipGetLastError());
    loss_finalize_kernel<1><<<1, 256, 0, ctx->stream>>>(part, nb, part + 512);
    TF_HIP(hipGetLastError());
    apr_bias_delta_kernel<<<(unsigned)((I + 255) / 256), 256, 0, ctx->stream>>>(d_G + (size_t)I * d, d_db, I, part + 512, eps);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

extern "C" int tfrec_apr_dense_adagrad_grads(tfrec_ctx *ctx, float *d_P, const float *d_Q, const float *d_b, float *d_accP,
                                             const float *d_dP, const float *d_dQ, const float *d_db, float *d_G, int64_t U,
                                             int64_t I, int d, int64_t B, int64_t B_total, float lr, float reg, float reg_adv,
                                             double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_P, d_P, d_P, B));
    TF_REQUIRE(d_accP && d_dP && d_dQ && d_db, TFREC_EINVAL, "null accumulator or delta table");
    TF_REQUIRE(d_G != nullptr && ((uintptr_t)d_G & 15u) == 0, TFREC_EINVAL, "gradient table missing or not 16-byte aligned");
    TF_REQUIRE(B_total >= B && B_total >= 1, TFREC_EINVAL, "the batch size over all GPUs is below this GPU's share");
    tfrec_batch_slot &sl = ctx->slots[2];
    TF_REQUIRE(sl.apr_pending == 1 && sl.B == B && sl.U == U && sl.I == I, TFREC_EINVAL,
               "tfrec_apr_dense_delta_grads of the same batch must be the previous step call on this context");
    sl.apr_pending = 0;
    TF_HIP(hipMemsetAsync(d_G, 0, sizeof(float) * ((size_t)I * d + (size_t)I), ctx->stream));
    if (B == 0) {
        sl.B = -1;
        if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 3 * sizeof(double), ctx->stream));
        return TFREC_OK;
    }
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    AprParams a{};
    RowCfg cfg;
    StepWs w;
    TF_TRY(apr_common(ctx, &a, &cfg, &w, sl, B, d, {d_P, d_Q, d_accP, d_dP, d_dQ}));
    a.P = d_P; a.Q = const_cast<float *>(d_Q); a.b = const_cast<float *>(d_b); a.accP = d_accP;
    a.dP = const_cast<float *>(d_dP); a.dQ = const_cast<float *>(d_dQ); a.db = const_cast<float *>(d_db);
    a.gQ = d_G; a.gb = d_G + (size_t)I * d;
    a.U = (uint32_t)U; a.lr = lr; a.reg_adv = reg_adv;
    a.reg_coef = (float)(4.0 * (double)reg / ((double)B_total * (double)d));
    const bool want_loss = d_loss != nullptr;
    return dispatch_cfg_tiles(cfg, B <= SHORT_TILES_MAX_B, [&]<int LPR, int NCH, int VEC, bool FULL, bool SHORT>() -> int {
        if (want_loss) {
            TF_TRY(run_pass(ctx, AprUser2Pol<LPR, NCH, VEC, FULL, true, SHORT>{a}, w, sl.keys, sl.vals, 0, B, w.n_tiles_u, w.blocks_u, true,
                            "user_tile", "user_fixup"));
            loss_finalize_kernel<3><<<1, 256, 0, ctx->stream>>>(w.loss_part, w.blocks_u, w.scalars + 4);
            TF_HIP(hipGetLastError());
        } else {
            TF_TRY(run_pass(ctx, AprUser2Pol<LPR, NCH, VEC, FULL, false, SHORT>{a}, w, sl.keys, sl.vals, 0, B, w.n_tiles_u, w.blocks_u, false,
                            "user_tile", "user_fixup"));
        }
        TF_TRY(run_pass(ctx, AprItem2Pol<LPR, NCH, VEC, FULL, true, SHORT>{a}, w, sl.keys, sl.vals, B, 2 * B, w.n_tiles_i, w.blocks_i, false,
                        "item_tile", "item_fixup"));
        if (want_loss) {
            // this GPU's share of the three sums; the third is divided by the entry count of the whole batch
            apr_loss_out_kernel<<<1, 1, 0, ctx->stream>>>(w.scalars + 4, d_loss, 1.0 / ((double)B_total * (double)d));
            TF_HIP(hipGetLastError());
        }
        return TFREC_OK;
    });
}

extern "C" int tfrec_adagrad_apply(tfrec_ctx *ctx, float *d_w, float *d_acc, const float *d_g, float lr, int64_t n)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_REQUIRE(n >= 0, TFREC_EINVAL, "negative length");
    if (n == 0) return TFREC_OK;
    TF_REQUIRE(d_w && d_acc && d_g, TFREC_EINVAL, "null array");
    adagrad_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_w, d_acc, d_g, lr, n);
    TF_HIP(hipGetLastError());
    return TFREC_OK;
}

// ------------------------------------------------------------------------------------------
// prediction and DNS candidate selection: one LPR-lane group per (user, item) pair / positive
// ------------------------------------------------------------------------------------------
template <int LPR, int NCH, int VEC>
__global__ __launch_bounds__(256) void predict_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                      const float *__restrict__ b, uint32_t U, uint32_t I, int d, int nchunk,
                                                      const int32_t *__restrict__ users, const int32_t *__restrict__ items,
                                                      int64_t n, float *__restrict__ out, uint32_t *__restrict__ flags)
{
    using RowT = Row<LPR, NCH, VEC>;
    const int gl = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (256 / LPR);
    for (int64_t t = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; t < n; t += stride) {
        int32_t u = users[t], i = items[t];
        if (u < 0 || (uint32_t)u >= U || i < 0 || (uint32_t)i >= I) {
            if (gl == 0) atomicOr(flags, TFREC_FLAG_BAD_INDEX);
            u = u < 0 ? 0 : ((uint32_t)u >= U ? (int32_t)(U - 1) : u);
            i = i < 0 ? 0 : ((uint32_t)i >= I ? (int32_t)(I - 1) : i);
        }
        const RowT p = RowT::load(P + (size_t)u * d, gl, nchunk);
        const RowT q = RowT::load(Q + (size_t)i * d, gl, nchunk);
        const float y = group_sum<LPR>(p.dot_partial(q)) + b[i];
        if (gl == 0) out[t] = y;
    }
}

extern "C" int tfrec_mf_predict(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, int64_t U, int64_t I, int d,
                                const int32_t *d_users, const int32_t *d_items, int64_t n, float *d_out)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_items, d_items, n));
    TF_REQUIRE(n == 0 || d_out, TFREC_EINVAL, "null output");
    if (n == 0) return TFREC_OK;
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_P, d_Q}), &cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    return dispatch_cfg<false>(cfg, [&]<int LPR, int NCH, int VEC, bool FULL>() -> int {
        const int64_t groups = (n + (256 / LPR) - 1) / (256 / LPR);
        const int blocks = (int)(groups < 4096 ? groups : 4096);
        predict_kernel<LPR, NCH, VEC><<<blocks, 256, 0, ctx->stream>>>(d_P, d_Q, d_b, (uint32_t)U, (uint32_t)I, d, d / VEC, d_users,
                                                                     d_items, n, d_out, ctx->d_flags);
        TF_HIP(hipGetLastError());
        return TFREC_OK;
    });
}

// model/DNSBPR.py:74-79.  The group keeps P[u] in registers and streams the C candidate rows,
// four gathers in flight; the first maximum wins, as numpy's argmax does.
template <int LPR, int NCH, int VEC>
__global__ __launch_bounds__(256) void dns_select_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                         const float *__restrict__ b, uint32_t U, uint32_t I, int d, int nchunk,
                                                         const int32_t *__restrict__ users, const int32_t *__restrict__ cand,
                                                         int64_t n_pos, int C, int32_t *__restrict__ out,
                                                         uint32_t *__restrict__ flags)
{
    using RowT = Row<LPR, NCH, VEC>;
    constexpr int UNR = 4;
    const int gl = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * (256 / LPR);
    for (int64_t t = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; t < n_pos; t += stride) {
        int32_t u = users[t];
        bool bad = u < 0 || (uint32_t)u >= U;
        if (bad) u = u < 0 ? 0 : (int32_t)(U - 1);
        const RowT p = RowT::load(P + (size_t)u * d, gl, nchunk);
        float best = -INFINITY;
        int32_t best_id = -1;
        const int32_t *cr = cand + t * C;
        for (int c0 = 0; c0 < C; c0 += UNR) {
            RowT q[UNR];
            float bb[UNR];
            int32_t id[UNR];
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int c = c0 + k < C ? c0 + k : C - 1;
                int32_t it = cr[c];
                if (it < 0 || (uint32_t)it >= I) { bad = true; it = it < 0 ? 0 : (int32_t)(I - 1); }
                id[k] = it;
                q[k] = RowT::load(Q + (size_t)it * d, gl, nchunk);
                bb[k] = b[it];
            }
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const float sc = group_sum<LPR>(p.dot_partial(q[k])) + bb[k];
                if (c0 + k < C && (best_id < 0 || sc > best)) { best = sc; best_id = id[k]; }
            }
        }
        if (gl == 0) {
            out[t] = best_id;
            if (bad) atomicOr(flags, TFREC_FLAG_BAD_INDEX);
        }
    }
}

extern "C" int tfrec_dns_select(tfrec_ctx *ctx, const float *d_P, const float *d_Q, const float *d_b, int64_t U, int64_t I, int d,
                                const int32_t *d_users, const int32_t *d_cand, int64_t n_pos, int C, int32_t *d_out_neg)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_cand, d_cand, n_pos));
    TF_REQUIRE(C >= 1, TFREC_EINVAL, "need at least one candidate per positive");
    TF_REQUIRE(n_pos == 0 || d_out_neg, TFREC_EINVAL, "null output");
    if (n_pos == 0) return TFREC_OK;
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_P, d_Q}), &cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    return dispatch_cfg<false>(cfg, [&]<int LPR, int NCH, int VEC, bool FULL>() -> int {
        const int64_t groups = (n_pos + (256 / LPR) - 1) / (256 / LPR);
        const int blocks = (int)(groups < 8192 ? groups : 8192);
        dns_select_kernel<LPR, NCH, VEC><<<blocks, 256, 0, ctx->stream>>>(d_P, d_Q, d_b, (uint32_t)U, (uint32_t)I, d, d / VEC, d_users,
                                                                        d_cand, n_pos, C, d_out_neg, ctx->d_flags);
        TF_HIP(hipGetLastError());
        return TFREC_OK;
    });
}
