// Training steps of the matrix-factorization ranking path: BPR-SGD, APR-Adagrad, DNS candidate
// selection and plain prediction.  The duplicate-index reduction lives in seg_pass.hpp; this
// file holds the arithmetic (policies) and the C entry points.
//
// One step, for a batch of B triples (u, i, j):
//   1. build_keys     3B (key, occurrence) pairs: key u for the user side, U+item for both
//                     item sides; indices are range-checked (clamped + sticky flag)
//   2. radix sort     by key: user occurrences land in [0,B), item occurrences in [B,3B)
//   3. user pass      per triple: gather P[u], Q[i], Q[j] -> s = sigma(-x); sum the user
//                     gradient per distinct u and update P[u] in place; leave s and the
//                     pre-step copy of P[u] in a per-triple staging area
//   4. item pass      per item occurrence: gather the staged s and P[u] -> sum the item
//                     gradient per distinct item and update Q, b in place
// Step 4 reads only the staging area, never P, so every gradient is taken at the pre-step
// parameters although both tables are updated in place.
#include "seg_pass.hpp"

// ------------------------------------------------------------------------------------------
// keys
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_keys_kernel(const int32_t *__restrict__ users, const int32_t *__restrict__ pos,
                                                         const int32_t *__restrict__ neg, int64_t B, uint32_t U, uint32_t I,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                         uint2 *__restrict__ pair, uint32_t *__restrict__ flags)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= B) return;
    int32_t u = users[t], i = pos[t], j = neg[t];
    const bool bad = u < 0 || (uint32_t)u >= U || i < 0 || (uint32_t)i >= I || j < 0 || (uint32_t)j >= I;
    if (bad) {
        atomicOr(flags, TFREC_FLAG_BAD_INDEX);
        u = u < 0 ? 0 : ((uint32_t)u >= U ? (int32_t)(U - 1) : u);
        i = i < 0 ? 0 : ((uint32_t)i >= I ? (int32_t)(I - 1) : i);
        j = j < 0 ? 0 : ((uint32_t)j >= I ? (int32_t)(I - 1) : j);
    }
    keys[t] = (uint32_t)u;
    vals[t] = (uint32_t)t;
    keys[B + 2 * t] = U + (uint32_t)i;
    vals[B + 2 * t] = (uint32_t)(2 * t);
    keys[B + 2 * t + 1] = U + (uint32_t)j;
    vals[B + 2 * t + 1] = (uint32_t)(2 * t + 1);
    pair[t] = make_uint2((uint32_t)i, (uint32_t)j);
}

// ------------------------------------------------------------------------------------------
// BPR-SGD policies.  Reference math: model/base/MatrixFactorization.py:26-32 (score),
// model/losses.py:4-12 (-log_sigmoid(x + 1e-7)), model/BPR.py:42-48 (loss, l2 on gathered rows, SGD)
// ------------------------------------------------------------------------------------------
struct BprParams {
    float *P, *Q, *b;
    float *stage;        // [B, d] pre-step P[u] of every triple
    float *sbuf;         // [B]    s = sigma(-(x + 1e-7)) of every triple
    const uint2 *pair;   // [B]    (i, j) of every triple, range-checked
    uint32_t U;
    int d, nchunk;
    float lr, reg;
};

template <int LPR_, int NCH_, int VEC_, bool LOSS>
struct BprUserPol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_ };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using Loss = LossAcc<LOSS ? 2 : 0>;
    BprParams a;
    struct Occ { RowT p, qi, qj; float bi, bj; };
    struct Seg { RowT p; };

    __device__ __forceinline__ uint2 aux(uint32_t, uint32_t val) const { return a.pair[val]; }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        Occ o;
        o.p = RowT::load(a.P + (size_t)en.x * a.d, gl, a.nchunk);
        o.qi = RowT::load(a.Q + (size_t)en.z * a.d, gl, a.nchunk);
        o.qj = RowT::load(a.Q + (size_t)en.w * a.d, gl, a.nchunk);
        o.bi = a.b[en.z];
        o.bj = a.b[en.w];
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t, int) const { return Seg{RowT::zero()}; }
    __device__ __forceinline__ void compute(Seg &seg, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &loss, int gl) const
    {
        const RowT diff = o.qi - o.qj;
        const float x = group_sum<LPR_>(o.p.dot_partial(diff)) + (o.bi - o.bj);
        const float z = x + 1e-7f;
        const float s = sigmoid_neg(z);
        acc.v.fma(s, diff);          // user gradient is  -sum s*(q_i - q_j) + n*reg*p
        acc.cnt += 1;
        seg.p = o.p;
        o.p.store(a.stage + (size_t)en.y * a.d, gl, a.nchunk);
        if (gl == 0) a.sbuf[en.y] = s;
        if constexpr (LOSS) {
            float r = o.p.dot_partial(o.p) + o.qi.dot_partial(o.qi) + o.qj.dot_partial(o.qj);
            if (gl == 0) {
                loss.t[0] += softplus_neg(z);
                r += o.bi * o.bi + o.bj * o.bj;
            }
            loss.t[1] += 0.5f * r;
        }
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        float *row = a.P + (size_t)key * a.d;
        const RowT p = seg ? seg->p : RowT::load(row, gl, a.nchunk);
        const float nreg = (float)acc.cnt * a.reg, lr = a.lr;
        const RowT out = p.map2(acc.v, [=](float pv, float av) { return fmaf(-lr, fmaf(nreg, pv, -av), pv); });
        out.store(row, gl, a.nchunk);
    }
};

template <int LPR_, int NCH_, int VEC_>
struct BprItemPol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_ };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using Loss = LossAcc<0>;
    BprParams a;
    struct Occ { RowT p; float s; };
    struct Seg { RowT q; float bk; };

    __device__ __forceinline__ uint2 aux(uint32_t, uint32_t) const { return make_uint2(0, 0); }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        const uint32_t t = en.y >> 1;
        Occ o;
        o.p = RowT::load(a.stage + (size_t)t * a.d, gl, a.nchunk);
        o.s = a.sbuf[t];
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        const uint32_t it = key - a.U;
        return Seg{RowT::load(a.Q + (size_t)it * a.d, gl, a.nchunk), a.b[it]};
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
        const uint32_t it = key - a.U;
        float *row = a.Q + (size_t)it * a.d;
        const RowT q = seg ? seg->q : RowT::load(row, gl, a.nchunk);
        const float bk = seg ? seg->bk : a.b[it];
        const float nreg = (float)acc.cnt * a.reg, lr = a.lr;
        const RowT out = q.map2(acc.v, [=](float qv, float av) { return fmaf(-lr, fmaf(nreg, qv, -av), qv); });
        out.store(row, gl, a.nchunk);
        if (gl == 0) a.b[it] = fmaf(-lr, fmaf(nreg, bk, -acc.s0), bk);
    }
};

// ------------------------------------------------------------------------------------------
// APR policies.  model/APR.py:64-136.
//   phase 1 (delta):  g = d L / d Theta with L = sum softplus(-clip(y_i - y_j, -80, 1e8)),
//                     delta_row = eps * g_row / sqrt(max(|g_row|^2, 1e-12)); the bias gradient is
//                     normalised as ONE vector, so its squared norm is a sum over all items.
//   phase 2 (Adagrad on L + reg_adv*L(Theta+delta) + 2*reg*mean-square).
// ------------------------------------------------------------------------------------------
struct AprParams {
    float *P, *Q, *b, *accP, *accQ, *accb, *dP, *dQ, *db;
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
template <int LPR_, int NCH_, int VEC_>
struct AprUser1Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_ };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using Loss = LossAcc<0>;
    AprParams a;
    struct Occ { RowT p, qi, qj; float bi, bj; };
    struct Seg { int unused; };
    __device__ __forceinline__ uint2 aux(uint32_t, uint32_t val) const { return a.pair[val]; }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        Occ o;
        o.p = RowT::load(a.P + (size_t)en.x * a.d, gl, a.nchunk);
        o.qi = RowT::load(a.Q + (size_t)en.z * a.d, gl, a.nchunk);
        o.qj = RowT::load(a.Q + (size_t)en.w * a.d, gl, a.nchunk);
        o.bi = a.b[en.z];
        o.bj = a.b[en.w];
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t, int) const { return Seg{0}; }
    __device__ __forceinline__ void compute(Seg &, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &, int gl) const
    {
        const RowT diff = o.qi - o.qj;
        const float r = group_sum<LPR_>(o.p.dot_partial(diff)) + (o.bi - o.bj);
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
        out.store(a.dP + (size_t)key * a.d, gl, a.nchunk);
    }
};

// phase 1, item side: delta_Q rows, raw bias gradient into db, and sum of its squares
template <int LPR_, int NCH_, int VEC_>
struct AprItem1Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_ };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using Loss = LossAcc<0>;
    AprParams a;
    struct Occ { RowT p; float s; };
    struct Seg { int unused; };
    __device__ __forceinline__ uint2 aux(uint32_t, uint32_t) const { return make_uint2(0, 0); }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        const uint32_t t = en.y >> 1;
        const int32_t ur = a.users[t];
        const uint32_t u = ur < 0 ? 0u : ((uint32_t)ur >= a.U ? a.U - 1 : (uint32_t)ur);
        return Occ{RowT::load(a.P + (size_t)u * a.d, gl, a.nchunk), a.sbuf[t]};
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
        const float n2 = group_sum<LPR_>(acc.v.dot_partial(acc.v));
        const float sc = a.eps * rsqrtf(fmaxf(n2, 1e-12f));
        RowT out = RowT::zero();
        out.fma(sc, acc.v);
        out.store(a.dQ + (size_t)it * a.d, gl, a.nchunk);
        if (gl == 0) a.db[it] = acc.s0;     // scaled once the global norm is known
    }
};

// phase 2, user side
template <int LPR_, int NCH_, int VEC_, bool LOSS>
struct AprUser2Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_ };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using Loss = LossAcc<LOSS ? 3 : 0>;
    AprParams a;
    struct Occ { RowT p, dp, qi, qj, dqi, dqj; float bi, bj, dbi, dbj; };
    struct Seg { RowT p; };
    __device__ __forceinline__ uint2 aux(uint32_t, uint32_t val) const { return a.pair[val]; }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        Occ o;
        o.p = RowT::load(a.P + (size_t)en.x * a.d, gl, a.nchunk);
        o.dp = RowT::load(a.dP + (size_t)en.x * a.d, gl, a.nchunk);
        o.qi = RowT::load(a.Q + (size_t)en.z * a.d, gl, a.nchunk);
        o.qj = RowT::load(a.Q + (size_t)en.w * a.d, gl, a.nchunk);
        o.dqi = RowT::load(a.dQ + (size_t)en.z * a.d, gl, a.nchunk);
        o.dqj = RowT::load(a.dQ + (size_t)en.w * a.d, gl, a.nchunk);
        o.bi = a.b[en.z]; o.bj = a.b[en.w];
        o.dbi = a.db[en.z]; o.dbj = a.db[en.w];
        return o;
    }
    __device__ __forceinline__ Seg load_seg(uint32_t, int) const { return Seg{RowT::zero()}; }
    __device__ __forceinline__ void compute(Seg &seg, const Occ &o, uint4 en, Acc<RowT> &acc, Loss &loss, int gl) const
    {
        const RowT diff = o.qi - o.qj;
        const RowT pa = o.p + o.dp;
        const RowT diffa = (o.qi + o.dqi) - (o.qj + o.dqj);
        const float r = group_sum<LPR_>(o.p.dot_partial(diff)) + (o.bi - o.bj);
        const float ra = group_sum<LPR_>(pa.dot_partial(diffa)) + ((o.bi + o.dbi) - (o.bj + o.dbj));
        const float s = clip_sigmoid_neg(r);
        const float sa = a.reg_adv * clip_sigmoid_neg(ra);
        // G_P += -s*(q_i-q_j) - sa*(q_i+dq_i - q_j-dq_j) + creg*p
        acc.v.fma(-s, diff);
        acc.v.fma(-sa, diffa);
        acc.v.fma(a.reg_coef, o.p);
        acc.cnt += 1;
        seg.p = o.p;
        // what the item side needs from this triple: c = s*p + sa*(p+dp)  and  s+sa
        RowT c = RowT::zero();
        c.fma(s, o.p);
        c.fma(sa, pa);
        c.store(a.stage + (size_t)en.y * a.d, gl, a.nchunk);
        if (gl == 0) a.sbuf[en.y] = s + sa;
        if constexpr (LOSS) {
            float m = o.p.dot_partial(o.p) + o.qi.dot_partial(o.qi) + o.qj.dot_partial(o.qj);
            if (gl == 0) {
                loss.t[0] += clip_softplus_neg(r);
                loss.t[1] += clip_softplus_neg(ra);
                m += (float)a.d * (o.bi * o.bi + o.bj * o.bj);
            }
            loss.t[2] += m;
        }
    }
    __device__ __forceinline__ void apply(uint32_t key, const Seg *seg, const Acc<RowT> &acc, int gl) const
    {
        float *row = a.P + (size_t)key * a.d, *arow = a.accP + (size_t)key * a.d;
        const RowT p = seg ? seg->p : RowT::load(row, gl, a.nchunk);
        const RowT ac = RowT::load(arow, gl, a.nchunk);
        const RowT an = ac.map2(acc.v, [](float av, float g) { return fmaf(g, g, av); });
        an.store(arow, gl, a.nchunk);
        const float lr = a.lr;
        // w -= lr * g / sqrt(acc_new)
        RowT step = acc.v.map2(an, [=](float g, float av) { return lr * g / sqrtf(av); });
        (p - step).store(row, gl, a.nchunk);
    }
};

// phase 2, item side
template <int LPR_, int NCH_, int VEC_>
struct AprItem2Pol {
    enum { LPR = LPR_, NCH = NCH_, VEC = VEC_ };
    using RowT = Row<LPR_, NCH_, VEC_>;
    using Loss = LossAcc<0>;
    AprParams a;
    struct Occ { RowT c; float s; };
    struct Seg { RowT q; float bk; };
    __device__ __forceinline__ uint2 aux(uint32_t, uint32_t) const { return make_uint2(0, 0); }
    __device__ __forceinline__ Occ load_occ(uint4 en, int gl) const
    {
        const uint32_t t = en.y >> 1;
        return Occ{RowT::load(a.stage + (size_t)t * a.d, gl, a.nchunk), a.sbuf[t]};
    }
    __device__ __forceinline__ Seg load_seg(uint32_t key, int gl) const
    {
        const uint32_t it = key - a.U;
        return Seg{RowT::load(a.Q + (size_t)it * a.d, gl, a.nchunk), a.b[it]};
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
        float *row = a.Q + (size_t)it * a.d, *arow = a.accQ + (size_t)it * a.d;
        const RowT q = seg ? seg->q : RowT::load(row, gl, a.nchunk);
        const float bk = seg ? seg->bk : a.b[it];
        const float n = (float)acc.cnt, creg = a.reg_coef, lr = a.lr;
        // every occurrence adds creg*q (rows) and creg*d*b (bias) from the mean-square term
        const RowT g = acc.v.map2(q, [=](float gv, float qv) { return fmaf(n * creg, qv, gv); });
        const RowT ac = RowT::load(arow, gl, a.nchunk);
        const RowT an = ac.map2(g, [](float av, float gv) { return fmaf(gv, gv, av); });
        an.store(arow, gl, a.nchunk);
        RowT step = g.map2(an, [=](float gv, float av) { return lr * gv / sqrtf(av); });
        (q - step).store(row, gl, a.nchunk);
        if (gl == 0) {
            const float gb = fmaf(n * creg * (float)a.d, bk, acc.s0);
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
struct RowCfg { int lpr, nch, vec; };

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
        return true;
    }
    c->vec = 1; c->lpr = 64;
    if (d <= 64) c->nch = 1;
    else if (d <= 128) c->nch = 2;
    else if (d <= 256) c->nch = 4;
    else return false;
    return true;
}

// Calls f.template operator()<LPR,NCH,VEC>() for the chosen configuration.
template <class F>
static int dispatch_cfg(const RowCfg &c, F &&f)
{
    if (c.vec == 4) {
        switch (c.lpr) {
            case 8: return f.template operator()<8, 1, 4>();
            case 16: return f.template operator()<16, 1, 4>();
            case 32: return f.template operator()<32, 1, 4>();
            case 64: return f.template operator()<64, 1, 4>();
        }
    } else {
        switch (c.nch) {
            case 1: return f.template operator()<64, 1, 1>();
            case 2: return f.template operator()<64, 2, 1>();
            case 4: return f.template operator()<64, 4, 1>();
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

struct StepWs {
    uint32_t *keys_in, *vals_in, *keys, *vals;
    uint2 *pair;
    float *stage, *sbuf, *part_rows;
    float4 *part_sc;
    uint32_t *tile_flags;
    double *loss_part, *scalars;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    int64_t n_tiles_u, n_tiles_i;
    int blocks_u, blocks_i;
};

static size_t step_ws_bytes(int64_t B, int d, const RowCfg &c, int end_bit, size_t *sort_bytes)
{
    const int CH = c.lpr > 32 ? c.lpr : 32;
    const int64_t tiles = (2 * B + CH - 1) / CH + 1;      // the larger (item) side
    const size_t rowf = (size_t)c.lpr * c.nch * c.vec;
    *sort_bytes = tfrec_sort_temp_bytes((size_t)3 * B, end_bit);
    size_t s = 0;
    s += 4 * tf_align(sizeof(uint32_t) * 3 * B);           // keys_in, vals_in, keys, vals
    s += tf_align(sizeof(uint2) * B);
    s += tf_align(sizeof(float) * (size_t)B * d);          // stage
    s += tf_align(sizeof(float) * B);                      // sbuf
    s += tf_align(sizeof(float) * 2 * tiles * rowf);       // partial rows
    s += tf_align(sizeof(float4) * 2 * tiles);
    s += tf_align(sizeof(uint32_t) * tiles);
    s += tf_align(sizeof(double) * 4 * (tiles + 8));       // loss partials (<= one per block)
    s += tf_align(sizeof(double) * 16);
    s += tf_align(*sort_bytes);
    return s + 4096;
}

// upper bound over all row configurations, for tfrec_ctx_reserve
size_t tfrec_step_need_bytes(int64_t B, int d)
{
    RowCfg c;
    if (!pick_cfg(d, true, &c)) return 0;
    size_t sb = 0, best = step_ws_bytes(B, d, c, 32, &sb);
    RowCfg c1;
    if (pick_cfg(d | 1, false, &c1)) {      // the scalar fallback used for unaligned tables
        c1.nch = d <= 64 ? 1 : (d <= 128 ? 2 : 4);
        const size_t alt = step_ws_bytes(B, d, c1, 32, &sb);
        best = alt > best ? alt : best;
    }
    return best;
}

static int step_ws_carve(tfrec_ctx *ctx, int64_t B, int d, const RowCfg &c, int end_bit, StepWs *w)
{
    size_t sort_bytes = 0;
    const size_t need = step_ws_bytes(B, d, c, end_bit, &sort_bytes);
    TF_TRY(tfrec_ws_require(ctx, need));
    const int CH = c.lpr > 32 ? c.lpr : 32, GPB = 256 / c.lpr;
    const int64_t tiles = (2 * B + CH - 1) / CH + 1;
    const size_t rowf = (size_t)c.lpr * c.nch * c.vec;
    w->keys_in = (uint32_t *)tfrec_ws_take(ctx, sizeof(uint32_t) * 3 * B);
    w->vals_in = (uint32_t *)tfrec_ws_take(ctx, sizeof(uint32_t) * 3 * B);
    w->keys = (uint32_t *)tfrec_ws_take(ctx, sizeof(uint32_t) * 3 * B);
    w->vals = (uint32_t *)tfrec_ws_take(ctx, sizeof(uint32_t) * 3 * B);
    w->pair = (uint2 *)tfrec_ws_take(ctx, sizeof(uint2) * B);
    w->stage = (float *)tfrec_ws_take(ctx, sizeof(float) * (size_t)B * d);
    w->sbuf = (float *)tfrec_ws_take(ctx, sizeof(float) * B);
    w->part_rows = (float *)tfrec_ws_take(ctx, sizeof(float) * 2 * tiles * rowf);
    w->part_sc = (float4 *)tfrec_ws_take(ctx, sizeof(float4) * 2 * tiles);
    w->tile_flags = (uint32_t *)tfrec_ws_take(ctx, sizeof(uint32_t) * tiles);
    w->loss_part = (double *)tfrec_ws_take(ctx, sizeof(double) * 4 * (tiles + 8));
    w->scalars = (double *)tfrec_ws_take(ctx, sizeof(double) * 16);
    w->sort_tmp = tfrec_ws_take(ctx, sort_bytes);
    w->sort_tmp_bytes = sort_bytes;
    TF_REQUIRE(w->sort_tmp != nullptr, TFREC_ENOMEM, "internal: workspace carve failed");
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
    TF_REQUIRE((uint64_t)U + (uint64_t)I < (1ull << 32), TFREC_EUNSUP, "U+I must be below 2^32");
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
static int run_pass(tfrec_ctx *ctx, const Pol &pol, const StepWs &w, int64_t base, int64_t n, int64_t n_tiles, int blocks,
                    bool with_loss_part, const char *tile_name, const char *fix_name)
{
    if (n <= 0) return TFREC_OK;
    seg_tile_kernel<Pol><<<blocks, 256, 0, ctx->stream>>>(pol, w.keys, w.vals, base, n, w.part_rows, w.part_sc, w.tile_flags,
                                                         with_loss_part ? w.loss_part : nullptr);
    TF_HIP(hipGetLastError());
    tfrec_mark(ctx, tile_name);
    seg_fixup_kernel<Pol><<<blocks, 256, 0, ctx->stream>>>(pol, w.keys, base, n, w.part_rows, w.part_sc, w.tile_flags, n_tiles);
    TF_HIP(hipGetLastError());
    tfrec_mark(ctx, fix_name);
    return TFREC_OK;
}

static int sort_batch(tfrec_ctx *ctx, const StepWs &w, const int32_t *users, const int32_t *pos, const int32_t *neg, int64_t B,
                      int64_t U, int64_t I, int end_bit)
{
    tfrec_marks_reset(ctx);
    tfrec_mark(ctx, "start");
    build_keys_kernel<<<(int)((B + 255) / 256), 256, 0, ctx->stream>>>(users, pos, neg, B, (uint32_t)U, (uint32_t)I, w.keys_in,
                                                                      w.vals_in, w.pair, ctx->d_flags);
    TF_HIP(hipGetLastError());
    TF_TRY(tfrec_sort_pairs(w.sort_tmp, w.sort_tmp_bytes, w.keys_in, w.keys, w.vals_in, w.vals, (size_t)3 * B, end_bit, ctx->stream));
    tfrec_mark(ctx, "index");
    return TFREC_OK;
}

extern "C" int tfrec_bpr_sgd_step(tfrec_ctx *ctx, float *d_P, float *d_Q, float *d_b, int64_t U, int64_t I, int d,
                                  const int32_t *d_users, const int32_t *d_pos, const int32_t *d_neg, int64_t B, float lr,
                                  float reg, double *d_loss)
{
    TF_TRY(tfrec_ctx_enter(ctx));
    TF_TRY(check_step_args(d_P, d_Q, d_b, U, I, d, d_users, d_pos, d_neg, B));
    if (B == 0) {
        if (d_loss) TF_HIP(hipMemsetAsync(d_loss, 0, 2 * sizeof(double), ctx->stream));
        return TFREC_OK;
    }
    RowCfg cfg;
    TF_REQUIRE(pick_cfg(d, all_aligned16({d_P, d_Q}), &cfg), TFREC_EUNSUP, "factor size d=%d is not supported (max 256)", d);
    const int end_bit = bits_for((uint64_t)U + (uint64_t)I);
    StepWs w;
    TF_TRY(step_ws_carve(ctx, B, d, cfg, end_bit, &w));
    TF_TRY(sort_batch(ctx, w, d_users, d_pos, d_neg, B, U, I, end_bit));

    BprParams a{d_P, d_Q, d_b, w.stage, w.sbuf, w.pair, (uint32_t)U, d, d / cfg.vec, lr, reg};
    const bool want_loss = d_loss != nullptr;
    return dispatch_cfg(cfg, [&]<int LPR, int NCH, int VEC>() -> int {
        if (want_loss) {
            TF_TRY(run_pass(ctx, BprUserPol<LPR, NCH, VEC, true>{a}, w, 0, B, w.n_tiles_u, w.blocks_u, true, "user_tile", "user_fixup"));
            loss_finalize_kernel<2><<<1, 256, 0, ctx->stream>>>(w.loss_part, w.blocks_u, d_loss);
            TF_HIP(hipGetLastError());
        } else {
            TF_TRY(run_pass(ctx, BprUserPol<LPR, NCH, VEC, false>{a}, w, 0, B, w.n_tiles_u, w.blocks_u, false, "user_tile", "user_fixup"));
        }
        return run_pass(ctx, BprItemPol<LPR, NCH, VEC>{a}, w, B, 2 * B, w.n_tiles_i, w.blocks_i, false, "item_tile", "item_fixup");
    });
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
    const int end_bit = bits_for((uint64_t)U + (uint64_t)I);
    StepWs w;
    TF_TRY(step_ws_carve(ctx, B, d, cfg, end_bit, &w));
    TF_TRY(sort_batch(ctx, w, d_users, d_pos, d_neg, B, U, I, end_bit));

    AprParams a{};
    a.P = d_P; a.Q = d_Q; a.b = d_b; a.accP = d_accP; a.accQ = d_accQ; a.accb = d_accb;
    a.dP = d_dP; a.dQ = d_dQ; a.db = d_db;
    a.stage = w.stage; a.sbuf = w.sbuf; a.pair = w.pair; a.users = d_users;
    a.U = (uint32_t)U; a.d = d; a.nchunk = d / cfg.vec;
    a.lr = lr; a.reg_coef = (float)(4.0 * (double)reg / ((double)B * (double)d)); a.reg_adv = reg_adv; a.eps = eps;
    const bool want_loss = d_loss != nullptr;
    return dispatch_cfg(cfg, [&]<int LPR, int NCH, int VEC>() -> int {
        // phase 1: deltas
        TF_TRY(run_pass(ctx, AprUser1Pol<LPR, NCH, VEC>{a}, w, 0, B, w.n_tiles_u, w.blocks_u, false, "delta_user_tile", "delta_user_fixup"));
        TF_TRY(run_pass(ctx, AprItem1Pol<LPR, NCH, VEC>{a}, w, B, 2 * B, w.n_tiles_i, w.blocks_i, false, "delta_item_tile", "delta_item_fixup"));
        const int nb = (int)((2 * B + 255) / 256 < 512 ? (2 * B + 255) / 256 : 512);
        apr_bias_norm_kernel<<<nb, 256, 0, ctx->stream>>>(w.keys, B, 2 * B, (uint32_t)U, d_db, w.loss_part);
        TF_HIP(hipGetLastError());
        loss_finalize_kernel<1><<<1, 256, 0, ctx->stream>>>(w.loss_part, nb, w.scalars);
        TF_HIP(hipGetLastError());
        apr_bias_scale_kernel<<<(int)((2 * B + 255) / 256), 256, 0, ctx->stream>>>(w.keys, B, 2 * B, (uint32_t)U, d_db, w.scalars, eps);
        TF_HIP(hipGetLastError());
        tfrec_mark(ctx, "delta_bias_norm");
        // phase 2: Adagrad step at (Theta, delta)
        if (want_loss) {
            TF_TRY(run_pass(ctx, AprUser2Pol<LPR, NCH, VEC, true>{a}, w, 0, B, w.n_tiles_u, w.blocks_u, true, "user_tile", "user_fixup"));
            loss_finalize_kernel<3><<<1, 256, 0, ctx->stream>>>(w.loss_part, w.blocks_u, w.scalars + 4);
            TF_HIP(hipGetLastError());
        } else {
            TF_TRY(run_pass(ctx, AprUser2Pol<LPR, NCH, VEC, false>{a}, w, 0, B, w.n_tiles_u, w.blocks_u, false, "user_tile", "user_fixup"));
        }
        TF_TRY(run_pass(ctx, AprItem2Pol<LPR, NCH, VEC>{a}, w, B, 2 * B, w.n_tiles_i, w.blocks_i, false, "item_tile", "item_fixup"));
        if (want_loss) {
            // the third number is a mean over B*d entries
            apr_loss_out_kernel<<<1, 1, 0, ctx->stream>>>(w.scalars + 4, d_loss, 1.0 / ((double)B * (double)d));
            TF_HIP(hipGetLastError());
        }
        return TFREC_OK;
    });
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
    return dispatch_cfg(cfg, [&]<int LPR, int NCH, int VEC>() -> int {
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
    return dispatch_cfg(cfg, [&]<int LPR, int NCH, int VEC>() -> int {
        const int64_t groups = (n_pos + (256 / LPR) - 1) / (256 / LPR);
        const int blocks = (int)(groups < 8192 ? groups : 8192);
        dns_select_kernel<LPR, NCH, VEC><<<blocks, 256, 0, ctx->stream>>>(d_P, d_Q, d_b, (uint32_t)U, (uint32_t)I, d, d / VEC, d_users,
                                                                        d_cand, n_pos, C, d_out_neg, ctx->d_flags);
        TF_HIP(hipGetLastError());
        return TFREC_OK;
    });
}
