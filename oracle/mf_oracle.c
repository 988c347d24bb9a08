/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the reference's
 * matrix-factorization pairwise-ranking path.  Only tests/, the smoke check and
 * the cpu_baseline leg of the benchmark may build, load or call this file; the
 * product library never does.
 *
 * Parity status of each part:
 *   - evaluator part (top-k order aside): pinned against the reference's own C++
 *     (oracle/ref_harness.cpp builds evaluator/backend/cpp/{tools,metrics,evaluate}.h
 *     in place) and against the ML-100k known answer in tests/golden/.
 *   - training part (BPR / APR / DNS step math): the reference runs it through
 *     TensorFlow 1.x, which is not installed here and has no fixture, log or test
 *     in the reference recording a loss or a parameter: PARITY UNPINNED against
 *     TensorFlow itself.  The formulas below follow the reference's graph
 *     definitions line by line and are cross-checked by an independent
 *     torch-autograd implementation (tests/golden/make_golden.py).
 *
 * All training arithmetic is carried out in double precision from the float32
 * tables and rounded to float32 once per updated element, so the oracle is the
 * more accurate side of every comparison.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* workspace: dense double accumulators, cleared only where touched           */
/* ------------------------------------------------------------------------- */
typedef struct {
    int64_t U, I;
    int d;
    double *gP, *gQ, *gb;        /* gradient sums                               */
    double *gP2, *gQ2, *gb2;     /* second set (APR: adversarial-direction sums) */
    int32_t *cntU, *cntI;        /* occurrences of each row in the batch         */
    int32_t *touchedU, *touchedI;
    int64_t nTU, nTI;
} oracle_ws;

oracle_ws *oracle_ws_create(int64_t U, int64_t I, int d)
{
    oracle_ws *w = (oracle_ws *)calloc(1, sizeof(oracle_ws));
    if (!w) return NULL;
    w->U = U; w->I = I; w->d = d;
    w->gP = (double *)calloc((size_t)U * d, sizeof(double));
    w->gQ = (double *)calloc((size_t)I * d, sizeof(double));
    w->gb = (double *)calloc((size_t)I, sizeof(double));
    w->gP2 = (double *)calloc((size_t)U * d, sizeof(double));
    w->gQ2 = (double *)calloc((size_t)I * d, sizeof(double));
    w->gb2 = (double *)calloc((size_t)I, sizeof(double));
    w->cntU = (int32_t *)calloc((size_t)U, sizeof(int32_t));
    w->cntI = (int32_t *)calloc((size_t)I, sizeof(int32_t));
    w->touchedU = (int32_t *)malloc(sizeof(int32_t) * (size_t)U);
    w->touchedI = (int32_t *)malloc(sizeof(int32_t) * (size_t)I);
    return w;
}

void oracle_ws_destroy(oracle_ws *w)
{
    if (!w) return;
    free(w->gP); free(w->gQ); free(w->gb); free(w->gP2); free(w->gQ2); free(w->gb2);
    free(w->cntU); free(w->cntI); free(w->touchedU); free(w->touchedI); free(w);
}

static void touch_user(oracle_ws *w, int32_t u) { if (w->cntU[u]++ == 0) w->touchedU[w->nTU++] = u; }
static void touch_item(oracle_ws *w, int32_t i) { if (w->cntI[i]++ == 0) w->touchedI[w->nTI++] = i; }

static void ws_clear(oracle_ws *w)
{
    const int d = w->d;
    for (int64_t t = 0; t < w->nTU; ++t) {
        int32_t u = w->touchedU[t];
        memset(w->gP + (size_t)u * d, 0, sizeof(double) * d);
        memset(w->gP2 + (size_t)u * d, 0, sizeof(double) * d);
        w->cntU[u] = 0;
    }
    for (int64_t t = 0; t < w->nTI; ++t) {
        int32_t i = w->touchedI[t];
        memset(w->gQ + (size_t)i * d, 0, sizeof(double) * d);
        memset(w->gQ2 + (size_t)i * d, 0, sizeof(double) * d);
        w->gb[i] = 0.0; w->gb2[i] = 0.0;
        w->cntI[i] = 0;
    }
    w->nTU = 0; w->nTI = 0;
}

/* softplus(x) = log(1 + exp(x)), stable */
static double softplus(double x) { return (x > 0 ? x : 0) + log1p(exp(-fabs(x))); }
static double sigmoid(double x) { return x >= 0 ? 1.0 / (1.0 + exp(-x)) : exp(x) / (1.0 + exp(x)); }

static double dot_rows(const float *a, const float *b, int d)
{
    double s = 0;
    for (int f = 0; f < d; ++f) s += (double)a[f] * (double)b[f];
    return s;
}

/* ------------------------------------------------------------------------- */
/* BPR-MF minibatch step with plain SGD                                       */
/*   y(u,i) = sum_f P[u,f] Q[i,f] + b[i]      model/base/MatrixFactorization.py:26-32 */
/*   rank loss = sum_t -log_sigmoid(y_i - y_j + 1e-7)   model/losses.py:4-12, model/BPR.py:42 */
/*   reg loss  = l2_loss over the GATHERED rows/biases  MatrixFactorization.py:48-57, BPR.py:43-45 */
/*   update    = GradientDescent on rank + reg*regloss  BPR.py:46-48 (DNSBPR.py:35-47 identical) */
/* Every gradient is evaluated at the pre-step parameters; a row occurring n   */
/* times receives the sum of its n per-occurrence gradients.                   */
/* loss_out[0] = rank loss, loss_out[1] = un-scaled reg loss.                  */
/* ------------------------------------------------------------------------- */
int oracle_bpr_step(oracle_ws *w, float *P, float *Q, float *b,
                    const int32_t *users, const int32_t *pos, const int32_t *neg, int64_t B,
                    double lr, double reg, double *loss_out)
{
    const int d = w->d;
    double rank_loss = 0, reg_loss = 0;
    for (int64_t t = 0; t < B; ++t) {
        const int32_t u = users[t], i = pos[t], j = neg[t];
        if (u < 0 || u >= w->U || i < 0 || i >= w->I || j < 0 || j >= w->I) return -1;
        const float *p = P + (size_t)u * d, *qi = Q + (size_t)i * d, *qj = Q + (size_t)j * d;
        const double yi = dot_rows(p, qi, d) + (double)b[i];
        const double yj = dot_rows(p, qj, d) + (double)b[j];
        const double z = (yi - yj) + 1e-7;
        rank_loss += softplus(-z);
        const double s = sigmoid(-z);           /* -d(rank)/dz */
        touch_user(w, u); touch_item(w, i); touch_item(w, j);
        double *gp = w->gP + (size_t)u * d, *gqi = w->gQ + (size_t)i * d, *gqj = w->gQ + (size_t)j * d;
        for (int f = 0; f < d; ++f) {
            const double pf = p[f], qif = qi[f], qjf = qj[f];
            gp[f] += -s * (qif - qjf) + reg * pf;
            gqi[f] += -s * pf + reg * qif;
            gqj[f] += s * pf + reg * qjf;
            reg_loss += 0.5 * (pf * pf + qif * qif + qjf * qjf);
        }
        w->gb[i] += -s + reg * (double)b[i];
        w->gb[j] += s + reg * (double)b[j];
        reg_loss += 0.5 * ((double)b[i] * b[i] + (double)b[j] * b[j]);
    }
    for (int64_t t = 0; t < w->nTU; ++t) {
        const int32_t u = w->touchedU[t];
        float *p = P + (size_t)u * d; const double *g = w->gP + (size_t)u * d;
        for (int f = 0; f < d; ++f) p[f] = (float)((double)p[f] - lr * g[f]);
    }
    for (int64_t t = 0; t < w->nTI; ++t) {
        const int32_t i = w->touchedI[t];
        float *q = Q + (size_t)i * d; const double *g = w->gQ + (size_t)i * d;
        for (int f = 0; f < d; ++f) q[f] = (float)((double)q[f] - lr * g[f]);
        b[i] = (float)((double)b[i] - lr * w->gb[i]);
    }
    ws_clear(w);
    if (loss_out) { loss_out[0] = rank_loss; loss_out[1] = reg_loss; }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* The same step on several host threads (OpenMP), for the CPU baseline of    */
/* the benchmark.  Thread T owns the rows whose id is T modulo the thread     */
/* count and adds their gradient terms in batch order, so every parameter     */
/* comes out bit for bit as oracle_bpr_step leaves it; only the two loss sums */
/* are added in a different order.  s_buf: B doubles of scratch.              */
/* ------------------------------------------------------------------------- */
int oracle_bpr_step_mt(oracle_ws *w, float *P, float *Q, float *b,
                       const int32_t *users, const int32_t *pos, const int32_t *neg, int64_t B,
                       double lr, double reg, double *loss_out, double *s_buf, int nthreads)
{
    const int d = w->d;
    if (nthreads < 1) nthreads = 1;
    for (int64_t t = 0; t < B; ++t) {
        const int32_t u = users[t], i = pos[t], j = neg[t];
        if (u < 0 || u >= w->U || i < 0 || i >= w->I || j < 0 || j >= w->I) return -1;
    }
    double rank_loss = 0, reg_loss = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : rank_loss, reg_loss)
    for (int64_t t = 0; t < B; ++t) {
        const int32_t u = users[t], i = pos[t], j = neg[t];
        const float *p = P + (size_t)u * d, *qi = Q + (size_t)i * d, *qj = Q + (size_t)j * d;
        const double z = ((dot_rows(p, qi, d) + (double)b[i]) - (dot_rows(p, qj, d) + (double)b[j])) + 1e-7;
        rank_loss += softplus(-z);
        s_buf[t] = sigmoid(-z);
        double r = 0;
        for (int f = 0; f < d; ++f) r += 0.5 * ((double)p[f] * p[f] + (double)qi[f] * qi[f] + (double)qj[f] * qj[f]);
        reg_loss += r + 0.5 * ((double)b[i] * b[i] + (double)b[j] * b[j]);
    }
#pragma omp parallel num_threads(nthreads)
    {
        int T = 0, NT = 1;
#ifdef _OPENMP
        T = omp_get_thread_num(); NT = omp_get_num_threads();
#endif
        /* gradient sums of the rows this thread owns, all at the pre-step parameters */
        for (int64_t t = 0; t < B; ++t) {
            const int32_t u = users[t], i = pos[t], j = neg[t];
            const double s = s_buf[t];
            const float *p = P + (size_t)u * d, *qi = Q + (size_t)i * d, *qj = Q + (size_t)j * d;
            if (u % NT == T) {
                double *gp = w->gP + (size_t)u * d;
                for (int f = 0; f < d; ++f) gp[f] += -s * ((double)qi[f] - (double)qj[f]) + reg * (double)p[f];
                w->cntU[u]++;
            }
            if (i % NT == T) {
                double *g = w->gQ + (size_t)i * d;
                for (int f = 0; f < d; ++f) g[f] += -s * (double)p[f] + reg * (double)qi[f];
                w->gb[i] += -s + reg * (double)b[i];
                w->cntI[i]++;
            }
            if (j % NT == T) {
                double *g = w->gQ + (size_t)j * d;
                for (int f = 0; f < d; ++f) g[f] += s * (double)p[f] + reg * (double)qj[f];
                w->gb[j] += s + reg * (double)b[j];
                w->cntI[j]++;
            }
        }
#pragma omp barrier
        /* apply once per distinct row (the first occurrence met applies and clears) */
        for (int64_t t = 0; t < B; ++t) {
            const int32_t u = users[t];
            if (u % NT == T && w->cntU[u]) {
                float *p = P + (size_t)u * d; double *g = w->gP + (size_t)u * d;
                for (int f = 0; f < d; ++f) { p[f] = (float)((double)p[f] - lr * g[f]); g[f] = 0.0; }
                w->cntU[u] = 0;
            }
            for (int side = 0; side < 2; ++side) {
                const int32_t i = side ? neg[t] : pos[t];
                if (i % NT == T && w->cntI[i]) {
                    float *q = Q + (size_t)i * d; double *g = w->gQ + (size_t)i * d;
                    for (int f = 0; f < d; ++f) { q[f] = (float)((double)q[f] - lr * g[f]); g[f] = 0.0; }
                    b[i] = (float)((double)b[i] - lr * w->gb[i]);
                    w->gb[i] = 0.0;
                    w->cntI[i] = 0;
                }
            }
        }
    }
    if (loss_out) { loss_out[0] = rank_loss; loss_out[1] = reg_loss; }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* APR (adversarial personalized ranking) minibatch step                      */
/*   model/APR.py:64-84   inference / perturbed inference                     */
/*   model/APR.py:86-114  losses: r = clip(y_i - y_j, -80, 1e8), L = sum softplus(-r) */
/*   model/APR.py:116-132 delta = eps * l2_normalize(dL/dTheta); rows: per row, bias: whole vector */
/*   model/APR.py:134-136 Adagrad on L + reg_adv*L_adv + 2*reg*mean(p^2+q_i^2+q_j^2+b_i^2+b_j^2) */
/*   model/APR.py:156-164 two session calls per batch: delta update, then the optimizer */
/* Adagrad is TensorFlow-1.x sparse Adagrad: duplicate rows are summed first, */
/* then acc += g*g ; w -= lr * g / sqrt(acc) once per distinct row.            */
/* The reference keeps full-size delta tables; every row it reads in the       */
/* second call was written by the first, so only the batch rows matter.        */
/* dP, dQ, db (full-size delta tables) are updated for the batch rows when     */
/* non-NULL.  loss_out[0]=L, [1]=L_adv, [2]=the mean-square term (unscaled).   */
/* ------------------------------------------------------------------------- */
static double clip_r(double r) { return r < -80.0 ? -80.0 : (r > 1e8 ? 1e8 : r); }
static int clip_pass(double r) { return r >= -80.0 && r <= 1e8; }

int oracle_apr_step(oracle_ws *w, float *P, float *Q, float *b,
                    float *accP, float *accQ, float *accb,
                    float *dP, float *dQ, float *db,
                    const int32_t *users, const int32_t *pos, const int32_t *neg, int64_t B,
                    double lr, double reg, double reg_adv, double eps, double *loss_out)
{
    const int d = w->d;
    /* pass 1: gradient of the plain loss, summed per row (APR.py:122) */
    for (int64_t t = 0; t < B; ++t) {
        const int32_t u = users[t], i = pos[t], j = neg[t];
        if (u < 0 || u >= w->U || i < 0 || i >= w->I || j < 0 || j >= w->I) return -1;
        const float *p = P + (size_t)u * d, *qi = Q + (size_t)i * d, *qj = Q + (size_t)j * d;
        const double r = (dot_rows(p, qi, d) + (double)b[i]) - (dot_rows(p, qj, d) + (double)b[j]);
        const double s = clip_pass(r) ? sigmoid(-clip_r(r)) : 0.0;
        touch_user(w, u); touch_item(w, i); touch_item(w, j);
        double *gp = w->gP2 + (size_t)u * d, *gqi = w->gQ2 + (size_t)i * d, *gqj = w->gQ2 + (size_t)j * d;
        for (int f = 0; f < d; ++f) {
            gp[f] += -s * ((double)qi[f] - (double)qj[f]);
            gqi[f] += -s * (double)p[f];
            gqj[f] += s * (double)p[f];
        }
        w->gb2[i] += -s; w->gb2[j] += s;
    }
    /* deltas: rows normalised one by one, the bias gradient as one vector (APR.py:130-132);
       l2_normalize(x) = x * rsqrt(max(sum(x^2), 1e-12)).  gX2 now holds the delta. */
    for (int64_t t = 0; t < w->nTU; ++t) {
        double *g = w->gP2 + (size_t)w->touchedU[t] * d, n2 = 0;
        for (int f = 0; f < d; ++f) n2 += g[f] * g[f];
        const double sc = eps / sqrt(n2 > 1e-12 ? n2 : 1e-12);
        for (int f = 0; f < d; ++f) g[f] *= sc;
    }
    double nb2 = 0;
    for (int64_t t = 0; t < w->nTI; ++t) {
        const int32_t i = w->touchedI[t];
        double *g = w->gQ2 + (size_t)i * d, n2 = 0;
        for (int f = 0; f < d; ++f) n2 += g[f] * g[f];
        const double sc = eps / sqrt(n2 > 1e-12 ? n2 : 1e-12);
        for (int f = 0; f < d; ++f) g[f] *= sc;
        nb2 += w->gb2[i] * w->gb2[i];
    }
    {
        const double sc = eps / sqrt(nb2 > 1e-12 ? nb2 : 1e-12);
        for (int64_t t = 0; t < w->nTI; ++t) w->gb2[w->touchedI[t]] *= sc;
    }
    /* the reference stores the deltas in float32 variables (APR.py:55-60) */
    for (int64_t t = 0; t < w->nTU; ++t) {
        const int32_t u = w->touchedU[t];
        double *g = w->gP2 + (size_t)u * d;
        for (int f = 0; f < d; ++f) { g[f] = (double)(float)g[f]; if (dP) dP[(size_t)u * d + f] = (float)g[f]; }
    }
    for (int64_t t = 0; t < w->nTI; ++t) {
        const int32_t i = w->touchedI[t];
        double *g = w->gQ2 + (size_t)i * d;
        for (int f = 0; f < d; ++f) { g[f] = (double)(float)g[f]; if (dQ) dQ[(size_t)i * d + f] = (float)g[f]; }
        w->gb2[i] = (double)(float)w->gb2[i]; if (db) db[i] = (float)w->gb2[i];
    }
    /* pass 2: gradient of the optimised loss at (Theta, delta) */
    double L = 0, Ladv = 0, msq = 0;
    const double creg = 4.0 * reg / ((double)B * d);   /* d/dx of 2*reg*mean(x^2) over B*d entries */
    for (int64_t t = 0; t < B; ++t) {
        const int32_t u = users[t], i = pos[t], j = neg[t];
        const float *p = P + (size_t)u * d, *qi = Q + (size_t)i * d, *qj = Q + (size_t)j * d;
        const double *ep = w->gP2 + (size_t)u * d, *eqi = w->gQ2 + (size_t)i * d, *eqj = w->gQ2 + (size_t)j * d;
        const double bi = b[i], bj = b[j];
        const double r = (dot_rows(p, qi, d) + bi) - (dot_rows(p, qj, d) + bj);
        double ai = 0, aj = 0;
        for (int f = 0; f < d; ++f) {
            const double pp = (double)p[f] + ep[f];
            ai += pp * ((double)qi[f] + eqi[f]);
            aj += pp * ((double)qj[f] + eqj[f]);
        }
        const double ra = (ai + bi + w->gb2[i]) - (aj + bj + w->gb2[j]);
        L += softplus(-clip_r(r));
        Ladv += softplus(-clip_r(ra));
        const double s = clip_pass(r) ? sigmoid(-clip_r(r)) : 0.0;
        const double sa = reg_adv * (clip_pass(ra) ? sigmoid(-clip_r(ra)) : 0.0);
        double *gp = w->gP + (size_t)u * d, *gqi = w->gQ + (size_t)i * d, *gqj = w->gQ + (size_t)j * d;
        for (int f = 0; f < d; ++f) {
            const double pf = p[f], qif = qi[f], qjf = qj[f];
            const double pa = pf + ep[f], qia = qif + eqi[f], qja = qjf + eqj[f];
            gp[f] += -s * (qif - qjf) - sa * (qia - qja) + creg * pf;
            gqi[f] += -s * pf - sa * pa + creg * qif;
            gqj[f] += s * pf + sa * pa + creg * qjf;
            msq += pf * pf + qif * qif + qjf * qjf + bi * bi + bj * bj;
        }
        w->gb[i] += -s - sa + creg * d * bi;
        w->gb[j] += s + sa + creg * d * bj;
    }
    msq /= ((double)B * d);
    /* sparse Adagrad, one application per distinct row */
    for (int64_t t = 0; t < w->nTU; ++t) {
        const int32_t u = w->touchedU[t];
        float *p = P + (size_t)u * d, *a = accP + (size_t)u * d; const double *g = w->gP + (size_t)u * d;
        for (int f = 0; f < d; ++f) {
            const double an = (double)a[f] + g[f] * g[f];
            a[f] = (float)an;
            p[f] = (float)((double)p[f] - lr * g[f] / sqrt((double)a[f]));
        }
    }
    for (int64_t t = 0; t < w->nTI; ++t) {
        const int32_t i = w->touchedI[t];
        float *q = Q + (size_t)i * d, *a = accQ + (size_t)i * d; const double *g = w->gQ + (size_t)i * d;
        for (int f = 0; f < d; ++f) {
            const double an = (double)a[f] + g[f] * g[f];
            a[f] = (float)an;
            q[f] = (float)((double)q[f] - lr * g[f] / sqrt((double)a[f]));
        }
        const double gbi = w->gb[i];
        const double an = (double)accb[i] + gbi * gbi;
        accb[i] = (float)an;
        b[i] = (float)((double)b[i] - lr * gbi / sqrt((double)accb[i]));
    }
    ws_clear(w);
    if (loss_out) { loss_out[0] = L; loss_out[1] = Ladv; loss_out[2] = msq; }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* DNS-BPR: pick, for each positive, the best-scored of its C candidates       */
/*   model/DNSBPR.py:74-79: neg_logits = logits[neg_pool]; reshape [n_pos, C];  */
/*   argmax(axis=1) (numpy: the FIRST maximum wins)                            */
/* score_mode 0: double accumulation; 1: float32 fmaf chain from the bias      */
/*   (acc = b; for f ascending: acc = fmaf(p[f], q[f], acc))                   */
/* ------------------------------------------------------------------------- */
static float score_f32chain(const float *p, const float *q, float bias, int d)
{
    float acc = bias;
    for (int f = 0; f < d; ++f) acc = fmaf(p[f], q[f], acc);
    return acc;
}

int oracle_dns_select(const float *P, const float *Q, const float *b, int d,
                      const int32_t *users, const int32_t *cand, int64_t n_pos, int C,
                      int score_mode, int32_t *out_neg, double *out_margin)
{
    for (int64_t t = 0; t < n_pos; ++t) {
        const float *p = P + (size_t)users[t] * d;
        double best = -INFINITY, second = -INFINITY; int bc = 0;
        for (int c = 0; c < C; ++c) {
            const int32_t it = cand[t * C + c];
            const double sc = score_mode ? (double)score_f32chain(p, Q + (size_t)it * d, b[it], d)
                                         : dot_rows(p, Q + (size_t)it * d, d) + (double)b[it];
            if (sc > best) { second = best; best = sc; bc = c; }
            else if (sc > second && it != cand[t * C + bc]) second = sc;
        }
        out_neg[t] = cand[t * C + bc];
        if (out_margin) out_margin[t] = best - second;   /* gap to the best DIFFERENT item */
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Evaluator                                                                   */
/* ------------------------------------------------------------------------- */
/* scores of one user against all items: model/BPR.py:76-80 (P Q^T + b).       */
/* mode 0: double -> float32;  mode 1: float32 fmaf chain starting at the bias; */
/* mode 2: operands rounded to bfloat16 (round-to-nearest-even), double sum.   */
static float bf16_round(float x)
{
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;              /* NaN stays */
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    float y; memcpy(&y, &u, 4); return y;
}

void oracle_score_row(const float *p, const float *Q, const float *b, int64_t I, int d, int mode, float *out)
{
    for (int64_t i = 0; i < I; ++i) {
        const float *q = Q + (size_t)i * d;
        if (mode == 1) out[i] = score_f32chain(p, q, b[i], d);
        else if (mode == 2) {
            double s = 0; for (int f = 0; f < d; ++f) s += (double)bf16_round(p[f]) * (double)bf16_round(q[f]);
            out[i] = (float)(s + (double)b[i]);
        } else out[i] = (float)(dot_rows(p, q, d) + (double)b[i]);
    }
}

/* indices of the k largest scores under the total order (score desc, id asc).  */
/* The reference (evaluator/backend/cpp/tools.h:10-19) leaves the order of equal */
/* scores to std::partial_sort_copy; fixtures are tie-free where order matters. */
void oracle_topk_row(const float *score, int64_t I, int k, int32_t *out)
{
    /* insertion into a sorted buffer of k: O(I*k) worst case, fine for tests */
    int n = 0;
    for (int64_t i = 0; i < I; ++i) {
        const float s = score[i];
        if (n == k) {
            const float ls = score[out[k - 1]];
            if (!(s > ls)) continue;      /* equal score, larger id: loses */
        }
        int pos = n < k ? n : k - 1;
        while (pos > 0) {
            const float ps = score[out[pos - 1]];
            if (ps > s || (ps == s)) break;   /* earlier id already ahead on ties */
            out[pos] = out[pos - 1]; --pos;
        }
        out[pos] = (int32_t)i;
        if (n < k) ++n;
    }
    for (int t = n; t < k; ++t) out[t] = -1;
}

/* per-user curves, exactly as evaluator/backend/cpp/metrics.h:11-100 computes  */
/* them (double inner arithmetic, float running sums and stores).  out [5,k].   */
static int in_truth(const int32_t *truth, int n, int32_t x)
{
    for (int t = 0; t < n; ++t) if (truth[t] == x) return 1;
    return 0;
}

void oracle_metrics_user(const int32_t *rank, int k, const int32_t *truth, int n_truth, float *out)
{
    float *pre = out, *rec = out + k, *apv = out + 2 * k, *nd = out + 3 * k, *mr = out + 4 * k;
    int hits = 0; float sum_pre = 0, dcg = 0, idcg = 0; int first = -1;
    for (int i = 0; i < k; ++i) {
        const int h = in_truth(truth, n_truth, rank[i]);
        hits += h;
        pre[i] = (float)(1.0 * hits / (i + 1));
        rec[i] = (float)(1.0 * hits / n_truth);
        if (h) sum_pre += pre[i];
        apv[i] = (float)(1.0 * sum_pre / (n_truth < i + 1 ? n_truth : i + 1));
        if (h) dcg = (float)(dcg + 1.0 / log2((double)(i + 2)));
        idcg = (float)(idcg + 1.0 / log2((double)(i + 2)));
        nd[i] = dcg / idcg;
        if (h && first < 0) first = i;
        mr[i] = first >= 0 ? (float)(1.0 / (first + 1)) : 0.0f;
    }
}

/* mean over users: sequential float accumulation in user order, then divide,   */
/* as evaluator/backend/cpp/evaluate.h:40-92.  per_user is [n,5,k].             */
void oracle_metrics_mean(const float *per_user, int64_t n, int k, float *out)
{
    for (int c = 0; c < 5 * k; ++c) out[c] = 0.0f;
    for (int64_t u = 0; u < n; ++u)
        for (int c = 0; c < 5 * k; ++c) out[c] += per_user[(size_t)u * 5 * k + c];
    for (int c = 0; c < 5 * k; ++c) out[c] /= (float)n;
}

/* Full-rank evaluation in factor form: for each test user score all items,     */
/* mask its train items with -inf (evaluator/backend/cpp/evaluate.py:25-27),    */
/* take the top-k, compute the curves, and average.                            */
/* train/test are CSR over ALL users (indptr [U+1]); test_users lists the users */
/* to evaluate in ascending order (evaluate.py:30).  out_rank [n,k] optional,   */
/* out_per_user [n,5,k] optional, out_mean [5,k].                              */
int oracle_eval_fullrank(const float *P, const float *Q, const float *b, int64_t I, int d,
                         const int32_t *test_users, int64_t n,
                         const int64_t *train_ptr, const int32_t *train_idx,
                         const int64_t *test_ptr, const int32_t *test_idx,
                         int k, int score_mode, int32_t *out_rank, float *out_per_user, float *out_mean)
{
    float *score = (float *)malloc(sizeof(float) * (size_t)I);
    int32_t *rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    float *cur = (float *)malloc(sizeof(float) * 5 * (size_t)k);
    float *sum = (float *)calloc(5 * (size_t)k, sizeof(float));
    if (!score || !rank || !cur || !sum) return -1;
    for (int64_t r = 0; r < n; ++r) {
        const int32_t u = test_users[r];
        oracle_score_row(P + (size_t)u * d, Q, b, I, d, score_mode, score);
        if (train_ptr) for (int64_t t = train_ptr[u]; t < train_ptr[u + 1]; ++t) score[train_idx[t]] = -INFINITY;
        oracle_topk_row(score, I, k, rank);
        oracle_metrics_user(rank, k, test_idx + test_ptr[u], (int)(test_ptr[u + 1] - test_ptr[u]), cur);
        if (out_rank) memcpy(out_rank + (size_t)r * k, rank, sizeof(int32_t) * k);
        if (out_per_user) memcpy(out_per_user + (size_t)r * 5 * k, cur, sizeof(float) * 5 * k);
        for (int c = 0; c < 5 * k; ++c) sum[c] += cur[c];
    }
    for (int c = 0; c < 5 * k; ++c) out_mean[c] = sum[c] / (float)n;
    free(score); free(rank); free(cur); free(sum);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Device epoch pipeline (no numpy stream can be drawn in parallel, so the GPU sampler defines its own
 * counter-based draws; this is the same arithmetic on the CPU, the checker of
 * tfrec_sample_negatives / tfrec_shuffle_perm).  Distribution and exclusion rule follow
 * model/BPR.py:50-62 and utils/tools.py:39-49: per stored interaction, n_neg uniform draws with
 * replacement from the items NOT in the user's row.
 * ------------------------------------------------------------------------------------------ */
static uint64_t o_mix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
static uint64_t o_draw64(uint64_t seed, uint64_t stream, uint64_t index, uint32_t attempt)
{
    return o_mix64(o_mix64(seed ^ (stream * 0xd1342543de82ef95ull)) + index * 0x2545f4914f6cdd1dull + (uint64_t)attempt * 0x9e3779b97f4a7c15ull);
}
static int o_row_contains(const int32_t *idx, int64_t lo, int64_t hi, int32_t id)
{
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (idx[mid] == id) return 1;
        if (idx[mid] < id) lo = mid + 1; else hi = mid;
    }
    return 0;
}
/* returns the number of users whose row holds every item (nothing to draw; their negatives are 0) */
int64_t oracle_sample_negatives(const int64_t *indptr, const int32_t *indices, int64_t U, int64_t I, int n_neg, uint64_t seed,
                                uint64_t stream, int64_t first, int32_t *users_out, int32_t *pos_out, int32_t *neg_out)
{
    int64_t n_full = 0;
    for (int64_t u = 0; u < U; ++u) {
        const int64_t r0 = indptr[u], r1 = indptr[u + 1];
        const int full = (r1 - r0) >= I;
        if (full && r1 > r0) ++n_full;
        for (int64_t t = r0; t < r1; ++t) {
            users_out[t] = (int32_t)u;
            pos_out[t] = indices[t];
            for (int c = 0; c < n_neg; ++c) {
                const uint64_t slot = (uint64_t)t * (uint64_t)n_neg + (uint64_t)c;
                const uint64_t index = ((uint64_t)first + (uint64_t)t) * (uint64_t)n_neg + (uint64_t)c;
                uint32_t cand = 0;
                if (!full) {
                    int ok = 0;
                    for (uint32_t a = 0; a < 64u && !ok; ++a) {
                        cand = (uint32_t)(((unsigned __int128)o_draw64(seed, stream, index, a) * (unsigned __int128)(uint64_t)I) >> 64);
                        ok = !o_row_contains(indices, r0, r1, (int32_t)cand);
                    }
                    while (!ok) {
                        cand = cand + 1 == (uint32_t)I ? 0u : cand + 1;
                        ok = !o_row_contains(indices, r0, r1, (int32_t)cand);
                    }
                }
                neg_out[slot] = (int32_t)cand;
            }
        }
    }
    return n_full;
}
void oracle_shuffle_keys(int64_t n, uint64_t seed, uint64_t stream, uint64_t *keys_out)
{
    for (int64_t t = 0; t < n; ++t) keys_out[t] = o_draw64(seed, stream ^ 0x73687566666c65ull, (uint64_t)t, 0);
}
