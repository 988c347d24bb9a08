// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product library.
//
// C-callable harness around the reference's own header-only C++ evaluator.  The
// headers are compiled from where they lie (evaluator/backend/cpp/tools.h,
// evaluator/backend/cpp/evaluate.h, found through -I at build time); nothing of
// them is copied here.  Built by oracle/build.py into oracle/_ref/.
//
// It exposes the two free functions the reference's Cython shim binds
// (evaluator/backend/cpp/apt_evaluate.pyx:7-16) with flat arrays instead of
// pointer tables, so golden rank lists and metric curves can be produced and the
// reference code can be timed as the CPU baseline.
#include "tools.h"     // c_top_k_index, c_top_k_array_index   (tools.h:10-30)
#include "evaluate.h"  // evaluate                             (evaluate.h:13-93)

#include <cstring>
#include <vector>

extern "C" {

// Reference top-k on a dense row-major [rows, cols] score matrix.
// out_rank is [rows, top_k] int32.  Mirrors apt_evaluate.pyx:26-37.
void ref_topk_rows(const float *ratings, int rows, int cols, int top_k, int thread_num, int *out_rank)
{
    std::vector<float *> rp(rows);
    std::vector<int *> op(rows);
    for (int r = 0; r < rows; ++r) {
        rp[r] = const_cast<float *>(ratings) + (size_t)r * cols;
        op[r] = out_rank + (size_t)r * top_k;
    }
    c_top_k_array_index(rp.data(), cols, rows, top_k, thread_num, op.data());
}

// Reference metric curves from rank lists.  truth is the concatenation of the
// per-user ground-truth lists, truth_ptr its CSR-style offsets [rows+1].
// results is [5, top_k] float32, overwritten.  Mirrors apt_evaluate.pyx:40-51.
void ref_evaluate(const int *ranks, int rows, int top_k, const int *truth, const long long *truth_ptr,
                  int thread_num, float *results)
{
    std::vector<int *> rk(rows), gt(rows);
    std::vector<int> gn(rows);
    for (int r = 0; r < rows; ++r) {
        rk[r] = const_cast<int *>(ranks) + (size_t)r * top_k;
        gt[r] = const_cast<int *>(truth) + truth_ptr[r];
        gn[r] = (int)(truth_ptr[r + 1] - truth_ptr[r]);
    }
    std::memset(results, 0, sizeof(float) * 5 * (size_t)top_k);  // evaluate() accumulates with +=
    evaluate(rows, rk.data(), top_k, gt.data(), gn.data(), thread_num, results);
}

// Per-user curves straight from metrics.h (no averaging); out is [5, top_k].
void ref_metrics_one(const int *rank, int top_k, const int *truth, int truth_len, float *out)
{
    int *r = const_cast<int *>(rank);
    int *t = const_cast<int *>(truth);
    std::vector<float> v;
    v = precision(r, top_k, t, truth_len); std::memcpy(out + 0 * top_k, v.data(), sizeof(float) * top_k);
    v = recall(r, top_k, t, truth_len);    std::memcpy(out + 1 * top_k, v.data(), sizeof(float) * top_k);
    v = ap(r, top_k, t, truth_len);        std::memcpy(out + 2 * top_k, v.data(), sizeof(float) * top_k);
    v = ndcg(r, top_k, t, truth_len);      std::memcpy(out + 3 * top_k, v.data(), sizeof(float) * top_k);
    v = mrr(r, top_k, t, truth_len);       std::memcpy(out + 4 * top_k, v.data(), sizeof(float) * top_k);
}

}  // extern "C"
