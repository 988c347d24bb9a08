"""Test-only CPU stand-in for the local compute of tfrec_amd.sharded (float64 numpy + the oracle).
It lets the exchange logic (routing, deduplication, all-to-all, merge) run under gloo without a GPU.
Never used by the product path."""
import numpy as np
import torch

import oracle as orc


def _np(t):
    return t.numpy()


class CpuBackend:
    def uniform_rows(self, row_lo, row_hi, d, lo, hi, seed, device):
        full = np.random.default_rng(seed).uniform(lo, hi, (row_hi, d)).astype(np.float32)
        return torch.from_numpy(full[row_lo:row_hi].copy())

    def bpr_grad_step(self, P, Qc, bc, users, pos_slot, neg_slot, lr, reg):
        P_, Q_, b_ = _np(P), _np(Qc), _np(bc)
        u, i, j = _np(users).astype(np.int64), _np(pos_slot).astype(np.int64), _np(neg_slot).astype(np.int64)
        p, qi, qj = P_[u].astype(np.float64), Q_[i].astype(np.float64), Q_[j].astype(np.float64)
        z = (p * (qi - qj)).sum(1) + (b_[i].astype(np.float64) - b_[j]) + 1e-7
        s = 1.0 / (1.0 + np.exp(z))
        gP = np.zeros(P_.shape); gQ = np.zeros(Q_.shape); gb = np.zeros(b_.shape)
        np.add.at(gP, u, -s[:, None] * (qi - qj) + reg * p)
        np.add.at(gQ, i, -s[:, None] * p + reg * qi)
        np.add.at(gQ, j, s[:, None] * p + reg * qj)
        np.add.at(gb, i, -s + reg * b_[i]); np.add.at(gb, j, s + reg * b_[j])
        P_[:] = (P_ - lr * gP).astype(np.float32)
        Q_[:] = gQ.astype(np.float32)
        b_[:] = gb.astype(np.float32)

    def bpr_dense_grad_step(self, P, Q, b, G, users, pos, neg, lr, reg, slot=-1):
        assert slot == -1, "the CPU stand-in has no prepared batches"
        I, d = Q.shape
        Qc, bc = Q.clone(), b.clone()
        G.zero_()
        if users.shape[0] == 0:
            return
        self.bpr_grad_step(P, Qc, bc, users, pos, neg, lr, reg)
        touched = torch.zeros(I, dtype=torch.bool)
        touched[pos.to(torch.int64)] = True
        touched[neg.to(torch.int64)] = True
        G[: I * d].view(I, d)[touched] = Qc[touched]
        G[I * d:][touched] = bc[touched]

    def axpy(self, y, x, alpha):
        y_, x_ = _np(y), _np(x)
        y_[:] = (y_.astype(np.float64) + alpha * x_.astype(np.float64)).astype(np.float32)

    def sample_negatives(self, indptr, indices, n_users, n_items, n_neg, seed, stream, first):
        u, p, n, _ = orc.sample_negatives(_np(indptr), _np(indices), n_users, n_items, n_neg, seed, stream, first)
        return torch.from_numpy(u), torch.from_numpy(p), torch.from_numpy(n)

    def dns_select(self, P, Q, b, users, cand):
        sel, _ = orc.dns_select(_np(P), np.ascontiguousarray(_np(Q)), np.ascontiguousarray(_np(b)), _np(users), _np(cand))
        return torch.from_numpy(sel.astype(np.int32))

    def shuffle_perm(self, n, seed, stream, device):
        return torch.from_numpy(orc.shuffle_perm(n, seed, stream))

    def gather_rows(self, src, perm):
        return src[perm.to(torch.int64)].contiguous()

    def rows_sgd_apply(self, Q, b, ids, grad_rows, grad_b, lr):
        Q_, b_ = _np(Q), _np(b)
        g = np.zeros(Q_.shape); gb = np.zeros(b_.shape)
        np.add.at(g, _np(ids).astype(np.int64), _np(grad_rows).astype(np.float64))
        np.add.at(gb, _np(ids).astype(np.int64), _np(grad_b).astype(np.float64))
        Q_[:] = (Q_ - lr * g).astype(np.float32)
        b_[:] = (b_ - lr * gb).astype(np.float32)

    def score_topk_shard(self, P_rows, Q, b, item_lo, users, k, train, mode):
        P_, Q_, b_ = _np(P_rows), _np(Q), _np(b)
        n = P_.shape[0]
        ids = np.full((n, k), -1, np.int32); sc = np.full((n, k), -np.inf, np.float32)
        tp, ti = (_np(train[0]), _np(train[1])) if train is not None else (None, None)
        for r in range(n):
            row = orc.score_row(P_[r], np.ascontiguousarray(Q_), np.ascontiguousarray(b_), mode=1) if len(Q_) else np.zeros(0, np.float32)
            if tp is not None:
                u = int(users[r])
                m = ti[tp[u]:tp[u + 1]] - item_lo
                m = m[(m >= 0) & (m < len(row))]
                row[m] = -np.inf
            top = orc.topk_row(row, k) if len(row) else np.full(k, -1, np.int32)
            ok = (top >= 0)
            ok[ok] &= np.isfinite(row[top[ok]])
            ids[r, ok] = top[ok] + item_lo
            sc[r, ok] = row[top[ok]]
        return torch.from_numpy(ids), torch.from_numpy(sc)

    def topk_merge(self, ids, scores, G, n, k):
        ids_, sc_ = _np(ids), _np(scores)
        oi = np.full((n, k), -1, np.int32); os_ = np.full((n, k), -np.inf, np.float32)
        for r in range(n):
            cand = [(-float(sc_[g, r, t]), int(ids_[g, r, t])) for g in range(G) for t in range(k) if ids_[g, r, t] >= 0]
            cand.sort()
            for t, (ns, i) in enumerate(cand[:k]):
                oi[r, t] = i; os_[r, t] = -ns
        return torch.from_numpy(oi), torch.from_numpy(os_)

    def fill_masked(self, ids, users, k, train):
        ids_, tp, ti = _np(ids), _np(train[0]), _np(train[1])
        for r in range(ids_.shape[0]):
            c = int((ids_[r] >= 0).sum())
            row = ti[tp[int(users[r])]:tp[int(users[r]) + 1]]
            m = min(k - c, len(row))
            ids_[r, c:c + m] = row[:m]

    def metric_sums(self, ids, users, k, test):
        ids_, tp, ti = _np(ids), _np(test[0]), _np(test[1])
        s = np.zeros(5 * k)
        for r in range(ids_.shape[0]):
            u = int(users[r])
            s += orc.metrics_user(ids_[r], ti[tp[u]:tp[u + 1]]).reshape(-1).astype(np.float64)
        return torch.from_numpy(s)

    def sync(self):
        pass
