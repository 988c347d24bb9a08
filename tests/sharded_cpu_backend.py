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

    def items_dedup(self, pos, neg, I, items_per_owner, n_owners, counts_out):
        items = np.concatenate([_np(pos), _np(neg)]).astype(np.int64)
        uniq, inv = np.unique(items, return_inverse=True)
        B = pos.shape[0]
        owner = uniq // items_per_owner
        counts_out.copy_(torch.from_numpy(np.bincount(owner, minlength=n_owners).astype(np.int64)))
        return (torch.from_numpy(uniq.astype(np.int32)), torch.from_numpy(inv[:B].astype(np.int32)), torch.from_numpy(inv[B:].astype(np.int32)))

    def rows_gather(self, Q, b, ids):
        idx = ids.to(torch.int64)
        return Q.index_select(0, idx).contiguous(), b.index_select(0, idx).contiguous()

    def bpr_grad_step(self, P, Qc, bc, users, pos_slot, neg_slot, lr, reg, slot=-1):
        assert slot == -1, "the CPU stand-in has no prepared batches"
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

    # -- APR with a replicated item table: float64 restatement of model/APR.py:86-136 split where the
    #    product splits it (delta gradients | normalisation | Adagrad gradients | dense Adagrad) --
    @staticmethod
    def _clipsig(r):
        out = np.zeros_like(r)
        m = (r >= -80.0) & (r <= 1e8)
        out[m] = 1.0 / (1.0 + np.exp(r[m]))
        return out

    @staticmethod
    def _clipsoftplus(r):
        return np.logaddexp(0.0, -np.clip(r, -80.0, 1e8))

    def apr_dense_delta_grads(self, P, Q, b, dP, G, users, pos, neg, eps):
        P_, Q_, b_, dP_ = _np(P), _np(Q), _np(b), _np(dP)
        I, d = Q_.shape
        G.zero_()
        self._apr_batch = (_np(users).astype(np.int64), _np(pos).astype(np.int64), _np(neg).astype(np.int64))
        u, i, j = self._apr_batch
        if len(u) == 0:
            return
        p, qi, qj = P_[u].astype(np.float64), Q_[i].astype(np.float64), Q_[j].astype(np.float64)
        s = self._clipsig((p * (qi - qj)).sum(1) + (b_[i].astype(np.float64) - b_[j]))
        gP = np.zeros(P_.shape); gQ = np.zeros(Q_.shape); gb = np.zeros(b_.shape)
        np.add.at(gP, u, -s[:, None] * (qi - qj))
        np.add.at(gQ, i, -s[:, None] * p); np.add.at(gQ, j, s[:, None] * p)
        np.add.at(gb, i, -s); np.add.at(gb, j, s)
        rows = np.unique(u)
        n2 = (gP[rows] ** 2).sum(1, keepdims=True)
        dP_[rows] = (eps * gP[rows] / np.sqrt(np.maximum(n2, 1e-12))).astype(np.float32)
        G_ = _np(G)
        G_[: I * d] = gQ.astype(np.float32).reshape(-1)
        G_[I * d:] = gb.astype(np.float32)

    def apr_delta_from_grads(self, G, dQ, db, eps):
        I, d = dQ.shape
        G_ = _np(G).astype(np.float64)
        g, gb = G_[: I * d].reshape(I, d), G_[I * d:]
        _np(dQ)[:] = (eps * g / np.sqrt(np.maximum((g ** 2).sum(1, keepdims=True), 1e-12))).astype(np.float32)
        _np(db)[:] = (eps * gb / np.sqrt(max((gb ** 2).sum(), 1e-12))).astype(np.float32)

    def apr_dense_adagrad_grads(self, P, Q, b, accP, dP, dQ, db, G, n_local, n_total, lr, reg, reg_adv, loss=None):
        P_, Q_, b_, accP_ = _np(P), _np(Q), _np(b), _np(accP)
        dP_, dQ_, db_ = _np(dP), _np(dQ), _np(db)
        I, d = Q_.shape
        G.zero_()
        u, i, j = self._apr_batch
        assert len(u) == n_local
        if n_local == 0:
            if loss is not None:
                loss.zero_()
            return
        f = np.float64
        p, qi, qj = P_[u].astype(f), Q_[i].astype(f), Q_[j].astype(f)
        bi, bj = b_[i].astype(f), b_[j].astype(f)
        pa, qia, qja = p + dP_[u], qi + dQ_[i], qj + dQ_[j]
        r = (p * (qi - qj)).sum(1) + (bi - bj)
        ra = (pa * (qia - qja)).sum(1) + ((bi + db_[i]) - (bj + db_[j]))
        s, sa = self._clipsig(r), reg_adv * self._clipsig(ra)
        creg = 4.0 * reg / (n_total * d)
        gP = np.zeros(P_.shape); gQ = np.zeros(Q_.shape); gb = np.zeros(b_.shape)
        np.add.at(gP, u, -s[:, None] * (qi - qj) - sa[:, None] * (qia - qja) + creg * p)
        c = s[:, None] * p + sa[:, None] * pa
        np.add.at(gQ, i, -c + creg * qi); np.add.at(gQ, j, c + creg * qj)
        np.add.at(gb, i, -(s + sa) + creg * d * bi); np.add.at(gb, j, (s + sa) + creg * d * bj)
        rows = np.unique(u)
        an = accP_[rows].astype(f) + gP[rows] ** 2
        accP_[rows] = an.astype(np.float32)
        P_[rows] = (P_[rows].astype(f) - lr * gP[rows] / np.sqrt(an)).astype(np.float32)
        G_ = _np(G)
        G_[: I * d] = gQ.astype(np.float32).reshape(-1)
        G_[I * d:] = gb.astype(np.float32)
        if loss is not None:
            m = (p ** 2 + qi ** 2 + qj ** 2).sum(1) + d * (bi ** 2 + bj ** 2)
            _np(loss)[:] = [self._clipsoftplus(r).sum(), self._clipsoftplus(ra).sum(), m.sum() / (n_total * d)]

    def adagrad_apply(self, w, acc, g, lr):
        w_, a_, g_ = _np(w), _np(acc), _np(g).astype(np.float64)
        m = g_ != 0
        an = a_[m].astype(np.float64) + g_[m] ** 2
        a_[m] = an.astype(np.float32)
        w_[m] = (w_[m].astype(np.float64) - lr * g_[m] / np.sqrt(an)).astype(np.float32)

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
