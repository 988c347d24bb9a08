"""Multi-GPU training and evaluation of the matrix-factorization path: one process per GPU over
torch.distributed (backend "nccl" = RCCL over xGMI on a node; "gloo" on CPU for tests).

The reference is a single process (run_model.py:14-16); nothing here translates reference code.
The arithmetic is still that of model/BPR.py:36-48 and evaluator/evaluator.py:29-32.

Layout
  * user table P sharded by contiguous user ranges, item table Q / bias b by contiguous item
    ranges; rank r owns users [r*Upr, (r+1)*Upr) and items [r*Ipr, (r+1)*Ipr).
  * training: every triple is routed to the rank that owns its USER, so P rows never travel.  Per
    step a rank (1) deduplicates the items its batch touches, (2) all-to-all: requests them from
    their owners, (3) all-to-all: receives the rows, (4) runs the step on its P shard and the
    fetched copies, which leaves one summed gradient per distinct item, (5) all-to-all: returns
    those sums, (6) owners add the sums of all ranks in rank order and apply the SGD update.
    Wire traffic per triple is at most two item rows each way, less after deduplication.
  * small item tables ("dense" mode, chosen automatically below DENSE_ITEM_BYTES): every rank keeps
    a full copy of Q and b instead.  A step then needs no lookups: each rank runs its local step,
    which updates its P shard and leaves the gradient sums of the items it touched in a zeroed table
    of the shape of [Q | b]; one all-reduce sums the tables; every rank applies the same update to
    its copy.  No host synchronisation, one collective per step.  For evaluation each rank scores
    its own item range of the copy, as in the sharded layout.
  * APR (``ShardedAprTrainer``): the replicated-item layout with two all-reduces per step, one of the
    perturbation gradients and one of the Adagrad gradients.
  * evaluation with the replicated item table: user-sharded -- every rank evaluates the users it owns
    against all items, only the metric sums are all-reduced.
  * evaluation: item-sharded.  The factor rows of a round of evaluated users (a chunk per owner) are
    all-gathered, every rank takes the top-k over its own items (global ids), one all-to-all sends each
    list to the owner of its user, who merges the ranks' lists under the total order (score desc, id asc)
    -- so the result is that of an unsharded evaluation -- and the metric sums are all-reduced.

All device arithmetic goes through a *backend* object.  The product backend is ``HipBackend``
(the C ABI on this rank's GPU, torch CUDA tensors as buffers).  Tests inject a CPU backend to
run the exchange logic under gloo without a GPU; this module never falls back by itself.
"""
import numpy as np
import torch
import torch.distributed as dist


DENSE_ITEM_BYTES = 256 << 20     # item tables up to this size are replicated rather than sharded


def _via_host(t, group=None):
    """True under gloo with a device tensor: those collectives are staged through host memory here, with the copy
    on the current stream.  torch's gloo backend would copy a device tensor on a stream from its HIGH-priority pool
    and wait for it on the host; with two ranks on one GPU (the test set-up) that starves the other rank's
    ordinary-priority queues, and both ranks end up parked inside the collective (tools/stream_probe.hip replays it
    in plain HIP: the stall needs two processes, a high-priority stream and a device-to-host copy on it).  RCCL
    takes device tensors as they are and orders itself on the stream."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce(t, group=None, op=None):
    op = op if op is not None else dist.ReduceOp.SUM
    if _via_host(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_gather(x, world, group=None):
    """List of every rank's ``x`` (equal shapes)."""
    x = x.contiguous()
    if _via_host(x, group):
        h = x.cpu()
        outs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(outs, h, group=group)
        return [o.to(x.device) for o in outs]
    outs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(outs, x, group=group)
    return outs


def _broadcast(t, src, group=None):
    if _via_host(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def _all_to_all_rows(out, send, recv_counts, send_counts, group=None):
    """all_to_all_single with per-rank row counts."""
    if _via_host(send, group):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, send.contiguous().cpu(), list(recv_counts), list(send_counts), group=group)
        out.copy_(h_out)
        return
    dist.all_to_all_single(out, send.contiguous(), list(recv_counts), list(send_counts), group=group)


def _part(n, world):
    per = (n + world - 1) // world
    return per


class HipBackend:
    """Local compute through libtfrec_amd on this rank's GPU."""

    def __init__(self, dev):
        self.dev = dev

    def bpr_grad_step(self, P, Qc, bc, users, pos_slot, neg_slot, lr, reg, slot=-1, ut=None):
        """``ut``: the versioned table over ``P`` (used when the batch was prepared; ``P`` is then not read here)."""
        if slot >= 0 and ut is not None:
            self.dev.bpr_grad_step_prepared_v(slot, ut, Qc, bc, lr, reg)
        elif slot >= 0:
            self.dev.bpr_grad_step_prepared(slot, P, Qc, bc, lr, reg)
        else:
            self.dev.bpr_grad_step(P, Qc, bc, users, pos_slot, neg_slot, lr, reg)

    def items_dedup(self, pos, neg, I, items_per_owner, n_owners, counts_out):
        """(uniq buffer, pos_slot, neg_slot): the distinct items of a batch, ascending, in the first sum(counts) entries
        of the buffer; ``counts_out`` (device int64[n_owners]) receives the distinct items per owner."""
        B = pos.shape[0]
        uniq = torch.empty(max(min(2 * B, I), 1), dtype=torch.int32, device=pos.device)
        ps, ns = torch.empty(B, dtype=torch.int32, device=pos.device), torch.empty(B, dtype=torch.int32, device=pos.device)
        self.dev.items_dedup(pos, neg, I, items_per_owner, n_owners, uniq, ps, ns, counts_out)
        return uniq, ps, ns

    def rows_gather(self, Q, b, ids):
        n = ids.shape[0]
        rows = torch.empty((n, Q.shape[1]), dtype=torch.float32, device=Q.device)
        bias = torch.empty(n, dtype=torch.float32, device=Q.device)
        if n:
            self.dev.rows_gather(Q, b, ids, rows, bias)
        return rows, bias

    def rows_sgd_apply(self, Q, b, ids, grad_rows, grad_b, lr):
        self.dev.rows_sgd_apply(Q, b, ids, grad_rows, grad_b, lr)

    def bpr_dense_grad_step(self, P, Q, b, G, users, pos, neg, lr, reg, slot=-1):
        self.dev.bpr_dense_grad_step(P, Q, b, G, users, pos, neg, lr, reg, slot=slot)

    def prepare_batch(self, slot, U, I, users, pos, neg):
        self.dev.prepare_batch(slot, U, I, users, pos, neg)

    def user_table(self, P):
        """A versioned user table over ``P`` (tfrec_utable): steps on it stage no pre-step user rows."""
        return self.dev.user_table(P)

    def bpr_dense_grad_step_v(self, ut, Q, b, G, users, pos, neg, lr, reg, slot=-1):
        self.dev.bpr_dense_grad_step_v(ut, Q, b, G, users, pos, neg, lr, reg, slot=slot)

    def bpr_dense_grad_step_chunked(self, ut, Q, b, G, users, pos, neg, lr, reg, events, slot=-1):
        """``events``: torch.cuda.Event objects, recorded by the library as the item ranges of ``G`` complete."""
        self.dev.bpr_dense_grad_step_chunked(ut, Q, b, G, users, pos, neg, lr, reg, [e.cuda_event for e in events], slot=slot)

    def axpy(self, y, x, alpha):
        self.dev.axpy(y, x, alpha)

    def apr_dense_delta_grads(self, P, Q, b, dP, G, users, pos, neg, eps):
        self.dev.apr_dense_delta_grads(P, Q, b, dP, G, users, pos, neg, eps)

    def apr_delta_from_grads(self, G, dQ, db, eps):
        self.dev.apr_delta_from_grads(G, dQ, db, eps)

    def apr_dense_adagrad_grads(self, P, Q, b, accP, dP, dQ, db, G, n_local, n_total, lr, reg, reg_adv, loss=None):
        self.dev.apr_dense_adagrad_grads(P, Q, b, accP, dP, dQ, db, G, n_local, n_total, lr, reg, reg_adv, loss=loss)

    def adagrad_apply(self, w, acc, g, lr):
        self.dev.adagrad_apply(w, acc, g, lr)

    def sample_negatives(self, indptr, indices, n_users, n_items, n_neg, seed, stream, first):
        nnz = indices.shape[0]
        u = torch.empty(nnz, dtype=torch.int32, device=indices.device)
        p = torch.empty(nnz, dtype=torch.int32, device=indices.device)
        n = torch.empty((nnz, n_neg), dtype=torch.int32, device=indices.device)
        self.dev.sample_negatives(indptr, indices, n_users, n_items, n_neg, seed, stream, users_out=u, pos_out=p, neg_out=n,
                                  first_interaction=first)
        return u, p, n

    def dns_select(self, P, Q, b, users, cand):
        out = torch.empty(users.shape[0], dtype=torch.int32, device=users.device)
        self.dev.dns_select(P, Q, b, users, cand, out=out)
        return out

    def shuffle_perm(self, n, seed, stream, device):
        out = torch.empty(n, dtype=torch.int32, device=device)
        self.dev.shuffle_perm(n, seed, stream, out=out)
        return out

    def gather_rows(self, src, perm):
        out = torch.empty((perm.shape[0],) + tuple(src.shape[1:]), dtype=torch.int32, device=src.device)
        self.dev.gather_rows_i32(src.contiguous(), perm, out=out)
        return out

    def score_topk_shard(self, P_rows, Q, b, item_lo, users, k, train, mode):
        n = P_rows.shape[0]
        ids = torch.empty((n, k), dtype=torch.int32, device=P_rows.device)
        sc = torch.empty((n, k), dtype=torch.float32, device=P_rows.device)
        self.dev.score_topk_shard(P_rows, Q, b, item_lo, users, k, train=train, mode=mode, out_ids=ids, out_scores=sc)
        return ids, sc

    def topk_merge(self, ids, scores, G, n, k):
        oi = torch.empty((n, k), dtype=torch.int32, device=ids.device)
        os_ = torch.empty((n, k), dtype=torch.float32, device=ids.device)
        self.dev.topk_merge(ids, scores, G, n, k, out_ids=oi, out_scores=os_)
        return oi, os_

    def fill_masked(self, ids, users, k, train):
        self.dev.topk_fill_masked(ids, users, k, train)

    def metric_sums(self, ids, users, k, test):
        sums = torch.empty(5 * k, dtype=torch.float64, device=ids.device)
        self.dev.rank_metrics_sums(ids, users, k, test, sums=sums)
        return sums

    def uniform_rows(self, row_lo, row_hi, d, lo, hi, seed, device):
        """Rows [row_lo, row_hi) of the table the counter-based generator would fill for ``seed``."""
        t = torch.empty((row_hi - row_lo, d), dtype=torch.float32, device=device)
        from . import _lib
        _lib.check(self.dev.lib.tfrec_fill_uniform_at(self.dev.h, t.data_ptr(), t.numel(), float(lo), float(hi), int(seed),
                                                      int(row_lo) * int(d)))
        return t

    def sync(self):
        self.dev.synchronize()


class PhaseTimers:
    """Stream-event marks of the phases of a planned step (benchmarks only): ``mark(name)`` records an event on the
    current stream; ``totals()`` waits and returns the milliseconds spent before each named mark, summed over steps."""

    def __init__(self):
        self.events = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append((name, e))

    def totals(self):
        torch.cuda.synchronize()
        out = {}
        for (n0, e0), (n1, e1) in zip(self.events[:-1], self.events[1:]):
            if n1 != "start":
                out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out


def make_device(local_rank):
    """A Device whose stream is also torch's current stream on that GPU, so that torch ops (the
    exchange plumbing) and library kernels are ordered without extra synchronisation."""
    from .device import Device
    torch.cuda.set_device(local_rank)
    stream = torch.cuda.Stream(device=torch.device("cuda", local_rank))
    torch.cuda.set_stream(stream)
    dev = Device(local_rank, stream=stream.cuda_stream)
    dev._torch_stream = stream          # keep the stream alive as long as the context
    return dev


class ShardedBprTrainer:
    """BPR-MF with row-sharded tables.  ``step`` takes this rank's triples: global ids, every user
    inside ``user_range``."""

    def __init__(self, dev_or_backend, U, I, d, lr, reg, seed=2018, device=None, group=None, item_mode="auto", overlap_chunks=None):
        """``overlap_chunks`` (replicated item table): the item half of a step runs in this many ranges of item ids and
        the all-reduce of a range's gradient rows starts, on a second stream, while the next ranges are still being
        computed.  Default: 2 under RCCL with more than one rank (every further range costs the item half about 3 % in fill and
        drain of its launches, measured on one GPU), else 1 (one all-reduce after the step, and on one
        rank a result that equals the single-GPU step bit for bit).  The library's stream must be torch's current
        stream when steps are called (as ``bench.py`` arranges), because the collectives order themselves on that."""
        self.backend = dev_or_backend if hasattr(dev_or_backend, "uniform_rows") else HipBackend(dev_or_backend)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # collectives run whenever a process group exists, also one of a single rank (where they are trivial): the
        # single-rank run then takes the same path as the many-rank one
        self._coll = dist.is_initialized()
        self.U, self.I, self.d, self.lr, self.reg = int(U), int(I), int(d), float(lr), float(reg)
        self.upr, self.ipr = _part(self.U, self.world), _part(self.I, self.world)
        self.user_range = (min(self.rank * self.upr, self.U), min((self.rank + 1) * self.upr, self.U))
        self.item_range = (min(self.rank * self.ipr, self.I), min((self.rank + 1) * self.ipr, self.I))
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        if item_mode == "auto":
            item_mode = "dense" if 4 * self.I * (self.d + 1) <= DENSE_ITEM_BYTES else "sharded"
        if item_mode not in ("dense", "sharded"):
            raise ValueError("item_mode must be 'auto', 'dense' or 'sharded'")
        self.item_mode = item_mode
        nu, ni = self.user_range[1] - self.user_range[0], self.item_range[1] - self.item_range[0]
        self._ut = None
        # model/base/MatrixFactorization.py:17-21; the stream of random numbers is that of the unsharded
        # table, so the shards of any world size assemble into the same initial model
        self._P = self._init_rows(self.U, self.user_range, seed * 2 + 1)
        # a backend that offers it: the user shard is a versioned table (no staging of pre-step user rows in the
        # step); reading ``self.P`` flattens it first
        self._ut = self.backend.user_table(self._P) if hasattr(self.backend, "user_table") and nu > 0 else None
        self._ut_dirty = False
        if item_mode == "dense":
            # [Q | b] in one flat buffer, so that one all-reduce and one update cover both
            self._Qb = torch.zeros(self.I * (self.d + 1), dtype=torch.float32, device=device)
            self._Qb[: self.I * self.d].view(self.I, self.d).copy_(self._init_rows(self.I, (0, self.I), seed * 2 + 2))
            self._G = torch.zeros_like(self._Qb)
            self.Q_full = self._Qb[: self.I * self.d].view(self.I, self.d)
            self.b_full = self._Qb[self.I * self.d:]
            # the views the item-sharded evaluator works on: this rank's item range of the copy
            self.Q = self.Q_full[self.item_range[0]:self.item_range[1]]
            self.b = self.b_full[self.item_range[0]:self.item_range[1]]
        else:
            self.Q = self._init_rows(self.I, self.item_range, seed * 2 + 2)
            self.b = torch.zeros(max(ni, 1), dtype=torch.float32, device=device)[:ni]
        assert self.P.shape[0] == nu and self.Q.shape[0] == ni
        self._mark = None
        self._timers = None         # a PhaseTimers object while a benchmark wants the split of a planned step
        self._pending = {}          # slot -> index tensors of a prepared batch, alive until its step is enqueued
        if overlap_chunks is None:
            overlap_chunks = 2 if (self._coll and self.world > 1 and dist.get_backend(group) == "nccl") else 1
        self.overlap_chunks = int(overlap_chunks)
        if not (1 <= self.overlap_chunks <= 32):
            raise ValueError("overlap_chunks must be 1..32")
        if self.overlap_chunks > 1 and not (item_mode == "dense" and self._ut is not None and hasattr(self.backend, "bpr_dense_grad_step_chunked")):
            self.overlap_chunks = 1
        if self.overlap_chunks > 1:
            self._side = torch.cuda.Stream(device=device)
            self._chunk_events = [torch.cuda.Event() for _ in range(self.overlap_chunks)]
            for e in self._chunk_events:
                e.record()          # creates the underlying HIP event, which the library records from then on

    def _init_rows(self, n_total, rng, seed):
        return self.backend.uniform_rows(rng[0], rng[1], self.d, -0.05, 0.05, seed, self.device)

    @property
    def P(self):
        """This rank's user rows, complete and current (a versioned table is flattened first)."""
        if self._ut is not None and self._ut_dirty:
            self._ut.flatten()
            self._ut_dirty = False
        return self._P

    # ------------------------------------------------------------------------------------------
    def _a2a(self, send, send_counts, recv_counts):
        """all_to_all_single with per-rank row counts (lists of ints)."""
        out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if not self._coll:
            out.copy_(send)
            return out
        _all_to_all_rows(out, send, recv_counts, send_counts, self.group)
        return out

    def prepare(self, slot, users, pos, neg):
        """Dense mode: sorts the index arrays of an upcoming batch on the library's auxiliary stream
        (slot 0 or 1), beside whatever the main stream is doing; ``step_prepared(slot)`` consumes it."""
        assert self.item_mode == "dense"
        u = (users.to(torch.int32) - self.user_range[0]).contiguous()
        p, n = pos.to(torch.int32).contiguous(), neg.to(torch.int32).contiguous()
        # The library reads these arrays on its auxiliary stream, which torch's allocator knows nothing about:
        # they are kept until the step that consumes the slot has been enqueued on the main stream (that step
        # waits for the preparation, so whatever reuses the memory afterwards is ordered behind the reads).
        self._pending[slot] = (u, p, n)
        self.backend.prepare_batch(slot, self.user_range[1] - self.user_range[0], self.I, u, p, n)

    def step_prepared(self, slot, prepare_next=None):
        """Consumes the batch prepared in ``slot``.  ``prepare_next`` = (slot, users, pos, neg) of the
        batch after it: its index work is enqueued once this step's row passes are, so it runs beside
        the all-reduce (which leaves the GPU's memory system idle) rather than beside the passes."""
        self._dense_step(None, None, None, slot, prepare_next)

    def _dense_step(self, users, pos, neg, slot, prepare_next=None):
        if users is not None:
            users = (users.to(torch.int32) - self.user_range[0]).contiguous()
            pos, neg = pos.to(torch.int32).contiguous(), neg.to(torch.int32).contiguous()
        if self.overlap_chunks > 1:
            return self._dense_step_overlapped(users, pos, neg, slot, prepare_next)
        if self._ut is not None:
            self.backend.bpr_dense_grad_step_v(self._ut, self.Q_full, self.b_full, self._G, users, pos, neg, self.lr, self.reg, slot=slot)
            self._ut_dirty = True
        else:
            self.backend.bpr_dense_grad_step(self._P, self.Q_full, self.b_full, self._G, users, pos, neg, self.lr, self.reg, slot=slot)
        if slot >= 0:
            self._pending.pop(slot, None)        # consumed: the step that read the slot is enqueued
        if prepare_next is not None:
            self.prepare(*prepare_next)
        if self._coll:
            _all_reduce(self._G, self.group)
        self.backend.axpy(self._Qb, self._G, -self.lr)

    def _dense_step_overlapped(self, users, pos, neg, slot, prepare_next):
        """The dense step with the all-reduce of the item gradient table overlapped with the item half that produces
        it: rows of item range c are summed over the ranks (second stream) while ranges c+1.. are computed."""
        K, I, d = self.overlap_chunks, self.I, self.d
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)         # the previous step's update has read the table the library is about to zero
        self.backend.bpr_dense_grad_step_chunked(self._ut, self.Q_full, self.b_full, self._G, users, pos, neg, self.lr, self.reg,
                                                 self._chunk_events, slot=slot)
        self._ut_dirty = True
        if slot >= 0:
            self._pending.pop(slot, None)
        if prepare_next is not None:
            self.prepare(*prepare_next)
        ipc = (I + K - 1) // K
        with torch.cuda.stream(self._side):
            for c in range(K):
                lo, hi = min(c * ipc, I), min((c + 1) * ipc, I)
                self._side.wait_event(self._chunk_events[c])
                if hi > lo and self._coll:
                    _all_reduce(self._G[lo * d: hi * d], self.group)
            if self._coll:
                _all_reduce(self._G[I * d:], self.group)         # biases: complete with the last range
        main.wait_stream(self._side)
        self.backend.axpy(self._Qb, self._G, -self.lr)

    def step(self, users, pos, neg):
        """One global minibatch step; collective: every rank must call it (an empty batch is allowed).  With a
        sharded item table this plans the exchange of just this batch (one host synchronisation for the
        counts); ``plan_steps`` + ``step_planned`` plan many steps at once and then run them without any."""
        if self.item_mode == "dense":
            return self._dense_step(users, pos, neg, -1)
        self.step_planned(self.plan_steps([(users, pos, neg)])[0])

    def plan_steps(self, batches):
        """Plans the lookups of a list of upcoming steps (collective).  ``batches``: this rank's (users, pos, neg)
        per step, global ids, every user inside ``user_range``; empty batches allowed.  Per step the library
        de-duplicates the items (marks + prefix sum, ascending, hence grouped by owner) and counts them per owner;
        ALL the counts cross in one all-to-all and are read back once, and then ALL the requested ids cross in a
        second one (they depend on the batches alone, not on the parameters), so that a step itself
        (``step_planned``) exchanges only rows and gradient sums and never touches the host."""
        assert self.item_mode == "sharded"
        W, dev = self.world, self.device
        S = len(batches)
        counts = torch.zeros((S, W), dtype=torch.int64, device=dev)
        plans = []
        for s, (users, pos, neg) in enumerate(batches):
            users = (users.to(torch.int32) - self.user_range[0]).contiguous()
            pos, neg = pos.to(torch.int32).contiguous(), neg.to(torch.int32).contiguous()
            uniq, ps, ns = self.backend.items_dedup(pos, neg, self.I, self.ipr, W, counts[s])
            plans.append({"B": int(users.shape[0]), "users": users, "pos_slot": ps, "neg_slot": ns, "uniq": uniq})
        if self._coll:
            asked = torch.empty((W, S), dtype=torch.int64, device=dev)          # row r: what rank r asks of me, per step
            _all_to_all_rows(asked, counts.t().contiguous(), [1] * W, [1] * W, self.group)
        else:
            asked = counts.t().contiguous()
        both = torch.stack([counts, asked.t().contiguous()]).cpu()           # the one host read of the plan
        for s, p in enumerate(plans):
            p["send"], p["recv"] = both[0, s].tolist(), both[1, s].tolist()
            p["uniq"] = p["uniq"][: sum(p["send"])]
        # the ids of every step to their owners, in one exchange: the block for owner o is the steps' shares for o,
        # one after the other; what arrives from rank r splits the same way by the counts r announced
        if S:
            offs = [np.concatenate([[0], np.cumsum(p["send"])]).tolist() for p in plans]
            send = torch.cat([plans[s]["uniq"][offs[s][o]:offs[s][o + 1]] for o in range(W) for s in range(S)]) if S * W else plans[0]["uniq"][:0]
            send_counts = [sum(plans[s]["send"][o] for s in range(S)) for o in range(W)]
            recv_counts = [sum(plans[s]["recv"][r] for s in range(S)) for r in range(W)]
            got = self._a2a(send, send_counts, recv_counts)
            pos_, pieces = 0, [[None] * W for _ in range(S)]
            for r in range(W):
                for s in range(S):
                    n = plans[s]["recv"][r]
                    pieces[s][r] = got[pos_:pos_ + n]
                    pos_ += n
            for s, p in enumerate(plans):
                p["local"] = (torch.cat(pieces[s]) - self.item_range[0]).contiguous()      # my rows the ranks want, by rank
        return plans

    def _prepare_planned(self, plan, slot):
        """Index sort of a planned batch on the library's auxiliary stream (slot 0 or 1)."""
        if plan["B"] > 0 and hasattr(self.backend, "prepare_batch") and not plan.get("prepared_in") == slot:
            nu = self.user_range[1] - self.user_range[0]
            self.backend.prepare_batch(slot, nu, sum(plan["send"]), plan["users"], plan["pos_slot"], plan["neg_slot"])
            plan["prepared_in"] = slot

    def step_planned(self, plan, slot=0, next_plan=None):
        """One global minibatch step from a plan of ``plan_steps``: no host synchronisation, four exchanges (rows
        and biases to the ranks that asked, gradient sums back).  The batch's index sort runs on the library's
        auxiliary stream beside the exchange that fetches the rows; with ``next_plan`` the sort of the NEXT batch
        is enqueued (other slot) before this step's gradient sums travel back, so that it runs beside them."""
        be, sc, rc = self.backend, plan["send"], plan["recv"]
        B = plan["B"]
        ahead = B > 0 and hasattr(be, "prepare_batch")
        self._prepare_planned(plan, slot)
        tm = self._timers
        if tm is not None:
            tm.mark("start")
        # (1) the owners' rows and biases to the ranks that asked for them, in the order of plan["uniq"] there
        local = plan["local"]
        rows, bias = be.rows_gather(self.Q, self.b, local)
        Qc, bc = self._a2a(rows, rc, sc), self._a2a(bias, rc, sc)
        if tm is not None:
            tm.mark("fetch")
        # (2) local step: P shard updated in place, Qc / bc turned into gradient sums
        if B > 0 and ahead and self._ut is not None:
            be.bpr_grad_step(self._P, Qc, bc, plan["users"], plan["pos_slot"], plan["neg_slot"], self.lr, self.reg, slot=slot, ut=self._ut)
            self._ut_dirty = True
        elif B > 0:
            be.bpr_grad_step(self.P, Qc, bc, plan["users"], plan["pos_slot"], plan["neg_slot"], self.lr, self.reg, slot=slot if ahead else -1)
        if next_plan is not None:
            self._prepare_planned(next_plan, slot ^ 1)
        if tm is not None:
            tm.mark("compute")
        # (3) gradient sums back to the owners, who add them in rank order and apply
        gr, gb = self._a2a(Qc, sc, rc), self._a2a(bc, sc, rc)
        if tm is not None:
            tm.mark("return")
        if local.shape[0] > 0:
            be.rows_sgd_apply(self.Q, self.b, local, gr, gb, self.lr)
        if tm is not None:
            tm.mark("apply")

    def run_planned(self, plans):
        """The steps of ``plans`` in order, each batch's index sort enqueued one step ahead."""
        for s, plan in enumerate(plans):
            self.step_planned(plan, slot=s & 1, next_plan=plans[s + 1] if s + 1 < len(plans) else None)

    def full_item_table(self):
        """(Q, b) of ALL items on this rank: the local copy (replicated table) or an all-gather of the
        shards (collective).  Read-only use, e.g. the epoch-start candidate scoring of DNS."""
        if self.item_mode == "dense":
            return self.Q_full, self.b_full
        if not self._coll:
            return self.Q, self.b

        def cat(x):
            pad = torch.zeros((self.ipr,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            pad[: x.shape[0]] = x
            outs = _all_gather(pad, self.world, self.group)
            return torch.cat(outs)[: self.I].contiguous()
        return cat(self.Q), cat(self.b)

    def gather_full(self):
        """(P, Q, b) assembled on every rank (testing / checkpointing)."""
        def cat(x, n):
            if not self._coll:
                return x.clone()
            per = _part(n, self.world)
            pad = torch.zeros((per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            pad[: x.shape[0]] = x
            outs = _all_gather(pad, self.world, self.group)
            return torch.cat(outs)[:n]
        if self.item_mode == "dense":
            return cat(self.P, self.U), self.Q_full.clone(), self.b_full.clone()
        return cat(self.P, self.U), cat(self.Q, self.I), cat(self.b, self.I)


class ShardedAprTrainer:
    """APR (model/APR.py:86-136, 156-164) over several GPUs: P, its Adagrad accumulator and its delta
    table are sharded by user range; Q, b with their accumulators and deltas are replicated (the
    layout ``ShardedBprTrainer`` calls "dense").  Every triple goes to the rank that owns its user.
    One step is four library calls and two all-reduces of a table of the shape of [Q | b]:

      delta gradients of the local triples -> all-reduce -> delta_Q, delta_b on every copy
      Adagrad gradients of the local triples (P updated locally) -> all-reduce -> Adagrad on every copy

    The adversarial perturbation normalises the item gradients of the WHOLE step (rows one by one,
    the bias gradient as one vector), which is why the sums have to be complete before phase two."""

    item_mode = "dense"

    def __init__(self, dev_or_backend, U, I, d, lr, reg, reg_adv, eps, params=None, seed=2018, device=None, group=None,
                 initial_accumulator=0.1):
        self.backend = dev_or_backend if hasattr(dev_or_backend, "uniform_rows") else HipBackend(dev_or_backend)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # collectives run whenever a process group exists, also one of a single rank (where they are trivial): the
        # single-rank run then takes the same path as the many-rank one
        self._coll = dist.is_initialized()
        self.U, self.I, self.d = int(U), int(I), int(d)
        self.lr, self.reg, self.reg_adv, self.eps = float(lr), float(reg), float(reg_adv), float(eps)
        self.upr, self.ipr = _part(self.U, self.world), _part(self.I, self.world)
        self.user_range = (min(self.rank * self.upr, self.U), min((self.rank + 1) * self.upr, self.U))
        self.item_range = (min(self.rank * self.ipr, self.I), min((self.rank + 1) * self.ipr, self.I))
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        lo, hi = self.user_range
        n_item = self.I * (self.d + 1)
        self._Qb = torch.zeros(n_item, dtype=torch.float32, device=device)
        self.Q_full = self._Qb[: self.I * self.d].view(self.I, self.d)
        self.b_full = self._Qb[self.I * self.d:]
        if params is not None:
            # pretrained [P, Q, b] (model/APR.py:47-53): every rank takes its rows
            P0, Q0, b0 = (np.ascontiguousarray(x, dtype=np.float32) for x in params)
            assert P0.shape == (self.U, self.d) and Q0.shape == (self.I, self.d) and b0.shape == (self.I,)
            self.P = torch.from_numpy(P0[lo:hi].copy()).to(device)
            self.Q_full.copy_(torch.from_numpy(Q0))
            self.b_full.copy_(torch.from_numpy(b0))
        else:
            self.P = self.backend.uniform_rows(lo, hi, self.d, -0.05, 0.05, seed * 2 + 1, device)
            self.Q_full.copy_(self.backend.uniform_rows(0, self.I, self.d, -0.05, 0.05, seed * 2 + 2, device))
        self.accP = torch.full_like(self.P, float(initial_accumulator))
        self.dP = torch.zeros_like(self.P)
        self._accQb = torch.full_like(self._Qb, float(initial_accumulator))
        self._dQb = torch.zeros_like(self._Qb)
        self.dQ = self._dQb[: self.I * self.d].view(self.I, self.d)
        self.db = self._dQb[self.I * self.d:]
        self._G = torch.zeros_like(self._Qb)
        self.Q = self.Q_full[self.item_range[0]:self.item_range[1]]
        self.b = self.b_full[self.item_range[0]:self.item_range[1]]

    def _sum_over_ranks(self, t):
        if self._coll:
            _all_reduce(t, self.group)

    def step(self, users, pos, neg, global_batch=None, loss=None):
        """One global minibatch step; collective (an empty local batch is allowed).  ``global_batch``:
        triples of the step over all ranks (it scales the mean-square term); counted with one more
        all-reduce and a host read when not given.  ``loss``: optional float64[3] device tensor that
        receives the step's three loss terms summed over the ranks."""
        be = self.backend
        users = (users.to(torch.int32) - self.user_range[0]).contiguous()
        pos, neg = pos.to(torch.int32).contiguous(), neg.to(torch.int32).contiguous()
        n_local = int(users.shape[0])
        if global_batch is None:
            cnt = torch.tensor([n_local], dtype=torch.int64, device=self.device)
            self._sum_over_ranks(cnt)
            global_batch = int(cnt.item())
        if global_batch == 0:
            return
        be.apr_dense_delta_grads(self.P, self.Q_full, self.b_full, self.dP, self._G, users, pos, neg, self.eps)
        self._sum_over_ranks(self._G)
        be.apr_delta_from_grads(self._G, self.dQ, self.db, self.eps)
        be.apr_dense_adagrad_grads(self.P, self.Q_full, self.b_full, self.accP, self.dP, self.dQ, self.db, self._G, n_local,
                                   global_batch, self.lr, self.reg, self.reg_adv, loss=loss)
        self._sum_over_ranks(self._G)
        be.adagrad_apply(self._Qb, self._accQb, self._G, self.lr)
        if loss is not None:
            self._sum_over_ranks(loss)

    def full_item_table(self):
        return self.Q_full, self.b_full

    def gather_full(self):
        """(P, Q, b) assembled on every rank (testing / checkpointing)."""
        if not self._coll:
            return self.P.clone(), self.Q_full.clone(), self.b_full.clone()
        pad = torch.zeros((self.upr, self.d), dtype=self.P.dtype, device=self.device)
        pad[: self.P.shape[0]] = self.P
        outs = _all_gather(pad, self.world, self.group)
        return torch.cat(outs)[: self.U], self.Q_full.clone(), self.b_full.clone()


class ShardedEvaluator:
    """Item-sharded full-rank evaluation.  ``train`` / ``test`` are (indptr int64, indices int32)
    tensors over ALL users on this rank's device (replicated); ``test_users`` the sorted global ids
    of the users to evaluate (replicated)."""

    def __init__(self, dev_or_backend, trainer, train, test, test_users, top_k=50, mode=0, chunk_users=1 << 18, layout="auto"):
        self.backend = dev_or_backend if hasattr(dev_or_backend, "metric_sums") else HipBackend(dev_or_backend)
        self.t = trainer
        self.train, self.test = train, test
        self.test_users = test_users.to(torch.int32)
        self.k, self.mode, self.chunk = int(top_k), int(mode), int(chunk_users)
        # "items": every rank scores its item range for all users, lists gathered and merged (the only
        # choice when the item table is sharded).  "users": with a replicated item table every rank
        # evaluates the users it owns against all items -- no exchange but the final sum.
        if layout == "auto":
            layout = "users" if getattr(trainer, "item_mode", "sharded") == "dense" else "items"
        if layout not in ("items", "users"):
            raise ValueError("layout must be 'auto', 'items' or 'users'")
        if layout == "users" and getattr(trainer, "item_mode", "sharded") != "dense":
            raise ValueError("user-sharded evaluation needs the replicated item table")
        self.layout = layout

    def _evaluate_own_users(self):
        t, k = self.t, self.k
        sums = torch.zeros(5 * k, dtype=torch.float64, device=t.device)
        tu = self.test_users.to(torch.int64)
        own = self.test_users[(tu >= t.user_range[0]) & (tu < t.user_range[1])]
        for s in range(0, own.shape[0], self.chunk):
            eu = own[s:s + self.chunk].contiguous()
            rows = t.P.index_select(0, eu.to(torch.int64) - t.user_range[0]).contiguous()
            ids, _ = self.backend.score_topk_shard(rows, t.Q_full, t.b_full, 0, eu, k, self.train, self.mode)
            if self.train is not None:
                self.backend.fill_masked(ids, eu, k, self.train)
            sums += self.backend.metric_sums(ids, eu, k, self.test)
        if t._coll:
            _all_reduce(sums, t.group)
        return (sums / float(self.test_users.shape[0])).to(torch.float32).reshape(5, k)

    def evaluate(self):
        """float32 [5, k] on every rank: Precision, Recall, MAP, NDCG, MRR @1..k."""
        if self.layout == "users":
            return self._evaluate_own_users()
        # Item-sharded.  A round takes the next ``chunk`` users of EVERY owner: their factor rows are all-gathered,
        # every rank takes the top-k of all of the round's users over its own items, and one all-to-all sends each
        # list to the owner of its user; all ranks then merge, fill and score their own users at the same time.
        t, k, W = self.t, self.k, self.t.world
        dev = t.device
        sums = torch.zeros(5 * k, dtype=torch.float64, device=dev)
        owners = torch.div(self.test_users.to(torch.int64), t.upr, rounding_mode="floor")
        by_owner = [self.test_users[owners == o] for o in range(W)]
        longest = max(int(x.shape[0]) for x in by_owner) if W else 0
        for s in range(0, longest, self.chunk):
            parts = [x[s:s + self.chunk].contiguous() for x in by_owner]
            sizes = [int(x.shape[0]) for x in parts]
            mine = parts[t.rank]
            rows_mine = t.P.index_select(0, mine.to(torch.int64) - t.user_range[0]).contiguous()
            if t._coll and W > 1:
                pad = torch.zeros((max(sizes), t.d), dtype=torch.float32, device=dev)
                pad[: sizes[t.rank]] = rows_mine
                rows = torch.cat([g[:n] for g, n in zip(_all_gather(pad, W, t.group), sizes)])
            else:
                rows = rows_mine
            eu = torch.cat(parts).contiguous()
            ids, sc = self.backend.score_topk_shard(rows, t.Q, t.b, t.item_range[0], eu, k, self.train, self.mode)
            n_me = sizes[t.rank]
            if t._coll and W > 1:
                got_ids = torch.empty((W * n_me, k), dtype=ids.dtype, device=dev)
                got_sc = torch.empty((W * n_me, k), dtype=sc.dtype, device=dev)
                _all_to_all_rows(got_ids, ids, [n_me] * W, sizes, t.group)
                _all_to_all_rows(got_sc, sc, [n_me] * W, sizes, t.group)
            else:
                got_ids, got_sc = ids, sc
            if n_me > 0:
                mi, _ = self.backend.topk_merge(got_ids.view(W, n_me, k).contiguous(), got_sc.view(W, n_me, k).contiguous(), W, n_me, k)
                if self.train is not None:
                    self.backend.fill_masked(mi, mine, k, self.train)
                sums += self.backend.metric_sums(mi, mine, k, self.test)
        if t._coll:
            _all_reduce(sums, t.group)
        return (sums / float(self.test_users.shape[0])).to(torch.float32).reshape(5, k)


class ShardedEpochSampler:
    """Epochs of triples for a ``ShardedBprTrainer``, drawn on every rank for the users it owns
    (model/BPR.py:50-62; with ``n_neg > 1`` the dynamic negative sampling of model/DNSBPR.py:65-80:
    pools of ``n_neg`` candidates per positive, the best-scored one under the parameters at the
    start of the epoch).  A rank's draws are those the whole training matrix would draw for its rows
    (the draw counter is the global position of the interaction), so the sampled triples do not
    depend on the number of ranks; each rank shuffles its own triples."""

    def __init__(self, dev_or_backend, trainer, train_matrix, seed=2018, n_neg=1, shuffle="rank"):
        """``shuffle``: "rank" -- every rank shuffles its own triples and feeds ``batch_size`` of them per step (the
        throughput setting: a global batch is ``world * batch_size`` triples); "global" -- every rank derives the ONE
        permutation a single process would draw for the whole epoch (data/dataiterator.py:52-54) and takes, of
        every batch of ``batch_size`` consecutive positions, the triples of the users it owns: the global batches
        are then exactly the single-process batches whatever the number of ranks."""
        self.backend = dev_or_backend if hasattr(dev_or_backend, "uniform_rows") else HipBackend(dev_or_backend)
        self.t = trainer
        if shuffle not in ("rank", "global"):
            raise ValueError("shuffle must be 'rank' or 'global'")
        self.shuffle = shuffle
        from .utils.tools import csr_trained_rows
        csr = csr_trained_rows(train_matrix)       # the users the reference trains (utils/tools.py:52-62)
        lo, hi = trainer.user_range
        self.first = int(csr.indptr[lo])
        self.total_nnz = int(csr.nnz)
        sub = csr[lo:hi]
        dev = trainer.device
        self.indptr = torch.from_numpy(sub.indptr.astype(np.int64)).to(dev)
        self.indices = torch.from_numpy(sub.indices.astype(np.int32)).to(dev)
        self.nnz = int(sub.nnz)
        self.n_neg, self.seed = int(n_neg), int(seed)

    @classmethod
    def from_local_csr(cls, dev_or_backend, trainer, indptr, indices, first, total_nnz, seed=2018, n_neg=1, shuffle="rank"):
        """A sampler over this rank's rows given as device tensors (``indptr`` int64 [rows + 1] starting at 0,
        ``indices`` int32, sorted within a row), for callers that build the interaction lists on the device;
        ``first``: position of the rank's first interaction in the whole training set, ``total_nnz``: its size."""
        self = cls.__new__(cls)
        self.backend = dev_or_backend if hasattr(dev_or_backend, "uniform_rows") else HipBackend(dev_or_backend)
        self.t, self.shuffle = trainer, shuffle
        self.first, self.total_nnz = int(first), int(total_nnz)
        self.indptr, self.indices = indptr, indices
        self.nnz = int(indices.shape[0])
        self.n_neg, self.seed = int(n_neg), int(seed)
        return self

    def draw(self, e):
        """(local user rows, items, negatives) of epoch ``e`` in storage order; collective when the item
        table is sharded and ``n_neg > 1`` (the candidates are scored against all items)."""
        t, be = self.t, self.backend
        nu = t.user_range[1] - t.user_range[0]
        if self.nnz > 0:
            u, p, cand = be.sample_negatives(self.indptr, self.indices, nu, t.I, self.n_neg, self.seed, 2 * e, self.first)
        else:
            u = torch.empty(0, dtype=torch.int32, device=t.device); p = u.clone()
            cand = torch.empty((0, self.n_neg), dtype=torch.int32, device=t.device)
        if self.n_neg == 1:
            return u, p, cand.reshape(-1)
        Qf, bf = t.full_item_table()
        neg = be.dns_select(t.P, Qf, bf, u, cand) if self.nnz > 0 else u.clone()
        return u, p, neg

    def epoch(self, e):
        """Shuffled (users as GLOBAL ids, items, negatives) of this rank for epoch ``e``."""
        u, p, n = self.draw(e)
        if self.nnz == 0:
            return u, p, n
        perm = self.backend.shuffle_perm(self.nnz, self.seed, (2 * e + 1) + (self.t.rank << 40), self.t.device)
        be = self.backend
        return be.gather_rows(u, perm) + self.t.user_range[0], be.gather_rows(p, perm), be.gather_rows(n, perm)

    def epoch_global(self, e, batch_size):
        """This rank's share of the single-process epoch: (users as GLOBAL ids, items, negatives, bounds) where
        ``bounds[k]:bounds[k+1]`` are this rank's triples of global batch ``k`` (positions ``k*batch_size`` onwards of
        the one permutation of all ``total_nnz`` interactions), in the order they have in that batch.  One host read."""
        t, be, N = self.t, self.backend, self.total_nnz
        n_batches = (N + batch_size - 1) // batch_size
        u, p, n = self.draw(e)
        if N == 0:
            return u, p, n, [0] * (n_batches + 1)
        perm = be.shuffle_perm(N, self.seed, 2 * e + 1, t.device).to(torch.int64)       # position r holds storage triple perm[r]
        rel = perm - self.first
        where = torch.nonzero((rel >= 0) & (rel < self.nnz)).squeeze(1)                  # ascending positions of my triples
        sel = rel[where].to(torch.int32).contiguous()
        edges = torch.arange(0, n_batches + 1, device=where.device, dtype=torch.int64) * batch_size
        bounds = torch.searchsorted(where, edges).tolist()
        if sel.shape[0] == 0:
            return u[:0], p[:0], n[:0], bounds
        return be.gather_rows(u, sel) + t.user_range[0], be.gather_rows(p, sel), be.gather_rows(n, sel), bounds

    def train_epoch(self, e, batch_size):
        """One epoch of collective steps.  ``shuffle="rank"``: every rank feeds ``batch_size`` of its own triples per
        step; ranks that run out keep stepping with empty batches until the longest list is consumed.
        ``shuffle="global"``: the steps are the single-process batches of ``batch_size`` triples, each rank feeding
        the triples of its users."""
        t = self.t
        apr = not hasattr(t, "plan_steps")              # the APR trainer: its step wants the global batch size
        if self.shuffle == "global":
            users, pos, neg, bounds = self.epoch_global(e, batch_size)
            n_steps = len(bounds) - 1
            batches = [(users[bounds[s]:bounds[s + 1]], pos[bounds[s]:bounds[s + 1]], neg[bounds[s]:bounds[s + 1]]) for s in range(n_steps)]
            sizes = [min(batch_size, self.total_nnz - s * batch_size) for s in range(n_steps)]
        else:
            users, pos, neg = self.epoch(e)
            steps = torch.tensor([(self.nnz + batch_size - 1) // batch_size], dtype=torch.int64, device=t.device)
            if t._coll:
                _all_reduce(steps, t.group, op=dist.ReduceOp.MAX)
            n_steps = int(steps.item())
            batches = []
            for s in range(n_steps):
                lo, hi = min(s * batch_size, self.nnz), min((s + 1) * batch_size, self.nnz)
                batches.append((users[lo:hi], pos[lo:hi], neg[lo:hi]))
            sizes = [None] * n_steps
        if apr:
            for b, size in zip(batches, sizes):
                t.step(*b, global_batch=size)
        elif t.item_mode == "sharded":
            # every message size of the epoch is settled here, in one exchange and one host read
            t.run_planned(t.plan_steps(batches))
        else:
            for b in batches:
                t.step(*b)
        return n_steps
