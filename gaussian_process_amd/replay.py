"""One rank of a G-rank DistGP run, alone on ONE GPU -- a measurement tool (bench.py --replay-rank r --of G).

The row-block cyclic schedule of `dist.DistGP` (the reference loop it shards is
GP_regression.py:138-148) needs G GPUs to run.  What a single GPU CAN run is one rank's share of it: its T/G row
blocks, its panel solves, its update launches, its three streams and events, its pack copies and its Python issue
-- everything except the other ranks.  `ReplayComm` is a communicator (the five members DistGP talks to: rank,
size, broadcast, all_gather, all_reduce) for a world of G in which only rank r exists: a collective delivers the
bytes the absent ranks WOULD have sent, by device-to-device copies out of a stored factorisation of the same problem
(`ReplaySource`: the same block primitives on the same inputs, all blocks on this GPU), on the stream the
collective was issued on.  Sizes, streams, events and ordering are a real rank's; only the xGMI transfer time is
replaced by an HBM copy.  T(1 GPU) / max_r T_replay(r) is therefore an UPPER bound of the G-GPU speed-up: what the
compute side, the latency chain and the host allow before communication costs anything.

Not a product path: nothing here is reachable from the drop-in functions.
"""
from __future__ import annotations

import numpy as np
import torch

from .dist import DistGP, YB, block_layout


class SoloComm:
    """a world of one rank: DistGP issues no collective on it (the source factorisation of a replay)"""
    rank, size = 0, 1

    def broadcast(self, t, src, tag=None):
        pass

    def all_gather(self, out, inp, tag=None):
        out.view(-1)[:inp.numel()].copy_(inp.reshape(-1))

    def all_reduce(self, t, op="sum", tag=None):
        pass

    def describe(self):
        return {"backend": "none (one rank)", "ranks": 1, "env": {}}


class ReplaySource:
    """The whole problem factored on this GPU with the block primitives (block rows nb): L row blocks in natural
    order, m = L^-1 y, alpha, v^T -- what the absent ranks' messages are cut from."""

    def __init__(self, device_index, nb, X, y, Xs, sigma, ell, noise_var, ops=None, lookahead=2):
        self.gp = DistGP(device_index, nb=nb, comm=SoloComm(), ops=ops, lookahead=lookahead)
        gp = self.gp
        gp.set_train(X, y)
        self.lml = gp.factorize(sigma, ell, noise_var)
        self.alpha_h = gp.alpha()
        self.have_test = Xs is not None
        if self.have_test:
            gp.set_test(Xs)
            self.mu, self.var = gp.predict_resident(want_sd=False)
        self.NB, self.T, self.Np, self.N = gp.NB, gp.T, gp.Np, gp.N
        self.hyper = (float(sigma), float(ell), float(noise_var))
        dev = gp.dev
        self.alpha = torch.zeros(gp.Np, dtype=torch.float64, device=dev)
        self.alpha[:gp.N].copy_(torch.from_numpy(self.alpha_h).to(dev))
        # per block: log-diagonal sum (column 0), as every rank computes it for its own blocks
        self.logdiag = torch.zeros(gp.T, 2, dtype=torch.float64, device=dev)
        for b in range(gp.T):
            gp.ops.logdiag_sumsq(gp.A[b * gp.NB:(b + 1) * gp.NB, b * gp.NB:(b + 1) * gp.NB], gp.NB, None, 0, self.logdiag[b])
        self.mtm = torch.zeros(2, dtype=torch.float64, device=dev)
        gp.ops.logdiag_sumsq(None, 0, gp.A[gp.yrow], gp.N, self.mtm)
        gp.ops.sync()

    def L3(self):
        """(T, NB, ld) view of the row blocks of L"""
        gp = self.gp
        return gp.A[:gp.T * gp.NB].view(gp.T, gp.NB, gp.ld)


class ReplayComm:
    def __init__(self, rank, size, source: ReplaySource, layout="balanced"):
        if not (0 <= rank < size):
            raise ValueError("rank must be in [0, size)")
        self.rank, self.size, self.src = int(rank), int(size), source
        self.layout = layout
        self._own, self._li, self._blocks = block_layout(source.T, self.size, layout)
        dev = source.gp.dev
        self._bidx = [torch.tensor(b, dtype=torch.int64, device=dev) for b in self._blocks]     # for gathers out of the source
        self._prep = None
        self.bytes = {"bcast": 0, "allgather": 0}      # what the absent ranks delivered, per kind
        # null delivery (experiment, GPMI_REPLAY_NULL=1): the absent ranks' panel columns and diagonal blocks are NOT copied in
        # -- the receive buffers keep whatever they held, results are garbage, and the timing is the rank's own work alone:
        # the difference to a normal replay is what the replay's stand-in copies (15 GB read + written per fit) cost it
        import os
        self.null = os.environ.get("GPMI_REPLAY_NULL") == "1"

    # ---- the block ownership of dist.DistGP, for any rank q
    def _nblocks(self, q):
        return len(self._blocks[q])

    def _lstart(self, k, q):
        import bisect
        return bisect.bisect_right(self._blocks[q], k)

    def _prepare(self):
        """messages that are sums over an absent rank's blocks: computed once, with the same primitives a rank uses"""
        if self._prep is not None:
            return self._prep
        s, G, r = self.src, self.size, self.rank
        gp, ops, NB, T = s.gp, s.gp.ops, s.NB, s.T
        dev = gp.dev
        prep = {}
        # LML pieces (tag "lml"): rank q's (sum of its blocks' log-diagonal sums, m^T m if it carries the y block)
        lml = torch.zeros(G, 2, dtype=torch.float64, device=dev)
        for q in range(G):
            nq = self._nblocks(q)
            if nq:
                mine = s.logdiag.index_select(0, self._bidx[q]).contiguous()      # rank q's blocks, in its local order
                ops.sum_fixed(mine, nq, 2, 1, lml[q, 0:1])
            if q == self._own[T]:
                lml[q, 1:2].copy_(s.mtm[1:2])
        prep["lml"] = lml
        # backward solve (tag "alpha_part", k): for the blocks k this rank owns, the sum of the other ranks' contributions
        # sum_{j > k, j % G != r} L_jk^T alpha_j -- one gemv over the rows below with this rank's own alpha blocks zeroed
        am = s.alpha.clone()
        for b in self._blocks[r]:
            am[b * NB:(b + 1) * NB].zero_()
        others = torch.zeros(max(self._nblocks(r), 1), NB, dtype=torch.float64, device=dev)
        scratch = torch.empty(max((T * NB + 63) // 64, 1) * NB, dtype=torch.float64, device=dev)
        for li, k in enumerate(self._blocks[r]):
            r0 = (k + 1) * NB
            if r0 < T * NB:
                ops.gemv_t(gp.A[r0:T * NB, k * NB:(k + 1) * NB], am[r0:T * NB], others[li], scratch)
        prep["alpha_others"] = others
        # mean / variance partial row dots (tag "dots") of every absent rank over ITS column blocks of v^T
        if s.have_test:
            n_p = gp.n_p
            dots = torch.zeros(G, 2, n_p, dtype=torch.float64, device=dev)
            V3 = gp.V[:, :T * NB].view(n_p, T, NB)
            for q in range(G):
                nq = self._nblocks(q)
                if q == r or nq == 0:
                    continue
                Vq = torch.empty(n_p, nq * NB + 32, dtype=torch.float64, device=dev)
                Vq[:, :nq * NB].view(n_p, nq, NB).copy_(V3.index_select(1, self._bidx[q]))
                mq = torch.empty(nq * NB, dtype=torch.float64, device=dev)
                mq.view(nq, NB).copy_(gp.m[:T * NB].view(T, NB).index_select(0, self._bidx[q]))
                ops.row_dots(Vq, nq * NB, mq, dots[q, 0], dots[q, 1])
                ops.sync()
                del Vq, mq
            prep["dots"] = dots
            # one-pass form (tag "dots_rows"): rank q's own test points' v . m and v . v, from the source's results
            nr = (n_p + G - 1) // G
            nr = (nr + 127) // 128 * 128
            kss = gp._kss_diag()
            rows = torch.zeros(2, G * nr, dtype=torch.float64, device=dev)
            rows[0, :gp.n].copy_(torch.from_numpy(np.ascontiguousarray(s.mu)).to(dev))
            rows[1, :gp.n].copy_(torch.from_numpy(np.ascontiguousarray(kss - s.var)).to(dev))
            prep["dots_rows"] = rows.view(2, G, nr).permute(1, 0, 2).contiguous()
        ops.sync()
        self._prep = prep
        return prep

    # ---- the five members
    def broadcast(self, t, src, tag=None):
        kind = tag[0] if tag else None
        if kind in ("X", "y", "Xs"):
            return                      # replicated inputs: this rank uploaded the same arrays
        if src == self.rank:
            return                      # the owner's buffer already holds what it would send
        s = self.src
        NB = s.NB
        if kind == "Lkk":
            k = tag[1]
            if self.null:
                return
            t.copy_(s.gp.A[k * NB:(k + 1) * NB, k * NB:(k + 1) * NB])
        elif kind == "m":
            t.copy_(s.gp.m)
        elif kind == "vblock":
            k = tag[1]
            if self.null:
                return
            t.copy_(s.gp.V[:, k * NB:(k + 1) * NB])
        else:
            raise NotImplementedError("ReplayComm.broadcast: no replay for %r" % (tag,))
        self.bytes["bcast"] += t.numel() * 8

    def all_gather(self, out, inp, tag=None):
        kind = tag[0] if tag else None
        s, G, r = self.src, self.size, self.rank
        NB, T = s.NB, s.T
        n = inp.numel()
        flat = out.view(-1)
        if kind == "panel":
            k = tag[1]
            cmax = n // (NB * NB)
            O = flat[:G * cmax * NB * NB].view(G, cmax, NB, NB)
            L3 = s.L3()
            for q in range(G):
                cnt = self._nblocks(q) - self._lstart(k, q)
                if cnt <= 0:
                    continue
                if q == r:
                    O[q, :cnt].copy_(inp.view(cmax, NB, NB)[:cnt])
                elif not self.null:
                    ls = self._lstart(k, q)
                    torch.index_select(L3[:, :, k * NB:(k + 1) * NB], 0, self._bidx[q][ls:ls + cnt], out=O[q, :cnt])
                    self.bytes["allgather"] += cnt * NB * NB * 8
            return
        prep = self._prepare()
        if kind == "lml":
            O = flat[:G * 2].view(G, 2)
            O.copy_(prep["lml"])
            O[r].copy_(inp.reshape(-1))
        elif kind == "dots":
            n_p = n // 2
            O = flat[:G * n].view(G, 2, n_p)
            O.copy_(prep["dots"])
            O[r].copy_(inp.view(2, n_p))
        elif kind == "dots_rows":
            nr = n // 2
            O = flat[:G * n].view(G, 2, nr)
            O.copy_(prep["dots_rows"])
            O[r].copy_(inp.view(2, nr))
        elif kind == "alpha_part":
            k = tag[1]
            O = flat[:G * NB].view(G, NB)
            O.zero_()
            O[r].copy_(inp)
            if self._own[k] == r and G > 1:       # only the owner of block k reads the sum: the others' share goes into one slot
                O[(r + 1) % G].copy_(prep["alpha_others"][self._li[k]])
        elif kind == "alpha_full":
            cmax = n // NB
            O = flat[:G * cmax * NB].view(G, cmax, NB)
            A2 = s.alpha[:T * NB].view(T, NB)
            for q in range(G):
                nq = self._nblocks(q)
                if q == r:
                    O[q].copy_(inp.view(cmax, NB))
                elif nq:
                    torch.index_select(A2, 0, self._bidx[q], out=O[q, :nq])
        else:
            raise NotImplementedError("ReplayComm.all_gather: no replay for %r" % (tag,))

    def all_reduce(self, t, op="sum", tag=None):
        kind = tag[0] if tag else None
        if kind in ("info", "alpha_err"):
            return                      # the absent ranks report no failed pivot and no given-up wait
        raise NotImplementedError("ReplayComm.all_reduce: no replay for %r" % (tag,))

    def describe(self):
        return {"backend": "replay (rank %d of %d alone on one GPU; collectives = device copies out of a stored factorisation)"
                           % (self.rank, self.size), "ranks": self.size, "env": {}}


def replay_rank(device_index, source: ReplaySource, rank, size, X, y, Xs, lookahead=2, ops=None, layout="balanced"):
    """DistGP of rank `rank` in a world of `size`, its collectives served from `source`; train / test sets resident"""
    gp = DistGP(device_index, nb=source.NB, comm=ReplayComm(rank, size, source, layout), lookahead=lookahead, ops=ops,
                layout=layout)
    gp.set_train(X, y)
    if Xs is not None:
        gp.set_test(Xs)
    return gp


def check_rank(gp, source: ReplaySource):
    """max deviations of a replayed rank's results from the source factorisation it was fed from: its rows of L (lower
    triangle, relative to the largest entry), the y block's m, and -- if present -- alpha"""
    s = source
    NB, G, r = s.NB, gp.G, gp.rank
    out = {"L_rel": 0.0}
    scale = 0.0                     # block row by block row: the source can be most of the memory, no temporary of its size
    for b in range(s.T):            # ... and the lower triangle only: above it the buffer holds whatever the allocation held
        scale = max(scale, float(torch.tril(s.gp.A[b * NB:(b + 1) * NB, :(b + 1) * NB], diagonal=b * NB).abs().max().item()))
    for li, b in enumerate(gp.my_blocks):
        mine = torch.tril(gp.A[li * NB:(li + 1) * NB, :(b + 1) * NB], diagonal=b * NB)
        ref = torch.tril(s.gp.A[b * NB:(b + 1) * NB, :(b + 1) * NB], diagonal=b * NB)
        out["L_rel"] = max(out["L_rel"], float((mine - ref).abs().max().item()) / scale)
    if gp.yrow is not None:
        ms = float(s.gp.m.abs().max().item())
        out["m_rel"] = float((gp.A[gp.yrow, :s.Np] - s.gp.A[s.gp.yrow, :s.Np]).abs().max().item()) / ms
    return out


__all__ = ["SoloComm", "ReplaySource", "ReplayComm", "replay_rank", "check_rank", "YB"]
