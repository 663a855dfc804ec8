"""Multi-GPU GP fit + predict: the N x N covariance row-block partitioned (block-cyclic)
over the ranks of one node, one process per GPU, collectives through
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

Reference path: GP_regression.py:126-148 / tune_hyperparms_regression.py:306-312 (the
reference itself is single-process NumPy; SURVEY.md section 8e is the design this follows).

Layout.  NB-row blocks of K + sI; global block b lives on rank owner[b] (block_layout: the blocks
dealt one per rank in every group of G, in the order that evens out the ranks' shares of the update -- "balanced", the
default; boustrophedon with layout="snake", plain b % G with layout="cyclic"), stacked in increasing b in that rank's local matrix A (rows) x (Np + pad)
-- row-major, so the rank holds whole rows of L.  One extra 128-row block carries y (owner[T]): the
factorisation sweeps it like any other row block, so it ends as m = L^-1 y.

Right-looking step k (block column k):
  owner: Cholesky of the diagonal block -> L_kk         [gpmi_dev_potrf_block]
  broadcast L_kk (NB x NB)                                [RCCL broadcast]
  every rank: its rows below k  <-  rows * L_kk^-T        [gpmi_dev_trsm_block]
  all-gather of the panel column (each rank's rows)       [RCCL all_gather]
  every rank: trailing update of its own rows, one launch  [gpmi_dev_gemm_nt_rowmap]
With lookahead (DistGP(lookahead=...)) the update is split so that block column k+1 is ready
first and the next panel step overlaps the rest of the update; level 2 (default) also takes
the Cholesky of diagonal block k+1 off the collective chain (_factor_critical_path_first).
Predict: v^T = K_s^T L^-T with the COLUMN blocks of v^T distributed like the row
blocks of L; per step the owner solves its block, broadcasts it, and every rank
updates its own column blocks; mean / variance are ordered sums of per-rank partial
row dots.  No collective carries more than the panel column / one solved block.

torch is used for device memory, strided copies, streams and torch.distributed only;
all arithmetic -- the three small reductions over gathered partials included (gpmi_dev_sum_fixed,
gpmi_dev_axpy2d) -- goes through the C-ABI block primitives (`HipBlockOps`).  The driver
takes the primitives as an object so that the CPU tests (gloo, world_size 2) can run
the same schedule on NumPy stand-ins defined under tests/.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check

YB = 128                  # rows of the y block (one tile)
_SHARED_STREAMS = {}      # (device index, name, host thread) -> torch.cuda.Stream: see DistGP._named_stream
INT64_MAX = (1 << 63) - 1


def _round_up(x, m):
    return (x + m - 1) // m * m


LAYOUTS = ("cyclic", "snake", "balanced")


def block_layout(T, G, layout):
    """Which rank owns row block b (b = 0 .. T; block T is the y block), for the two row-block distributions:
      "cyclic"  b -> b % G.
      "snake"   the same dealt boustrophedon: blocks 0 .. G-1 to ranks 0 .. G-1, blocks G .. 2G-1 to ranks G-1 .. 0, and so on.
      "balanced" each group of G blocks dealt in the order that evens out the ranks' shares (below).
    Over the factorisation row block b is updated by b (b + 1) / 2 block products (step k < b touches its columns k + 1 .. b),
    so a rank's share of the work is the sum of that over its blocks, and cyclic dealing gives the last rank much more of it
    than the first: N = 65536, nb = 1024, 8 ranks -- rank 7 (blocks 7, 15, .. 63) carries 6384 units, rank 0 4592, the mean
    is 5460, and the step ends when rank 7 does (replay: 277 ms against 225).  The snake's worst rank carries 5600 there
    (+2.6 % over the mean instead of +16.9 %).  Either way at most one block per rank separates the ranks' row counts at
    any step, so the per-step balance of the panel solves is the cyclic one.
    Returns (owner[b] for b <= T, local index li[b], blocks[r] = that rank's blocks < T in increasing order)."""
    if layout not in LAYOUTS:
        raise ValueError("layout must be one of %s" % (LAYOUTS,))
    if layout == "balanced":
        # every group of G consecutive blocks is still dealt one block per rank (so the property above holds), but in the
        # order that evens the shares out: groups from the heaviest (last) down, the group's heaviest block to the rank
        # that carries least so far.  64 blocks, 8 ranks: worst share +0.4 % over the mean (snake +2.6 %)
        own = [0] * (T + 1)
        load = [0] * G
        for g0 in range((T // G) * G, -1, -G):
            grp = list(range(min(T, g0 + G - 1), g0 - 1, -1))            # heaviest first
            for b, r in zip(grp, sorted(range(G), key=lambda r: (load[r], r))):
                own[b] = r
                load[r] += b * (b + 1) // 2 if b < T else 0
    else:
        own = []
        for b in range(T + 1):
            q, i = divmod(b, G)
            own.append(G - 1 - i if (layout == "snake" and q % 2) else i)
    blocks = [[b for b in range(T) if own[b] == r] for r in range(G)]
    li = [0] * (T + 1)
    for r in range(G):
        for j, b in enumerate(blocks[r]):
            li[b] = j
    li[T] = len(blocks[own[T]])
    return own, li, blocks


class TorchComm:
    """The collectives the schedule issues, on a torch.distributed process group ("nccl" = RCCL on ROCm; gloo in
    the CPU tests).  DistGP only ever talks to an object with this interface (rank, size, broadcast, all_gather,
    all_reduce), so a test can hand it an in-process stand-in and run G ranks as G threads of one process."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("DistGP needs torch.distributed (init_process_group) -- one rank per GPU")
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def _src(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    # `tag` names what a collective carries -- ("panel", k), ("Lkk", k), ("vblock", k), ... -- and is ignored here: a
    # stand-in communicator (replay.ReplayComm: one rank of G alone on a GPU) uses it to know which bytes the absent
    # ranks would have delivered
    def broadcast(self, t, src, tag=None):
        dist.broadcast(t, src=self._src(src), group=self.group)

    def all_gather(self, out, inp, tag=None):
        """out (size * len(inp)) <- every rank's inp, in rank order"""
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce(self, t, op="sum", tag=None):
        ops = {"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}
        dist.all_reduce(t, op=ops[op], group=self.group)

    def describe(self):
        """the backend and the RCCL / NCCL knobs in effect (bench line)"""
        import os
        knobs = {k: v for k, v in os.environ.items()
                 if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC", "HSA_FORCE_FINE_GRAIN", "TORCH_NCCL_"))}
        return {"backend": dist.get_backend(self.group), "ranks": self.size, "env": knobs}


class RcclComm:
    """The same five members straight on RCCL through the C-ABI (gpmi_comm_*: libgpmi355x.so opens librccl itself and
    issues every collective on the CALLER'S stream -- no process-group layer, no internal communication stream).
    north_star: "a thin C-ABI ... RCCL broadcast/all-gather of panel columns over xGMI".

    A communicator must not run two collectives at once, and the schedule issues collectives from three streams (main,
    side, crit) so that a broadcast of the next diagonal block does not queue behind the all-gather of the current panel:
    there is one communicator per stream, handed out in the order the streams first call in -- the same order on every
    rank, since every rank runs the same program.  torch.distributed (any backend; gloo will do) is only the side
    channel that carries rank 0's 128-byte ncclUniqueIds to the others; a world of one rank needs none."""

    NCOMM = 3

    def __init__(self, device_index, rank=None, size=None, group=None):
        self.lib = _lib.load()
        if rank is None:
            if dist.is_initialized():
                rank, size = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, size = 0, 1
        self.rank, self.size, self.group = int(rank), int(size), group
        self.device_index = int(device_index)
        if self.size > 1 and not dist.is_initialized():
            raise RuntimeError("RcclComm over more than one rank needs torch.distributed (any backend) for the id exchange")
        self._comms = []
        for i in range(self.NCOMM):
            idb = C.create_string_buffer(128)
            if self.rank == 0:
                check(self.lib.gpmi_comm_unique_id(idb))
            if self.size > 1:
                box = [idb.raw]
                dist.broadcast_object_list(box, src=self._src(0), group=group)
                idb = C.create_string_buffer(box[0], 128)
            h = C.c_void_p()
            check(self.lib.gpmi_comm_create(idb, self.rank, self.size, self.device_index, C.byref(h)))
            self._comms.append(h)
        self._by_stream = {}

    def _src(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def close(self):
        for h in self._comms:
            self.lib.gpmi_comm_destroy(h)
        self._comms = []

    def _where(self):
        """(communicator of the current stream, the stream)"""
        st = torch.cuda.current_stream().cuda_stream
        idx = self._by_stream.setdefault(st, min(len(self._by_stream), self.NCOMM - 1))
        return self._comms[idx], C.c_void_p(st)

    @staticmethod
    def _flat(t):
        if not t.is_contiguous():
            raise ValueError("RcclComm: collectives take contiguous tensors")
        return C.c_void_p(t.data_ptr()), t.numel() * t.element_size()

    def broadcast(self, t, src, tag=None):
        comm, st = self._where()
        p, nbytes = self._flat(t)
        check(self.lib.gpmi_comm_broadcast(comm, st, p, nbytes, int(src)))

    def all_gather(self, out, inp, tag=None):
        comm, st = self._where()
        pi, nb_in = self._flat(inp)
        po, nb_out = self._flat(out)
        if nb_out < nb_in * self.size:
            raise ValueError("RcclComm.all_gather: the receive buffer is too small")
        check(self.lib.gpmi_comm_all_gather(comm, st, pi, po, nb_in))

    def all_reduce(self, t, op="sum", tag=None):
        comm, st = self._where()
        p, _ = self._flat(t)
        dt = {torch.float64: 0, torch.int64: 1}[t.dtype]
        check(self.lib.gpmi_comm_all_reduce(comm, st, p, t.numel(), dt, {"sum": 0, "min": 1, "max": 2}[op]))

    def describe(self):
        import os
        buf, ver = C.create_string_buffer(512), C.c_int()
        self.lib.gpmi_comm_library(buf, 512, C.byref(ver))
        knobs = {k: v for k, v in os.environ.items()
                 if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC", "HSA_FORCE_FINE_GRAIN"))}
        return {"backend": "rccl through the C-ABI (gpmi_comm_*), one communicator per stream", "ranks": self.size,
                "library": buf.value.decode("utf-8", "replace"), "nccl_version": ver.value, "env": knobs}


class HipBlockOps:
    """Block primitives on torch CUDA tensors through libgpmi355x.so (gpmi_dev_*).
    Views must be float64 with unit column stride; ld = view.stride(0)."""

    def __init__(self, device_index):
        self.lib = _lib.load()
        self.device = torch.device("cuda", device_index)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def _ld(t):
        assert t.dtype == torch.float64 and (t.dim() == 1 or t.stride(-1) == 1)
        return t.stride(0) if t.dim() == 2 else t.shape[0]

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def rbf_rows(self, X, N, d, row0, nrows, ncols, sigma, ell, noise_var, out):
        check(self.lib.gpmi_dev_rbf_rows(self._stream(), self._p(X), N, d, row0, nrows, ncols,
                                         float(sigma), float(ell), float(noise_var), self._p(out), self._ld(out)))

    def rbf_cross(self, Xs, n, Xcols, ncols_real, d, nrows, ncols, sigma, ell, out):
        """out[i][j] = k(Xs[i], Xcols[j]); rows >= n and cols >= ncols_real are zero."""
        check(self.lib.gpmi_dev_rbf_cross(self._stream(), self._p(Xs), n, self._p(Xcols), max(ncols_real, 0), d,
                                          0, nrows, ncols, float(sigma), float(ell), self._p(out), self._ld(out)))

    def cov_rows(self, kind, params, X, N, d, row0, nrows, ncols, noise_var, out):
        """rbf_rows for any covariance function: kind 0 rbf (sigma, l), 1 lin (c), 2 per (p, l), 3 CO2 composite (11)"""
        pr = np.ascontiguousarray(params, dtype=np.float64)
        check(self.lib.gpmi_dev_cov_rows(self._stream(), int(kind), pr.ctypes.data_as(C.POINTER(C.c_double)), pr.shape[0],
                                         self._p(X), N, d, row0, nrows, ncols, float(noise_var), self._p(out), self._ld(out)))

    def cov_cross(self, kind, params, Xs, n, Xcols, ncols_real, d, col0, square, nrows, ncols, out):
        """rbf_cross likewise; Xcols is a window of the column inputs starting at input col0, square: n == N"""
        pr = np.ascontiguousarray(params, dtype=np.float64)
        check(self.lib.gpmi_dev_cov_cross(self._stream(), int(kind), pr.ctypes.data_as(C.POINTER(C.c_double)), pr.shape[0],
                                          self._p(Xs), n, self._p(Xcols), max(ncols_real, 0), d, int(col0), 1 if square else 0,
                                          nrows, ncols, self._p(out), self._ld(out)))

    def potrf_block(self, A, col_offset, info):
        check(self.lib.gpmi_dev_potrf_block(self._stream(), self._p(A), self._ld(A), A.shape[0], col_offset,
                                            self._p(info)))

    def trsm_block(self, L, X):
        check(self.lib.gpmi_dev_trsm_block(self._stream(), self._p(L), self._ld(L), self._p(X), self._ld(X),
                                           X.shape[0], X.shape[1]))

    def gemm_nt(self, Cm, A, B):
        check(self.lib.gpmi_dev_gemm_nt(self._stream(), self._p(Cm), self._ld(Cm), self._p(A), self._ld(A),
                                        self._p(B), self._ld(B), Cm.shape[0], Cm.shape[1], A.shape[1], 0, 0))

    def gemm_nt_rowmap(self, Cm, A, B, row_ncols, row_block_rows, row_ncols_host=None):
        """row_ncols_host: the same map as a contiguous int32 NumPy array -- only the supertiles with live
        tiles are then launched"""
        if row_ncols_host is not None:
            check(self.lib.gpmi_dev_gemm_nt_rowmap_host(
                self._stream(), self._p(Cm), self._ld(Cm), self._p(A), self._ld(A), self._p(B), self._ld(B),
                Cm.shape[0], Cm.shape[1], A.shape[1], C.c_void_p(row_ncols.data_ptr()),
                C.c_void_p(row_ncols_host.ctypes.data), row_ncols_host.shape[0], row_block_rows))
            return
        check(self.lib.gpmi_dev_gemm_nt_rowmap(self._stream(), self._p(Cm), self._ld(Cm), self._p(A), self._ld(A),
                                               self._p(B), self._ld(B), Cm.shape[0], Cm.shape[1], A.shape[1],
                                               C.c_void_p(row_ncols.data_ptr()), row_block_rows))

    def gemm_nt_blocks(self, Cm, A, Bflat, ldb, boff, brows, row_ncols=None, row_block_rows=128, row_ncols_host=None):
        """Cm -= A * B^T with B = the row blocks (brows x K, leading dimension ldb) at Bflat[boff[i]:] -- the
        all-gather's receive buffer read in natural block order; optional row map as in gemm_nt_rowmap"""
        assert boff.dtype == torch.int64 and boff.is_contiguous() and boff.shape[0] * brows >= Cm.shape[1]
        check(self.lib.gpmi_dev_gemm_nt_blocks(
            self._stream(), self._p(Cm), self._ld(Cm), self._p(A), self._ld(A), self._p(Bflat), int(ldb),
            C.c_void_p(boff.data_ptr()), int(brows), Cm.shape[0], Cm.shape[1], A.shape[1],
            C.c_void_p(row_ncols.data_ptr()) if row_ncols is not None else None,
            C.c_void_p(row_ncols_host.ctypes.data) if row_ncols_host is not None else None,
            row_ncols_host.shape[0] if row_ncols_host is not None else 0, row_block_rows))

    def logdiag_sumsq(self, A, n, x, nx, out2):
        check(self.lib.gpmi_dev_logdiag_sumsq(self._stream(), self._p(A) if A is not None else None,
                                              self._ld(A) if A is not None else 0, n,
                                              self._p(x) if x is not None else None, nx, self._p(out2)))

    def row_dots(self, V, ncols, m, dot, sq):
        check(self.lib.gpmi_dev_row_dots(self._stream(), self._p(V), self._ld(V), V.shape[0], ncols,
                                         self._p(m), self._p(dot), self._p(sq)))

    def gemv_t(self, A, x, y, scratch):
        check(self.lib.gpmi_dev_gemv_t(self._stream(), self._p(A) if A is not None else None,
                                       self._ld(A) if A is not None else 0,
                                       A.shape[0] if A is not None else 0, y.shape[0],
                                       self._p(x) if x is not None else None, self._p(y), self._p(scratch)))

    def trsv_lt(self, L, b, inverted=False, vside=None):
        """L^T x = b, x overwrites b (L as potrf_block leaves it).  The first call on a factored block writes the
        inverses of its 128 x 128 diagonal blocks into their upper triangles; inverted=True says they are there.
        vside (n * 128 doubles, kept by the caller with the block): the one-launch form (gpmi_dev_trsv_lt_chain)."""
        n = b.shape[0]
        if n % 128 == 0 and vside is not None:
            x = torch.empty_like(b)
            if getattr(self, "_err", None) is None:
                self._err = torch.zeros(16, dtype=torch.int32, device=self.device)
            check(self.lib.gpmi_dev_trsv_lt_chain(self._stream(), self._p(L), self._ld(L), self._p(vside), self._p(b), self._p(x),
                                                  n, 0 if inverted else 1, self._p(self._err)))
            b.copy_(x)
            return
        if n % 128 == 0:
            x = torch.empty_like(b)
            check(self.lib.gpmi_dev_trsv_lt_vinv(self._stream(), self._p(L), self._ld(L), self._p(b), self._p(x), n,
                                                 0 if inverted else 1))
            b.copy_(x)
            return
        check(self.lib.gpmi_dev_trsv_lt(self._stream(), self._p(L), self._ld(L), self._p(b), n))

    def grad_trace(self, X, N, d, row0, nrows, alpha_r, alpha_c, Kinv, kinv_sign, sigma, ell, partial, out2):
        """out2[0:2] += sum_ij (alpha_r[i] alpha_c[j] - kinv_sign * Kinv[i - row0][j]) * (dK_ij/dl, dK_ij/dsigma)
        over rows row0 .. row0 + nrows and all N columns (tune_hyperparms_regression.py:43-57)"""
        check(self.lib.gpmi_dev_grad_trace(self._stream(), self._p(X), N, d, row0, nrows, self._p(alpha_r), self._p(alpha_c),
                                           self._p(Kinv), self._ld(Kinv), float(kinv_sign), float(sigma), float(ell),
                                           self._p(partial), self._p(out2)))

    def sum_fixed(self, inp, count, stride, n, out, base=None, scale=1.0):
        """out[i] = (base[i] if base is not None else 0) + scale * sum_{q < count} inp[q * stride + i], added in index order"""
        check(self.lib.gpmi_dev_sum_fixed(self._stream(), self._p(inp), int(count), int(stride), int(n),
                                          self._p(base) if base is not None else None, float(scale), self._p(out)))

    def axpy2d(self, Y, X, a):
        """Y += a * X (2-D views of equal shape)"""
        assert Y.shape == X.shape and Y.dim() == 2
        check(self.lib.gpmi_dev_axpy2d(self._stream(), self._p(Y), self._ld(Y), self._p(X), self._ld(X), Y.shape[0], Y.shape[1],
                                       float(a)))

    def set_concurrent(self, on):
        check(self.lib.gpmi_dev_set_concurrent(1 if on else 0))

    def set_option(self, name, value):
        """kernel-selection option of the block primitives called from this thread (gpmi_dev_set_option)"""
        check(self.lib.gpmi_dev_set_option(name.encode(), int(value)))

    def sync(self):
        torch.cuda.synchronize(self.device)


class _Timed:
    """events (CUDA) or host seconds (CPU stand-ins) around a piece of the schedule; see DistGP.profile"""

    def __init__(self, gp, key, stream):
        self.gp, self.key, self.stream = gp, key, stream

    def _mark(self):
        import time
        if not self.gp._cuda():
            return time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.gp.dev) if self.stream is None else self.gp._stream_of(self.stream))
        return ev

    def __enter__(self):
        self.a = self._mark()

    def __exit__(self, *exc):
        if self.gp._prof is not None:
            self.gp._prof.setdefault(self.key, []).append((self.a, self._mark()))


class DistGP:
    """Row-block cyclic GP fit / predict over the ranks of `group` (default: WORLD)."""

    def __init__(self, device_index=0, nb=512, ld_pad=32, ops=None, group=None, lookahead=2,
                 force_collectives=False, comm=None, layout=None):
        if nb <= 0 or nb % 128:
            raise ValueError("nb must be a positive multiple of 128")
        import os
        # how the row blocks are dealt to the ranks (block_layout): "balanced" evens out the ranks' shares of the update
        self.layout = layout if layout is not None else os.environ.get("GPMI_DIST_LAYOUT", "balanced")
        if self.layout not in LAYOUTS:
            raise ValueError("layout must be one of %s" % (LAYOUTS,))
        # collectives: an object with TorchComm's five members; default by $GPMI_DIST_COMM: "torch" (RCCL through
        # torch.distributed's process group) or "rccl" (RCCL through this library's own C-ABI, RcclComm)
        if comm is None:
            comm = RcclComm(device_index, group=group) if os.environ.get("GPMI_DIST_COMM", "torch") == "rccl" else TorchComm(group)
        self.comm = comm
        self.group = group
        self.rank = self.comm.rank
        self.G = self.comm.size
        # a world of one rank needs no collective; force_collectives issues them anyway (every broadcast,
        # all-gather and all-reduce of the multi-rank schedule, on a communicator of size 1) so that the
        # RCCL call path can be exercised on a single GPU
        self.coll = self.G >= 2 or bool(force_collectives)
        self.ops = ops if ops is not None else HipBlockOps(device_index)
        self.dev = self.ops.device
        self.NB = int(nb)
        self.ld_pad = int(ld_pad) // 2 * 2
        # 0: none; 1: panel k+1 (diagonal block, broadcast, solves, all-gather) behind the trailing
        # update of step k; 2: additionally "critical path first" -- the owner of diagonal block k+1
        # solves and updates its own block row and factors the block on a third stream before the
        # rest of panel k is gathered, so the latency-bound potrf leaves the collective chain
        self.lookahead = int(lookahead)
        # large update launches in the ticket form of the GEMM (resident workgroups that draw tiles from per-XCD counters and
        # take over the other XCDs' tails -- a rank's staircase of row blocks is dealt unevenly to the XCDs): on wherever no
        # kernel of another stream needs a whole CU meanwhile, i.e. not while this rank factors a diagonal block
        # -- measured on the 8-rank replay: the update itself runs 5-10 % faster, the panel solves that share the CUs with
        # the resident workgroups 2-3x slower, the step not faster (profiles/r04_replay_ticket_ab.txt): off by default.  With
        # real RCCL it may be the other way round: a per-tile launch loses 12 % for every shader engine in which another
        # kernel holds a CU exclusively (LAB_NOTES.md), the ticket form only the CU's share; bench.py measures both forms
        # before its timed region and takes the faster (self.ticket may be switched between steps)
        _tk = os.environ.get("GPMI_DIST_TICKET", "0")
        self.ticket = int(_tk) if _tk.lstrip("-").isdigit() else 0      # "auto" (bench.py decides by measurement) starts per-tile
        if os.environ.get("GPMI_DIST_BALANCE") == "0" and hasattr(self.ops, "set_option"):     # A/B switch of the XCD-balanced launch geometry
            self.ops.set_option("gemm_balance", 0)
        self.have_factor = False
        self._vinv_blocks = set()    # local diagonal blocks whose 128 x 128 inverses are in place (backward solve)
        self._vside = {}             # local diagonal block -> its inverses' side buffer (one-launch backward solve)
        self._kind = None            # covariance function other than the squared exponential: (kind number, parameters)
        self.have_test = False
        self.stage_ms = {}
        self._prof = None            # profile(True): {key: [(start, end) events or (t0, t1) seconds]}

    # ------------------------------------------------------------------ self-diagnosis (bench line)
    def profile(self, on=True):
        """Switch per-kind timing of the schedule on or off.  While on, every collective, every update launch, the
        panel solves and the diagonal-block factorisations are bracketed by events on the stream they run on, the
        main stream's wait for the panel of each step is bracketed too (that elapsed time IS the stall), and the host
        clocks each step's issue.  A handful of event records per step: meant for one diagnostic step outside the
        timed region (bench.py), not for the timed steps."""
        self._prof = {} if on else None

    def _timed(self, key, stream="main"):
        """context manager: events around the enclosed work on the named stream (seconds on the host for CPU ops)"""
        import contextlib
        if self._prof is None:
            return contextlib.nullcontext()
        return _Timed(self, key, stream)

    def profile_summary(self):
        """{key: {"ms": summed elapsed, "n": count, "max_ms": largest}} after a synchronisation; keys: update (every
        trailing-update launch of the fit), allgather, pack, bcast, panel_solve, diag, stall_panel (main stream
        waiting for the panel chain), update_v / bcast_v / solve_v (predict sweep), alpha_gather / alpha_gemv /
        alpha_solve, host_issue (host seconds per step of the fit loop, reported in ms)."""
        if self._prof is None:
            return {}
        if self._cuda():
            torch.cuda.synchronize(self.dev)
        out = {}
        for key, pairs in self._prof.items():
            vals = [(a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3) for a, b in pairs]
            out[key] = {"ms": float(sum(vals)), "n": len(vals), "max_ms": float(max(vals)) if vals else 0.0}
        return out

    def _lstart(self, k, r=None):
        """first local block index of rank r whose global block index is > k"""
        import bisect
        r = self.rank if r is None else r
        return bisect.bisect_right(self._blocks[r], k)

    def _nblocks(self, r):
        return len(self._blocks[r])

    def _tensor(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    # ------------------------------------------------------------------ covariance function (f4)
    KINDS = {"rbf": 0, "lin": 1, "per": 2, "co2": 3}

    def set_kernel(self, kind, p0=0.0, p1=0.0):
        """Covariance function of the following fits (kernel_choice of prediction(), GP_regression.py:125-136; 'co2':
        CO2_example.py:66-90 with p0 = its 11 hyper-parameters) -- the same choices as GPContext.set_kernel.  'rbf' takes
        its sigma and l from factorize()."""
        if kind not in self.KINDS:
            raise ValueError("kernel must be 'rbf', 'lin', 'per' or 'co2', got %r" % (kind,))
        if kind == "rbf":
            new = None
        elif kind == "lin":
            new = (1, np.array([float(p0)]))
        elif kind == "per":
            if not (float(p0) != 0.0 and float(p1) != 0.0):
                raise ValueError("period and lengthscale must be non-zero")
            new = (2, np.array([float(p0), float(p1)]))
        else:
            th = np.ascontiguousarray(np.asarray(p0, dtype=np.float64).reshape(-1))
            if th.shape[0] != 11:
                raise ValueError("the CO2 composite kernel takes 11 hyper-parameters")
            new = (3, th)
        same = (new is None and self._kind is None) or (
            new is not None and self._kind is not None and new[0] == self._kind[0] and np.array_equal(new[1], self._kind[1]))
        self._kind = new
        if not same:                      # a resident factor belongs to the covariance function it was built with
            self.have_factor = False
            self.have_v = False

    def _cov_rows(self, X, N, row0, nrows, ncols, noise_var, out):
        if self._kind is None:
            self.ops.rbf_rows(X, N, self.d, row0, nrows, ncols, self.sigma, self.ell, noise_var, out)
        else:
            self.ops.cov_rows(self._kind[0], self._kind[1], X, N, self.d, row0, nrows, ncols, noise_var, out)

    def _cov_cross(self, Xs, n, Xcols, ncols_real, col0, nrows, ncols, out):
        if self._kind is None:
            self.ops.rbf_cross(Xs, n, Xcols, ncols_real, self.d, nrows, ncols, self.sigma, self.ell, out)
        else:
            self.ops.cov_cross(self._kind[0], self._kind[1], Xs, n, Xcols, ncols_real, self.d, col0, n == self.N, nrows,
                               ncols, out)

    def _kss_diag(self):
        """diag(K_ss) of GP_regression.py:147 without building K_ss"""
        if self._kind is None:
            return self.sigma ** 2
        kind, pr = self._kind
        if kind == 2:
            return 1.0                                        # exp(0)
        if kind == 3:                                         # every factor is 1 at distance 0; K_ss is square: + theta_11^2
            return ((pr[0] * pr[0] + pr[2] * pr[2]) + pr[5] * pr[5]) + (pr[8] * pr[8] + pr[10] * pr[10])
        Xs = self.Xs.cpu().numpy()
        kss = np.zeros(self.n)
        for k in range(self.d):                               # (x - c).(x - c), summed in input order
            e = Xs[:, k] - pr[0]
            kss = kss + e * e
        return kss

    # ------------------------------------------------------------------ data
    def set_train(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        if X.ndim != 2 or y.shape[0] != X.shape[0]:
            raise ValueError("X_train must be (N, d) and y_train (N,)")
        self.N, self.d = X.shape
        NB, G = self.NB, self.G
        self.Np = _round_up(self.N, NB)
        self.T = self.Np // NB
        self._own, self._li, self._blocks = block_layout(self.T, G, self.layout)
        self.my_blocks = list(self._blocks[self.rank])
        self.nloc = len(self.my_blocks)
        self.ry = self._own[self.T]                  # rank that carries the y block
        self.yrow = self.nloc * NB if self.rank == self.ry else None
        # rows_base: my blocks (+ the y rows); `rows` = rows_base + the test rows that ride along in the one-pass form
        # (fit_predict_resident), 0 of them otherwise -- _set_tail
        self.rows_base = self.nloc * NB + (YB if self.rank == self.ry else 0)
        self.rows = self.rows_base
        self._tail_cache = {}
        self.ld = self.Np + self.ld_pad
        # X, y replicated (N*(d+1)*8 bytes); broadcast from rank 0 so every rank factors the same data
        self.X = torch.from_numpy(X).to(self.dev)
        self.y = torch.from_numpy(y).to(self.dev)
        if self.coll:
            self.comm.broadcast(self.X, 0, tag=("X",))
            self.comm.broadcast(self.y, 0, tag=("y",))
        self.A = self._tensor(max(self.rows, 1), self.ld)
        self.Lkk = self._tensor(NB, NB)
        cmax = max(self._nblocks(r) for r in range(G))
        self.send = self._tensor(max(cmax, 1) * NB * NB)
        self.Lk = [self.Lkk, self._tensor(NB, NB)]
        # The panel column of a step lives where the all-gather delivers it: one contiguous chunk per rank
        # (rank r's solved blocks below k, in its local order).  Nothing re-orders it: the update kernels read
        # the blocks in natural order through a per-step offset table (block k+1+i -> chunk of rank
        # (k+1+i) % G, position among that rank's blocks below k).  Two buffers: panel k+1 is gathered while
        # the update with panel k still runs (lookahead).
        self.Pbuf = [self._tensor(G * max(cmax, 1) * NB * NB) for _ in range(2 if self.lookahead else 1)] \
            if self.coll else [None, None]
        offs, starts = [], []
        for k in range(max(self.T - 1, 0)):
            cnts = [self._nblocks(r) - self._lstart(k, r) for r in range(G)]
            ck = max(cnts)
            starts.append(len(offs))
            for b in range(k + 1, self.T):
                r = self._own[b]
                offs.append((r * ck + self._li[b] - self._lstart(k, r)) * NB * NB)
        self.boff_h = np.ascontiguousarray(offs or [0], dtype=np.int64)
        self.boff = torch.from_numpy(self.boff_h).to(self.dev)
        self.boff_start = starts
        self.info = torch.full((1,), INT64_MAX, dtype=torch.int64, device=self.dev)
        self.red = self._tensor(max(self.nloc, 1) + 1, 2)
        self.m = self._tensor(self.Np)
        self._set_tail(0)
        self.have_factor = False
        self.have_test = False

    def _build_rowmaps(self, vrows):
        """the row maps of my update launches when `vrows` full-width rows (test rows riding along) follow my blocks
        (and the y rows): per step, per 128-row band, the number of columns the band updates"""
        NB = self.NB
        tail_bands = (YB // 128 if self.rank == self.ry else 0) + vrows // 128
        tb = {}
        # per step: number of columns each 128-row band of my rows below block k updates (the part of the
        # trailing matrix on or below the diagonal, to the 128-column tile)
        bands = NB // 128
        tabs, offs = [], []
        for k in range(self.T - 1):
            ls = self._lstart(k)
            offs.append(sum(len(t) for t in tabs))
            t = []
            for li in range(ls, self.nloc):        # a block reaches its own diagonal: lower triangle only, per 128-row band
                t += [(self.my_blocks[li] - k - 1) * NB + (q + 1) * 128 for q in range(bands)]
            t += [self.Np - (k + 1) * NB] * tail_bands
            tabs.append(t)
        flat = [v for t in tabs for v in t] or [0]
        tb["rowmap"] = torch.tensor(flat, dtype=torch.int32, device=self.dev)
        tb["rowmap_h"] = np.ascontiguousarray(flat, dtype=np.int32)
        tb["rowmap_off"] = offs
        tb["rowmap_len"] = [len(t) for t in tabs]
        # lookahead part (b) of step k: my rows of blocks > k+1 update columns from block k+2 on
        tabs, offs = [], []
        for k in range(self.T - 2):
            ls = self._lstart(k + 1)
            offs.append(sum(len(t) for t in tabs))
            t = []
            for li in range(ls, self.nloc):
                t += [(self.my_blocks[li] - k - 2) * NB + (q + 1) * 128 for q in range(bands)]
            t += [self.Np - (k + 2) * NB] * tail_bands
            tabs.append(t)
        flat = [v for t in tabs for v in t] or [0]
        tb["rowmapB"] = torch.tensor(flat, dtype=torch.int32, device=self.dev)
        tb["rowmapB_h"] = np.ascontiguousarray(flat, dtype=np.int32)
        tb["rowmapB_off"] = offs
        tb["rowmapB_len"] = [len(t) for t in tabs]
        # critical-path-first schedule: the diagonal block of block row k+2 is updated ahead of part (b)
        # (it is the next but one to be factored), so its bands update nothing in (b)
        flatC = []
        for k, t in enumerate(tabs):
            t = list(t)
            ls1 = self._lstart(k + 1)
            if ls1 < self.nloc and self.my_blocks[ls1] == k + 2:
                t[:bands] = [0] * bands
            flatC += t
        tb["rowmapC"] = torch.tensor(flatC or [0], dtype=torch.int32, device=self.dev)
        tb["rowmapC_h"] = np.ascontiguousarray(flatC or [0], dtype=np.int32)
        return tb

    def _set_tail(self, vrows):
        """make `vrows` (a multiple of 128) the number of test rows that ride below my blocks: row count, row maps, room in A"""
        if vrows % 128:
            raise ValueError("tail rows must be a multiple of 128")
        tb = self._tail_cache.get(vrows)
        if tb is None:
            tb = self._tail_cache[vrows] = self._build_rowmaps(vrows)
        for key, v in tb.items():
            setattr(self, key, v)
        self.rows = self.rows_base + vrows
        if self.A.shape[0] < max(self.rows, 1):
            self.A = None                                 # free first: a rank's matrix can be most of the memory
            if self.dev.type == "cuda":
                torch.cuda.empty_cache()                  # ... and hand the block back, or the allocator keeps it beside the new one
            self.A = self._tensor(max(self.rows, 1), self.ld)
            self.have_factor = False

    def _ticket(self, on):
        """context manager: the enclosed update launches use the ticket form of the GEMM (HIP primitives only)"""
        import contextlib
        ops = self.ops

        @contextlib.contextmanager
        def scope():
            if on and self.ticket and hasattr(ops, "set_option"):
                ops.set_option("gemm_ticket", 2)
                try:
                    yield
                finally:
                    ops.set_option("gemm_ticket", 0)
            else:
                yield
        return scope()

    def _concurrent(self, on):
        """context manager: the block primitives' beside-an-update forms (gpmi_dev_set_concurrent) for the enclosed sweep"""
        import contextlib
        ops = self.ops

        @contextlib.contextmanager
        def scope():
            if hasattr(ops, "set_concurrent") and on:
                ops.set_concurrent(True)
                try:
                    yield
                finally:
                    ops.set_concurrent(False)
            else:
                yield
        return scope()

    # ------------------------------------------------------------------ streams (no-ops on CPU)
    def _cuda(self):
        return self.dev.type == "cuda"

    def _side(self):
        """context manager: run on the side (panel + collectives) stream"""
        import contextlib
        if not self._cuda():
            return contextlib.nullcontext()
        return torch.cuda.stream(self._named_stream("side"))

    def _named_stream(self, name):
        """The auxiliary streams ('side', 'crit') are ONE set per process and device, shared by every DistGP instance: the
        runtime maps streams onto a small pool of hardware queues as they are first used, and a second instance's fresh
        pair can land on ONE queue -- its panel chain and its diagonal-block chain then run one after the other instead
        of side by side (measured on the 8-rank replay: the same rank 259 ms as the first instance of its process,
        315 ms as the second; profiles/r04_stream_mapping.txt).  Ordering between instances is the streams' own."""
        if not self._cuda():
            return None
        import threading
        # (per host thread: the thread-rank tests run several ranks of one world in one process, each with streams of its own)
        key = (self.dev.index if self.dev.index is not None else torch.cuda.current_device(), name, threading.get_ident())
        st = _SHARED_STREAMS.get(key)
        if st is None:
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=self.dev, priority=hi)
        return st

    def _on(self, name):
        """context manager: run on the named auxiliary stream ('side', 'crit'); no-op on CPU"""
        import contextlib
        if not self._cuda():
            return contextlib.nullcontext()
        if name == "side":
            return self._side()
        return torch.cuda.stream(self._named_stream(name))

    def _stream_of(self, name):
        if name == "main":
            return torch.cuda.current_stream(self.dev)
        if name == "side":
            with self._side():
                return torch.cuda.current_stream(self.dev)
        return self._named_stream(name)

    def _record(self, name):
        """event after everything queued so far on the named stream (None on CPU)"""
        if not self._cuda():
            return None
        ev = torch.cuda.Event()
        ev.record(self._stream_of(name))
        return ev

    def _wait(self, name, ev):
        if ev is not None:
            self._stream_of(name).wait_event(ev)

    def _order(self, first_is_side):
        """make the other stream wait for everything queued so far on the first one"""
        if not self._cuda():
            return
        main = torch.cuda.current_stream(self.dev)
        with self._side():
            side = torch.cuda.current_stream(self.dev)
        ev = torch.cuda.Event()
        if first_is_side:
            ev.record(side)
            main.wait_event(ev)
        else:
            ev.record(main)
            side.wait_event(ev)

    # ------------------------------------------------------------------ fit
    def _gather_panel(self, k, buf, r0):
        """all-gather of block column k below the diagonal: this rank's solved rows (from local row r0) are
        packed into the send buffer, every rank's chunk lands in buf at rank * cmax_k blocks; runs on the
        current stream.  Consumers address the blocks through the offset table of step k."""
        NB, G, A = self.NB, self.G, self.A
        c0 = k * NB
        cnts = [self._nblocks(r) - self._lstart(k, r) for r in range(G)]
        cmax = max(cnts)
        cnt = cnts[self.rank]
        if cnt:
            with self._timed("pack", None):
                self.send[:cnt * NB * NB].view(cnt * NB, NB).copy_(A[r0:r0 + cnt * NB, c0:c0 + NB])
        with self._timed("allgather", None):
            self.comm.all_gather(buf[:G * cmax * NB * NB], self.send[:cmax * NB * NB], tag=("panel", k))

    def _pblock(self, k, buf, i):
        """block k + 1 + i of panel column k as an NB x NB tensor"""
        NB = self.NB
        if not self.coll:
            r0 = (self._lstart(k) + i) * NB
            return self.A[r0:r0 + NB, k * NB:(k + 1) * NB]
        off = int(self.boff_h[self.boff_start[k] + i])
        return buf[off:off + NB * NB].view(NB, NB)

    def _update(self, k, buf, Cm, Am, first, rowmap=None, rowmap_h=None):
        """Cm -= Am * (panel column k from its block `first` on)^T, columns of Cm in natural block order"""
        ops, NB = self.ops, self.NB
        with self._timed("update", None):
            if not self.coll:
                r0 = (self._lstart(k) + first) * NB
                nrow = Cm.shape[1]
                P = self.A[r0:r0 + nrow, k * NB:(k + 1) * NB]
                if rowmap is None:
                    ops.gemm_nt(Cm, Am, P)
                else:
                    ops.gemm_nt_rowmap(Cm, Am, P, rowmap, 128, rowmap_h)
                return
            s0 = self.boff_start[k] + first
            nblk = Cm.shape[1] // NB
            ops.gemm_nt_blocks(Cm, Am, buf, NB, self.boff[s0:s0 + nblk], NB, rowmap, 128, rowmap_h)

    def _panel_step(self, k, buf):
        """Block column k: owner factors the diagonal block, broadcast, every rank solves its
        rows below, all-gather of the panel column into buf."""
        ops, NB, G, A = self.ops, self.NB, self.G, self.A
        owner = self._own[k]
        c0 = k * NB
        if self.rank == owner:
            li = self._li[k]
            diag = A[li * NB:(li + 1) * NB, c0:c0 + NB]
            with self._timed("diag", None):
                ops.potrf_block(diag, c0, self.info)
                self.Lkk.copy_(diag)
        if self.coll:
            with self._timed("bcast", None):
                self.comm.broadcast(self.Lkk, owner, tag=("Lkk", k))
        ls = self._lstart(k)
        r0 = ls * NB
        m = self.rows - r0
        if m > 0:
            with self._timed("panel_solve", None):
                ops.trsm_block(self.Lkk, A[r0:r0 + m, c0:c0 + NB])
        if k == self.T - 1 or not self.coll:
            return
        self._gather_panel(k, buf, r0)

    def _factor_critical_path_first(self):
        """Right-looking sweep with the diagonal chain decoupled from the panel chain.

        Per step k, three streams:
          crit  owner of block k+1 only: its block (k+1, k) <- block * L_kk^-T, the diagonal block
                (k+1, k+1) -= that block times its transpose, Cholesky of the diagonal block;
                then (all ranks) the broadcast of L_(k+1)(k+1)
          side  every rank: its other rows of block column k <- rows * L_kk^-T, all-gather of the column
          main  (a) block column k+1 of the rows below block k+1, (b) the remaining columns
        Dependencies across streams are events; every rank issues the collectives in the same
        order (broadcast k, all-gather k, broadcast k+1, ...)."""
        ops, NB, G, A, T = self.ops, self.NB, self.G, self.A, self.T
        Lk = self.Lk
        ev_a = self._record("main")                      # the K build
        self._wait("crit", ev_a)
        with self._on("crit"):
            if self.rank == self._own[0]:
                diag = A[0:NB, 0:NB]
                with self._timed("diag", None):
                    ops.potrf_block(diag, 0, self.info)
                    Lk[0].copy_(diag)
            if self.coll:
                with self._timed("bcast", None):
                    self.comm.broadcast(Lk[0], self._own[0], tag=("Lkk", 0))
        ev_bcast = self._record("crit")
        ev_panel_prev = None                             # side-stream work of step k-1 (last reader of Lk[(k+1) % 2])
        import time
        for k in range(T):
            t_issue = time.perf_counter()
            Lcur = Lk[k % 2]
            buf = self.Pbuf[k % 2]
            c0, c1 = k * NB, (k + 1) * NB
            ls = self._lstart(k)                         # my first local block below k
            own_next = (k + 1 < T) and (self.rank == self._own[k + 1])
            ev_row = None
            # Lk[(k+1) % 2] is overwritten below (owner: copy of its new diagonal block; everyone: the
            # broadcast); its last reader is the side stream's solve of step k-1
            self._wait("crit", ev_panel_prev)
            if own_next:                                 # block k+1 is my local block `ls`
                self._wait("crit", ev_a)                 # its columns <= k+1 are final up to step k-1
                with self._on("crit"):
                    blk = A[ls * NB:(ls + 1) * NB, c0:c0 + NB]
                    with self._timed("diag", None):
                        ops.trsm_block(Lcur, blk)
                    ev_row = self._record("crit")
                    dk = A[ls * NB:(ls + 1) * NB, c1:c1 + NB]
                    with self._timed("diag", None):
                        ops.gemm_nt(dk, blk, blk)
                        ops.potrf_block(dk, c1, self.info)
                        Lk[(k + 1) % 2].copy_(dk)
            self._wait("side", ev_bcast)                 # L_kk has arrived
            self._wait("side", ev_a)                     # block column k carries every update before step k
            with self._on("side"):
                r_rest = (ls + 1) * NB if own_next else ls * NB
                m = self.rows - r_rest
                if m > 0:
                    with self._timed("panel_solve", None):
                        ops.trsm_block(Lcur, A[r_rest:r_rest + m, c0:c0 + NB])
                if k < T - 1 and self.coll:
                    self._wait("side", ev_row)
                    self._gather_panel(k, buf, ls * NB)
            ev_panel = self._record("side")
            ev_panel_prev = ev_panel
            if k == T - 1:
                self._wait("main", ev_panel)
                break
            with self._on("crit"):                       # queued behind the owner's potrf on this stream
                if self.coll:
                    with self._timed("bcast", None):
                        self.comm.broadcast(Lk[(k + 1) % 2], self._own[k + 1], tag=("Lkk", k + 1))
            ev_bcast = self._record("crit")
            with self._timed("stall_panel", "main"):      # elapsed between these two events = main-stream time lost to the panel chain
                self._wait("main", ev_panel)
            if not self.coll:
                self._wait("main", ev_row)
            r1 = self._lstart(k + 1) * NB                # my rows of blocks > k+1 (and the y rows)
            m1 = self.rows - r1
            if m1 > 0:                                   # (a) block column k+1 below its diagonal block
                with self._timed("update", None):
                    ops.gemm_nt(A[r1:r1 + m1, c1:c1 + NB], A[r1:r1 + m1, c0:c0 + NB], self._pblock(k, buf, 0))
            ls1 = self._lstart(k + 1)
            if k + 2 < T and ls1 < self.nloc and self.my_blocks[ls1] == k + 2:
                # (a2) the diagonal block of my block row k+2: everything the next critical step reads
                rows = slice(ls1 * NB, (ls1 + 1) * NB)
                with self._timed("update", None):
                    ops.gemm_nt(A[rows, c1 + NB:c1 + 2 * NB], A[rows, c0:c0 + NB], self._pblock(k, buf, 1))
            ev_a = self._record("main")
            if k + 2 < T and m1 > 0:                     # (b) the remaining columns
                off, ln = self.rowmapB_off[k], self.rowmapB_len[k]
                # ticket form unless this rank factors diagonal block k+1 meanwhile (potrf128 needs a whole CU, and
                # resident workgroups would keep every CU until the launch ends)
                with self._ticket(not own_next):
                    self._update(k, buf, A[r1:r1 + m1, c1 + NB:self.Np], A[r1:r1 + m1, c0:c0 + NB], 1,
                                 self.rowmapC[off:off + ln], self.rowmapC_h[off:off + ln])
            if self._prof is not None:
                self._prof.setdefault("host_issue", []).append((t_issue, time.perf_counter()))
        self._wait("main", ev_bcast)
        self._wait("main", self._record("crit"))

    def factorize(self, sigma, ell, noise_var, _ride=False):
        """K + sI -> L (distributed), m = L^-1 y; returns the log-marginal-likelihood
        (tune_hyperparms_regression.py:312) on every rank.  Raises LinAlgError on every
        rank if a pivot is not positive.

        Lookahead level 1: the trailing update of step k is split: (a) block column k+1 first,
        then the side stream runs the whole panel step k+1 (diagonal block, broadcast, solves,
        all-gather) while (b) the remaining columns update on the main stream -- collectives and
        the latency-bound panel kernels hide behind the MFMA work.  Level 2 (default): see
        _factor_critical_path_first."""
        import time
        t_begin = time.perf_counter()
        # _ride (fit_predict_resident): my share of the test set's rows K(X*, X) sits below my blocks (and the y rows) and is
        # carried through every panel solve and update like the y rows
        self._set_tail(self.v_rows if _ride else 0)
        ops, NB, G, A = self.ops, self.NB, self.G, self.A
        self.sigma, self.ell = float(sigma), float(ell)
        self.have_factor = False
        self.have_v = False
        self._vinv_blocks = set()
        self.info.fill_(INT64_MAX)
        # K + sI: my row blocks, lower part
        if self._kind is not None and self._kind[0] == 2 and self.d != 1:
            raise ValueError("the periodic kernel is 1-D only (GP_regression.py:48)")
        for li, b in enumerate(self.my_blocks):
            self._cov_rows(self.X, self.N, b * NB, NB, (b + 1) * NB, noise_var, A[li * NB:(li + 1) * NB])
        if self.yrow is not None:
            A[self.yrow:self.yrow + YB, :self.Np].zero_()
            A[self.yrow, :self.N].copy_(self.y)
        if _ride and self.v_rows:
            v0 = self.rows_base
            self._cov_cross(self.Xs[self.v_t0:self.v_t0 + max(self.v_cnt, 1)], self.v_cnt, self.X, self.N, 0, self.v_rows, self.Np,
                            A[v0:v0 + self.v_rows, :self.Np])
        T = self.T
        # panel primitives run beside the update on other streams: their small-LDS forms while the sweep is in flight;
        # switched back on every way out (an exception in a collective must not leave the thread-local flag on)
        with self._concurrent(bool(self.lookahead)):
            if self.lookahead >= 2:
                self._factor_critical_path_first()
            elif not self.lookahead:
                for k in range(T):
                    self._panel_step(k, self.Pbuf[0])
                    if k == T - 1:
                        break
                    r0 = self._lstart(k) * NB
                    m = self.rows - r0
                    if m > 0:
                        off, ln = self.rowmap_off[k], self.rowmap_len[k]
                        self._update(k, self.Pbuf[0], A[r0:r0 + m, (k + 1) * NB:self.Np], A[r0:r0 + m, k * NB:(k + 1) * NB], 0,
                                     self.rowmap[off:off + ln], self.rowmap_h[off:off + ln])
            else:
                self._order(first_is_side=False)              # side waits for the K build
                with self._side():
                    self._panel_step(0, self.Pbuf[0])
                for k in range(T - 1):
                    with self._timed("stall_panel", "main"):
                        self._order(first_is_side=True)       # main waits for panel k
                    buf = self.Pbuf[k % 2]
                    c0, c1 = k * NB, (k + 1) * NB
                    r0 = self._lstart(k) * NB                 # my rows of blocks > k (and the y rows)
                    m = self.rows - r0
                    if m > 0:                                 # (a) block column k+1
                        with self._timed("update", None):
                            ops.gemm_nt(A[r0:r0 + m, c1:c1 + NB], A[r0:r0 + m, c0:c0 + NB], self._pblock(k, buf, 0))
                    self._order(first_is_side=False)          # side waits for (a)
                    with self._side():
                        self._panel_step(k + 1, self.Pbuf[(k + 1) % 2])
                    if k + 2 < T:                             # (b) the remaining columns
                        r1 = self._lstart(k + 1) * NB         # my rows of blocks > k+1 (and the y rows)
                        m1 = self.rows - r1
                        if m1 > 0:
                            off, ln = self.rowmapB_off[k], self.rowmapB_len[k]
                            self._update(k, buf, A[r1:r1 + m1, c1 + NB:self.Np], A[r1:r1 + m1, c0:c0 + NB], 1,
                                         self.rowmapB[off:off + ln], self.rowmapB_h[off:off + ln])
                self._order(first_is_side=True)               # main waits for the last panel
        # not-PD: smallest failing global column over all ranks
        if self.coll:
            self.comm.all_reduce(self.info, "min", tag=("info",))
        info = int(self.info.item())
        if info != INT64_MAX and info < self.N:
            err = np.linalg.LinAlgError("Matrix is not positive definite")
            err.bad_pivot = info + 1
            raise err
        # LML pieces: per-rank sum of log diag over owned blocks, m^T m on the y rank (fixed order)
        self.red.zero_()
        for li, b in enumerate(self.my_blocks):
            ops.logdiag_sumsq(A[li * NB:(li + 1) * NB, b * NB:(b + 1) * NB], NB, None, 0, self.red[li])
        if self.yrow is not None:
            ops.logdiag_sumsq(None, 0, A[self.yrow], self.N, self.red[self.nloc])
            self.m.copy_(A[self.yrow, :self.Np])
        # this rank's piece: its log-diagonal sums added in block order (fixed order, own kernel), m^T m on the y rank
        part = self._tensor(2)
        ops.sum_fixed(self.red, self.nloc, 2, 1, part[0:1])
        part[1:2].copy_(self.red[self.nloc, 1:2])
        if self.coll:
            allp = self._tensor(G * 2)
            self.comm.all_gather(allp, part, tag=("lml",))
            self.comm.broadcast(self.m, self.ry, tag=("m",))
        else:
            allp = part
        allp = allp.view(G, 2).cpu().numpy()
        logsum = 0.0
        for r in range(G):
            logsum += float(allp[r, 0])
        mtm = float(allp[self.ry, 1])
        self.have_factor = True
        self._rode = bool(_ride)
        self.stage_ms["fit"] = (time.perf_counter() - t_begin) * 1e3
        return -.5 * mtm - logsum - self.N / 2.0 * math.log(2 * math.pi)

    # ------------------------------------------------------------------ predict
    def set_test(self, Xs):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        if Xs.ndim != 2 or Xs.shape[1] != self.d:
            raise ValueError("X_test must be (n, d) with the training d")
        self.n = Xs.shape[0]
        self.n_p = _round_up(self.n, 128)
        self.Xs = torch.from_numpy(Xs).to(self.dev)
        if self.coll:
            self.comm.broadcast(self.Xs, 0, tag=("Xs",))
        self.ldv = max(self.nloc, 1) * self.NB + self.ld_pad
        self.V = self._tensor(self.n_p, self.ldv)
        self.Xk = [self._tensor(self.n_p, self.NB) for _ in range(2)]
        self.dots = self._tensor(2, self.n_p)
        self.m_loc = self._tensor(max(self.nloc, 1) * self.NB)
        # one-pass form (fit_predict_resident): the test points are dealt to the ranks in contiguous runs of v_nr (a multiple
        # of 128); mine are v_t0 .. v_t0 + v_cnt (v_rows with the padding to whole 128-row bands)
        self.v_nr = _round_up((self.n_p + self.G - 1) // self.G, 128)
        self.v_t0 = min(self.rank * self.v_nr, self.n_p)
        self.v_rows = min(self.v_nr, self.n_p - self.v_t0)
        self.v_cnt = max(0, min(self.v_nr, self.n - self.v_t0))
        self.have_test = True
        self.have_v = False

    def predict_resident(self, want_sd=True):
        """mu = K_s^T alpha (as v^T m), var = sigma^2 - sum(v^2) (GP_regression.py:143-148)."""
        if not self.have_factor:
            raise ValueError("no factorisation resident (call factorize)")
        if not self.have_test:
            raise ValueError("no test set (call set_test)")
        import time
        t_begin = time.perf_counter()
        ops, NB, G, A, V = self.ops, self.NB, self.G, self.A, self.V
        with self._concurrent(bool(self.lookahead)):
            for li, b in enumerate(self.my_blocks):
                self._cov_cross(self.Xs, self.n, self.X[b * NB:], self.N - b * NB, b * NB, self.n_p, NB,
                                V[:, li * NB:(li + 1) * NB])
                self.m_loc[li * NB:(li + 1) * NB].copy_(self.m[b * NB:(b + 1) * NB])
            T = self.T

            def solve_block(k, Xk):
                """owner: v^T block k <- block * L_kk^-T; everyone receives it in Xk"""
                if self.rank == self._own[k]:
                    li = self._li[k]
                    blk = V[:, li * NB:(li + 1) * NB]
                    with self._timed("solve_v", None):
                        ops.trsm_block(A[li * NB:(li + 1) * NB, k * NB:(k + 1) * NB], blk)
                        if self.coll:
                            Xk.copy_(blk)
                if self.coll and k < T - 1:
                    with self._timed("bcast_v", None):
                        self.comm.broadcast(Xk, self._own[k], tag=("vblock", k))

            def xk_view(k):
                return self.Xk[k % 2] if self.coll else V[:, k * NB:(k + 1) * NB]

            if not self.lookahead:
                for k in range(T):
                    solve_block(k, self.Xk[0])
                    if k == T - 1:
                        break
                    ls = self._lstart(k)
                    if self.nloc - ls > 0:
                        with self._timed("update_v", None):
                            ops.gemm_nt(V[:, ls * NB:self.nloc * NB], self.Xk[0] if self.coll else xk_view(k),
                                        A[ls * NB:self.nloc * NB, k * NB:(k + 1) * NB])
            else:
                self._order(first_is_side=False)
                with self._side():
                    solve_block(0, self.Xk[0])
                for k in range(T - 1):
                    with self._timed("stall_v", "main"):
                        self._order(first_is_side=True)        # main waits for solved block k
                    Xk = xk_view(k)
                    c0 = k * NB
                    ls = self._lstart(k)
                    own_next = (self.rank == self._own[k + 1])
                    if own_next:                               # (a) my block k+1 first
                        li = self._li[k + 1]
                        with self._timed("update_v", None):
                            ops.gemm_nt(V[:, li * NB:(li + 1) * NB], Xk, A[li * NB:(li + 1) * NB, c0:c0 + NB])
                    self._order(first_is_side=False)
                    with self._side():
                        solve_block(k + 1, self.Xk[(k + 1) % 2])
                    lb = self._lstart(k + 1)                   # (b) my blocks beyond k+1
                    if self.nloc - lb > 0:
                        with self._timed("update_v", None), self._ticket(True):      # the sweep has no whole-CU kernel
                            ops.gemm_nt(V[:, lb * NB:self.nloc * NB], Xk, A[lb * NB:self.nloc * NB, c0:c0 + NB])
                self._order(first_is_side=True)
        self.dots.zero_()
        if self.nloc:
            ops.row_dots(V, self.nloc * NB, self.m_loc, self.dots[0], self.dots[1])
        if self.coll:
            alld = self._tensor(G * 2 * self.n_p)
            self.comm.all_gather(alld, self.dots.view(-1), tag=("dots",))
        else:
            alld = self.dots
        alld = alld.view(G, 2, self.n_p).cpu().numpy()
        mu = np.zeros(self.n_p)
        sq = np.zeros(self.n_p)
        for r in range(G):                     # fixed order: bitwise reproducible
            mu += alld[r, 0]
            sq += alld[r, 1]
        var = self._kss_diag() - sq[:self.n]
        self.have_v = True
        self.stage_ms["predict"] = (time.perf_counter() - t_begin) * 1e3
        with np.errstate(invalid="ignore"):
            out2 = np.sqrt(var) if want_sd else var
        return mu[:self.n].copy(), out2

    def predict(self, Xs, want_sd=True):
        self.set_test(Xs)
        return self.predict_resident(want_sd)

    def fit_predict_resident(self, sigma, ell, noise_var, want_sd=True):
        """prediction() in one pass (GP_regression.py:109-156; the single-GPU gpmi_fit_predict_resident): the test points are
        dealt to the ranks, each rank's rows K(X*_mine, X) ride below its row blocks through the factorisation -- its panel
        solves and update launches carry them with no launch and NO MESSAGE of their own (the panel column every rank
        receives for its update is all they need) -- and come out as its rows of v^T; mean and variance of its points are
        row dots with m (which every rank holds), one all-gather of 2 n / G doubles per rank collects them.  Against
        predict_resident(): 63 broadcasts of n x nb blocks (2.1 GB per rank at N = 65536, n = 4096, nb = 1024) and the
        sweep's latency chain are gone.  Returns (lml, mu, sd or var); results agree with factorize() + predict_resident()
        to rounding.  post_chol() afterwards runs the column-distributed sweep first (it needs v by columns)."""
        if not self.have_test:
            raise ValueError("no test set (call set_test)")
        if self._kind is not None and self._kind[0] == 3 and self.n == self.N and self.G > 1:
            # kernel_4's delta sits on the diagonal of a SQUARE K_s (CO2_example.py:58): a row offset the cross build
            # does not take -- the two-call form serves this one case
            lml = self.factorize(sigma, ell, noise_var)
            mu, out2 = self.predict_resident(want_sd)
            return lml, mu, out2
        import time
        lml = self.factorize(sigma, ell, noise_var, _ride=True)
        t_begin = time.perf_counter()
        ops, G, nr = self.ops, self.G, self.v_nr
        dots = self._tensor(2, nr)
        dots.zero_()
        if self.v_rows:
            v0 = self.rows_base
            ops.row_dots(self.A[v0:v0 + self.v_rows], self.Np, self.m, dots[0, :self.v_rows], dots[1, :self.v_rows])
        if self.coll:
            alld = self._tensor(G * 2 * nr)
            self.comm.all_gather(alld, dots.view(-1), tag=("dots_rows",))
        else:
            alld = dots
        alld = alld.view(G, 2, nr).cpu().numpy()
        mu = np.zeros(G * nr)
        sq = np.zeros(G * nr)
        for r in range(G):
            mu[r * nr:(r + 1) * nr] = alld[r, 0]
            sq[r * nr:(r + 1) * nr] = alld[r, 1]
        var = self._kss_diag() - sq[:self.n]
        self.stage_ms["predict"] = (time.perf_counter() - t_begin) * 1e3
        with np.errstate(invalid="ignore"):
            out2 = np.sqrt(var) if want_sd else var
        return lml, mu[:self.n].copy(), out2

    def fit_predict(self, X, y, Xs, sigma, ell, noise_var, want_sd=True):
        self.set_train(X, y)
        self.set_test(Xs)
        return self.fit_predict_resident(sigma, ell, noise_var, want_sd)

    # ------------------------------------------------------------------ posterior covariance factor (f1)
    def post_chol(self, jitter):
        """L_ = cholesky(K_ss + jitter * I - v^T v) (GP_regression.py:153-154) with v distributed: every rank forms
        the contribution of its own column blocks of v^T (one MFMA SYRK), the contributions are summed by one
        all-reduce of n_p x n_p doubles, and every rank factors the small matrix itself (same bits on every
        rank).  Needs predict_resident() first.  Returns the n x n lower factor; raises LinAlgError if not PD."""
        if self.have_factor and self.have_test and not getattr(self, "have_v", False) and getattr(self, "_rode", False):
            self.predict_resident()                       # after the one-pass form: v by columns is this sweep's product
        if not (self.have_factor and self.have_test and getattr(self, "have_v", False)):
            raise ValueError("post_chol needs a factorisation and predict_resident() first")
        ops, NB, n_p = self.ops, self.NB, self.n_p
        P = self._tensor(n_p, n_p + self.ld_pad)
        Gm = self._tensor(n_p, n_p + self.ld_pad)
        Gm.zero_()
        if self.nloc:
            ops.gemm_nt(Gm[:, :n_p], self.V[:, :self.nloc * NB], self.V[:, :self.nloc * NB])     # Gm = -v_loc^T v_loc
        if self.coll:
            self.comm.all_reduce(Gm, "sum", tag=("vtv",))
        self._cov_rows(self.Xs, self.n, 0, n_p, n_p, float(jitter), P)
        ops.axpy2d(P[:, :n_p], Gm[:, :n_p], 1.0)
        info = torch.full((1,), INT64_MAX, dtype=torch.int64, device=self.dev)
        ops.potrf_block(P[:, :n_p], 0, info)
        bad = int(info.item())
        if bad != INT64_MAX and bad < self.n:
            err = np.linalg.LinAlgError("Matrix is not positive definite")
            err.bad_pivot = bad + 1
            raise err
        return np.tril(P[:self.n, :self.n].cpu().numpy())

    # ------------------------------------------------------------------ LML gradient (f2)
    def lml_grad(self):
        """(dLML/dl, dLML/dsigma) = .5 * trace((alpha alpha^T - K_y^-1) dK/dtheta) at the resident factorisation
        (tune_hyperparms_regression.py:43-57, :144) with L distributed.  U = L^-T is formed by the predict sweep on
        the identity (rank r ends up with ITS column blocks of U, upper triangular); K_y^-1 = U U^T is a sum over
        columns, so every rank owns an additive part of it: per row block it forms -U_r[rows] U_r^T with one MFMA
        GEMM into an nb x N scratch and feeds the fused trace kernel -- nothing N x N is ever summed across ranks,
        the alpha alpha^T term is added on rank 0 only, and the 2 x G partial traces are summed in rank order."""
        if not self.have_factor:
            raise ValueError("no factorisation resident (call factorize)")
        if self._kind is not None:
            raise ValueError("lml_grad: squared-exponential kernel only (tune_hyperparms_regression.py:54)")
        ops, NB, G, A, T, Np = self.ops, self.NB, self.G, self.A, self.T, self.Np
        alpha = self.alpha()                                   # full vector, every rank
        a_full = self._tensor(Np)
        a_full.zero_()
        a_full[:self.N].copy_(torch.from_numpy(alpha).to(self.dev))
        a_row = a_full if self.rank == 0 else torch.zeros_like(a_full)
        ncl = max(self.nloc, 1) * NB
        U = self._tensor(Np, ncl + self.ld_pad)
        U.zero_()
        for li, b in enumerate(self.my_blocks):                # my columns of the identity
            U[b * NB:(b + 1) * NB, li * NB:(li + 1) * NB].diagonal().fill_(1.0)
        Xk = self._tensor(Np, NB)
        # ---- U^T sweep: column block k of U (rows < (k+1) NB are non-zero) <- block * L_kk^-T, then the later blocks
        for k in range(T):
            m = (k + 1) * NB
            if self.rank == self._own[k]:
                li = self._li[k]
                blk = U[:m, li * NB:(li + 1) * NB]
                ops.trsm_block(A[li * NB:(li + 1) * NB, k * NB:(k + 1) * NB], blk)
                if self.coll:
                    Xk[:m].copy_(blk)
            if k == T - 1:
                break
            if self.coll:
                self.comm.broadcast(Xk[:m], self._own[k], tag=("ublock", k))
            src = Xk[:m] if self.coll else U[:m, self._li[k] * NB:(self._li[k] + 1) * NB]
            ls = self._lstart(k)
            if self.nloc - ls > 0:
                ops.gemm_nt(U[:m, ls * NB:self.nloc * NB], src, A[ls * NB:self.nloc * NB, k * NB:(k + 1) * NB])
        # ---- partial traces, one row block of (my part of) -K_y^-1 at a time
        scratch = self._tensor(NB, Np + self.ld_pad)
        nblk = (NB // 128) * (Np // 128)
        partial = self._tensor(2 * nblk)
        out2 = self._tensor(2)
        out2.zero_()
        for a in range(T):
            lo = self._lstart(a - 1) if a > 0 else 0           # my first column block with global index >= a
            scratch.zero_()
            if self.nloc - lo > 0:                             # row a of U is zero left of its own diagonal block
                ops.gemm_nt(scratch[:, :Np], U[a * NB:(a + 1) * NB, lo * NB:self.nloc * NB], U[:, lo * NB:self.nloc * NB])
            nrows = min(NB, self.N - a * NB)
            if nrows > 0:
                ops.grad_trace(self.X, self.N, self.d, a * NB, nrows, a_row, a_full, scratch, -1.0, self.sigma, self.ell,
                               partial, out2)
        if self.coll:
            allp = self._tensor(G * 2)
            self.comm.all_gather(allp, out2, tag=("grad",))
        else:
            allp = out2
        allp = allp.view(G, 2).cpu().numpy()
        sl = ss = 0.0
        for r in range(G):                                     # fixed order
            sl += float(allp[r, 0])
            ss += float(allp[r, 1])
        return .5 * sl, .5 * ss

    # ------------------------------------------------------------------ alpha
    def alpha(self):
        """alpha = solve(L.T, m) (GP_regression.py:140) with L distributed by row blocks.
        Blocks from the bottom up: every rank forms the contribution of its rows below
        block k, sum_j L_jk^T alpha_j (its own alpha_j, which it solved itself); the
        contributions are gathered and summed in rank order; the owner of block k solves
        L_kk^T alpha_k = m_k - sum.  One 8*G*nb-byte all-gather per block: latency-bound,
        L never moves.  Returns the full alpha (N,) on every rank."""
        if not self.have_factor:
            raise ValueError("no factorisation resident (call factorize)")
        import time
        t_begin = time.perf_counter()
        ops, NB, G, A, T = self.ops, self.NB, self.G, self.A, self.T
        aloc = self._tensor(max(self.nloc, 1) * NB)
        aloc.zero_()
        part = self._tensor(NB)
        allp = self._tensor(G * NB)
        rhs = self._tensor(NB)
        scratch = self._tensor(max((self.nloc * NB + 63) // 64, 1) * NB)
        for k in range(T - 1, -1, -1):
            c0 = k * NB
            ls = self._lstart(k)
            r0, r1 = ls * NB, self.nloc * NB
            with self._timed("alpha_gemv", None):
                if r1 > r0:
                    ops.gemv_t(A[r0:r1, c0:c0 + NB], aloc[r0:r1], part, scratch)
                else:
                    part.zero_()
            if self.coll:
                with self._timed("alpha_gather", None):
                    self.comm.all_gather(allp, part, tag=("alpha_part", k))
            else:
                allp.copy_(part)
            if self.rank == self._own[k]:
                li = self._li[k]
                # rhs = m_k - (part_0 + part_1 + ... ): ONE launch of the library's own fixed-order sum (contributions added
                # in rank order whatever the launch geometry: the same bits on every run and every rank)
                with self._timed("alpha_solve", None):
                    ops.sum_fixed(allp, G, NB, NB, rhs, base=self.m[c0:c0 + NB], scale=-1.0)
                    if self._cuda():             # the one-launch backward solve keeps each block's inverses in a side buffer
                        vs = self._vside.get(li)
                        if vs is None:
                            vs = self._vside[li] = self._tensor(NB * 128)
                        ops.trsv_lt(A[li * NB:(li + 1) * NB, c0:c0 + NB], rhs, inverted=li in self._vinv_blocks, vside=vs)
                    else:
                        ops.trsv_lt(A[li * NB:(li + 1) * NB, c0:c0 + NB], rhs, inverted=li in self._vinv_blocks)
                    self._vinv_blocks.add(li)
                    aloc[li * NB:(li + 1) * NB].copy_(rhs)
        # assemble the full vector in natural block order
        cmax = max(self._nblocks(r) for r in range(G))
        send = self._tensor(cmax * NB)
        send.zero_()
        send[:self.nloc * NB].copy_(aloc[:self.nloc * NB])
        if self.coll:
            recv = self._tensor(G * cmax * NB)
            self.comm.all_gather(recv, send, tag=("alpha_full",))
        else:
            recv = send
        R = recv.view(G, cmax, NB).cpu().numpy()
        # a poll of the one-launch diagonal-block solve that gave up (non-finite factor) is known to the block's owner only:
        # the flag is max-reduced so that EVERY rank raises, none walks on to the next collective with a NaN alpha
        err = getattr(ops, "_err", None)
        if self._cuda():
            if err is None:
                err = ops._err = torch.zeros(16, dtype=torch.int32, device=self.dev)
            flag = err[0:1].to(torch.float64)
            if self.coll:
                self.comm.all_reduce(flag, "max", tag=("alpha_err",))
            if float(flag.item()) != 0.0:
                err.zero_()
                raise RuntimeError("DistGP.alpha: the one-launch backward solve gave up waiting for a block (non-finite factor?)")
        out = np.empty(self.Np)
        for b in range(T):
            out[b * NB:(b + 1) * NB] = R[self._own[b], self._li[b]]
        self.stage_ms["alpha"] = (time.perf_counter() - t_begin) * 1e3
        return out[:self.N].copy()

    def fit(self, X, y, sigma, ell, noise_var):
        self.set_train(X, y)
        return self.factorize(sigma, ell, noise_var)

    def timers(self):
        return dict(self.stage_ms)


_default = {}


def default_dist(n_gpus=None):
    """The DistGP over WORLD that the drop-in functions use for `n_gpus=`: one per process,
    created on first use on this rank's device.  torch.distributed must already be initialised by the launcher
    (one process per GPU); n_gpus, when given, must be the world size."""
    if not dist.is_initialized():
        raise RuntimeError("n_gpus needs torch.distributed initialised by the launcher (one process per GPU, e.g. "
                           "python -m torch.distributed.run --nproc-per-node N ...)")
    world = dist.get_world_size()
    if n_gpus is not None and int(n_gpus) != world:
        raise ValueError("n_gpus=%s but the process group has %d ranks" % (n_gpus, world))
    gp = _default.get("gp")
    if gp is None:
        import os
        dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
        nb = int(os.environ.get("GPMI_DIST_NB", "512"))
        gp = _default["gp"] = DistGP(dev, nb=nb)
    return gp


def sharded_lml_batch(triples, evaluate, group=None):
    """Batched log-marginal-likelihood over hyper-parameter triples (BASELINE config 5;
    the reference's loops at tune_hyperparms_regression.py:368-369, 385-386), sharded over
    the ranks: triple t is evaluated by rank t % G with the single-GPU path -- the
    factorisations are independent, so there is no data-path collective, only one
    all-gather of the T results.

    triples: (T, 3) rows of (l, sigma_f, noise_var), identical on every rank.
    evaluate: callable(sub_triples (t, 3)) -> (lml (t,), status (t,)), e.g.
              GPContext.lml_batch after GPContext.set_train on every rank.
    Returns (lml (T,), status (T,)) on every rank, NaN where a factorisation failed.
    """
    t = np.ascontiguousarray(triples, dtype=np.float64)
    if t.ndim != 2 or t.shape[1] != 3:
        raise ValueError("triples must be (T, 3) = (l, sigma_f, noise_var)")
    T = t.shape[0]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        lml, st = evaluate(t)
        return np.asarray(lml, dtype=np.float64), np.asarray(st, dtype=np.int32)
    G, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (T + G - 1) // G
    mine = t[rank::G]
    lml = np.full(per, np.nan)
    st = np.zeros(per)
    if len(mine):
        l, s_ = evaluate(mine)
        lml[:len(mine)] = l
        st[:len(mine)] = s_
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    send = torch.from_numpy(np.concatenate([lml, st])).to(dev)
    recv = torch.empty(G * 2 * per, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    R = recv.cpu().numpy().reshape(G, 2, per)
    out = np.empty(T)
    out_s = np.empty(T, dtype=np.int32)
    for r in range(G):
        cnt = len(range(r, T, G))
        out[r::G] = R[r, 0, :cnt]
        out_s[r::G] = R[r, 1, :cnt].astype(np.int32)
    return out, out_s
