"""G ranks as G threads of ONE process -- TEST INFRASTRUCTURE.

`gaussian_process_amd.dist.DistGP` talks to its communicator through five members (rank, size, broadcast,
all_gather, all_reduce; `TorchComm` is the product's).  `ThreadWorld(G).comm(r)` is an in-process stand-in with the
same semantics, so that north_star's own topology (8 ranks) can run the real HIP block primitives -- offset tables,
staircase row maps, three streams per rank -- on the ONE GPU of the test box, where the pool allows at most six
processes on the card.  Collectives are rendezvous points: every rank drains the stream it is on, meets the others at
a barrier, copies device-to-device, drains again.  Reductions sum in rank order (same bits on every rank).
"""
import threading

import torch


class ThreadWorld:
    def __init__(self, size, timeout=300.0):
        self.size = size
        self.barrier = threading.Barrier(size, timeout=timeout)
        self.slots = [None] * size

    def comm(self, rank):
        return ThreadComm(self, rank)

    def run(self, fn):
        """fn(rank, comm) on `size` threads; returns the list of results, re-raises the first failure"""
        out, err = [None] * self.size, [None] * self.size

        def body(r):
            try:
                out[r] = fn(r, self.comm(r))
            except BaseException as e:          # noqa: BLE001 -- a dead rank must not leave the others at the barrier
                err[r] = e
                self.barrier.abort()
        ts = [threading.Thread(target=body, args=(r,)) for r in range(self.size)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        first = next((e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
        if first is None:
            first = next((e for e in err if e is not None), None)
        if first is not None:
            raise first
        return out


class ThreadComm:
    def __init__(self, world, rank):
        self.world, self.rank, self.size = world, rank, world.size

    @staticmethod
    def _drain(t):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def _meet(self):
        self.world.barrier.wait()

    def broadcast(self, t, src, tag=None):
        w = self.world
        self._drain(t)
        if self.rank == src:
            w.slots[0] = t
        self._meet()
        if self.rank != src:
            t.copy_(w.slots[0])
            self._drain(t)
        self._meet()

    def all_gather(self, out, inp, tag=None):
        w = self.world
        self._drain(inp)
        w.slots[self.rank] = inp
        self._meet()
        n = inp.numel()
        flat = out.view(-1)
        for r in range(self.size):
            flat[r * n:(r + 1) * n].copy_(w.slots[r].reshape(-1))
        self._drain(out)
        self._meet()

    def all_reduce(self, t, op="sum", tag=None):
        w = self.world
        self._drain(t)
        w.slots[self.rank] = t.clone()
        self._meet()
        acc = w.slots[0].clone()
        for r in range(1, self.size):
            acc = torch.minimum(acc, w.slots[r]) if op == "min" else torch.maximum(acc, w.slots[r]) if op == "max" \
                else acc + w.slots[r]
        t.copy_(acc)
        self._drain(t)
        self._meet()

    def describe(self):
        return {"backend": "threads (test stand-in)", "ranks": self.size, "env": {}}
