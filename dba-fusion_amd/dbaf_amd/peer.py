"""One-shot all-reduce of the reduced camera system by direct peer reads (csrc/peer_allreduce.hip), and a stand-in for
the `dist` argument of dbaf_amd.sharded that routes float64 all-reduces through it.

Opt-in: `PeerDist.wrap(dist)` returns `dist` unchanged unless DBA_PEER_ALLREDUCE=1.  The exchange regions are set up
once per process group (one hipIpc handle per rank, exchanged with `dist.all_gather_object`); every other collective
(reduce, broadcast, all_gather_into_tensor, barrier) is forwarded to the wrapped module."""
import ctypes
import os

import torch

from . import _lib


class PeerAllReduce:
    def __init__(self, rank, world, max_doubles, exchange_handles):
        """exchange_handles(my_handle: bytes) -> list of every rank's handle (bytes), in rank order."""
        self.lib = _lib.load()
        self.rank, self.world, self.max_doubles = int(rank), int(world), int(max_doubles)
        nbytes = self.lib.dba_peer_exchange_bytes(self.max_doubles)
        mine = ctypes.c_void_p()
        hbuf = (ctypes.c_ubyte * 64)()
        _lib.check(self.lib.dba_peer_exchange_create(nbytes, ctypes.byref(mine), ctypes.cast(hbuf, ctypes.c_void_p)),
                   "dba_peer_exchange_create")
        self._mine = mine
        handles = exchange_handles(bytes(hbuf))
        assert len(handles) == self.world
        self._opened = []
        regions = []
        for r, h in enumerate(handles):
            if r == self.rank:
                regions.append(mine.value)
                continue
            ptr = ctypes.c_void_p()
            hb = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            _lib.check(self.lib.dba_peer_exchange_open(ctypes.cast(hb, ctypes.c_void_p), ctypes.byref(ptr)),
                       "dba_peer_exchange_open")
            self._opened.append(ptr)
            regions.append(ptr.value)
        self._regions = (ctypes.c_void_p * self.world)(*regions)
        self._status = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.epoch = 0

    def all_reduce(self, t):
        """sum over the ranks, in place; t: contiguous float64 CUDA tensor with at most max_doubles elements"""
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() <= self.max_doubles
        self.epoch += 1
        _lib.check(self.lib.dba_peer_allreduce_f64(ctypes.c_void_p(t.data_ptr()), t.numel(),
                                                   ctypes.cast(self._regions, ctypes.c_void_p), self.rank, self.world,
                                                   self.epoch, self.max_doubles, ctypes.c_void_p(self._status.data_ptr()),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "dba_peer_allreduce_f64")

    def timed_out(self):
        """True if any all-reduce so far gave up waiting for a peer (synchronises)"""
        return bool(self._status.item())

    def check(self):
        """Raises if an all-reduce since the last check gave up waiting for a peer -- its buffer then holds a partial sum
        and whatever was computed from it (solve, retraction) is wrong.  Synchronises; clears the sticky status word."""
        if self._status.item():
            self._status.zero_()
            raise RuntimeError("peer all-reduce: a rank waited longer than DBA_PEER_TIMEOUT_MS for a peer's contribution; "
                               "the reduced camera system of this call is incomplete (state must be discarded)")

    def close(self):
        torch.cuda.synchronize()
        for ptr in self._opened:
            self.lib.dba_peer_exchange_close(ptr, 1)
        self._opened = []
        if self._mine is not None:
            self.lib.dba_peer_exchange_close(self._mine, 0)
            self._mine = None


class PeerDist:
    """`dist` stand-in for dbaf_amd.sharded: float64 all-reduces that fit the exchange regions go through PeerAllReduce,
    everything else through the wrapped torch.distributed module."""

    def __init__(self, dist, max_doubles=1 << 18):
        self._dist = dist

        def exchange(mine):
            out = [None] * dist.get_world_size()
            dist.all_gather_object(out, mine)
            return out

        self.peer = PeerAllReduce(dist.get_rank(), dist.get_world_size(), max_doubles, exchange)
        dist.barrier()   # every rank has mapped every region before the first epoch

    @staticmethod
    def wrap(dist):
        if dist is None or os.environ.get("DBA_PEER_ALLREDUCE") != "1" or dist.get_world_size() < 2:
            return dist
        return PeerDist(dist)

    def all_reduce(self, t, *args, **kwargs):
        if not args and not kwargs and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() \
                and t.numel() <= self.peer.max_doubles:
            return self.peer.all_reduce(t)
        return self._dist.all_reduce(t, *args, **kwargs)

    def check(self):
        self.peer.check()

    def __getattr__(self, name):
        return getattr(self._dist, name)
