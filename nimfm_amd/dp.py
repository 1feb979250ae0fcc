"""Data parallelism: thin host-side handles over the library's groups (include/nimfm_hip.h, nfm_dp_*).

The reference's only parallel strategy is shared-memory Hogwild over contiguous sample slices
(optimizer/sgd_multi.nim:83-101): every thread owns a slice, all threads share one model.  Across GPUs the
slices become per-rank shards (resident in each GPU's HBM) and the shared model becomes replicas that the
LIBRARY reconciles -- every `sync_period` mini-batches, overlapped with the next period's mini-batches on a
second stream, and exactly at the end of every nfm_opt_epoch call (csrc/dp.hip):

  SGD      the replicas' increments averaged (default) or summed        (Optimizer.setDataParallel(..., combine=))
  AdaGrad  the replicas' g_sum / g_norm increments summed               (optimizer/adagrad.nim:113-134 is additive)

One process per GPU: the collective is RCCL's ncclAllReduce over xGMI on the single parameter / state arena,
through a communicator the library owns.  This module only bootstraps it: rank 0 asks the library for an id
and the bytes travel over whatever the host already has (torch.distributed here; a Nim host would use a
file, the environment or MPI).  `Group.local` makes the ranks of ONE process (threads) instead -- several
GPUs with peer access, or one GPU shared by all ranks, which is how the exchange rules are tested on a
one-GPU box (tests/test_gpu_dp.py).
"""
import ctypes as C

from . import _capi as capi

ID_BYTES = 128


class Group:
    """one rank's handle of a data-parallel group (nfm_dp)"""

    def __init__(self, handle, ctx):
        self.h, self.ctx = handle, ctx

    @staticmethod
    def unique_id():
        buf = (C.c_char * ID_BYTES)()
        capi.check(capi.lib().nfm_dp_unique_id(buf))
        return bytes(buf)

    @classmethod
    def rccl(cls, ctx, id_bytes, rank, world):
        """collective over the ranks (one process per GPU): ncclCommInitRank inside the library"""
        if len(id_bytes) != ID_BYTES:
            raise ValueError("the group id is %d bytes" % ID_BYTES)
        h = C.c_void_p()
        capi.check(capi.lib().nfm_dp_create(ctx.h, id_bytes, rank, world, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_torch(cls, ctx, dist):
        """bootstrap over an initialised torch.distributed process group (any backend): only the 128-byte id
        travels through it; the exchange itself is the library's own RCCL communicator"""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.rccl(ctx, box[0], rank, world)

    @classmethod
    def local(cls, ctxs):
        """the ranks of this process, one context (and one host thread) each"""
        world = len(ctxs)
        arr = (C.c_void_p * world)(*[c.h for c in ctxs])
        out = (C.c_void_p * world)()
        capi.check(capi.lib().nfm_dp_create_local(arr, world, out))
        return [cls(C.c_void_p(out[r]), ctxs[r]) for r in range(world)]

    def info(self):
        r, w, n, b = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().nfm_dp_info(self.h, C.byref(r), C.byref(w), C.byref(n), C.byref(b)))
        return {"rank": r.value, "world": w.value, "collectives": n.value, "bytes": b.value}

    def close(self):
        if self.h and capi.alive:
            capi.lib().nfm_dp_destroy(self.h)
        self.h = None


def shard_bounds(n, rank, world):
    """the reference's thread partition (optimizer/sgd_multi.nim:85-88): contiguous slices of n div world samples, the
    last one takes the remainder"""
    lo = rank * (n // world)
    hi = n if rank == world - 1 else lo + n // world
    return lo, hi
