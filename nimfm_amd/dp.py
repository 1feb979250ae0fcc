"""Data parallelism: one process per GPU, sample-sharded, replicated parameters, RCCL exchange.

The reference's only parallel strategy is shared-memory Hogwild over contiguous sample slices
(optimizer/sgd_multi.nim:83-101): every thread owns a slice, all threads share one model.  Across
GPUs the slices become per-rank shards (resident in each GPU's HBM) and the shared model becomes
replicas that are reconciled by ONE collective per exchange over xGMI (torch.distributed backend
"nccl" = RCCL); there is no collective inside an epoch.

Exchange rules:
  SGD      replicas are averaged:  theta <- (1/N) sum_r theta_r        (local SGD / model averaging)
           All ranks advance `it` identically, so their global L2 scales are identical and the
           stored tensors (theta / scale) can be averaged directly.
  AdaGrad  the state is additive over samples (optimizer/adagrad.nim:113-134), so the replicas'
           increments since the last exchange are summed:  G <- G_prev + sum_r (G_r - G_prev);
           the result is the state one process would hold after seeing all shards' samples at the
           parameters each replica used.

The tensors below alias the library's device buffers (nfm_model_device_buffers /
nfm_opt_device_state) through __cuda_array_interface__: no copies, torch only supplies the
collective.  exchange() itself is backend-agnostic and is exercised with gloo on CPU tensors in
tests/test_dp_gloo.py.
"""
import ctypes as C

from . import _capi as capi


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def alias(torch, dev, ptr, n):
    """float64 torch tensor of n elements over raw device memory (no copy)."""
    return torch.as_tensor(_DevArray(ptr, n), device=dev)


def exchange(tensors, dist, world, rule, prevs=None, force=False):
    """Reconcile replicas in place.  rule: "average" | "sum_deltas" (needs prevs, updated in place).
    force runs the collective even for a single replica (used to test the plumbing on one GPU)."""
    if world <= 1 and not force:
        return
    if rule == "average":
        avg = dist.get_backend() == "nccl"  # RCCL averages inside the collective; gloo has no AVG
        for t in tensors:
            if avg:
                try:
                    dist.all_reduce(t, op=dist.ReduceOp.AVG)
                    continue
                except RuntimeError:  # a build without ncclAvg for this dtype: sum, then scale
                    avg = False
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(world)
    elif rule == "sum_deltas":
        for t, p in zip(tensors, prevs):
            t.sub_(p)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.add_(p)
            p.copy_(t)
    else:
        raise ValueError("unknown exchange rule %r" % (rule,))


class ParamViews:
    """torch views of a model's (and, for AdaGrad, an optimizer's) device buffers on one GPU."""

    def __init__(self, torch, dev, fm, opt):
        self.torch, self.dev = torch, dev
        L = capi.lib()
        P, w, sc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nP, nw, ns = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(L.nfm_model_device_buffers(fm._h, C.byref(P), C.byref(nP), C.byref(w), C.byref(nw), C.byref(sc),
                                              C.byref(ns)))
        self.params = [alias(torch, dev, P.value, nP.value), alias(torch, dev, w.value, nw.value),
                       alias(torch, dev, sc.value, ns.value)]
        # [P | w | scalars] is one allocation (include/nimfm_hip.h): one collective per exchange
        self.arena = [alias(torch, dev, P.value, (sc.value + 8 * ns.value - P.value) // 8)]
        self.state, self.state_prev = [], []
        self.is_adagrad = type(opt).__name__ == "AdaGrad"
        if self.is_adagrad:
            G, N, Gw, Nw, gs = (C.c_void_p() for _ in range(5))
            n1, n2 = C.c_int64(), C.c_int64()
            capi.check(L.nfm_opt_device_state(opt._h, C.byref(G), C.byref(N), C.byref(n1), C.byref(Gw), C.byref(Nw),
                                              C.byref(n2), C.byref(gs)))
            self.state = [alias(torch, dev, G.value, n1.value), alias(torch, dev, N.value, n1.value),
                          alias(torch, dev, Gw.value, n2.value), alias(torch, dev, Nw.value, n2.value),
                          alias(torch, dev, gs.value, 2)]
            # [G | N | Gw | Nw | gscalars] is one allocation as well
            self.state_arena = [alias(torch, dev, G.value, (gs.value + 16 - G.value) // 8)]
            self.state_prev = [t.clone() for t in self.state_arena]

    def average(self, dist, world, force=False):
        """Called between epochs; the library has synchronised its stream when nfm_opt_epoch returns."""
        if self.is_adagrad:
            exchange(self.state_arena, dist, world, "sum_deltas", self.state_prev, force)
        else:
            exchange(self.arena, dist, world, "average", None, force)
        self.torch.cuda.synchronize(self.dev)
