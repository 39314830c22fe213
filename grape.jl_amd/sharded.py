"""Trajectory-sharded evaluation: one process per GPU, ``torch.distributed`` for the two
cross-trajectory reductions of the path:

1. between the forward and the backward sweep: all-reduce(sum) of the partial sums
   ``[Re f, Im f, sum_k w_k |tau_k|^2, Re sum_k w_k tau_k, sum_k J_b,k]`` with ``f = sum_k w_k tau_k``
   (needed by chi_sm, docs/src/tutorial.md:402, by every J_T and by the state running cost);
2. at the end: all-reduce(sum) of the partial gradient -- the sum over k of
   ``_grad_J_T_via_chi!`` (/root/reference/src/optimize.jl:579).

With the ``nccl`` backend (RCCL on ROCm) both collectives run on device buffers that alias the
library's inputs/outputs, on the same HIP stream as the kernels; with ``gloo`` (CPU tests) the
host-pointer split-phase API is used.
"""
from __future__ import annotations

import numpy as np


def functional_value(functional, sums, K_total, lambda_b=0.0):
    """J from the all-reduced sums [Re f, Im f, sum w|tau|^2, Re sum w tau, sum_k J_b,k, ...]."""
    fr, fi, ss, re = (float(v) for v in sums[:4])
    jb = lambda_b * float(sums[4]) if len(sums) > 4 else 0.0
    if functional == 0:
        return 1.0 - (fr * fr + fi * fi) / (K_total * K_total) + jb
    if functional == 1:
        return 1.0 - ss / K_total + jb
    return 1.0 - re / K_total + jb


def shard_range(K_total, world_size, rank):
    """Contiguous block of trajectories owned by ``rank`` (k in [g K/G, (g+1) K/G))."""
    base, rem = divmod(K_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedEvaluator:
    """fg! over a trajectory-sharded ensemble.  ``handle`` exposes the split-phase API of
    ``GrapeHip`` (forward/backward or forward_device/backward_device) for the LOCAL trajectories
    and was created with ``K_total`` = size of the whole ensemble."""

    def __init__(self, handle, K_total, functional, weights_local=None, dist=None, device=None):
        self.h = handle
        self.K_total = int(K_total)
        self.functional = functional
        # the handle owns the weights (they enter chi on the device): ``weights_local`` is only cross-checked
        hw = getattr(handle, "_weights", None)
        if weights_local is not None and hw is not None and not np.array_equal(np.asarray(weights_local), hw):
            raise ValueError("weights_local differs from the weights the handle was created with")
        self.dist = dist  # torch.distributed module (initialised) or None for a single process
        self.device = device  # torch.device for the RCCL path, None for the host (gloo) path
        self._buf = None

    # -- host path (gloo / single process) ---------------------------------------------------
    def eval_host(self, pulsevals):
        """fg! over all shards with host buffers: J (including lambda_b * J_b of the state running cost), the full
        gradient and this shard's tau."""
        import torch
        tau = self.h.forward(pulsevals)
        # the shard's partial sums as the library formed them (its weights, and sum_k J_b,k): grape_get_sums
        sums = torch.from_numpy(np.array(self.h.sums(), dtype=np.float64))
        if self.dist is not None:
            self.dist.all_reduce(sums)
        G = torch.from_numpy(self.h.backward(complex(sums[0].item(), sums[1].item())))
        if self.dist is not None:
            self.dist.all_reduce(G)
        J = functional_value(self.functional, sums.tolist(), self.K_total, getattr(self.h, "lambda_b", 0.0))
        return J, G.numpy(), tau

    # -- device path (nccl == RCCL) ------------------------------------------------------------
    def alloc_device(self, L, N_T, K_local):
        import torch
        self._x = torch.empty(L * N_T, dtype=torch.float64, device=self.device)
        self._out = torch.zeros(2 * K_local + 8, dtype=torch.float64, device=self.device)
        self._G = torch.zeros(L * N_T, dtype=torch.float64, device=self.device)
        self._K = K_local
        return self._x, self._out, self._G

    def eval_device(self, stream_ptr):
        """One evaluation with inputs already resident in ``self._x``; everything stays on the
        device and on ``stream_ptr``.  Returns nothing (results in self._out / self._G)."""
        K = self._K
        self.h.forward_device(self._x.data_ptr(), self._out.data_ptr(), stream_ptr)
        sums = self._out[2 * K:2 * K + 8]   # f, sum w|tau|^2, Re sum w tau, sum J_b (+ padding)
        if self.dist is not None:
            self.dist.all_reduce(sums)
        self.h.backward_device(sums.data_ptr(), self._G.data_ptr(), stream_ptr)
        if self.dist is not None:
            self.dist.all_reduce(self._G)

    def check_collective(self, stream_ptr):
        """``grape_check`` on every rank, with the one recoverable status decided TOGETHER: GRAPE_ERR_AGAIN (blocked
        exponential, N > 64: the asynchronous squaring plan was too short and has been adapted) makes a rank repeat the
        evaluation, and an evaluation contains collectives -- a rank repeating alone would pair its all-reduces with the
        other ranks' next step.  Returns True when ALL ranks have to repeat the evaluation (any rank saw AGAIN); every
        other error is raised.  One extra all-reduce of one integer, only for handles on the blocked path."""
        from .api import GrapeHipError
        state, err = 0, None   # 0 fine, 1 repeat, 2 failed (a rank that fails must not leave the others in the collective)
        try:
            self.h.check(stream_ptr)
        except GrapeHipError as exc:
            state, err = (1, None) if exc.code == -7 else (2, exc)
        if self.dist is not None and getattr(self.h, "N", 0) > 64:
            import torch
            flag = torch.tensor([state], dtype=torch.int32, device=self.device if self.device is not None else "cpu")
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            state = int(flag.item())
        if err is not None:
            raise err
        if state == 2:
            raise GrapeHipError(-3, "another rank's shard failed in this evaluation")
        return state == 1

    def J_device(self):
        K = self._K
        return functional_value(self.functional, self._out[2 * K:2 * K + 8].tolist(), self.K_total,
                                getattr(self.h, "lambda_b", 0.0))
