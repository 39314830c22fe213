"""world_size-2 gloo test of the trajectory-sharded path (CPU): the ShardedEvaluator protocol
(all-reduce of the tau partial sums between the sweeps, all-reduce of the partial gradient) must
reproduce the single-process result.  The per-shard evaluator is an oracle-backed stand-in with the
split-phase signature of GrapeHip.forward/backward."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import grape_oracle as go
from grape_jl_amd import synth
from grape_jl_amd.sharded import ShardedEvaluator, functional_value, shard_range


class OracleShard:
    def __init__(self, pr, lo, hi, K_total, functional):
        self.pr, self.lo, self.hi, self.K_total, self.functional = pr, lo, hi, K_total, functional

    def _args(self):
        p = self.pr
        sl = slice(self.lo, self.hi)
        return p["H0"][sl], p["Hc"], p["tlist"]

    def forward(self, x):
        p, sl = self.pr, slice(self.lo, self.hi)
        self.x = np.array(x)
        _, tau, _ = go.evaluate_functional(*self._args(), x, p["psi0"][sl], p["target"][sl], p["weights"][sl])
        self.tau = tau
        return tau

    def sums(self):   # grape_get_sums: the shard's partial sums with the handle's own weights
        w, tau = self.pr["weights"][self.lo:self.hi], self.tau
        f = np.sum(w * tau)
        return np.array([f.real, f.imag, np.sum(w * np.abs(tau) ** 2), f.real, 0.0, 0.0, 0.0, 0.0])

    def backward(self, f_total):
        p, sl = self.pr, slice(self.lo, self.hi)
        _, G, _ = go.evaluate_gradient(*self._args(), self.x, p["psi0"][sl], p["target"][sl], p["weights"][sl],
                                       functional=self.functional, K_total=self.K_total, f_total=f_total)
        return G


def _worker(rank, world, port, functional, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = synth.make_problem(6, 2, 5, 5, seed=42)
    pr["weights"] = np.array([0.5, 1.0, 1.5, 2.0, 0.25])
    lo, hi = shard_range(5, world, rank)
    ev = ShardedEvaluator(OracleShard(pr, lo, hi, 5, functional), 5, functional, dist=dist)
    J, G, tau = ev.eval_host(pr["pulsevals"])
    if rank == 0:
        q.put((J, G, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("functional", [0, 1, 2])
def test_two_rank_gloo_matches_single_process(functional):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, functional, q)) for r in range(2)]
    for p in procs:
        p.start()
    J, G, lo, hi = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pr = synth.make_problem(6, 2, 5, 5, seed=42)
    pr["weights"] = np.array([0.5, 1.0, 1.5, 2.0, 0.25])
    Jr, Gr, _ = go.evaluate_gradient(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"],
                                     pr["weights"], functional=functional)
    assert (lo, hi) == (0, 3)
    assert abs(J - Jr) < 1e-14 and np.abs(G - Gr).max() < 1e-14


def test_shard_range_and_functional_value():
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert abs(functional_value(0, [3.0, 4.0, 0, 0], 10) - 0.75) < 1e-15
    assert abs(functional_value(1, [0, 0, 2.0, 0], 4) - 0.5) < 1e-15
    assert abs(functional_value(2, [0, 0, 0, 1.0], 4) - 0.75) < 1e-15
    _ = torch


class _AgainShard:
    """Stand-in whose check() reports GRAPE_ERR_AGAIN (blocked path: the squaring plan was too short) on chosen ranks for
    the first evaluation only, as GrapeHip.check does."""
    N = 100   # blocked path: the collective decision applies

    def __init__(self, again_first):
        self.calls, self.again_first = 0, again_first

    def check(self, stream):
        from grape_jl_amd.api import GrapeHipError
        self.calls += 1
        if self.calls == 1 and self.again_first:
            raise GrapeHipError(-7, "plan too short")


def _again_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # only rank 1 sees AGAIN: BOTH ranks must be told to repeat (an evaluation contains collectives; a rank repeating alone
    # would pair its all-reduces with the other rank's next step), and nobody repeats the second time
    ev = ShardedEvaluator(_AgainShard(again_first=(rank == 1)), 4, 0, dist=dist, device=None)
    first = ev.check_collective(0)
    second = ev.check_collective(0)
    q.put((rank, first, second))
    dist.barrier()
    dist.destroy_process_group()


def test_again_is_decided_by_all_ranks_together():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_again_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get() for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(0, True, False), (1, True, False)]


class _AgainOracleShard(OracleShard):
    """A real (oracle-backed) shard on the blocked-path protocol: its FIRST evaluation ends in GRAPE_ERR_AGAIN on the chosen
    rank only -- as a handle does whose squaring plan was too short -- and the partial gradient it handed out for that
    evaluation is garbage (the propagators were never finished)."""
    N = 100

    def __init__(self, *a, again=False):
        super().__init__(*a)
        self.again, self.evals, self.checks = again, 0, 0

    def backward(self, f_total):
        self.evals += 1
        G = super().backward(f_total)
        return G + 1e3 if (self.again and self.evals == 1) else G

    def check(self, stream):
        from grape_jl_amd.api import GrapeHipError
        self.checks += 1
        if self.again and self.checks == 1:
            raise GrapeHipError(-7, "plan too short")


def _again_loop_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = synth.make_problem(6, 2, 5, 5, seed=42)
    pr["weights"] = np.array([0.5, 1.0, 1.5, 2.0, 0.25])
    lo, hi = shard_range(5, world, rank)
    sh = _AgainOracleShard(pr, lo, hi, 5, 0, again=(rank == 1))
    ev = ShardedEvaluator(sh, 5, 0, dist=dist)
    # the step of bench.py / of an optimizer on the sharded path: evaluate, decide TOGETHER, repeat together
    for attempt in (0, 1):
        J, G, _ = ev.eval_host(pr["pulsevals"])
        if not ev.check_collective(0):
            break
    # a further, ordinary step: its collectives must pair with the other rank's (they would not if one rank had repeated alone)
    J2, G2, _ = ev.eval_host(pr["pulsevals"] * 1.01)
    assert not ev.check_collective(0)
    q.put((rank, attempt, sh.evals, J, G, J2, G2))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_repeating_keeps_the_collectives_of_all_ranks_paired():
    """The deadlock / mis-pairing the collective decision exists to prevent, end to end: rank 1's shard reports
    GRAPE_ERR_AGAIN after the first evaluation.  Both ranks repeat the evaluation (attempt == 1 on BOTH), every rank ran
    three shard evaluations in all, and the gradients of the repeated and of the following step equal the single-process
    ones -- no all-reduce was paired with another step's."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_again_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get() for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pr = synth.make_problem(6, 2, 5, 5, seed=42)
    pr["weights"] = np.array([0.5, 1.0, 1.5, 2.0, 0.25])
    args = (pr["H0"], pr["Hc"], pr["tlist"])
    Jr, Gr, _ = go.evaluate_gradient(*args, pr["pulsevals"], pr["psi0"], pr["target"], pr["weights"], functional=0)
    Jr2, Gr2, _ = go.evaluate_gradient(*args, pr["pulsevals"] * 1.01, pr["psi0"], pr["target"], pr["weights"], functional=0)
    for rank, attempt, evals, J, G, J2, G2 in got:
        assert attempt == 1 and evals == 3, (rank, attempt, evals)
        assert abs(J - Jr) < 1e-14 and np.abs(G - Gr).max() < 1e-14
        assert abs(J2 - Jr2) < 1e-14 and np.abs(G2 - Gr2).max() < 1e-14


def test_in_handle_reduction_order_is_the_shard_order():
    """Several devices behind ONE handle (grape_problem.ndev): the library adds the shard sums and the partial gradients in
    shard order on the calling thread, whatever order the per-shard host threads finish their enqueue halves in -- the
    result equals the left-to-right sum over contiguous trajectory blocks bit for bit, and differs from other orders only
    in the last bits.  Restated here with the oracle shards: (a) blocks of shard_range reproduce the unsharded result,
    (b) the fixed order is what makes repeated evaluations identical."""
    pr = synth.make_problem(6, 2, 5, 7, seed=9)
    pr["weights"] = np.linspace(0.5, 2.0, 7)
    G_parts, sums = [], np.zeros(8)
    shards = []
    for g in range(3):
        lo, hi = shard_range(7, 3, g)
        sh = OracleShard(pr, lo, hi, 7, 0)
        sh.forward(pr["pulsevals"])
        shards.append(sh)
        sums += sh.sums()                       # shard order, as multi_forward does
    for sh in shards:
        G_parts.append(sh.backward(complex(sums[0], sums[1])))
    G = G_parts[0].copy()
    for part in G_parts[1:]:
        G += part                               # shard order, as multi_backward does
    Jr, Gr, _ = go.evaluate_gradient(pr["H0"], pr["Hc"], pr["tlist"], pr["pulsevals"], pr["psi0"], pr["target"],
                                     pr["weights"], functional=0)
    assert abs(functional_value(0, sums, 7) - Jr) < 1e-14 and np.abs(G - Gr).max() < 1e-14
    G2 = G_parts[0].copy()
    for part in G_parts[1:]:
        G2 += part
    assert np.array_equal(G, G2)
    assert np.abs((G_parts[2] + G_parts[1]) + G_parts[0] - G).max() < 1e-15   # another order: equal to rounding only
