"""Deterministic synthetic inputs for the BASELINE.json configurations.

A counter-based generator (SplitMix64 output function over ``seed + (i+1)*gamma``) feeds
uniforms and Box-Muller normals, so the same arrays can be regenerated bit-identically in any
language.  Distributions:

* ``H0 = (X + X^dagger) / (4 sqrt(N))``, ``X_ij ~ CN(0, 2)`` (GUE scaled to spectral radius ~ 1);
  ensemble member ``k``: ``H0_k = H0 + 0.05 * GUE_k``
* control operators ``H_l``: same distribution, shared by all trajectories
* ``psi0_k``, ``target_k``: random unit vectors;  weights 1
* ``tlist = 0, dt, ..., N_T dt`` with ``dt = 1``;  ``eps_nl = 0.1 + 0.2 u``, ``u ~ U(-1, 1)``
"""
from __future__ import annotations

import numpy as np

_MASK = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15
BASE_SEED = 0x6772617065  # "grape"

CONFIGS = {
    # id: (N, L, N_T, K)
    "C1": (2, 1, 500, 1),       # README two-level problem (literal, see readme_tls())
    "C2": (16, 1, 500, 32),
    "C3": (64, 2, 1000, 128),   # headline
    "C4": (64, 2, 1000, 1024),  # 8 GPUs x 128 trajectories
    "C5": (256, 4, 2000, 64),
    # not BASELINE configurations: the headline shape at the two other register-tile counts (timing tools only)
    "X48": (48, 2, 1000, 128),
    "X32": (32, 2, 1000, 128),
    "C3L6": (64, 6, 1000, 128),   # six controls (the control count of /root/reference/test/test_lbfgsb_saddle_point.jl:89-124)
}


def _mix(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Outputs offset .. offset+n-1 of the SplitMix64 stream started at ``seed``."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        state = np.uint64(seed & _MASK) + idx * np.uint64(_GAMMA)
        return _mix(state)


def subseed(seed: int, tag: int) -> int:
    return int(splitmix64(seed ^ (tag * 0xD1342543DE82EF95 & _MASK), 1)[0])


def uniform01(seed: int, n: int) -> np.ndarray:
    """n doubles in the open interval (0, 1)."""
    u = splitmix64(seed, n)
    return ((u >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, n: int) -> np.ndarray:
    """n standard normals (Box-Muller on consecutive uniform pairs)."""
    m = (n + 1) // 2
    u = uniform01(seed, 2 * m)
    r = np.sqrt(-2.0 * np.log(u[0::2]))
    th = 2.0 * np.pi * u[1::2]
    z = np.empty(2 * m)
    z[0::2] = r * np.cos(th)
    z[1::2] = r * np.sin(th)
    return z[:n]


def gue(seed: int, N: int) -> np.ndarray:
    z = normal(seed, 2 * N * N)
    X = (z[0::2] + 1j * z[1::2]).reshape(N, N)
    return (X + X.conj().T) / (4.0 * np.sqrt(N))


def unit_vectors(seed: int, K: int, N: int) -> np.ndarray:
    z = normal(seed, 2 * K * N)
    v = (z[0::2] + 1j * z[1::2]).reshape(K, N)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make_problem(N: int, L: int, N_T: int, K: int, seed: int = BASE_SEED, dt: float = 1.0,
                 k_offset: int = 0, hermitian: bool = True):
    """Synthetic ensemble problem.  ``k_offset`` selects the global trajectory index of the first
    local trajectory, so that shards of one job regenerate exactly their own members."""
    H_base = gue(subseed(seed, 1), N)
    H0 = np.empty((K, N, N), dtype=np.complex128)
    for k in range(K):
        H0[k] = H_base + 0.05 * gue(subseed(seed, 1000 + k_offset + k), N)
    Hc = np.stack([gue(subseed(seed, 100 + l), N) for l in range(L)])
    if not hermitian:  # non-Hermitian variant used by the parity tests (Liouvillian-like generators)
        z = normal(subseed(seed, 7), 2 * N * N)
        H0 = H0 + 0.1 * (z[0::2] + 1j * z[1::2]).reshape(N, N) / np.sqrt(N)
    psi0 = np.concatenate([unit_vectors(subseed(seed, 2000 + k_offset + k), 1, N) for k in range(K)])
    target = np.concatenate([unit_vectors(subseed(seed, 3000 + k_offset + k), 1, N) for k in range(K)])
    tlist = dt * np.arange(N_T + 1, dtype=np.float64)
    u = 2.0 * uniform01(subseed(seed, 4), L * N_T) - 1.0
    pulsevals = 0.1 + 0.2 * u  # control-major [l * N_T + n]
    return dict(N=N, L=L, N_T=N_T, K=K, H0=H0, Hc=Hc, psi0=psi0, target=target, tlist=tlist,
                pulsevals=pulsevals, weights=np.ones(K))


def make_config(cid: str, K: int | None = None, k_offset: int = 0, hermitian: bool = True):
    N, L, N_T, K0 = CONFIGS[cid]
    if cid == "C1":
        return readme_tls()
    tag = int("".join(ch for ch in cid if ch.isdigit()))   # C3 -> 3, X48 -> 48, C3L6 -> 36
    return make_problem(N, L, N_T, K0 if K is None else K, seed=BASE_SEED ^ tag, k_offset=k_offset, hermitian=hermitian)


def readme_tls(eps0: float = 0.2, T: float = 5.0, nt: int = 501):
    """The README problem (/root/reference/README.md:37-43): H = sigma_z + eps(t) sigma_x, |0> -> |1>."""
    sz = np.array([[1, 0], [0, -1]], dtype=np.complex128)
    sx = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    tlist = np.linspace(0.0, T, nt)
    return dict(N=2, L=1, N_T=nt - 1, K=1, H0=sz[None], Hc=sx[None],
                psi0=np.array([[1, 0]], dtype=np.complex128), target=np.array([[0, 1]], dtype=np.complex128),
                tlist=tlist, pulsevals=np.full(nt - 1, eps0), weights=np.ones(1))
