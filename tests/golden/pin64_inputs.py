"""Inputs of the high-precision pins at the sizes of the headline kernels (tests/golden/make_mpmath_pin64.py,
mpmath_pin_herm64.json / mpmath_pin_herm100.json): Hermitian generators whose entries are dyadic rationals, produced by a
linear congruential generator written out here -- no library random stream enters, so the doubles the tests feed to the oracles
and to the GPU ARE the numbers mpmath worked with, on every machine and numpy version."""
import numpy as np

SHAPES = {"herm64": dict(N=64, L=2, N_T=4, K=2, seed=0x6772617065_64), "herm100": dict(N=100, L=2, N_T=2, K=1, seed=0x6772617065_100),
          # general (non-Hermitian) generators: expm_t18g_asm + deriv3g_asm
          "gen64": dict(N=64, L=2, N_T=3, K=1, seed=0x6772617065_65, general=True)}


class Lcg:
    """Knuth's MMIX generator, 64 bits; integers from the top bits"""

    def __init__(self, seed):
        self.x = seed & 0xFFFFFFFFFFFFFFFF

    def ints(self, n, lo, hi):
        out = np.empty(n, dtype=np.int64)
        for i in range(n):
            self.x = (6364136223846793005 * self.x + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            out[i] = lo + (self.x >> 33) % (hi - lo + 1)
        return out


def make_inputs(name):
    sh = SHAPES[name]
    N, L, N_T, K = sh["N"], sh["L"], sh["N_T"], sh["K"]
    g = Lcg(sh["seed"])
    q = 4096.0

    def herm(scale):   # (X + X^dagger) / 2 with X = (a + i b) / 4096, |a|, |b| <= scale * 4096: entries are multiples of 2^-13
        a = g.ints(N * N, -int(scale * q), int(scale * q)).reshape(N, N)
        b = g.ints(N * N, -int(scale * q), int(scale * q)).reshape(N, N)
        X = (a + 1j * b) / q
        if sh.get("general"):
            return X * 1.5                   # (spectral radius ~ 0.45 ... 0.9; an exact factor)
        return (X + X.conj().T) / 2

    s0 = 1.0 / (2.0 * np.sqrt(N))                  # spectral radius of the drift ~ 1, of a control ~ 0.5
    H0 = np.stack([herm(s0) for _ in range(K)])
    Hc = np.stack([herm(0.5 * s0) for _ in range(L)])

    def state():
        return (g.ints(N, -4096, 4096) + 1j * g.ints(N, -4096, 4096)) / q / np.sqrt(2.0 * N) * np.sqrt(3.0)

    # (the states are NOT normalised -- nothing in the path needs it, optimize.jl:722 -- but scaled to a norm near 1; the
    # scale factor is a double, and the doubles below are what every consumer reads)
    psi0 = np.stack([state() for _ in range(K)])
    target = np.stack([state() for _ in range(K)])
    steps = np.array([0.75, 1.0, 0.875, 1.125][:N_T])
    tlist = np.concatenate([[0.0], np.cumsum(steps)])
    pulse = g.ints(L * N_T, -1024, 1024) / q + 0.125      # control-major [l][n] (workspace.jl:159-162)
    weights = np.array([0.75, 1.25][:K])
    return dict(N=N, L=L, N_T=N_T, K=K, H0=H0, Hc=Hc, psi0=psi0, target=target, tlist=tlist, pulsevals=pulse, weights=weights)
