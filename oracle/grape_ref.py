"""ctypes wrapper of oracle/libgrape_ref.so (the C restatement, see grape_ref.c).

TEST INFRASTRUCTURE ONLY (checker in tests/ and smoke(); timed CPU baseline in bench.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GRADGEN, TAYLOR = 0, 1


def build(force=False):
    out = os.path.join(_HERE, "libgrape_ref.so")
    src = os.path.join(_HERE, "grape_ref.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libgrape_ref.so"], check=True, capture_output=True)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libgrape_ref.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.grape_ref_eval.restype = C.c_int
        _lib.grape_ref_expm.restype = C.c_int
        _lib.grape_ref_max_threads.restype = C.c_int
    return _lib


def set_taylor(max_order=100, tolerance=1e-16, check_convergence=True):
    """taylor_grad_max_order / _tolerance / _check_convergence of the reference (optimize.jl:914-918) for the calls that
    follow (process-wide, like set_balance); call without arguments to restore the defaults."""
    lib().grape_ref_set_taylor(C.c_int(int(max_order)), C.c_double(float(tolerance)), C.c_int(int(bool(check_convergence))))


def expm(A):
    A = np.asarray(A, dtype=np.complex128)
    n = A.shape[0]
    Af = np.asfortranarray(A.copy())
    E = np.zeros((n, n), dtype=np.complex128, order="F")
    w = np.zeros(6 * n * n, dtype=np.complex128)
    s = C.c_int(0)
    order = lib().grape_ref_expm(n, Af.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p),
                                 w.ctypes.data_as(C.c_void_p), C.byref(s))
    return np.ascontiguousarray(E), order, s.value


def evaluate(H0, Hc, tlist, pulsevals, psi0, target, weights=None, functional=0,
             gradient_method=GRADGEN, gradient=True, nthreads=0, want_parts=False, D=None, lambda_b=1.0):
    """Same array conventions as grape_jl_amd.GrapeHip (H0[k][row, col])."""
    H0 = np.asarray(H0)
    K, N = H0.shape[0], H0.shape[1]
    Hc = np.asarray(Hc)
    per_traj = Hc.ndim == 4
    L = Hc.shape[1] if per_traj else Hc.shape[0]
    tl = np.ascontiguousarray(tlist, dtype=np.float64)
    N_T = len(tl) - 1
    H0c = np.ascontiguousarray(np.swapaxes(H0, -1, -2), dtype=np.complex128)  # column-major
    Hcc = np.ascontiguousarray(np.swapaxes(Hc, -1, -2), dtype=np.complex128)
    p0 = np.ascontiguousarray(psi0, dtype=np.complex128)
    tg = np.ascontiguousarray(target, dtype=np.complex128)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    x = np.ascontiguousarray(pulsevals, dtype=np.float64)
    J = C.c_double(0.0)
    G = np.zeros(L * N_T) if gradient else None
    tau = np.zeros(K, dtype=np.complex128)
    psiT = np.zeros((K, N), dtype=np.complex128)
    tgr = np.zeros((K, L, N_T), dtype=np.complex128)
    vp = C.c_void_p
    if D is not None:
        D = np.asarray(D)
        dper = D.ndim == 3
        Dc = np.ascontiguousarray(np.swapaxes(D, -1, -2), dtype=np.complex128)
        lib().grape_ref_eval_b.restype = C.c_int
        rc = lib().grape_ref_eval_b(
            C.c_int(N), C.c_int(L), C.c_int(K), C.c_int(N_T), tl.ctypes.data_as(vp), H0c.ctypes.data_as(vp),
            Hcc.ctypes.data_as(vp), C.c_int(int(per_traj)), p0.ctypes.data_as(vp), tg.ctypes.data_as(vp),
            None if w is None else w.ctypes.data_as(vp), C.c_int(functional), C.c_int(gradient_method),
            x.ctypes.data_as(vp), C.byref(J), None if G is None else G.ctypes.data_as(vp),
            tau.ctypes.data_as(vp), psiT.ctypes.data_as(vp), tgr.ctypes.data_as(vp), C.c_int(nthreads),
            Dc.ctypes.data_as(vp), C.c_int(int(dper)), C.c_double(lambda_b))
        if rc:
            raise RuntimeError(f"grape_ref_eval_b failed with code {rc}")
        if want_parts:
            return J.value, G, tau, dict(psiT=psiT, tau_grads=tgr)
        return J.value, G, tau
    rc = lib().grape_ref_eval(
        C.c_int(N), C.c_int(L), C.c_int(K), C.c_int(N_T), tl.ctypes.data_as(vp), H0c.ctypes.data_as(vp),
        Hcc.ctypes.data_as(vp), C.c_int(int(per_traj)), p0.ctypes.data_as(vp), tg.ctypes.data_as(vp),
        None if w is None else w.ctypes.data_as(vp), C.c_int(functional), C.c_int(gradient_method),
        x.ctypes.data_as(vp), C.byref(J), None if G is None else G.ctypes.data_as(vp),
        tau.ctypes.data_as(vp), psiT.ctypes.data_as(vp), tgr.ctypes.data_as(vp), C.c_int(nthreads))
    if rc:
        raise RuntimeError(f"grape_ref_eval failed with code {rc}")
    if want_parts:
        return J.value, G, tau, dict(psiT=psiT, tau_grads=tgr)
    return J.value, G, tau


def evaluate_chi(H0, Hc, tlist, pulsevals, psi0, target, chi, weights=None, gradient_method=GRADGEN, nthreads=0,
                 D=None, lambda_b=1.0):
    """Gradient from caller-supplied boundary states chi[k] = chi_k(T) (user-defined J_T / chi pair,
    src/optimize.jl:845-855).  Returns (G, tau, psiT, tau_grads)."""
    H0 = np.asarray(H0)
    K, N = H0.shape[0], H0.shape[1]
    Hc = np.asarray(Hc)
    per_traj = Hc.ndim == 4
    L = Hc.shape[1] if per_traj else Hc.shape[0]
    tl = np.ascontiguousarray(tlist, dtype=np.float64)
    N_T = len(tl) - 1
    H0c = np.ascontiguousarray(np.swapaxes(H0, -1, -2), dtype=np.complex128)
    Hcc = np.ascontiguousarray(np.swapaxes(Hc, -1, -2), dtype=np.complex128)
    p0 = np.ascontiguousarray(psi0, dtype=np.complex128)
    tg = np.ascontiguousarray(target, dtype=np.complex128)
    ch = np.ascontiguousarray(chi, dtype=np.complex128)
    assert ch.shape == (K, N)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    x = np.ascontiguousarray(pulsevals, dtype=np.float64)
    J = C.c_double(0.0)
    G = np.zeros(L * N_T)
    tau = np.zeros(K, dtype=np.complex128)
    psiT = np.zeros((K, N), dtype=np.complex128)
    tgr = np.zeros((K, L, N_T), dtype=np.complex128)
    vp = C.c_void_p
    Dc, dper = None, False
    if D is not None:
        D = np.asarray(D)
        dper = D.ndim == 3
        Dc = np.ascontiguousarray(np.swapaxes(D, -1, -2), dtype=np.complex128)
    lib().grape_ref_eval_chi.restype = C.c_int
    rc = lib().grape_ref_eval_chi(
        C.c_int(N), C.c_int(L), C.c_int(K), C.c_int(N_T), tl.ctypes.data_as(vp), H0c.ctypes.data_as(vp),
        Hcc.ctypes.data_as(vp), C.c_int(int(per_traj)), p0.ctypes.data_as(vp), tg.ctypes.data_as(vp),
        None if w is None else w.ctypes.data_as(vp), C.c_int(gradient_method), x.ctypes.data_as(vp), C.byref(J),
        G.ctypes.data_as(vp), tau.ctypes.data_as(vp), psiT.ctypes.data_as(vp), tgr.ctypes.data_as(vp),
        C.c_int(nthreads), None if Dc is None else Dc.ctypes.data_as(vp), C.c_int(int(dper)), C.c_double(lambda_b),
        ch.ctypes.data_as(vp))
    if rc:
        raise RuntimeError(f"grape_ref_eval_chi failed with code {rc}")
    return G, tau, psiT, tgr


def max_threads():
    return lib().grape_ref_max_threads()


_blas = None


def use_openblas(on=True):
    """Put the OpenBLAS that scipy bundles (zgemm_, zgesv_; ONE BLAS thread: the parallelism stays the OpenMP loop over
    trajectories) underneath the dense kernels of the restatement, or go back to the plain C loops.  Returns a description
    of the library, or None if it cannot be found.  Timed CPU baseline of bench.py only."""
    global _blas
    if not on:
        lib().grape_ref_set_blas(None, None)
        return None
    if _blas is None:
        import glob
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
        for path in cands:
            try:
                so = C.CDLL(path, mode=C.RTLD_GLOBAL)
                zgemm, zgesv = C.cast(so.scipy_zgemm_, C.c_void_p), C.cast(so.scipy_zgesv_, C.c_void_p)
                so.scipy_openblas_set_num_threads(1)
                so.scipy_openblas_get_config.restype = C.c_char_p
                _blas = (so, zgemm, zgesv, "scipy-bundled " + so.scipy_openblas_get_config().decode().strip())
                break
            except (OSError, AttributeError):
                continue
    if _blas is None:
        return None
    lib().grape_ref_set_blas(_blas[1], _blas[2])
    return _blas[3]
