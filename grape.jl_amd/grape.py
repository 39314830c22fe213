"""Host-side mirror of the reference interface around the hot path.

Mirrors (names, argument meaning, side effects, error behaviour) of
/root/reference/src/optimize.jl and /root/reference/src/workspace.jl for the part of the
interface that touches the accelerated path:

* ``Trajectory`` / ``hamiltonian``   -- QuantumControl.Trajectory, QuantumPropagators.hamiltonian
* ``GrapeWrk``                       -- src/workspace.jl:78-362 (pulsevals layout :159-162, bounds :199-212)
* ``evaluate_functional``            -- src/optimize.jl:696-768
* ``evaluate_gradient_b`` (``evaluate_gradient!``) -- src/optimize.jl:824-1014
* ``optimize`` / ``GrapeResult``     -- src/optimize.jl:73-144, 185-228, src/result.jl:43-116,
                                       L-BFGS-B loop of ext/GRAPELBFGSBExt.jl:18-147 (via scipy's
                                       L-BFGS-B, the same L-BFGS-B 3.0 code family as LBFGSB.jl)

The arithmetic is never done here: every evaluation goes through the C ABI
(``GrapeHip.eval`` == ``fg!``).  ``backend=`` exists so that CPU-only tests can drive this host
logic with a stand-in evaluator; the default backend is the HIP library and there is no fallback.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import api


# ---- controls / generators ------------------------------------------------------------------
def discretize_on_midpoints(control, tlist):
    """QuantumPropagators.Controls.discretize_on_midpoints (docs/src/background.md:55)."""
    tlist = np.asarray(tlist, dtype=np.float64)
    nt = len(tlist)
    if callable(control):
        tm = np.empty(nt - 1)
        tm[0], tm[-1] = tlist[0], tlist[-1]
        if nt > 3:
            tm[1:-1] = tlist[1:-2] + 0.5 * (tlist[2:-1] - tlist[1:-2])
        return np.array([float(control(t)) for t in tm])
    c = np.asarray(control, dtype=np.float64)
    if len(c) == nt - 1:
        return c.copy()
    if len(c) == nt:
        v = np.empty(nt - 1)
        v[0], v[-1] = c[0], c[-1]
        if nt > 3:
            v[1:-1] = 0.5 * (c[1:-2] + c[2:-1])
        return v
    raise ValueError("control array must have length(tlist) or length(tlist)-1 elements")


def discretize(vals, tlist):
    """Pulse values on the intervals -> values on the points of ``tlist`` (finalize_result!, :219-228)."""
    vals = np.asarray(vals, dtype=np.float64)
    nt = len(tlist)
    if len(vals) == nt:
        return vals.copy()
    out = np.empty(nt)
    out[0], out[-1] = vals[0], vals[-1]
    out[1:-1] = 0.5 * (vals[:-1] + vals[1:])
    return out


class ShapedAmplitude:
    """``QuantumPropagators.Amplitudes.ShapedAmplitude(control; shape)``: the physical amplitude a(t) = S(t) eps(t) of
    an optimisable control eps(t) under a static shape S(t) (docs/src/tutorial.md:75-107).  GRAPE optimises the pulse
    values of ``control``; the shape reaches the kernels as ``grape_problem.shape[l][n]`` (it scales H_l in the
    propagator and mu_l = dH/d eps_l in the gradient)."""

    def __init__(self, control, shape):
        self.control = control
        self.shape = shape


class NonlinearAmplitude:
    """Control amplitude a(eps, t) = S(t) f(eps(t)) with a non-linear dependence on the optimisable control (the
    reference obtains mu = dH/d eps from ``get_control_derivs``, src/workspace.jl:286; docs/src/background.md:402).
    The kernels see the values f(eps_nl) as their pulse (and S as the shape); the gradient with respect to eps
    follows from the chain rule, dJ/d eps_nl = f'(eps_nl) dJ/d f_nl, applied on the host."""

    def __init__(self, control, func, dfunc, shape=None):
        self.control = control
        self.func = func
        self.dfunc = dfunc
        self.shape = shape


class FixedAmplitude:
    """A time-dependent coefficient that is NOT an optimisable control: an amplitude type whose ``get_controls`` is empty
    and whose value the reference obtains per interval from ``evaluate(ampl, tlist, n)`` (the generator is re-evaluated on
    every step, /root/reference/src/optimize.jl:732, 881, 937-945; ExpProp's ``evaluate!``).  ``values``: a callable a(t)
    or an array on ``tlist`` / on its midpoints.  ``grape_problem`` knows one constant drift per trajectory, so the term
    crosses the boundary as a PSEUDO-CONTROL: an extra control operator whose pulse values are fixed on the host and whose
    rows of the gradient are dropped."""

    def __init__(self, values):
        self.values = values


class _PseudoControlBackend:
    """Innermost wrapper of a handle that carries P pseudo-controls behind the L optimised ones (see FixedAmplitude): the
    fixed pulse values are appended to every pulse vector on the way in, the gradient rows of the pseudo-controls are cut on
    the way out.  The handle itself works on L + P controls."""

    def __init__(self, inner, fixed_vals, L, N_T):
        self.inner, self.L_opt, self.N_T = inner, L, N_T
        self.fixed = np.ascontiguousarray(fixed_vals, dtype=np.float64).reshape(-1)

    def _x(self, x):
        return np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1), self.fixed])

    def _g(self, G):
        return None if G is None else G[: self.L_opt * self.N_T].copy()

    def eval(self, x, gradient=True, want_psiT=False):
        res = list(self.inner.eval(self._x(x), gradient=gradient, want_psiT=want_psiT))
        res[1] = self._g(res[1])
        return tuple(res)

    def forward(self, x):
        return self.inner.forward(self._x(x))

    def backward(self, f_total):
        return self._g(self.inner.backward(f_total))

    def backward_chi(self, chi):
        return self._g(self.inner.backward_chi(chi))

    def backward_xi(self, xi, lambda_b, f_total=None, chi=None):
        return self._g(self.inner.backward_xi(xi, lambda_b, f_total=f_total, chi=chi))

    def tau_grads(self):
        return self.inner.tau_grads()[:, : self.L_opt]

    def __getattr__(self, name):
        return getattr(self.inner, name)


class _AmplitudeBackend:
    """Wraps a backend whose pulse values are the amplitudes f_l(eps): maps eps -> f(eps) on the way in and applies
    the chain rule to the gradient on the way out."""

    def __init__(self, inner, funcs, dfuncs, N_T):
        self.inner, self.funcs, self.dfuncs, self.N_T = inner, funcs, dfuncs, N_T

    def _map(self, x, fs):
        out = np.array(x, dtype=np.float64, copy=True)
        for l, f in enumerate(fs):
            if f is not None:
                seg = slice(l * self.N_T, (l + 1) * self.N_T)
                out[seg] = np.vectorize(f, otypes=[np.float64])(x[seg])
        return out

    def eval(self, x, gradient=True, want_psiT=False):
        res = list(self.inner.eval(self._map(x, self.funcs), gradient=gradient, want_psiT=want_psiT))
        if gradient and res[1] is not None:
            d = self._map(x, self.dfuncs)
            for l, f in enumerate(self.dfuncs):
                if f is None:
                    d[l * self.N_T:(l + 1) * self.N_T] = 1.0
            res[1] = res[1] * d
        return tuple(res)

    def __getattr__(self, name):
        return getattr(self.inner, name)


def hamiltonian(H0, *terms):
    """``hamiltonian(H0, (H1, eps1), (H2, eps2), ...)``: H(t) = H0 + sum_l a_l(t) H_l, a_l a control or a
    ``ShapedAmplitude`` of one."""
    ops, ctrls, shapes, nonlinear, fixed = [], [], [], {}, []
    for op, ctrl in terms:
        if isinstance(ctrl, FixedAmplitude):      # time-dependent, but not a control (get_controls(ampl) == ())
            fixed.append((np.asarray(op, dtype=np.complex128), ctrl))
            continue
        ops.append(np.asarray(op, dtype=np.complex128))
        if isinstance(ctrl, (ShapedAmplitude, NonlinearAmplitude)):
            ctrls.append(ctrl.control)
            shapes.append(ctrl.shape)
            if isinstance(ctrl, NonlinearAmplitude):
                nonlinear[id(ctrl.control)] = (ctrl.func, ctrl.dfunc)
        else:
            ctrls.append(ctrl)
            shapes.append(None)
    return Generator(np.asarray(H0, dtype=np.complex128), ops, ctrls, shapes, nonlinear, fixed)


@dataclass
class Generator:
    drift: np.ndarray
    ops: List[np.ndarray]
    controls: list
    shapes: list = None   # per term: None or the static shape S(t) of a ShapedAmplitude
    nonlinear: dict = None  # id(control) -> (f, f') for controls that enter through a NonlinearAmplitude
    fixed: list = None      # [(operator, FixedAmplitude)]: time-dependent terms that are not controls


@dataclass
class Trajectory:
    initial_state: np.ndarray
    generator: Generator
    target_state: Optional[np.ndarray] = None
    weight: float = 1.0
    # per-step propagation callbacks (the `callback` / `observables` propagation keyword arguments of a trajectory, called at
    # src/optimize.jl:733-737 forward, :882-887 and :973-978 backward): `prop_callback(propagator, observables)` is called
    # once per time step with a PropagatorView.  On the HIP path the steps do not run on the host: the calls are synthesised
    # AFTER the sweep from the stored states (grape_get_storage), in the reference's order
    prop_callback: Optional[Callable] = None
    prop_observables: object = None


@dataclass
class PropagatorView:
    """what a propagation callback sees of the propagator (QuantumPropagators' `propagator.state`, `.t`, `.tlist`, `.backward`)"""
    state: np.ndarray
    t: float
    tlist: np.ndarray
    n: int                 # index into tlist of the time point just reached
    k: int                 # trajectory
    backward: bool = False


def get_controls(trajectories):
    """Unique controls (by identity) in order of first appearance."""
    controls = []
    for traj in trajectories:
        for c in traj.generator.controls:
            if not any(c is d for d in controls):
                controls.append(c)
    return controls


# ---- functionals (QuantumControl.Functionals) ---------------------------------------------------
def J_T_sm(Psi, trajectories, tau=None):
    K = len(trajectories)
    tau = _taus(Psi, trajectories) if tau is None else tau
    f = sum(t.weight * tk for t, tk in zip(trajectories, tau))
    return 1.0 - abs(f) ** 2 / K**2


def J_T_ss(Psi, trajectories, tau=None):
    K = len(trajectories)
    tau = _taus(Psi, trajectories) if tau is None else tau
    return 1.0 - sum(t.weight * abs(tk) ** 2 for t, tk in zip(trajectories, tau)) / K


def J_T_re(Psi, trajectories, tau=None):
    K = len(trajectories)
    tau = _taus(Psi, trajectories) if tau is None else tau
    return 1.0 - sum(t.weight * np.real(tk) for t, tk in zip(trajectories, tau)) / K


def _taus(Psi, trajectories):
    return [np.vdot(t.target_state, p) for t, p in zip(trajectories, Psi)]


_FUNCTIONAL_CODE = {J_T_sm: api.J_T_SM, J_T_ss: api.J_T_SS, J_T_re: api.J_T_RE}


# ---- result -----------------------------------------------------------------------------------
@dataclass
class GrapeResult:
    """src/result.jl:43-67."""
    tlist: np.ndarray
    iter_start: int = 0
    iter_stop: int = 5000
    iter: int = 0
    secs: float = 0.0
    tau_vals: np.ndarray = None
    J_T: float = 0.0
    J_T_prev: float = 0.0
    J_a: float = 0.0
    J_a_prev: float = 0.0
    J_b: float = 0.0
    J_b_prev: float = 0.0
    guess_controls: list = field(default_factory=list)
    optimized_controls: list = field(default_factory=list)
    states: list = field(default_factory=list)
    start_local_time: float = 0.0
    end_local_time: float = 0.0
    records: list = field(default_factory=list)
    converged: bool = False
    f_calls: int = 0
    fg_calls: int = 0
    message: str = "in progress"

    def __repr__(self):
        return f"GrapeResult<{self.message}>"


# ---- workspace --------------------------------------------------------------------------------
class GrapeWrk:
    """src/workspace.jl:78-362 -- owns ``pulsevals`` (control-major) and the device handle."""

    def __init__(self, trajectories: Sequence[Trajectory], tlist, backend=None, **kwargs):
        self.trajectories = list(trajectories)
        self.tlist = np.asarray(tlist, dtype=np.float64)
        self.kwargs = dict(kwargs)
        self.controls = get_controls(self.trajectories)
        if len(self.controls) == 0:
            raise ValueError("no controls in trajectories: cannot optimize")  # workspace.jl:155-157
        if "J_T" not in self.kwargs:
            raise ValueError("`optimize` for `method=GRAPE` must be passed the functional `J_T`.")  # :298-303
        if any(t.prop_callback is not None for t in self.trajectories) and self.kwargs.get("lambda_b", 1.0) != 0.0 and \
                (self.kwargs.get("g_b") is not None or self.kwargs.get("state_penalty") is not None):
            # the reference calls a backward callback right after prop_step!, BEFORE the xi term of the running cost is added to
            # the state (src/optimize.jl:973-985); the stored backward states the callbacks are synthesised from already hold
            # that term (round-5 advisor finding): refused rather than handed different states than upstream
            raise ValueError("per-step propagation callbacks cannot be combined with a state running cost (g_b / state_penalty) "
                             "on the HIP path: the stored backward states already include the xi term")
        L, N_T, K = len(self.controls), len(self.tlist) - 1, len(self.trajectories)
        self.L, self.N_T, self.K = L, N_T, K
        # pulsevals = vcat(discretize_on_midpoints(ctrl, tlist)...)   (workspace.jl:159-162)
        self.pulsevals = np.concatenate([discretize_on_midpoints(c, self.tlist) for c in self.controls])
        self.pulsevals_guess = self.pulsevals.copy()
        self.gradient = np.zeros(L * N_T)
        self.grad_J_Tb = np.zeros(L * N_T)
        self.grad_J_a = np.zeros(L * N_T)
        self.J_parts = np.zeros(3)
        self.fg_count = [0, 0]
        lo = self.kwargs.get("lower_bound", -np.inf)
        hi = self.kwargs.get("upper_bound", np.inf)
        self.lower_bounds = np.full(L * N_T, lo, dtype=np.float64)
        self.upper_bounds = np.full(L * N_T, hi, dtype=np.float64)
        pb = self.kwargs.get("pulse_options", None)
        if pb:
            # per-control bounds: pulse_options[control][:upper_bounds / :lower_bounds] (workspace.jl:204-214; the
            # singular keys are accepted as well).  The reference writes them through the views `[l:L:end]`, i.e.
            # INTERLEAVED, although pulsevals is control-major (workspace.jl:159-162) -- for L > 1 the bounds of
            # control l land on every L-th value of the whole vector.  Default here: the evident intent (the values
            # of control l); `reference_bounds_layout=True` reproduces the reference's stride literally
            # (INTEGRATION.md section 5).  For L = 1 both are the same.
            literal = bool(self.kwargs.get("reference_bounds_layout", False))
            for l, c in enumerate(self.controls):
                opt = next((v for k, v in pb if k is c), {})
                for keys, arr in ((("lower_bounds", "lower_bound"), self.lower_bounds),
                                  (("upper_bounds", "upper_bound"), self.upper_bounds)):
                    for key in keys:
                        if key in opt:
                            if literal:
                                arr[l::L] = opt[key]
                            else:
                                arr[l * N_T:(l + 1) * N_T] = opt[key]
        prev = self.kwargs.get("continue_from")
        if prev is not None:
            # src/workspace.jl:167-186: continue a previous optimization -- its result object is reused (counters and
            # records kept), the pulses are re-discretised from its optimized controls
            self.result = prev
            prev.iter_stop = self.kwargs.get("iter_stop", 5000)
            prev.converged = False
            prev.message = "in progress"
            self.pulsevals = np.concatenate([discretize_on_midpoints(c, prev.tlist) for c in prev.optimized_controls])
            self.pulsevals_guess = self.pulsevals.copy()
            prev.start_local_time = prev.end_local_time = time.time()
            self.backend = backend if backend is not None else self._make_hip_backend()
            if any(t.prop_callback is not None for t in self.trajectories) and hasattr(self.backend, "set_fused_sweeps"):
                self.backend.set_fused_sweeps(False)
            return
        self.result = GrapeResult(tlist=self.tlist.copy(), iter_start=self.kwargs.get("iter_start", 0),
                                  iter_stop=self.kwargs.get("iter_stop", 5000))
        self.result.iter = self.result.iter_start
        self.result.tau_vals = np.zeros(K, dtype=np.complex128)
        self.result.guess_controls = [discretize(self.pulsevals[l * N_T:(l + 1) * N_T], self.tlist) for l in range(L)]
        self.result.optimized_controls = [g.copy() for g in self.result.guess_controls]
        self.result.states = [np.zeros_like(t.initial_state, dtype=np.complex128) for t in self.trajectories]
        self.result.start_local_time = self.result.end_local_time = time.time()
        self.backend = backend if backend is not None else self._make_hip_backend()
        if any(t.prop_callback is not None for t in self.trajectories) and hasattr(self.backend, "set_fused_sweeps"):
            self.backend.set_fused_sweeps(False)     # (the callbacks see the backward states of the reference: see _propagation_callbacks)

    def _make_hip_backend(self):
        J_T = self.kwargs["J_T"]
        custom = J_T not in _FUNCTIONAL_CODE
        if custom and not callable(self.kwargs.get("chi")):
            # the reference derives chi by automatic differentiation when it is missing (workspace.jl:306-308);
            # here a user-defined J_T has to come with its chi
            raise ValueError("a user-defined J_T needs the matching `chi(Psi, trajectories; tau)` (optimize.jl:845-855)")
        trajs = self.trajectories
        H0 = np.stack([t.generator.drift for t in trajs])
        per_traj = []
        for t in trajs:
            ops = []
            for c in self.controls:
                op = [o for o, cc in zip(t.generator.ops, t.generator.controls) if cc is c]
                ops.append(sum(op) if op else np.zeros_like(t.generator.drift))
            per_traj.append(np.stack(ops))
        # time-dependent terms that are not controls (FixedAmplitude): one pseudo-control per amplitude object, behind the
        # optimised ones; its operator per trajectory, its pulse values fixed here
        fixed_amps = []
        for t in trajs:
            for _, a in (t.generator.fixed or []):
                if not any(a is b for b in fixed_amps):
                    fixed_amps.append(a)
        if fixed_amps:
            for i, t in enumerate(trajs):
                extra = []
                for a in fixed_amps:
                    op = [o for o, aa in (t.generator.fixed or []) if aa is a]
                    extra.append(sum(op) if op else np.zeros_like(t.generator.drift))
                per_traj[i] = np.concatenate([per_traj[i], np.stack(extra)])
            fixed_vals = np.stack([discretize_on_midpoints(a.values, self.tlist) for a in fixed_amps])
        Hc = np.stack(per_traj)
        if all(np.array_equal(Hc[0], h) for h in Hc[1:]):
            Hc = Hc[0]
        # static shapes of ShapedAmplitudes: one S_l per control (the ABI scales H_l and mu_l by shape[l][n])
        shape = None
        for l, c in enumerate(self.controls):
            found = []
            for t in trajs:
                shp = t.generator.shapes or [None] * len(t.generator.controls)
                found += [sh for sh, cc in zip(shp, t.generator.controls) if cc is c]
            shaped = [sh for sh in found if sh is not None]
            if shaped:
                vals = [discretize_on_midpoints(sh, self.tlist) for sh in found if sh is not None]
                if len(shaped) != len(found) or any(not np.array_equal(vals[0], v) for v in vals[1:]):
                    raise ValueError("a control must enter every term with the same shape")
                if shape is None:
                    shape = np.ones((self.L + len(fixed_amps), self.N_T))
                shape[l] = vals[0]
        method = {"gradgen": api.GRAD_GRADGEN, "taylor": api.GRAD_TAYLOR}.get(
            self.kwargs.get("gradient_method", "gradgen"))
        if method is None:
            raise ValueError(f"Invalid gradient_method={self.kwargs.get('gradient_method')!r} not in (gradgen, taylor)")
        # prop_method keyword of the reference (src/workspace.jl:222-232): ExpProp materialises the propagators;
        # the polynomial methods (Cheby / Newton, README.md:55) are served by the matrix-free series kernel
        pm = self.kwargs.get("prop_method", "ExpProp")
        pm = getattr(pm, "__name__", pm)
        prop = {"expprop": api.PROP_EXP, "exp": api.PROP_EXP, "cheby": api.PROP_SERIES, "newton": api.PROP_SERIES,
                "series": api.PROP_SERIES}.get(str(pm).lower().lstrip(":"))
        if prop is None:
            raise ValueError(f"prop_method={pm!r} not in (ExpProp, Cheby, Newton, series)")
        funcs, dfuncs = [], []
        for c in self.controls:
            fd = [t.generator.nonlinear[id(c)] for t in trajs if t.generator.nonlinear and id(c) in t.generator.nonlinear]
            funcs.append(fd[0][0] if fd else None)
            dfuncs.append(fd[0][1] if fd else None)
        wrap = (lambda b: _AmplitudeBackend(b, funcs, dfuncs, self.N_T)) if any(f is not None for f in funcs) else (lambda b: b)
        # keyword arguments g_b / xi of the reference (src/docstring.jl): an arbitrary state running cost.  (state_penalty=D
        # selects the device-side fast path of the family g_b = <Psi|D|Psi>.)
        g_b, xi_fn = self.kwargs.get("g_b"), self.kwargs.get("xi")
        if g_b is not None and self.kwargs.get("lambda_b", 1.0) != 0.0:
            if not callable(xi_fn):
                # (the reference builds xi by automatic differentiation when it is missing, src/workspace.jl:313-316)
                raise ValueError("a state running cost g_b needs the matching `xi(state, trajectory, tlist, n)`")
            if self.kwargs.get("state_penalty") is not None:
                raise ValueError("give either state_penalty (g_b = <Psi|D|Psi>) or the callbacks g_b / xi, not both")
            inner_wrap_rc = wrap
            tl, lam = self.tlist, self.kwargs.get("lambda_b", 1.0)
            wrap = lambda b: inner_wrap_rc(_CustomRunningCostBackend(  # noqa: E731
                b, g_b, xi_fn, lam, trajs, tl, J_T if custom else None, self.kwargs["chi"] if custom else None))
        elif custom:
            inner_wrap = wrap
            wrap = lambda b: inner_wrap(_CustomChiBackend(b, J_T, self.kwargs["chi"], trajs))  # noqa: E731
        # trajectories without a target_state (optimize.jl:753: tau = NaN; legal with a user-defined J_T / chi): no target
        # array crosses the boundary.  A mix of trajectories with and without one keeps zeros in the gaps.
        if fixed_amps:       # innermost (directly on the handle): every other wrapper sees a backend of the L optimised controls
            outer_wrap_pc = wrap
            wrap = lambda b: outer_wrap_pc(_PseudoControlBackend(b, fixed_vals, self.L, self.N_T))  # noqa: E731
        if custom and all(t.target_state is None for t in trajs):
            targets = None
        else:
            targets = np.stack([t.target_state if t.target_state is not None else np.zeros_like(t.initial_state) for t in trajs])
        return wrap(api.GrapeHip(H0, Hc, self.tlist, np.stack([t.initial_state for t in trajs]),
                            targets,
                            prop_method=prop, prop_tolerance=self.kwargs.get("prop_tolerance", 0.0), shape=shape,
                            weights=np.array([t.weight for t in trajs], dtype=np.float64),
                            functional=_FUNCTIONAL_CODE.get(J_T, api.J_T_SM), gradient_method=method,
                            devices=self.kwargs.get("devices"),
                            chi_min_norm=self.kwargs.get("chi_min_norm", 0.0),
                            taylor_max_order=self.kwargs.get("taylor_grad_max_order", 0),
                            taylor_tolerance=self.kwargs.get("taylor_grad_tolerance", 0.0),
                            taylor_check_convergence=self.kwargs.get("taylor_grad_check_convergence", True),  # optimize.jl:917-918
                            device=self.kwargs.get("device", 0),
                            # g_b / xi of the expectation-value family (test_state_running_cost.jl:32-40):
                            # g_b = <Psi|D|Psi>, xi = -D Psi, given as the operator D itself
                            D=self.kwargs.get("state_penalty"), lambda_b=self.kwargs.get("lambda_b", 1.0)))


class _CustomChiBackend:
    """A user-defined functional on the HIP path (optimize.jl:757-760, 845-855): the forward sweep runs on the device,
    ``J_T(Psi, trajectories; tau)`` and ``chi(Psi, trajectories; tau)`` are evaluated on the host from the final
    states, and the backward sweep + gradient run on the device from the chi the user's function returned
    (``grape_forward`` / ``grape_get_final_states`` / ``grape_backward_chi``)."""

    def __init__(self, inner, J_T, chi, trajectories):
        self.inner, self.J_T, self.chi, self.trajectories = inner, J_T, chi, trajectories
        # the forward call must not run the (unit-target) backward sweep in the same launch: the caller's chi arrives
        # afterwards and the backward sweep runs then, once
        self.inner.set_fused_sweeps(False)

    def eval(self, x, gradient=True, want_psiT=False):
        tau = self.inner.forward(x)
        psiT = self.inner.final_states()
        J = float(self.J_T(list(psiT), self.trajectories, tau=list(tau)))
        J += getattr(self.inner, "lambda_b", 0.0) * float(self.inner.sums()[4])
        G = None
        if gradient:
            chi = np.stack([np.asarray(c, dtype=np.complex128) for c in self.chi(list(psiT), self.trajectories, tau=list(tau))])
            G = self.inner.backward_chi(chi)
        return (J, G, tau, psiT) if want_psiT else (J, G, tau)

    def __getattr__(self, name):
        return getattr(self.inner, name)


class _CustomRunningCostBackend:
    """An arbitrary state running cost on the HIP path (the ``g_b`` / ``xi`` keyword arguments of the reference,
    src/docstring.jl; used at src/optimize.jl:727-750, 856-866, 897-908): the forward sweep runs on the device, ``g_b`` and
    ``xi`` are evaluated on the host on the stored forward states (``grape_get_storage``), and the backward sweep takes
    the ``xi`` array (``grape_backward_xi``).  ``chi`` (optional): the user's chi of a user-defined J_T."""

    def __init__(self, inner, g_b, xi, lambda_b, trajectories, tlist, J_T=None, chi=None):
        self.inner, self.g_b, self.xi, self.lambda_b = inner, g_b, xi, float(lambda_b)
        self.trajectories, self.tlist, self.J_T, self.chi = trajectories, np.asarray(tlist, dtype=np.float64), J_T, chi
        self.inner.set_fused_sweeps(False)

    def eval(self, x, gradient=True, want_psiT=False):
        tau = self.inner.forward(x)
        fw = self.inner.storage(0)                      # Psi_k(t_n), [K, N_T+1, N]
        K, NT1, _ = fw.shape
        tl = self.tlist
        J_b = 0.0
        for k in range(K):                              # trapezoid rule of optimize.jl:727-750
            traj = self.trajectories[k]
            J_b += float(self.g_b(fw[k, 0], traj, tl, 1)) * (tl[1] - tl[0]) / 2.0
            for n in range(1, NT1):
                dt = 0.5 * (tl[n + 1] - tl[n - 1]) if n < NT1 - 1 else (tl[-1] - tl[-2]) / 2.0
                J_b += float(self.g_b(fw[k, n], traj, tl, n + 1)) * dt
        psiT = fw[:, -1, :]
        if self.J_T is not None:
            J = float(self.J_T(list(psiT), self.trajectories, tau=list(tau)))
        else:
            from .sharded import functional_value
            J = functional_value(self.inner.functional, self.inner.sums(), self.inner.K_total, 0.0)
        J += self.lambda_b * J_b
        G = None
        if gradient:
            xi = np.zeros_like(fw)
            for k in range(K):
                for n in range(1, NT1):
                    xi[k, n] = np.asarray(self.xi(fw[k, n], self.trajectories[k], tl, n + 1), dtype=np.complex128)
            chi = None
            if self.chi is not None:
                chi = np.stack([np.asarray(c, dtype=np.complex128) for c in self.chi(list(psiT), self.trajectories, tau=list(tau))])
            G = self.inner.backward_xi(xi, self.lambda_b, chi=chi)
        return (J, G, tau, psiT) if want_psiT else (J, G, tau)

    def __getattr__(self, name):
        return getattr(self.inner, name)


def _split_functional(wrk, J, tau):
    """J_parts[1] = J_T from the overlaps, J_parts[3] = lambda_b * sum_k J_b,k (src/optimize.jl:757-766): the backend
    returns their sum; J_T is evaluated again on the host from the final states / tau."""
    J_T = wrk.kwargs["J_T"](getattr(wrk, "_states", None), wrk.trajectories, tau=list(tau))
    wrk.J_parts[0] = float(J_T)
    on = (wrk.kwargs.get("state_penalty") is not None or wrk.kwargs.get("g_b") is not None) and wrk.kwargs.get("lambda_b", 1.0) != 0.0
    wrk.J_parts[2] = float(J - J_T) if on else 0.0


def _propagation_callbacks(wrk, backward):
    """the per-step callbacks of the trajectories (src/optimize.jl:733-737; :882-887 / :973-978), from the stored states: forward
    after every step n = 1 .. N_T with Psi_k(t_n); backward after every step n = N_T .. 1 with chi_k(t_(n-1)) (normalised as the
    backward propagator holds it, src/optimize.jl:867-868).  The backend runs its sweeps one after the other while callbacks
    are registered (the concurrent backward sweep starts from unit targets: other states)."""
    if not any(t.prop_callback is not None for t in wrk.trajectories):
        return
    st = wrk.backend.storage(1 if backward else 0)
    N_T = wrk.N_T
    for k, traj in enumerate(wrk.trajectories):
        if traj.prop_callback is None:
            continue
        steps = range(N_T - 1, -1, -1) if backward else range(1, N_T + 1)
        for n in steps:
            traj.prop_callback(PropagatorView(np.array(st[k, n]), float(wrk.tlist[n]), wrk.tlist, n, k, backward), traj.prop_observables)


def evaluate_functional(pulsevals, wrk: GrapeWrk, count_call=True):
    """src/optimize.jl:696-768 (side effects on wrk as documented there)."""
    if pulsevals is not wrk.pulsevals:
        wrk.pulsevals[:] = pulsevals
    if count_call:
        wrk.result.f_calls += 1
        wrk.fg_count[1] += 1
    J, _, tau, psiT = wrk.backend.eval(wrk.pulsevals, gradient=False, want_psiT=True)
    _propagation_callbacks(wrk, backward=False)
    wrk.result.tau_vals[:] = tau
    wrk._states = psiT
    _split_functional(wrk, J, tau)
    J_a = wrk.kwargs.get("J_a")
    if J_a is not None:
        wrk.J_parts[1] = wrk.kwargs.get("lambda_a", 1.0) * J_a(wrk.pulsevals, wrk.tlist)
    return float(np.sum(wrk.J_parts))


def evaluate_gradient_b(G, pulsevals, wrk: GrapeWrk):
    """``evaluate_gradient!`` -- src/optimize.jl:824-1014."""
    if pulsevals is not wrk.pulsevals:
        wrk.pulsevals[:] = pulsevals
    wrk.result.fg_calls += 1
    wrk.fg_count[0] += 1
    J, g, tau, psiT = wrk.backend.eval(wrk.pulsevals, gradient=True, want_psiT=True)
    _propagation_callbacks(wrk, backward=False)
    _propagation_callbacks(wrk, backward=True)
    wrk.result.tau_vals[:] = tau
    wrk._states = psiT
    _split_functional(wrk, J, tau)
    wrk.grad_J_Tb[:] = g
    G[:] = g
    J_a = wrk.kwargs.get("J_a")
    if J_a is not None:
        lam = wrk.kwargs.get("lambda_a", 1.0)
        wrk.J_parts[1] = lam * J_a(wrk.pulsevals, wrk.tlist)
        grad_J_a = wrk.kwargs.get("grad_J_a")
        if grad_J_a is not None:
            wrk.grad_J_a[:] = grad_J_a(wrk.pulsevals, wrk.tlist)
            G += lam * wrk.grad_J_a
    return float(np.sum(wrk.J_parts))


def update_result(wrk: GrapeWrk, i: int):
    """src/optimize.jl:185-216."""
    res = wrk.result
    for k in range(wrk.K):
        res.states[k] = np.array(wrk._states[k])
    res.J_T_prev, res.J_T = res.J_T, float(wrk.J_parts[0])
    res.J_a_prev, res.J_a = res.J_a, float(wrk.J_parts[1])
    if res.J_a > 0.0:
        res.J_a /= wrk.kwargs.get("lambda_a", 1.0)
    res.J_b_prev, res.J_b = res.J_b, float(wrk.J_parts[2])     # src/optimize.jl:199-204
    lam_b = wrk.kwargs.get("lambda_b", 1.0)
    if res.J_b != 0.0 and lam_b != 0.0:
        res.J_b /= lam_b
    if i > 0:
        res.iter = i
    if i >= res.iter_stop:
        res.converged = True
        res.message = "Reached maximum number of iterations"
    prev = res.end_local_time
    res.end_local_time = time.time()
    res.secs = res.end_local_time - prev


def _apply_convergence_check(result, check_convergence):
    """src/optimize.jl:154-182."""
    if result.converged:
        return
    c = check_convergence(result)
    if isinstance(c, bool):
        result.converged = c
        if c:
            result.message = "Convergence check returned true"
    elif isinstance(c, str):
        if c:
            result.converged = True
            result.message = c


class _Stop(Exception):
    pass


def optimize(trajectories, tlist, backend=None, **kwargs):
    """``GRAPE.optimize(trajectories, tlist; kwargs...)`` -- src/optimize.jl:73-144."""
    from scipy.optimize import minimize

    # callbacks: one callable or a tuple of them, called in order after every iteration; what they return (tuples) is
    # concatenated into one record.  `print_iters` / `print_iter_info` / `store_iter_info` append the iteration table as
    # the last callback, as QuantumControl.optimize does around GRAPE.optimize (set-up of `callback`, src/optimize.jl:92-103;
    # test/test_iterations.jl:43-150).  (A callback that modifies wrk.pulsevals does not feed back into scipy's
    # L-BFGS-B iterate -- LBFGSB.jl works on wrk.pulsevals itself, test_iterations.jl:128-148.)
    cbs = kwargs.get("callback", ())
    cbs = tuple(cbs) if isinstance(cbs, (tuple, list)) else (cbs,)
    if kwargs.get("print_iters", False) or "print_iter_info" in kwargs or "store_iter_info" in kwargs:
        info_print = kwargs.get("print_iter_info", ("iter.", "J_T", "ǁ∇Jǁ", "ǁΔϵǁ", "ΔJ", "FG(F)", "secs"))
        if not kwargs.get("print_iters", True):
            info_print = ()
        cbs = cbs + (make_grape_print_iters(print_iter_info=info_print, store_iter_info=kwargs.get("store_iter_info", ()),
                                            out=kwargs.get("print_iters_out")),)

    def callback(wrk_, it):
        rec = ()
        for cb in cbs:
            r = cb(wrk_, it)
            if r is not None:
                rec = rec + (tuple(r) if isinstance(r, (tuple, list)) else (r,))
        return rec or None

    check_convergence = kwargs.get("check_convergence", lambda res: res)
    wrk = GrapeWrk(trajectories, tlist, backend=backend, **kwargs)
    G = np.zeros_like(wrk.pulsevals)
    state = {"first": True}

    def fg(x):
        J = evaluate_gradient_b(G, x, wrk)
        if state["first"]:  # FG_START: iteration 0 (ext/GRAPELBFGSBExt.jl:100-109)
            state["first"] = False
            wrk.gradient[:] = G
            update_result(wrk, 0)
            info = callback(wrk, 0)
            wrk.fg_count = [0, 0]
            if info:
                wrk.result.records.append(info)
        return J, G.copy()

    def new_x(xk):  # NEW_X (ext/GRAPELBFGSBExt.jl:110-127)
        wrk.pulsevals[:] = xk
        update_result(wrk, wrk.result.iter + 1)
        info = callback(wrk, wrk.result.iter)
        wrk.fg_count = [0, 0]
        if info:
            wrk.result.records.append(info)
        _apply_convergence_check(wrk.result, check_convergence)
        if wrk.result.converged:
            raise _Stop()
        wrk.pulsevals_guess[:] = xk
        wrk.gradient[:] = G

    bounds = None
    if np.any(np.isfinite(wrk.lower_bounds)) or np.any(np.isfinite(wrk.upper_bounds)):
        bounds = list(zip(np.where(np.isfinite(wrk.lower_bounds), wrk.lower_bounds, None),
                          np.where(np.isfinite(wrk.upper_bounds), wrk.upper_bounds, None)))
    try:
        res = minimize(fg, wrk.pulsevals.copy(), jac=True, method="L-BFGS-B", bounds=bounds, callback=new_x,
                       options=dict(maxcor=kwargs.get("lbfgsb_m", 10), ftol=kwargs.get("lbfgsb_factr", 1e1) * 2.2e-16,
                                    gtol=kwargs.get("lbfgsb_pgtol", 1e-15), maxiter=wrk.result.iter_stop + 1,
                                    maxfun=10 * (wrk.result.iter_stop + 1) + 100, maxls=50))
        if wrk.result.message == "in progress":
            wrk.result.message = str(res.message)
    except _Stop:
        pass
    except Exception as exc:  # src/optimize.jl:125-135
        if kwargs.get("rethrow_exceptions", False):
            raise
        wrk.result.message = f"Exception: {exc}"
    # finalize_result! (src/optimize.jl:219-228)
    res_ = wrk.result
    res_.end_local_time = time.time()
    for l in range(wrk.L):
        res_.optimized_controls[l] = discretize(wrk.pulsevals[l * wrk.N_T:(l + 1) * wrk.N_T], res_.tlist)
    return res_


# ---- iteration table ----------------------------------------------------------------------------
_DELTA_HEADERS = {"ΔJ_T", "ΔJ_a", "ΔJ_b", "λ_a⋅ΔJ_a", "λ_b⋅ΔJ_b", "ΔJ", "ǁΔϵǁ", "max|Δϵ|", "ǁΔϵǁ/ǁϵǁ", "∫Δϵ²dt"}


def make_grape_print_iters(print_iter_info=("iter.", "J_T", "ǁ∇Jǁ", "ǁΔϵǁ", "ΔJ", "FG(F)", "secs"), store_iter_info=(),
                           out=None):
    """``make_grape_print_iters`` (src/optimize.jl:310-537): the callback that prints one table row per iteration and
    returns the tuple of ``store_iter_info`` fields for ``result.records``.  Columns, widths (11; ``iter.`` 6, ``FG(F)``
    and ``secs`` 8) and number formats (``%.2e``, ``n/a`` for differences in iteration 0, ``FG(F)`` as ``fg(f)``) follow
    the reference.  Supported fields: iter., J_T, J_a, J_b, λ_a⋅J_a, λ_b⋅J_b, J, ǁ∇J_Tǁ, ǁ∇(J_T+λ_b·J_b)ǁ, ǁ∇J_aǁ, λ_aǁ∇J_aǁ,
    ǁ∇Jǁ, ǁΔϵǁ, ǁϵǁ, max|Δϵ|, max|ϵ|, ǁΔϵǁ/ǁϵǁ, ∫Δϵ²dt, ΔJ_T, ΔJ_a, ΔJ_b, λ_a⋅ΔJ_a, λ_b⋅ΔJ_b, ΔJ, FG(F), secs (the
    line-search columns ǁsǁ, ∠°, α are properties of LBFGSB.jl's internals and are not available from scipy)."""
    import sys
    out = out or sys.stdout
    fields = list(print_iter_info) + [f for f in store_iter_info if f not in print_iter_info]
    unsupported = [f for f in fields if f in ("ǁsǁ", "∠°", "α")]
    if unsupported:
        raise ValueError(f"iteration-table fields {unsupported} need the optimizer's search direction (not available)")

    def print_table(wrk, iteration, *args):
        res, kw = wrk.result, wrk.kwargs
        lam_a, lam_b = kw.get("lambda_a", 1.0), kw.get("lambda_b", 1.0)
        eps, eps0 = wrk.pulsevals, wrk.pulsevals_guess
        d = eps - eps0
        dt = np.diff(wrk.tlist)
        v = {"iter.": iteration, "J_T": res.J_T, "ΔJ_T": res.J_T - res.J_T_prev, "J_a": res.J_a,
             "λ_a⋅J_a": float(wrk.J_parts[1]), "ΔJ_a": res.J_a - res.J_a_prev, "λ_a⋅ΔJ_a": lam_a * (res.J_a - res.J_a_prev),
             "J_b": res.J_b, "λ_b⋅J_b": float(wrk.J_parts[2]), "ΔJ_b": res.J_b - res.J_b_prev,
             "λ_b⋅ΔJ_b": lam_b * (res.J_b - res.J_b_prev), "J": res.J_T + lam_a * res.J_a + lam_b * res.J_b,
             "ǁ∇J_Tǁ": float(np.linalg.norm(wrk.grad_J_Tb)), "ǁ∇(J_T+λ_b·J_b)ǁ": float(np.linalg.norm(wrk.grad_J_Tb)),
             "ǁ∇J_aǁ": float(np.linalg.norm(wrk.grad_J_a)), "λ_aǁ∇J_aǁ": lam_a * float(np.linalg.norm(wrk.grad_J_a)),
             "ǁ∇Jǁ": float(np.linalg.norm(wrk.gradient)),
             "ΔJ": (res.J_T + lam_a * res.J_a + lam_b * res.J_b) - (res.J_T_prev + lam_a * res.J_a_prev + lam_b * res.J_b_prev),
             "ǁΔϵǁ": float(np.linalg.norm(d)), "ǁϵǁ": float(np.linalg.norm(eps)), "max|Δϵ|": float(np.abs(d).max()),
             "max|ϵ|": float(np.abs(eps).max()),
             "ǁΔϵǁ/ǁϵǁ": float(np.linalg.norm(d) / max(np.linalg.norm(eps), 1e-300)),
             "∫Δϵ²dt": float(sum(np.sum(d[l * wrk.N_T:(l + 1) * wrk.N_T] ** 2 * dt) for l in range(wrk.L))),
             "FG(F)": tuple(wrk.fg_count), "secs": res.secs}
        width = {"iter.": max(len(str(kw.get("iter_stop", 5000))), 6), "FG(F)": 8, "secs": 8, "ǁ∇(J_T+λ_b·J_b)ǁ": 17}
        if print_iter_info:
            if iteration == 0:
                out.write("".join(h.rjust(width.get(h, 11)) for h in print_iter_info) + "\n")
            row = []
            for h in print_iter_info:
                if h == "iter.":
                    sv = str(v[h])
                elif h == "FG(F)":
                    sv = "%d(%d)" % v[h]
                elif h == "secs":
                    sv = "%.1f" % v[h]
                elif h in _DELTA_HEADERS:
                    sv = ("%.2e" % v[h]) if iteration > 0 else "n/a"
                else:
                    sv = "%.2e" % v[h]
                row.append(sv.rjust(width.get(h, 11)))
            out.write("".join(row) + "\n")
            out.flush()
        return tuple(v[f] for f in store_iter_info)

    return print_table
