# GrapeHIP.jl -- thin `ccall` glue between GRAPE.jl and libgrape_hip.so (include/grape_hip.h, ABI v6).
#
# NOT EXECUTED in this repository's CI: the build image has no Julia toolchain.  It is written
# against the C ABI and the reference's own interfaces and shows exactly what a GRAPE.jl maintainer
# would add.  Drop-in point: the closure `fg!(F, G, pulsevals)` of GRAPE.optimize
# (/root/reference/src/optimize.jl:105-111), which run_optimizer calls at
# ext/GRAPELBFGSBExt.jl:99 (`f = fg!(f, obj.g, x)`) and ext/GRAPEOptimExt.jl:31.
module GrapeHIP

using LinearAlgebra
import QuantumControl
import QuantumControl.QuantumPropagators
using QuantumControl.QuantumPropagators.Controls: discretize_on_midpoints, evaluate, get_controls
using QuantumControl.QuantumPropagators.Amplitudes: ShapedAmplitude
using QuantumControl.Functionals: J_T_sm, J_T_ss, J_T_re

const libgrape = get(ENV, "GRAPE_HIP_LIB", "libgrape_hip.so")
const ABI_VERSION = 6

# mirror of `grape_problem` (include/grape_hip.h); field order and types must match the C struct
# (tests/test_abi.py compares the field lists)
struct GrapeProblem
    abi_version::Int32
    N::Int32
    L::Int32
    K::Int32
    K_total::Int32
    N_T::Int32
    functional::Int32        # 0 = J_T_sm, 1 = J_T_ss, 2 = J_T_re
    gradient_method::Int32   # 0 = :gradgen, 1 = :taylor
    hc_per_traj::Int32
    device::Int32
    tlist::Ptr{Float64}
    H0::Ptr{ComplexF64}      # K matrices, column-major (Julia's own layout: no copy)
    Hc::Ptr{ComplexF64}
    shape::Ptr{Float64}
    psi0::Ptr{ComplexF64}
    target::Ptr{ComplexF64}
    weights::Ptr{Float64}
    chi_min_norm::Float64
    taylor_max_order::Int32
    taylor_tolerance::Float64
    Dpen::Ptr{ComplexF64}    # state running cost g_b = <Psi|D|Psi> (C_NULL = off)
    dpen_per_traj::Int32
    lambda_b::Float64
    prop_method::Int32       # 0 = ExpProp (materialised Pade propagators), 1 = matrix-free series (Cheby/Newton role)
    prop_tolerance::Float64  # <= 0: 1e-17
    ndev::Int32              # > 1: the trajectories are dealt to several GPUs behind this one handle
    devices::Ptr{Int32}      # C_NULL: device, device+1, ...
    taylor_no_check::Int32   # ABI v6: 1 = taylor_grad_check_convergence = false (optimize.jl:917-918)
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    keepalive::Vector{Any}   # arrays whose pointers were handed over during grape_create
    K::Int
    N::Int
    functional::Int          # -1: user-supplied J_T / chi (split-phase calls + grape_backward_chi)
    no_target::Vector{Int}   # trajectories without a target_state: tau_vals[k] = NaN on the host (optimize.jl:753)
    L::Int                   # optimised controls (length(wrk.controls))
    N_T::Int
    fixed::Vector{Float64}   # pulse values of the pseudo-controls ([P*N_T], control-major; empty: none), see problem_arrays
    x_full::Vector{Float64}  # [(L+P)*N_T] staging: wrk.pulsevals followed by `fixed`
    G_full::Vector{Float64}  # [(L+P)*N_T] staging of the gradient (the first L*N_T entries are the caller's)
end

last_error(ptr) = unsafe_string(ccall((:grape_last_error, libgrape), Cstring, (Ptr{Cvoid},), ptr))

function check(h::Handle, rc::Integer)
    # becomes result.message = "Exception: ..." via src/optimize.jl:125-135 (or is rethrown with rethrow_exceptions)
    rc == 0 || error(last_error(h.ptr))
end


# ---- the static problem out of the reference's own types -----------------------------------------------------------

"""
    drift, control_ops, amplitudes = split_generator(generator)

`QuantumPropagators.Generators.Generator` keeps `ops` (drift terms first, then one operator per amplitude) and
`amplitudes` (`hamiltonian(H0, (H1, ϵ1), ...)`, docs/src/tutorial.md:55-73).
"""
function split_generator(gen)
    ops, amps = gen.ops, gen.amplitudes
    nd = length(ops) - length(amps)
    n = size(ops[1], 1)
    drift = zeros(ComplexF64, n, n)
    for op in ops[1:nd]
        drift .+= Matrix{ComplexF64}(op)
    end
    return drift, [Matrix{ComplexF64}(op) for op in ops[nd+1:end]], collect(amps)
end

control_of(a) = a
control_of(a::ShapedAmplitude) = a.control

"""
    H0, Hc, hc_per_traj, shape, fixed = problem_arrays(wrk)

Drift `H0[k]` of every trajectory, the operator `Hc[k][l]` = ∂H_k/∂ϵ_l multiplying control `l` of `wrk.controls`
(src/workspace.jl:152-157; a control that enters a generator through several amplitudes gets the sum of their
operators), and the static shapes `S_l(t_n)` of `ShapedAmplitude`s discretised on the midpoints of the time grid
(docs/src/tutorial.md:75-107) -- `nothing` if no amplitude is shaped.  Control amplitudes that depend non-linearly on
their control are not expressible in `grape_problem` (INTEGRATION.md: host-side chain rule).

A time-dependent amplitude that is NOT a control (`get_controls(a) == ()`: the reference re-evaluates the generator on
every interval, `evaluate(a, tlist, n)`, src/optimize.jl:732, 881, 937-945) becomes a PSEUDO-CONTROL: an extra operator
slot `L + j` in every `Hc[k]` and a row of fixed pulse values in `fixed` ([N_T, P]); `make_fg!` appends those values to
every pulse vector and drops the pseudo-controls' rows of the gradient.  (`grape_problem.H0` is one constant matrix per
trajectory; the library limits L + P to 8.)
"""
function problem_arrays(wrk)
    tlist, controls = wrk.result.tlist, wrk.controls
    K, L, N_T = length(wrk.trajectories), length(controls), length(tlist) - 1
    # pass 1: the amplitudes without a control, by identity, in order of first appearance
    fixed_amps = Any[]
    for traj in wrk.trajectories
        for a in split_generator(traj.generator)[3]
            isempty(get_controls(a)) && !any(b -> b === a, fixed_amps) && push!(fixed_amps, a)
        end
    end
    P = length(fixed_amps)
    fixed = Float64[real(evaluate(a, tlist, n)) for n = 1:N_T, a in fixed_amps]   # [N_T, P]
    H0 = Matrix{ComplexF64}[]
    Hc = Vector{Matrix{ComplexF64}}[]
    shape = ones(Float64, N_T, L + P)
    shaped = false
    for traj in wrk.trajectories
        drift, ops, amps = split_generator(traj.generator)
        n = size(drift, 1)
        per_l = [zeros(ComplexF64, n, n) for _ = 1:(L + P)]
        for (op, a) in zip(ops, amps)
            j = findfirst(b -> b === a, fixed_amps)
            if !isnothing(j)
                per_l[L + j] .+= op
                continue
            end
            l = findfirst(c -> c === control_of(a), controls)
            isnothing(l) && error("GrapeHIP: amplitude of type $(typeof(a)) is neither a (shaped) control of the problem nor control-free")
            per_l[l] .+= op
            if a isa ShapedAmplitude
                shape[:, l] .= discretize_on_midpoints(a.shape, tlist)
                shaped = true
            end
        end
        push!(H0, drift)
        push!(Hc, per_l)
    end
    hc_per_traj = any(Hc[k] != Hc[1] for k = 2:K)
    return H0, Hc, hc_per_traj, (shaped ? shape : nothing), fixed
end

"""J_T_sm / J_T_ss / J_T_re have a device-side χ (fast path); any other functional goes through `grape_backward_chi`."""
function functional_code(J_T)
    J_T === J_T_sm && return 0
    J_T === J_T_ss && return 1
    J_T === J_T_re && return 2
    return -1
end


"""
    Handle(wrk; device = 0, devices = nothing, prop_method = 0, D = nothing)

Uploads the static problem of a `GrapeWrk` (replaces the buffer set-up of src/workspace.jl:147-362).  Options are
taken from `wrk.kwargs` exactly where the reference reads them: `gradient_method` (workspace.jl:150), `chi_min_norm`
(optimize.jl:846), `taylor_grad_max_order` / `taylor_grad_tolerance` / `taylor_grad_check_convergence` (optimize.jl:915-918),
`lambda_b` (:833).
`D` (one matrix or one per trajectory) selects the state running cost of the family `g_b(Ψ) = ⟨Ψ|D|Ψ⟩`, `ξ = −DΨ`
(test/test_state_running_cost.jl:32-40).  `devices = [0, 1, ...]` spreads the trajectories over several GPUs behind this
one handle -- the analogue of `use_threads` (optimize.jl:720, 876).
"""
function Handle(wrk; device = 0, devices = nothing, prop_method = 0, D = nothing)
    tlist = Vector{Float64}(wrk.result.tlist)
    kw = wrk.kwargs
    H0, Hc, hc_per_traj, shape, fixed = problem_arrays(wrk)
    K, N, L, N_T = length(H0), size(H0[1], 1), length(wrk.controls), length(tlist) - 1
    P = size(fixed, 2)                                                      # pseudo-controls behind the L optimised ones
    H0f = reduce(hcat, vec.(H0))                                            # [N*N, K]: K column-major matrices
    Hcf = hc_per_traj ? reduce(hcat, [reduce(hcat, vec.(Hc[k])) for k = 1:K]) : reduce(hcat, vec.(Hc[1]))
    p0 = reduce(hcat, [Vector{ComplexF64}(t.initial_state) for t in wrk.trajectories])
    # trajectories without a target_state (optimize.jl:753: tau_k = NaN for THOSE k only, legal with a user J_T): no target
    # array at all when no trajectory has one; otherwise zeros in the gaps and tau_vals[k] = NaN set on the host (make_fg!)
    no_target = findall(t -> isnothing(t.target_state), wrk.trajectories)
    tg = length(no_target) == K ? nothing :
         reduce(hcat, [isnothing(t.target_state) ? zeros(ComplexF64, N) : Vector{ComplexF64}(t.target_state) for t in wrk.trajectories])
    weights = Float64[hasproperty(t, :weight) ? t.weight : 1.0 for t in wrk.trajectories]
    shp = isnothing(shape) ? nothing : Matrix{Float64}(shape)               # [N_T, L] column-major == [l][n]
    Df = isnothing(D) ? nothing : (D isa AbstractMatrix ? Matrix{ComplexF64}(D) : reduce(hcat, vec.(Matrix{ComplexF64}.(D))))
    devs = isnothing(devices) ? nothing : Vector{Int32}(devices)
    functional = functional_code(kw[:J_T])
    keep = Any[H0f, Hcf, p0, tg, tlist, weights, shp, Df, devs]
    prob = Ref(GrapeProblem(
        ABI_VERSION, N, L + P, K, 0, N_T, max(functional, 0),
        get(kw, :gradient_method, :gradgen) == :taylor ? 1 : 0, hc_per_traj ? 1 : 0, device,
        pointer(tlist), pointer(H0f), pointer(Hcf), isnothing(shp) ? C_NULL : pointer(shp), pointer(p0), isnothing(tg) ? C_NULL : pointer(tg),
        pointer(weights), get(kw, :chi_min_norm, 0.0), get(kw, :taylor_grad_max_order, 0),
        get(kw, :taylor_grad_tolerance, 0.0), isnothing(Df) ? C_NULL : pointer(Df),
        (D isa AbstractMatrix || isnothing(D)) ? 0 : 1, isnothing(D) ? 0.0 : get(kw, :lambda_b, 1.0), prop_method, 0.0,
        isnothing(devs) ? 0 : length(devs), isnothing(devs) ? C_NULL : pointer(devs),
        get(kw, :taylor_grad_check_convergence, true) ? 0 : 1))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve keep ccall((:grape_create, libgrape), Cint, (Ref{Ptr{Cvoid}}, Ref{GrapeProblem}), out, prob)
    rc == 0 || error(last_error(C_NULL))
    h = Handle(out[], keep, K, N, functional, no_target, L, N_T, vec(fixed), zeros(Float64, (L + P) * N_T), zeros(Float64, (L + P) * N_T))
    h.x_full[(L * N_T + 1):end] .= h.fixed
    finalizer(x -> ccall((:grape_destroy, libgrape), Cvoid, (Ptr{Cvoid},), x.ptr), h)
    # caller-supplied chi (functional == -1) or xi (a g_b that is not given as the operator D): the backward sweep runs
    # when that data arrives -- the forward call must not run a (unit-target) backward sweep in the same launch
    if functional < 0 || (!isnothing(get(kw, :g_b, nothing)) && isnothing(D))
        ccall((:grape_set_fused_sweeps, libgrape), Cint, (Ptr{Cvoid}, Cint), h.ptr, 0)
    end
    return h
end


"""
    fg! = make_fg!(h, wrk)

Replacement for the closure at src/optimize.jl:105-111.  Keeps the side effects callers rely on (SURVEY.md 8b):
`wrk.pulsevals`, the call counters, `wrk.result.tau_vals`, `wrk.J_parts[1:3]`, `wrk.grad_J_Tb`, `wrk.grad_J_a` and the
final states that `update_result!` reads (src/optimize.jl:187-189).
"""
function make_fg!(h::Handle, wrk)
    K, N = h.K, h.N
    kw = wrk.kwargs
    tlist = wrk.result.tlist
    psiT = Matrix{ComplexF64}(undef, N, K)
    chiT = Matrix{ComplexF64}(undef, N, K)
    sums = zeros(Float64, 8)
    J_T, chi = kw[:J_T], kw[:chi]
    J_a, grad_J_a = get(kw, :J_a, nothing), get(kw, :grad_J_a, nothing)
    λₐ, λ_b = get(kw, :lambda_a, 1.0), get(kw, :lambda_b, 1.0)
    has_gb = !isnothing(get(kw, :g_b, nothing))
    # g_b / xi as callbacks (src/optimize.jl:727-750, 856-866, 897-908) when no operator D was handed to create_handle:
    # evaluated here on the stored forward states, the array xi_k(t_n) goes to grape_backward_xi (ABI v5)
    g_b, xi = get(kw, :g_b, nothing), get(kw, :xi, nothing)
    xi_route = has_gb && isnothing(h.keep[8])                              # (keep[8]: the operator D handed to Handle(...))
    # (the reference derives a missing xi by automatic differentiation, src/workspace.jl:313-316; this glue has no AD
    # hook: say so HERE, at construction, not as a MethodError on `nothing` inside fg!)
    xi_route && isnothing(xi) && error("GrapeHIP.make_fg!: a running cost g_b given as a callback needs xi " *
                                       "(xi(state, trajectory, tlist, n) = -∂g_b/∂⟨Ψ|); pass xi = ... or hand the operator D to Handle(...)")
    N_T = length(tlist) - 1
    fw = xi_route ? Array{ComplexF64}(undef, N, N_T + 1, K) : nothing      # Ψ_k(t_n): [K][N_T+1][N] on the C side
    xi_arr = xi_route ? zeros(ComplexF64, N, N_T + 1, K) : nothing
    states() = [view(psiT, :, k) for k = 1:K]
    function J_b_and_xi!(want_xi)
        check(h, ccall((:grape_get_storage, libgrape), Cint, (Ptr{Cvoid}, Cint, Ptr{ComplexF64}), h.ptr, 0, fw))
        J_b = 0.0
        for k = 1:K
            traj = wrk.trajectories[k]
            J_b += g_b(view(fw, :, 1, k), traj, tlist, 1) * (tlist[2] - tlist[1]) / 2          # :728-730
            for n = 2:(N_T + 1)                                                                 # :739-749
                dt = n <= N_T ? (tlist[n + 1] - tlist[n - 1]) / 2 : (tlist[end] - tlist[end - 1]) / 2
                J_b += g_b(view(fw, :, n, k), traj, tlist, n) * dt
                want_xi && (xi_arr[:, n, k] .= xi(view(fw, :, n, k), traj, tlist, n))
            end
        end
        return J_b
    end

    return function fg!(F, G, pulsevals)
        (pulsevals !== wrk.pulsevals) && (wrk.pulsevals .= pulsevals)      # src/optimize.jl:706-713
        if isnothing(G)
            wrk.result.f_calls += 1; wrk.fg_count[2] += 1                  # :715-716
        else
            wrk.result.fg_calls += 1; wrk.fg_count[1] += 1                 # :838-839
        end
        tau = wrk.result.tau_vals
        # pseudo-controls (problem_arrays): the library sees L + P controls -- the fixed pulse values ride behind the caller's,
        # and the gradient lands in a staging vector whose first L*N_T entries are the caller's rows
        pseudo = !isempty(h.fixed)
        pseudo && copyto!(h.x_full, 1, wrk.pulsevals, 1, h.L * h.N_T)
        x = pseudo ? h.x_full : wrk.pulsevals
        Gbuf = pseudo ? h.G_full : wrk.grad_J_Tb
        if h.functional >= 0 && xi_route
            # built-in J_T with a g_b given as callbacks: split-phase calls, xi on the host
            check(h, GC.@preserve x tau h ccall((:grape_forward, libgrape), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}), h.ptr, pointer(x), pointer(tau)))
            check(h, ccall((:grape_get_final_states, libgrape), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), h.ptr, psiT))
            check(h, ccall((:grape_get_sums, libgrape), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, sums))
            Ψ = states()
            wrk.J_parts[1] = wrk.J_T_takes_tau ? J_T(Ψ, wrk.trajectories; tau = tau) : J_T(Ψ, wrk.trajectories)
            wrk.J_parts[3] = λ_b * J_b_and_xi!(!isnothing(G))
            if !isnothing(G)
                f_total = Float64[sums[1], sums[2]]                        # Σ_k w_k τ_k of this (unsharded) handle
                check(h, GC.@preserve f_total xi_arr wrk h ccall((:grape_backward_xi, libgrape), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Cdouble, Ptr{Float64}),
                    h.ptr, pointer(f_total), C_NULL, pointer(xi_arr), λ_b, pointer(Gbuf)))
            end
        elseif h.functional >= 0
            # built-in functional: one call, χ is formed on the device
            J = Ref{Float64}(0.0)
            Gp = isnothing(G) ? Ptr{Float64}(C_NULL) : pointer(Gbuf)
            rc = GC.@preserve x tau psiT wrk h ccall((:grape_eval, libgrape), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                h.ptr, pointer(x), J, Gp, pointer(tau), pointer(psiT))
            check(h, rc)
            J_b = 0.0
            if has_gb                                                      # J_parts[3] = λ_b Σ_k J_b,k, :764-766
                check(h, ccall((:grape_get_sums, libgrape), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, sums))
                J_b = λ_b * sums[5]
                wrk.J_parts[3] = J_b
            end
            wrk.J_parts[1] = J[] - J_b                                     # :757-760
        else
            # user-supplied J_T / chi (optimize.jl:757-760, 845-855): forward on the device, J_T and χ(T) on the host,
            # backward sweep and gradient on the device from the χ the user's function returns
            check(h, GC.@preserve x tau h ccall((:grape_forward, libgrape), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}), h.ptr, pointer(x), pointer(tau)))
            check(h, ccall((:grape_get_final_states, libgrape), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), h.ptr, psiT))
            Ψ = states()
            wrk.J_parts[1] = wrk.J_T_takes_tau ? J_T(Ψ, wrk.trajectories; tau = tau) : J_T(Ψ, wrk.trajectories)
            if has_gb
                check(h, ccall((:grape_get_sums, libgrape), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, sums))
                wrk.J_parts[3] = λ_b * sums[5]
            end
            xi_route && (wrk.J_parts[3] = λ_b * J_b_and_xi!(!isnothing(G)))
            if !isnothing(G)
                χ = wrk.chi_takes_tau ? chi(Ψ, wrk.trajectories; tau = tau) : chi(Ψ, wrk.trajectories)
                for k = 1:K
                    chiT[:, k] .= χ[k]
                end
                if xi_route
                    check(h, GC.@preserve chiT xi_arr wrk h ccall((:grape_backward_xi, libgrape), Cint,
                        (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Cdouble, Ptr{Float64}),
                        h.ptr, C_NULL, pointer(chiT), pointer(xi_arr), λ_b, pointer(Gbuf)))
                else
                    check(h, GC.@preserve chiT wrk h ccall((:grape_backward_chi, libgrape), Cint,
                        (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{Float64}), h.ptr, pointer(chiT), pointer(Gbuf)))
                end
            end
        end
        for k in h.no_target                                               # :753 -- only the trajectories without a target
            tau[k] = NaN
        end
        (pseudo && !isnothing(G)) && copyto!(wrk.grad_J_Tb, 1, h.G_full, 1, h.L * h.N_T)
        if !isnothing(J_a)
            wrk.J_parts[2] = λₐ * J_a(wrk.pulsevals, tlist)                # :761-763
        end
        for k = 1:K
            copyto!(wrk.fw_propagators[k].state, view(psiT, :, k))         # read by update_result! :187-189
        end
        if !isnothing(G)
            copyto!(G, wrk.grad_J_Tb)                                      # :1002-1003
            if !isnothing(grad_J_a)                                        # :1004-1011
                wrk.grad_J_a = grad_J_a(wrk.pulsevals, tlist)
                axpy!(λₐ, wrk.grad_J_a, G)
            end
        end
        return sum(wrk.J_parts)
    end
end

end # module
