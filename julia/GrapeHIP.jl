# GrapeHIP.jl -- thin `ccall` glue between GRAPE.jl and libgrape_hip.so (include/grape_hip.h).
#
# NOT EXECUTED in this repository's CI: the build image has no Julia toolchain.  It is written
# against the C ABI and the reference's own interfaces and shows exactly what a GRAPE.jl maintainer
# would add.  Drop-in point: the closure `fg!(F, G, pulsevals)` of GRAPE.optimize
# (/root/reference/src/optimize.jl:105-111), which run_optimizer calls at
# ext/GRAPELBFGSBExt.jl:99 (`f = fg!(f, obj.g, x)`) and ext/GRAPEOptimExt.jl:31.
module GrapeHIP

using LinearAlgebra

const libgrape = get(ENV, "GRAPE_HIP_LIB", "libgrape_hip.so")

# mirror of `grape_problem` (include/grape_hip.h); field order and types must match the C struct
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
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    keepalive::Vector{Any}   # arrays whose pointers were handed over during grape_create
end

function check(h::Handle, rc::Integer)
    if rc != 0
        msg = unsafe_string(ccall((:grape_last_error, libgrape), Cstring, (Ptr{Cvoid},), h.ptr))
        error(msg)  # becomes result.message = "Exception: ..." via src/optimize.jl:125-135
    end
end

"""
    Handle(trajectories, tlist, controls, pulse_ops; functional, gradient_method, device)

Uploads the static problem (replaces the buffer set-up of GrapeWrk, src/workspace.jl:147-362).
`H0[k]` is the drift of trajectory `k`, `Hc[l]` the operator multiplying control `l`.
"""
function Handle(H0::Vector{Matrix{ComplexF64}}, Hc::Vector{Matrix{ComplexF64}}, tlist::Vector{Float64},
                psi0::Vector{Vector{ComplexF64}}, target::Vector{Vector{ComplexF64}};
                weights = ones(length(H0)), functional = 0, gradient_method = 0, device = 0, K_total = 0,
                prop_method = 0)
    K, N, L = length(H0), size(H0[1], 1), length(Hc)
    H0f = reduce(hcat, vec.(H0)); Hcf = reduce(hcat, vec.(Hc))
    p0 = reduce(hcat, psi0); tg = reduce(hcat, target)
    keep = Any[H0f, Hcf, p0, tg, tlist, weights]
    prob = Ref(GrapeProblem(3, N, L, K, K_total, length(tlist) - 1, functional, gradient_method, 0, device,
                            pointer(tlist), pointer(H0f), pointer(Hcf), C_NULL, pointer(p0), pointer(tg),
                            pointer(weights), 0.0, 0, 0.0, C_NULL, 0, 0.0, prop_method, 0.0))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve keep ccall((:grape_create, libgrape), Cint, (Ref{Ptr{Cvoid}}, Ref{GrapeProblem}), out, prob)
    h = Handle(out[], keep)
    rc == 0 || error(unsafe_string(ccall((:grape_last_error, libgrape), Cstring, (Ptr{Cvoid},), C_NULL)))
    finalizer(x -> ccall((:grape_destroy, libgrape), Cvoid, (Ptr{Cvoid},), x.ptr), h)
    return h
end

"""
    fg!(h, wrk)(F, G, pulsevals) -> J

Replacement for the closure at src/optimize.jl:105-111.  Keeps the side effects callers rely on
(SURVEY.md 8b): wrk.pulsevals, counters, wrk.result.tau_vals, wrk.J_parts[1], wrk.grad_J_Tb and
the final states for update_result! (src/optimize.jl:187-189).
"""
function make_fg!(h::Handle, wrk)
    K = length(wrk.trajectories)
    N = length(wrk.trajectories[1].initial_state)
    psiT = Matrix{ComplexF64}(undef, N, K)
    return function fg!(F, G, pulsevals)
        (pulsevals !== wrk.pulsevals) && (wrk.pulsevals .= pulsevals)      # src/optimize.jl:706-713
        J = Ref{Float64}(0.0)
        if isnothing(G)
            wrk.result.f_calls += 1; wrk.fg_count[2] += 1                  # :715-716
            Gp = Ptr{Float64}(C_NULL)
        else
            wrk.result.fg_calls += 1; wrk.fg_count[1] += 1                 # :838-839
            Gp = pointer(wrk.grad_J_Tb)
        end
        rc = GC.@preserve pulsevals psiT ccall((:grape_eval, libgrape), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{ComplexF64}),
            h.ptr, pointer(wrk.pulsevals), J, Gp, pointer(wrk.result.tau_vals), pointer(psiT))
        check(h, rc)
        wrk.J_parts[1] = J[]                                               # :757-760
        for k = 1:K
            copyto!(wrk.fw_propagators[k].state, view(psiT, :, k))         # read by update_result! :187-189
        end
        isnothing(G) || copyto!(G, wrk.grad_J_Tb)                          # :1002-1003
        return sum(wrk.J_parts)
    end
end

end # module
