# make_reference_fixtures.jl -- the ONE route by which the parity grade of this repository can leave "unpinned".
#
# The build image has no Julia (SURVEY.md 8c), so tests/golden/*.npz hold outputs of the repository's OWN restatement.
# This script feeds the same inputs to the REFERENCE -- GRAPE.jl's GrapeWrk / evaluate_gradient!
# (src/workspace.jl:147-362, src/optimize.jl:824-1014) with ExpProp -- and writes what it returns next to them as
# tests/golden/ref_<name>.json.  tests/test_oracle.py::test_reference_outputs_when_present and
# tests/test_gpu_parity.py::test_reference_outputs_when_present_gpu pick those files up (skipped while there are none)
# and hold the oracles and the HIP path to them at the tolerances of SURVEY.md 8c.  NOT runnable in the build image.
#
#   julia --project=<env with GRAPE, QuantumControl, QuantumPropagators, NPZ, JSON> julia/make_reference_fixtures.jl
using GRAPE, QuantumControl, NPZ, JSON
using QuantumControl: Trajectory, hamiltonian
using QuantumControl.Functionals: J_T_sm, J_T_ss, J_T_re
using QuantumPropagators: ExpProp

const GOLDEN = joinpath(@__DIR__, "..", "tests", "golden")
const FUNCTIONALS = (J_T_sm, J_T_ss, J_T_re)          # codes 0, 1, 2 of include/grape_hip.h

function reference_outputs(path; gradient_method = :gradgen)
    z = npzread(path)
    H0, Hc, tlist = z["H0"], z["Hc"], Vector{Float64}(z["tlist"])       # H0[k, i, j], Hc[l, i, j] (row, column)
    per_traj = ndims(Hc) == 4                                            # Hc[k, l, i, j]: control operators per trajectory
    hc(k, l) = Matrix{ComplexF64}(per_traj ? Hc[k, l, :, :] : Hc[l, :, :])
    K, N, L, N_T = size(H0, 1), size(H0, 2), size(Hc, per_traj ? 2 : 1), length(tlist) - 1
    pulses = reshape(Vector{Float64}(z["pulsevals"]), N_T, L)            # control-major: column l = control l
    # one control per l, given by its values on the N_T intervals (discretize_on_midpoints keeps such a vector as it is,
    # src/workspace.jl:162); the SAME vector object in every trajectory, so that get_controls finds L controls
    controls = [pulses[:, l] for l = 1:L]
    weights = haskey(z, "weights") ? Vector{Float64}(z["weights"]) : ones(K)
    trajectories = [
        Trajectory(
            Vector{ComplexF64}(z["psi0"][k, :]),
            hamiltonian(Matrix{ComplexF64}(H0[k, :, :]), [(hc(k, l), controls[l]) for l = 1:L]...);
            target_state = Vector{ComplexF64}(z["target"][k, :]), weight = weights[k]
        ) for k = 1:K
    ]
    J_T = FUNCTIONALS[Int(z["functional"]) + 1]
    kwargs = Dict{Symbol,Any}(:J_T => J_T, :prop_method => ExpProp, :gradient_method => gradient_method)
    wrk = GRAPE.GrapeWrk(trajectories, tlist, kwargs)                    # src/optimize.jl:83
    G = zeros(length(wrk.pulsevals))
    J = GRAPE.evaluate_gradient!(G, wrk.pulsevals, wrk)                  # src/optimize.jl:110, 824
    tau = wrk.result.tau_vals                                            # src/optimize.jl:753
    psiT = [wrk.fw_propagators[k].state for k = 1:K]                     # src/optimize.jl:752
    return Dict(
        "source" => "GRAPE.jl $(pkgversion(GRAPE)), QuantumControl $(pkgversion(QuantumControl)), Julia $(VERSION)",
        "gradient_method" => String(gradient_method), "J" => J, "G" => G,
        "pulsevals_as_discretized" => Vector{Float64}(wrk.pulsevals),    # must equal the fixture's (checked by the tests)
        "tau_re" => real.(tau), "tau_im" => imag.(tau),
        "psiT_re" => [real.(p) for p in psiT], "psiT_im" => [imag.(p) for p in psiT],
        "tau_grads_re" => [real.(wrk.tau_grads[k]) for k = 1:K],        # [k][n, l]  (src/workspace.jl:236-237)
        "tau_grads_im" => [imag.(wrk.tau_grads[k]) for k = 1:K],
    )
end

for path in sort(filter(endswith(".npz"), readdir(GOLDEN; join = true)))
    name = splitext(basename(path))[1]
    out = Dict(String(m) => reference_outputs(path; gradient_method = m) for m in (:gradgen, :taylor))
    open(joinpath(GOLDEN, "ref_$(name).json"), "w") do io
        JSON.print(io, out, 1)
    end
    println(name, ": J = ", out["gradgen"]["J"])
end
