# Usage of julia/NEPMI355X.jl from NonlinearEigenproblems.jl (the snippets of INTEGRATION.md; not executed in this image)

# config C2, the caller's code unchanged except for the wrapper around the problem: `iar` dispatches on DeviceSPMF to the method of
# julia/NEPMI355X.jl, which is ONE ccall of nep_iar_run (recurrence, eigen-decompositions, Ritz blocks, residual batches, convergence
# test and extraction inside the library -- the pipeline bench.py measures).  The native gun problem is an SPMFSumNEP(PEP, SPMF_NEP)
# (src/gallery_extra/NLEVP_native.jl:4-18): DeviceSPMF wraps it as it is.
gun  = nep_gallery("nlevp_native_gun")
nep  = MI355X.DeviceSPMF(shift_and_scale(gun, shift=250^2, scale=330^2-220^2))
λ, Q = iar(nep; maxit=100, neigs=Inf, v=ones(size(nep,1)), tol=1e-10)            # default creator = device factors + UMFPACK refinement
λ, Q, Z = tiar(nep; maxit=60, neigs=10, v=ones(size(nep,1)))                       # same for tiar (basis Z resident in HBM: Matrix(Z))
# anything the fast path does not cover keeps working through the reference's own loop over the four seams:
λ, Q = iar(nep; maxit=100, neigs=5, proj_solve=true, orthmethod=MI355X.DeviceDGKS(), linsolvercreator=MI355X.DeviceLinSolverCreator())

# config C4, one Julia process per GPU (MPI.jl only launches the ranks and carries RCCL's 128-byte id); contour_beyn itself is
# the reference's, unchanged -- MatrixTrapezoidalSharded is selected by its third positional argument (method_beyncontour.jl:48-51)
MPI.Init(); comm = MPI.COMM_WORLD
MI355X.init_comm!(MPI.Comm_rank(comm), MPI.Comm_size(comm), buf -> MPI.Bcast!(buf, 0, comm))
gunspmf = DeviceSPMF(SPMF_NEP(get_Av(gun), get_fv(gun)))
λ, V = contour_beyn(gunspmf, MI355X.MatrixTrapezoidalSharded; σ=250.0^2, radius=1e4, N=64, k=32, neigs=typemax(Int), tol=1e-6,
                    linsolvercreator=MI355X.DeviceLinSolverCreator(umfpack_refinements=0, resident=true))

# V: DevBuf, (m+1) columns of n(m+1); Ctab: DevBuf m x mt, row j = alpha_j/j * f^(j)(sigma); H: DevBuf m x (m+4), zero-filled;
# Hpin: pinned host Matrix{ComplexF64}(undef, m+4, m) (hipHostMalloc'ed, read by the eig task)
h = Ref{Ptr{Cvoid}}()
chk(ccall((:nep_iar_create, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid},
          Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}, Int32, Ptr{Cvoid}, Ptr{ComplexF64}, Int32, Ref{Ptr{Cvoid}}),
          spmf.h, lu.h, n, m, V.p, n*(m+1), Ctab.p, m, active.p, work3n.p, abs.(fσ), fσ, length(fσ), H.p, Hpin, 0, h))
for k in 1:m
    chk(ccall((:nep_iar_step, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}), h[], k, sweeps, C_NULL))
    @async begin                                   # eig(H_k) overlaps the following steps, as in iar.py
        ccall((:nep_iar_wait, LIB), Int32, (Ptr{Cvoid}, Int32), h[], k)
        row = @view Hpin[:, k]                     # h[1:k], beta, (passes, flags), then 4 Float64: omega of x_0..x_sweeps
        ω = reinterpret(Float64, row[k+3:k+4])
        out = zeros(Int32, 4)                      # UMFPACK's stopping rule replayed on the record (nep_refine_review)
        ccall((:nep_refine_review, LIB), Int32, (Int32, Int32, Int32, Ptr{Float64}, Int32, Ptr{Int32}), 10, sweeps, 1, ω, -1, out)
        out[1] == 1 || error("refinement miss: rerun with checked solves")
    end
end


# residuals of all k Ritz pairs of a convergence check, Q = VV*Z column-major on the device (DevBuf n x k), F[t, s] = f_t(λ_s)
# (nep_resid_batch_cm_dev: the tiled K2 kernel for column-major blocks; row0 = -1: the whole residual enters the norms)
out = DevBuf(k, 1)                                       # 2k Float64 = k ComplexF64 slots: squared norms |r_s|², then |q_s|²
chk(ccall((:nep_resid_batch_cm_dev, LIB), Int32,
          (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
          nep.h, k, F, Q.ptr, n, -1, out.ptr, C_NULL, 0, C_NULL))
