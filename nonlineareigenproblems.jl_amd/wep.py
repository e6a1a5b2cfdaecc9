"""Waveguide eigenvalue problem (WEP) inputs and NEP type for the device backend.

Generators restate src/gallery_extra/waveguide/waveguide_FD.jl:10-64 (FD matrices), :91-182 (wavenumbers
TAUSCH / JARLEBRING) and src/gallery_extra/waveguide/Waveguide.jl:9-106 (SPMF structure, R/Rinv, S-functions).
`nep_gallery("WEP", nx=, nz=, benchmark_problem=, delta=)` mirrors src/gallery_extra/GalleryWaveguide.jl:60-94.

Device representation: the literal SPMF has 3 + 2 nz terms whose last 2 nz matrices are dense
rank-one nz x nz blocks (2 nz^3 = 2e9 non-zeros at nz = 999).  Here the NEP is
    M(lam) = A1 + lam A2 + lam^2 A3  +  [0 0; 0 P(lam)],   P(lam) = blkdiag(Rm, Rm) diag(s(lam)) blkdiag(Rm, Rm)^H / nz
i.e. three big REAL sparse matrices in one stacked CSR (the HBM-bound part) plus a factored corner term applied as
two small dense products with Rm (nz x nz, the scaled DFT matrix of Waveguide.jl:53-65).
"""
import numpy as np
import scipy.sparse as sp

from . import funcs


def generate_fd_interior_mat(nx, nz, hx, hz):
    ex = np.ones(nx); ez = np.ones(nz)
    Dxx = sp.diags([ex[:-1], -2 * ex, ex[:-1]], [-1, 0, 1], format="lil")
    Dzz = sp.diags([ez[:-1], -2 * ez, ez[:-1]], [-1, 0, 1], format="lil")
    Dzz[0, nz - 1] = 1; Dzz[nz - 1, 0] = 1                       # periodicity in z
    Dz = sp.diags([-ez[:-1], ez[:-1]], [-1, 1], format="lil")
    Dz[0, nz - 1] = -1; Dz[nz - 1, 0] = 1
    return sp.csc_matrix(Dxx) / hx ** 2, sp.csc_matrix(Dzz) / hz ** 2, sp.csc_matrix(Dz) / (2 * hz)


def generate_fd_boundary_mat(nx, nz, hx, hz):
    Iz = sp.identity(nz, format="csc")
    e1 = sp.csc_matrix(([1.0], ([0], [0])), shape=(nx, 1))
    en = sp.csc_matrix(([1.0], ([nx - 1], [0])), shape=(nx, 1))
    C1 = sp.hstack([sp.kron(e1, Iz), sp.kron(en, Iz)]) / hx ** 2
    d1 = 2 / hx; d2 = -1 / (2 * hx)
    vm = sp.csc_matrix(([d1, d2], ([0, 0], [0, 1])), shape=(1, nx))
    vp = sp.csc_matrix(([d1, d2], ([0, 0], [nx - 1, nx - 2])), shape=(1, nx))
    C2T = sp.vstack([sp.kron(vm, Iz), sp.kron(vp, Iz)])
    return sp.csc_matrix(C1), sp.csc_matrix(C2T)


def generate_wavenumber_fd(nx, nz, wg, delta):
    wg = wg.upper()
    if wg == "TAUSCH":
        xm, xp = 0.0 - delta, 2 / np.pi + 0.4 + delta
        k1 = np.sqrt(2.3) * np.pi; k2 = np.sqrt(3) * np.pi; k3 = np.pi

        def k(x, z):
            x, z = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(z, dtype=float))
            return (k1 * (x <= 0) + k2 * (x > 0) * (x <= 2 / np.pi) +
                    k2 * (x > 2 / np.pi) * (x <= 2 / np.pi + 0.4) * (z > 0.5) +
                    k3 * (x > 2 / np.pi) * (z <= 0.5) * (x <= 2 / np.pi + 0.4) + k3 * (x > 2 / np.pi + 0.4))
    elif wg == "JARLEBRING":
        xm, xp = -1.0 - delta, 1.0 + delta
        k1 = np.sqrt(2.3) * np.pi; k2 = 2 * np.sqrt(3) * np.pi; k3 = 4 * np.sqrt(3) * np.pi; k4 = np.pi

        def k(x, z):
            x, z = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(z, dtype=float))
            return (k1 * (x <= -1) + k4 * (x > 1) + k4 * (x > 0.5) * (x <= 1) * (z <= 0.4) +
                    k3 * (x > 0) * (x <= 0.5) + k3 * (x > 0.5) * (x <= 1) * (z > 0.4) +
                    k3 * (x > -1) * (x <= 0) * (z > 0.5) * (z - x / 2 <= 1) +
                    k2 * (x > -1) * (x <= 0) * (z > 0.5) * (z - x / 2 > 1) +
                    k3 * (x > -1) * (x <= 0) * (z <= 0.5) * (z + x / 2 > 0) +
                    k2 * (x > -1) * (x <= 0) * (z <= 0.5) * (z + x / 2 <= 0))
    else:
        raise ValueError("No wavenumber loaded: The given Waveguide '%s' is not supported in 'FD' discretization." % wg)
    X = np.linspace(xm, xp, nx + 2); hx = X[1] - X[0]
    Z = np.linspace(0.0, 1.0, nz + 1); hz = Z[1] - Z[0]
    K = k(X[None, 1:-1], Z[1:, None]) ** 2                 # nz x nx, vectorised column-major (z fastest)
    return K, hx, hz, float(k(-np.inf, 0.5)), float(k(np.inf, 0.5))


class WaveguideData:
    """all problem data of one waveguide discretisation"""

    def __init__(self, nx, nz, benchmark_problem="TAUSCH", delta=0.1):
        if nz % 2 == 0:
            raise ValueError("Variable nz must be odd! You have used nz = %d." % nz)
        self.nx, self.nz = int(nx), int(nz)
        self.K, self.hx, self.hz, self.Km, self.Kp = generate_wavenumber_fd(nx, nz, benchmark_problem, delta)
        self.n = nx * nz + 2 * nz
        p = (nz - 1) / 2
        self.d0 = -3 / (2 * self.hx)
        self.b = 4 * np.pi * 1j * np.arange(-p, p + 1)
        self.cM = self.Km ** 2 - 4 * np.pi ** 2 * np.arange(-p, p + 1) ** 2
        self.cP = self.Kp ** 2 - 4 * np.pi ** 2 * np.arange(-p, p + 1) ** 2
        self.bb = np.exp(-2j * np.pi * np.arange(nz) * (-p) / nz)

    def big_matrices(self):
        """[A1, A2, A3] of Waveguide.jl:17-19 as real CSR matrices (n x n)"""
        nx, nz = self.nx, self.nz
        Dxx, Dzz, Dz = generate_fd_interior_mat(nx, nz, self.hx, self.hz)
        C1, C2T = generate_fd_boundary_mat(nx, nz, self.hx, self.hz)
        Ix = sp.identity(nx, format="csr"); Iz = sp.identity(nz, format="csr")
        Q0 = sp.kron(Ix, Dzz, format="csr") + sp.kron(Dxx, Iz, format="csr") + sp.diags(self.K.ravel(order="F"))
        Q1 = sp.kron(Ix, 2 * Dz, format="csr")
        Q2 = sp.identity(nx * nz, format="csr")
        N = nx * nz
        Z12 = sp.csr_matrix((N, 2 * nz)); Z21 = sp.csr_matrix((2 * nz, N)); Z22 = sp.csr_matrix((2 * nz, 2 * nz))
        A1 = sp.bmat([[Q0, C1], [C2T, Z22]], format="csr")
        A2 = sp.bmat([[Q1, Z12], [Z21, Z22]], format="csr")
        A3 = sp.bmat([[Q2, Z12], [Z21, Z22]], format="csr")
        return [A1, A2, A3]

    def Rmat(self):
        """Rm[:, j] = R(e_j) = reverse(bb .* fft(e_j))  (Waveguide.jl:53-59) as a dense nz x nz matrix"""
        nz = self.nz
        F = np.fft.fft(np.eye(nz), axis=0)
        return (self.bb[:, None] * F)[::-1, :].copy()

    def corner_funs(self):
        """the 2 nz corner functions s_j(lam) = 1im*sqrt(lam^2 + b_j lam + c_j) + d0 (branch Im sqrt >= 0)"""
        return ([funcs.WEPSqrt(self.b[j], self.cM[j], self.d0) for j in range(self.nz)] +
                [funcs.WEPSqrt(self.b[j], self.cP[j], self.d0) for j in range(self.nz)])
