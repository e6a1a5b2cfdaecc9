"""Oracle for the waveguide eigenvalue problem (WEP) -- test infrastructure only.

Follows
  src/gallery_extra/waveguide/waveguide_FD.jl:10-64      FD matrices (Dxx, Dzz, Dz, C1, C2T)
  src/gallery_extra/waveguide/waveguide_FD.jl:91-182     wavenumbers (TAUSCH, JARLEBRING)
  src/gallery_extra/waveguide/Waveguide.jl:9-46          SPMF assembly (3 sparse + 2 nz rank-one terms)
  src/gallery_extra/waveguide/Waveguide.jl:53-106        R / Rinv (scaled FFT), S-functions
  src/gallery_extra/waveguide/Waveguide.jl:115-157       sqrt on the branch Im >= 0 (scalar and Schur form)
  src/gallery_extra/waveguide/Waveguide.jl:204-379       WEP_FD and its compute_Mlincomb (FFT form)
  src/gallery_extra/waveguide/Waveguide.jl:580-616       sqrt_derivative recurrence
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from . import neps


# ------------------------------------------------------------------------------------------------
def generate_fd_interior_mat(nx, nz, hx, hz):
    ex = np.ones(nx); ez = np.ones(nz)
    Dxx = sp.diags([ex[:-1], -2 * ex, ex[:-1]], [-1, 0, 1], format="lil")
    Dzz = sp.diags([ez[:-1], -2 * ez, ez[:-1]], [-1, 0, 1], format="lil")
    Dzz[0, nz - 1] = 1; Dzz[nz - 1, 0] = 1
    Dxx = sp.csc_matrix(Dxx) / hx ** 2
    Dzz = sp.csc_matrix(Dzz) / hz ** 2
    Dz = sp.diags([-ez[:-1], ez[:-1]], [-1, 1], format="lil")
    Dz[0, nz - 1] = -1; Dz[nz - 1, 0] = 1
    Dz = sp.csc_matrix(Dz) / (2 * hz)
    return Dxx, Dzz, Dz


def generate_fd_boundary_mat(nx, nz, hx, hz):
    e1 = sp.csc_matrix(([1.0], ([0], [0])), shape=(nx, 1))
    en = sp.csc_matrix(([1.0], ([nx - 1], [0])), shape=(nx, 1))
    Iz = sp.identity(nz, format="csc")
    C1 = sp.hstack([sp.kron(e1, Iz), sp.kron(en, Iz)]) / hx ** 2
    d1 = 2 / hx; d2 = -1 / (2 * hx)
    vm = sp.csc_matrix(([d1, d2], ([0, 0], [0, 1])), shape=(1, nx))
    vp = sp.csc_matrix(([d1, d2], ([0, 0], [nx - 1, nx - 2])), shape=(1, nx))
    C2T = sp.vstack([sp.kron(vm, Iz), sp.kron(vp, Iz)])
    return sp.csc_matrix(C1), sp.csc_matrix(C2T)


def _grid(nx, nz, xm, xp, delta):
    xm = xm - delta; xp = xp + delta
    X = np.linspace(xm, xp, nx + 2); hx = X[1] - X[0]
    Z = np.linspace(0.0, 1.0, nz + 1); hz = Z[1] - Z[0]
    return X[1:-1], Z[1:], hx, hz


def generate_wavenumber_fd(nx, nz, wg, delta):
    wg = wg.upper()
    if wg == "TAUSCH":
        X, Z, hx, hz = _grid(nx, nz, 0.0, 2 / np.pi + 0.4, delta)
        k1 = np.sqrt(2.3) * np.pi; k2 = np.sqrt(3) * np.pi; k3 = np.pi

        def k(x, z):
            x = np.asarray(x, dtype=float); z = np.asarray(z, dtype=float)
            o = np.ones(np.broadcast(x, z).shape)
            return (k1 * (x <= 0) * o + k2 * (x > 0) * (x <= 2 / np.pi) * o +
                    k2 * (x > 2 / np.pi) * (x <= 2 / np.pi + 0.4) * (z > 0.5) +
                    k3 * (x > 2 / np.pi) * (z <= 0.5) * (x <= 2 / np.pi + 0.4) +
                    k3 * (x > 2 / np.pi + 0.4) * o)
    elif wg == "JARLEBRING":
        X, Z, hx, hz = _grid(nx, nz, -1.0, 1.0, delta)
        k1 = np.sqrt(2.3) * np.pi; k2 = 2 * np.sqrt(3) * np.pi; k3 = 4 * np.sqrt(3) * np.pi; k4 = np.pi

        def k(x, z):
            x = np.asarray(x, dtype=float); z = np.asarray(z, dtype=float)
            o = np.ones(np.broadcast(x, z).shape)
            xx = x * o; zz = z * o
            return (k1 * (xx <= -1) + k4 * (xx > 1) +
                    k4 * (xx > 0.5) * (xx <= 1) * (zz <= 0.4) +
                    k3 * (xx > 0) * (xx <= 0.5) +
                    k3 * (xx > 0.5) * (xx <= 1) * (zz > 0.4) +
                    k3 * (xx > -1) * (xx <= 0) * (zz > 0.5) * (zz - xx / 2 <= 1) +
                    k2 * (xx > -1) * (xx <= 0) * (zz > 0.5) * (zz - xx / 2 > 1) +
                    k3 * (xx > -1) * (xx <= 0) * (zz <= 0.5) * (zz + xx / 2 > 0) +
                    k2 * (xx > -1) * (xx <= 0) * (zz <= 0.5) * (zz + xx / 2 <= 0))
    else:
        raise ValueError("No wavenumber loaded: The given Waveguide '%s' is not supported in 'FD' discretization." % wg)
    K = k(X[None, :], Z[:, None]) ** 2            # nz x nx
    Km = float(k(-np.inf, 0.5)); Kp = float(k(np.inf, 0.5))
    return K, hx, hz, Km, Kp


# ------------------------------------------------------------------------------------------------
def sqrt_pos_imag(a):
    a = complex(a)
    s = np.sign(a.imag)
    return np.sqrt(a) if s == 0 else s * np.sqrt(a)


def sqrt_schur_pos_imag(A):
    """Waveguide.jl:115-141"""
    if not (isinstance(A, np.ndarray) and A.ndim == 2):
        return sqrt_pos_imag(A)
    n = A.shape[0]
    T, Q = sla.schur(A.astype(complex), output="complex")
    U = np.zeros((n, n), dtype=complex)
    for i in range(n):
        U[i, i] = sqrt_pos_imag(T[i, i])
    for j in range(1, n):
        for i in range(j - 1, -1, -1):
            temp = U[i, i + 1:j] @ U[i + 1:j, j]
            U[i, j] = (T[i, j] - temp) / (U[i, i] + U[j, j])
    return Q @ U @ Q.conj().T


def sqrt_derivative(a, b, c, d=0, x=0):
    """all d derivatives of sqrt(a z^2 + b z + c) at z = x  (Waveguide.jl:580-616)"""
    aa = a; bb = b + 2 * a * x; cc = c + a * x ** 2 + b * x
    der = np.zeros(d + 1, dtype=complex)
    yi = sqrt_pos_imag(cc)
    der[0] = yi
    if d == 0:
        return der[0]
    yip1 = bb / (2 * sqrt_pos_imag(cc))
    fact = 1.0
    der[1] = yip1 * fact
    for i in range(2, d + 1):
        m = i - 2
        yip2 = -(2 * aa * (m - 1) * yi + bb * (1 + 2 * m) * yip1) / (2 * cc * (2 + m))
        fact *= i
        yi = yip1; yip1 = yip2
        der[i] = yip2 * fact
    return der


class WaveguideData:
    def __init__(self, nx, nz, benchmark_problem="TAUSCH", delta=0.1):
        if nz % 2 == 0:
            raise ValueError("Variable nz must be odd! You have used nz = %d." % nz)
        self.nx, self.nz = nx, nz
        self.K, self.hx, self.hz, self.Km, self.Kp = generate_wavenumber_fd(nx, nz, benchmark_problem, delta)
        self.Dxx, self.Dzz, self.Dz = generate_fd_interior_mat(nx, nz, self.hx, self.hz)
        self.C1, self.C2T = generate_fd_boundary_mat(nx, nz, self.hx, self.hz)
        p = (nz - 1) / 2
        self.p = p
        self.d0 = -3 / (2 * self.hx)
        self.b = 4 * np.pi * 1j * np.arange(-p, p + 1)
        self.cM = self.Km ** 2 - 4 * np.pi ** 2 * np.arange(-p, p + 1) ** 2
        self.cP = self.Kp ** 2 - 4 * np.pi ** 2 * np.arange(-p, p + 1) ** 2
        self.bb = np.exp(-2j * np.pi * np.arange(nz) * (-p) / nz)
        self.n = nx * nz + 2 * nz

    def R(self, X):
        return (self.bb * np.fft.fft(np.asarray(X).ravel()))[::-1]

    def Rinv(self, X):
        return np.fft.ifft((1.0 / self.bb) * np.asarray(X).ravel()[::-1])

    def Rmat(self):
        """dense nz x nz matrix with columns R(e_j)"""
        nz = self.nz
        return np.column_stack([self.R(np.eye(nz)[:, j]) for j in range(nz)])

    def big_matrices(self):
        """A[1..3] of Waveguide.jl:17-19 (real sparse)"""
        nx, nz = self.nx, self.nz
        Ix = sp.identity(nx, format="csc"); Iz = sp.identity(nz, format="csc")
        Q0 = sp.kron(Ix, self.Dzz) + sp.kron(self.Dxx, Iz) + sp.diags(self.K.ravel(order="F"))
        Q1 = sp.kron(Ix, 2 * self.Dz)
        Q2 = sp.kron(Ix, Iz)
        Z12 = sp.csc_matrix((nx * nz, 2 * nz)); Z21 = sp.csc_matrix((2 * nz, nx * nz)); Z22 = sp.csc_matrix((2 * nz, 2 * nz))
        A1 = sp.bmat([[Q0, self.C1], [self.C2T, Z22]], format="csc")
        A2 = sp.bmat([[Q1, Z12], [Z21, Z22]], format="csc")
        A3 = sp.bmat([[Q2, Z12], [Z21, Z22]], format="csc")
        return [A1, A2, A3]

    def S_scalar(self, lam):
        """the 2 nz corner function values s_j(lam) (incl. d0)"""
        lam = complex(lam)
        out = np.empty(2 * self.nz, dtype=complex)
        for j in range(self.nz):
            out[j] = 1j * sqrt_pos_imag(lam ** 2 + self.b[j] * lam + self.cM[j]) + self.d0
            out[self.nz + j] = 1j * sqrt_pos_imag(lam ** 2 + self.b[j] * lam + self.cP[j]) + self.d0
        return out


def assemble_waveguide_spmf_fd(wd):
    """literal SPMF with 3 + 2 nz terms (Waveguide.jl:9-46); only for small nz"""
    nx, nz = wd.nx, wd.nz
    A = wd.big_matrices()
    f = [neps.f_one(), neps.f_id(), neps.f_pow(2)]

    def make_S(b, c, d0):
        def S(lam):
            if isinstance(lam, np.ndarray) and lam.ndim == 2:
                I = np.eye(lam.shape[0])
                return 1j * sqrt_schur_pos_imag(lam @ lam + b * lam + c * I) + d0 * I
            return 1j * sqrt_pos_imag(lam ** 2 + b * lam + c) + d0
        return S

    N = nx * nz
    for half, cvec in ((0, wd.cM), (1, wd.cP)):
        for j in range(nz):
            ej = np.zeros(nz); ej[j] = 1
            r = wd.R(ej)
            E = np.zeros(2 * nz, dtype=complex)
            E[half * nz:(half + 1) * nz] = r
            Ej = np.outer(E, (E / nz).conj())
            Afull = sp.lil_matrix((wd.n, wd.n), dtype=complex)
            Afull[N:, N:] = Ej
            A.append(sp.csc_matrix(Afull))
            f.append(make_S(wd.b[j], cvec[j], wd.d0))
    return neps.SPMF_NEP([sp.csc_matrix(M, dtype=complex) for M in A], f)


class WEP_FD(neps.NEP):
    """Waveguide.jl:204-379 (matrix-free form with FFTs) plus an explicit M(lam) for the LU-based solve."""

    def __init__(self, nx, nz, benchmark_problem="TAUSCH", delta=0.1):
        self.wd = WaveguideData(nx, nz, benchmark_problem, delta)
        self.n = self.wd.n
        self.k_bar = np.mean(self.wd.K)
        self.K_scaled = self.wd.K - self.k_bar

    def _A(self, lam, d=0):
        wd = self.wd
        Iz = sp.identity(wd.nz, format="csc")
        if d == 0:
            return wd.Dzz + 2 * lam * wd.Dz + (lam ** 2 + self.k_bar) * Iz
        if d == 1:
            return 2 * wd.Dz + 2 * lam * Iz
        if d == 2:
            return 2 * Iz
        return sp.csc_matrix((wd.nz, wd.nz))

    def _mlincomb(self, lam, V, a):
        wd = self.wd
        nx, nz = wd.nx, wd.nz
        na = len(a)
        max_d = na - 1
        N = nx * nz
        V1 = V[:N, :]; V2 = V[N:, :]
        V1m = [V1[:, j].reshape((nz, nx), order="F") for j in range(na)]
        y1 = (self._A(lam) @ V1m[0] + V1m[0] @ wd.Dxx + self.K_scaled * V1m[0]) * a[0]
        for d in range(1, min(max_d, 3) + 1):
            y1 = y1 + (self._A(lam, d) @ V1m[d]) * a[d]
        y1 = np.asarray(y1).ravel(order="F").astype(complex)
        y1 += wd.C1 @ V2[:, 0] * a[0]
        D = np.zeros((2 * nz, na), dtype=complex)
        cMP = np.concatenate([wd.cM, wd.cP])
        for j in range(2 * nz):
            der = 1j * np.atleast_1d(sqrt_derivative(1, wd.b[j % nz], cMP[j], max_d, lam))
            D[j, :] = der[:na]
        y2t = (D[:, 0] + wd.d0) * np.concatenate([wd.Rinv(V2[:nz, 0]), wd.Rinv(V2[nz:, 0])]) * a[0]
        for jj in range(1, na):
            y2t = y2t + D[:, jj] * np.concatenate([wd.Rinv(V2[:nz, jj]), wd.Rinv(V2[nz:, jj])]) * a[jj]
        y2 = np.concatenate([wd.R(y2t[:nz]), wd.R(y2t[nz:])])
        y2 = y2 + wd.C2T @ V1[:, 0] * a[0]
        return np.concatenate([y1, y2])

    def corner(self, lam):
        """dense 2nz x 2nz corner  Rfull diag(s(lam)) Rfull^H / nz"""
        wd = self.wd; nz = wd.nz
        Rm = wd.Rmat(); s = wd.S_scalar(lam)
        P = np.zeros((2 * nz, 2 * nz), dtype=complex)
        P[:nz, :nz] = (Rm * s[:nz][None, :]) @ Rm.conj().T / nz
        P[nz:, nz:] = (Rm * s[nz:][None, :]) @ Rm.conj().T / nz
        return P

    def compute_Mder(self, lam, i=0):
        if i != 0:
            raise NotImplementedError
        wd = self.wd
        A1, A2, A3 = wd.big_matrices()
        M = sp.lil_matrix(A1 + lam * A2 + lam ** 2 * A3, dtype=complex)
        N = wd.nx * wd.nz
        M[N:, N:] = self.corner(lam)
        return sp.csc_matrix(M)
