"""Scalar functions f_i(lambda) of an SPMF with closed-form derivative tables.

The reference passes Julia closures that must work on scalars AND matrices and obtains scaled
derivatives by evaluating the matrix function of a bidiagonal matrix at every call
(src/NEPTypes.jl:993-1004, O(k^3) host work on the critical path; DerSPMF :1108-1128 caches it).
Here every function knows its derivatives in closed form, so the coefficient block
C[j,i] = a_j f_i^(j-1)(lambda) handed to the device kernel (include/nepmi355.h nep_mlincomb) costs
O(k) per function.  `matfun` gives the matrix function needed by compute_MM with a general S.
Dynamic range: derivatives reach 1e164 for gun at order 100 -> float64 only, and all
recurrences below are arranged to avoid intermediate overflow/underflow.
"""
import math

import numpy as np
import scipy.linalg as sla


class ScalarFun:
    """f: C -> C with derivatives."""

    def __call__(self, lam):
        return self.derivs(lam, 1)[0]

    def derivs(self, lam, k, scale=1.0):
        """array [scale^j * f^(j)(lam), j = 0..k-1] (complex128).  `scale` (chain-rule factor of an
        affine change of variable) is folded into the recurrences so that scale^j never appears alone:
        for gun, scale^j = 60500^j overflows at j = 65 while scale^j f^(j) stays below 1e165."""
        raise NotImplementedError

    def matfun(self, S):
        """f(S) for a square matrix S (host, small)."""
        raise NotImplementedError

    def values(self, lams):
        """vectorised f(lam) over an array of points"""
        return np.array([self(l) for l in np.asarray(lams, dtype=np.complex128)], dtype=np.complex128)

    def affine(self, scale, shift):
        """g(lam) = f(scale*lam + shift)  (shift_and_scale, src/NEPTransformations.jl:92-105)"""
        return Affine(self, scale, shift)

    def __mul__(self, c):
        return Scaled(c, self)

    __rmul__ = __mul__

    def __neg__(self):
        return Scaled(-1.0, self)

    def real_on_reals(self):
        """True when f maps real arguments (scalars and real matrices) to real values with real derivatives -- what decides the
        element type of compute_Mder / compute_Mlincomb / compute_MM results for real input (test/compute_types.jl)"""
        return False


class Monomial(ScalarFun):
    """lam^p  (p=0: one(S); p=1: S)   src/types_poly.jl:83-98"""

    def __init__(self, p):
        self.p = int(p)

    def real_on_reals(self):
        return True

    def derivs(self, lam, k, scale=1.0):
        lam = complex(lam)
        out = np.zeros(k, dtype=np.complex128)
        for j in range(min(k, self.p + 1)):
            out[j] = (math.factorial(self.p) // math.factorial(self.p - j)) * lam ** (self.p - j) * scale ** j
        return out

    def matfun(self, S):
        return np.linalg.matrix_power(np.asarray(S, dtype=complex), self.p)

    def values(self, lams):
        return np.asarray(lams, dtype=np.complex128) ** self.p


class Exp(ScalarFun):
    """exp(c*lam)   (DEP: c = -tau, src/NEPTypes.jl:505-509)"""

    def __init__(self, c):
        self.c = c

    def real_on_reals(self):
        return not np.iscomplexobj(self.c)

    def derivs(self, lam, k, scale=1.0):
        return np.exp(self.c * complex(lam)) * np.power(complex(self.c * scale), np.arange(k))

    def matfun(self, S):
        return sla.expm(self.c * np.asarray(S, dtype=complex))

    def values(self, lams):
        return np.exp(self.c * np.asarray(lams, dtype=np.complex128))


class ISqrt(ScalarFun):
    """1im*sqrt(alpha*lam + beta), principal branch   (gun: src/gallery_extra/NLEVP_native.jl:13-14)"""

    def __init__(self, alpha=1.0, beta=0.0):
        self.alpha, self.beta = alpha, beta

    def derivs(self, lam, k, scale=1.0):
        u = complex(self.alpha * complex(lam) + self.beta)
        out = np.zeros(k, dtype=np.complex128)
        d = 1j * np.sqrt(u)
        out[0] = d
        r = scale * self.alpha / u
        for j in range(k - 1):
            # scale^(j+1) f^(j+1) = scale^j f^(j) * scale*alpha*(1/2 - j)/u
            d = d * (r * (0.5 - j))
            out[j + 1] = d
        return out

    def matfun(self, S):
        S = np.asarray(S, dtype=complex)
        return 1j * sla.sqrtm(self.alpha * S + self.beta * np.eye(S.shape[0]))

    def values(self, lams):
        return 1j * np.sqrt(self.alpha * np.asarray(lams, dtype=np.complex128) + self.beta)


class Scaled(ScalarFun):
    def __init__(self, c, f):
        self.c, self.f = c, f

    def real_on_reals(self):
        return not np.iscomplexobj(self.c) and self.f.real_on_reals()

    def derivs(self, lam, k, scale=1.0):
        return self.c * self.f.derivs(lam, k, scale)

    def matfun(self, S):
        return self.c * self.f.matfun(S)

    def values(self, lams):
        return self.c * self.f.values(lams)


class Affine(ScalarFun):
    """lam -> f(scale*lam + shift)"""

    def __init__(self, f, scale, shift):
        self.f, self.scale, self.shift = f, scale, shift

    def derivs(self, lam, k, scale=1.0):
        # chain rule: d^j/dlam^j f(scale*lam+shift) = scale^j f^(j)(.)
        return self.f.derivs(self.scale * complex(lam) + self.shift, k, scale * self.scale)

    def matfun(self, S):
        S = np.asarray(S, dtype=complex)
        return self.f.matfun(self.scale * S + self.shift * np.eye(S.shape[0]))

    def values(self, lams):
        return self.f.values(self.scale * np.asarray(lams, dtype=np.complex128) + self.shift)


class Sum(ScalarFun):
    def __init__(self, *fs):
        self.fs = fs

    def real_on_reals(self):
        return all(f.real_on_reals() for f in self.fs)

    def derivs(self, lam, k, scale=1.0):
        return sum(f.derivs(lam, k, scale) for f in self.fs)

    def matfun(self, S):
        return sum(f.matfun(S) for f in self.fs)

    def values(self, lams):
        return sum(f.values(lams) for f in self.fs)


class WEPSqrt(ScalarFun):
    """Waveguide corner functions  f(lam) = 1im*sqrt(lam^2 + b*lam + c)*sign-rule + d0
    with the branch Im sqrt >= 0 (src/gallery_extra/waveguide/Waveguide.jl:143-157) and derivatives
    by the three-term recurrence of sqrt_derivative (:580-616), restated here from the ODE
    g^2 = q(lam): with g = sqrt(q), g*g' = q'/2 and Leibniz' rule gives all higher derivatives."""

    def __init__(self, b, c, d0=0.0):
        self.b, self.c, self.d0 = complex(b), complex(c), complex(d0)

    def _sqrt(self, lam):
        a = lam * lam + self.b * lam + self.c
        s = np.sqrt(complex(a))
        # branch rule: multiply by sign(imag(a)) unless imag(a) == 0
        if a.imag != 0:
            s = s * np.sign(a.imag)
        return s

    def derivs(self, lam, k, scale=1.0):
        lam = complex(lam)
        # Taylor coefficients t_j = scale^j g^(j)/j! of g = sqrt(q) around lam, q = q0 + q1 h + q2 h^2
        q1 = (2 * lam + self.b) * scale
        q2 = 1.0 * scale * scale
        t = np.zeros(max(k, 1), dtype=np.complex128)
        t[0] = self._sqrt(lam)
        # g^2 = q  ->  sum_{i=0..m} t_i t_{m-i} = q_m
        for m in range(1, k):
            qm = q1 if m == 1 else (q2 if m == 2 else 0.0)
            s = 0.0 + 0j
            for i in range(1, m):
                s += t[i] * t[m - i]
            t[m] = (qm - s) / (2 * t[0])
        out = np.empty(k, dtype=np.complex128)
        fact = 1.0
        for j in range(k):
            if j > 0:
                fact *= j
            out[j] = 1j * t[j] * fact
        out[0] += self.d0
        return out

    def matfun(self, S):
        raise NotImplementedError("WEPSqrt.matfun (compute_MM with non-diagonal S) is not supported")


class FromMatrixFunction(ScalarFun):
    """Generic fallback for user functions given the reference way: a callable valid for scalars
    and square matrices.  Derivatives via the Jordan-matrix trick of src/NEPTypes.jl:376-388
    (host, O(k^3)); use only for functions without a closed form."""

    def __init__(self, f):
        self.fun = f

    def derivs(self, lam, k, scale=1.0):
        if k == 1:
            return np.array([self.fun(complex(lam))], dtype=np.complex128)
        S = np.zeros((k, k), dtype=complex)
        S[np.arange(k), np.arange(k)] = lam
        S[np.arange(1, k), np.arange(k - 1)] = scale * np.arange(1, k)
        return np.asarray(self.fun(S), dtype=np.complex128)[:, 0].copy()

    def matfun(self, S):
        return np.asarray(self.fun(np.asarray(S, dtype=complex)))


class ExpSqrt(FromMatrixFunction):
    """lam -> exp(gamma * sqrt(alpha * lam + beta)): the wave-number functions of the "particle in a canyon" example,
    test/nleigs/particle_test_utils.jl:148-155 -- exp(i sqrt(m (lam - c))) for branch points c below the interval of
    interest (gamma = i, alpha = m, beta = -m c) and exp(-sqrt(m (c - lam))) from there on (gamma = -1, alpha = -m,
    beta = m c).  Value and matrix function in closed form (principal square root), higher derivatives through the
    Jordan-block evaluation of the base class."""

    def __init__(self, gamma, alpha, beta):
        self.gamma, self.alpha, self.beta = complex(gamma), float(alpha), float(beta)
        super().__init__(self._eval)

    def _eval(self, S):
        if isinstance(S, np.ndarray) and S.ndim == 2:
            return sla.expm(self.gamma * sla.sqrtm(self.alpha * S.astype(complex) + self.beta * np.eye(S.shape[0])))
        return np.exp(self.gamma * np.sqrt(self.alpha * complex(S) + self.beta))

    def values(self, lams):
        return np.exp(self.gamma * np.sqrt(self.alpha * np.asarray(lams, dtype=np.complex128) + self.beta))


def one():
    return Monomial(0)


def ident():
    return Monomial(1)
