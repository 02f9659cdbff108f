"""Polynomial form of the ridge inverse for large alphas (host side, numpy; tiny).

Under ``normalpha`` the penalty is ``a^2 = alpha^2 lambda_max``, so with ``x = K[tr,tr] / lambda_max`` (spectrum in
[0, 1])

    K[va,tr] (K[tr,tr] + a^2 I)^-1  =  (K[va,tr] / lambda_max) (x + alpha^2)^-1  ~=  sum_j c_j P'_j ,
    P'_j = K[va,tr] K[tr,tr]^j / lambda_max^(j+1)

for any polynomial ``q(x) = sum_j c_j x^j`` close to ``1 / (x + alpha^2)`` on [0, 1].  The relative error of every
spectral component of the hat matrix is the residual ``|1 - (x + alpha^2) q(x)|``; the polynomial that minimises its
maximum over [0, 1] is the classical Chebyshev one,

    1 - (x + alpha^2) q(x) = T_d(2x - 1) / T_d(-(1 + 2 alpha^2)),      max residual = 1 / T_d(1 + 2 alpha^2),

``d`` = number of terms.  At alpha = 7.85 four terms leave 5e-10 where the truncated Neumann (Taylor) series needs
five for 1.1e-9; the coefficients differ from the Taylor ones ((-1)^j alpha^-2(j+1)) only in their last digits.
The ``P'_j`` do not depend on alpha: they are shared by every alpha of an inner fold.
"""
import functools
import math

import numpy as np
from numpy.polynomial import chebyshev as _C
from numpy.polynomial import polynomial as _P


def residual_bound(alpha: float, terms: int) -> float:
    """max over the spectrum of the relative error of the ``terms``-term minimax polynomial: 1 / T_d(1 + 2 alpha^2)."""
    c = 1.0 + 2.0 * float(alpha) ** 2
    return 1.0 / math.cosh(terms * math.acosh(c))


def minimax_inverse_coefficients(alpha: float, terms: int) -> np.ndarray:
    """c_0 .. c_{terms-1} (float64) of the polynomial q minimising max_{x in [0,1]} |1 - (x + alpha^2) q(x)|."""
    return np.array(_minimax_cached(float(alpha), int(terms)))


@functools.lru_cache(maxsize=4096)
def _minimax_cached(alpha: float, terms: int) -> tuple:
    a2 = float(alpha) ** 2
    t_poly = _C.cheb2poly([0.0] * terms + [1.0])            # T_d as a monomial polynomial in t
    px, powk, lin = np.zeros(1), np.ones(1), np.array([-1.0, 2.0])
    for ck in t_poly:                                        # substitute t = 2x - 1
        px = _P.polyadd(px, ck * powk)
        powk = _P.polymul(powk, lin)
    r = px / _P.polyval(-a2, px)                             # residual polynomial, r(-alpha^2) = 1
    q, rem = _P.polydiv(_P.polysub(np.ones(1), r), np.array([a2, 1.0]))
    if not np.all(np.abs(rem) < 1e-9):                       # exact division up to rounding
        raise ArithmeticError("minimax polynomial: (x + alpha^2) does not divide 1 - r(x)")
    out = np.zeros(terms, dtype=np.float64)
    out[: q.size] = q
    return tuple(out.tolist())
