"""Convex quadratic programmes with published optima (W. Hock, K. Schittkowski, Test Examples for Nonlinear
Programming Codes, Springer 1981): problems 35 (Beale) and 76.  Both are QPs, so one call of the sub-problem seat at
x_k = 0 with a wide trust region must return p = x*.  Data only (objective matrices, rows, bounds, optimum)."""
import numpy as np

inf = np.inf

HS_QPS = {
    # min 9 - 8x1 - 6x2 - 4x3 + 2x1^2 + 2x2^2 + x3^2 + 2x1x2 + 2x1x3   s.t.  x1 + x2 + 2x3 <= 3,  x >= 0
    "hs035": dict(
        H=np.array([[4., 2, 2], [2, 4, 0], [2, 0, 2]]), c=np.array([-8., -6, -4]), f0=9.0,
        A=np.array([[1., 1, 2]]), gL=np.array([-inf]), gU=np.array([3.]), xL=np.zeros(3), xU=np.full(3, inf),
        x=np.array([4 / 3, 7 / 9, 4 / 9]), f=1 / 9,
        lam=np.array([-2 / 9])),                 # JuMP sign: a row active at its upper side has lambda <= 0
    # min x1^2 + .5x2^2 + x3^2 + .5x4^2 - x1x3 + x3x4 - x1 - 3x2 + x3 - x4
    # s.t. x1 + 2x2 + x3 + x4 <= 5,  3x1 + x2 + 2x3 - x4 <= 4,  x2 + 4x3 >= 1.5,  x >= 0
    "hs076": dict(
        H=np.array([[2., 0, -1, 0], [0, 1, 0, 0], [-1, 0, 2, 1], [0, 0, 1, 1]]), c=np.array([-1., -3, 1, -1]), f0=0.0,
        A=np.array([[1., 2, 1, 1], [3, 1, 2, -1], [0, 1, 4, 0]]), gL=np.array([-inf, -inf, 1.5]),
        gU=np.array([5., 4, inf]), xL=np.zeros(4), xU=np.full(4, inf),
        x=np.array([3 / 11, 23 / 11, 0.0, 6 / 11]), f=-103 / 22,
        lam=np.array([-5 / 11, 0.0, 0.0])),
}


def structure(q):
    """1-based COO structures (Jacobian; lower-triangular Hessian) and their values, as the seat takes them."""
    jr, jc = np.nonzero(q["A"])
    hr, hc = np.nonzero(np.tril(q["H"]))
    return dict(n=len(q["c"]), m=len(q["gL"]), jrow=jr + 1, jcol=jc + 1, hrow=hr + 1, hcol=hc + 1,
                jval=q["A"][jr, jc], hval=q["H"][hr, hc])
