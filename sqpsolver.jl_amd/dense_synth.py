"""A synthetic NLP with a DENSE Lagrangian Hessian, for the dense MFMA LDL^T path in situ (bench.py --workload dense;
BASELINE.json north_star: "dense LDL^T tiled on MFMA where the Hessian is dense"):

    min  1/2 x'Qx + c'x + kappa/4 sum_i x_i^4     s.t.  A x = b,   -1 <= x <= 1

Q = G G' / r + diag(d) is dense and positive definite, A dense with unit rows, b = A x_feas for an interior x_feas; the
scenarios of a batch share Q and A and differ in c and b (like the load scaling of the ACOPF contingency scenarios).  No
reference counterpart -- the reference's examples are ACOPF models with sparse Hessians -- but the shape SqpSolver.Model
describes (/root/reference/src/model.jl:3-35).  Device callbacks: csrc/acopf_dev.hpp dense_eval; CPU twin: oracle/problems.c
ora_problem_dense."""
from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass
class DenseNlp:
    n: int
    m: int
    Q: np.ndarray          # [n][n] symmetric
    A: np.ndarray          # [m][n] row-major
    c: np.ndarray          # [n]
    b: np.ndarray          # [m]
    kappa: float
    seed: int


@dataclasses.dataclass
class DenseLayout:
    """What SqpSolver.Model holds for this problem (1-based COO structures, bounds, start)."""
    n: int
    m: int
    num_linear: int
    jrow: np.ndarray
    jcol: np.ndarray
    hrow: np.ndarray
    hcol: np.ndarray
    xL: np.ndarray
    xU: np.ndarray
    gL: np.ndarray
    gU: np.ndarray
    x0: np.ndarray


def dense_synth(n: int = 1920, m: int = 128, seed: int = 7, kappa: float = 1.0) -> DenseNlp:
    rng = np.random.default_rng(seed)
    r = max(8, n // 4)
    G = rng.standard_normal((n, r))
    Q = G @ G.T / r + np.diag(rng.uniform(0.5, 1.5, n))
    Q = 0.5 * (Q + Q.T)
    A = rng.standard_normal((m, n))
    A /= np.linalg.norm(A, axis=1)[:, None]
    x_feas = rng.uniform(-0.5, 0.5, n)
    return DenseNlp(n, m, Q, A, 0.5 * rng.standard_normal(n), A @ x_feas, kappa, seed)


def dense_scenario(base: DenseNlp, s: int) -> DenseNlp:
    """Scenario s of a base problem: the same Q and A (shared by the batch on the device), linear cost and right-hand sides
    perturbed from seed base.seed * 1000 + s."""
    if s == 0:
        return base
    rng = np.random.default_rng(base.seed * 1000 + s)
    x_feas = rng.uniform(-0.5, 0.5, base.n)
    return dataclasses.replace(base, c=base.c * rng.uniform(0.8, 1.2) + 0.1 * rng.standard_normal(base.n), b=base.A @ x_feas)


def dense_layout(P: DenseNlp) -> DenseLayout:
    n, m = P.n, P.m
    jrow = np.repeat(np.arange(1, m + 1, dtype=np.int64), n)              # A row-major
    jcol = np.tile(np.arange(1, n + 1, dtype=np.int64), m)
    hcol = np.concatenate([np.full(n - j, j + 1, dtype=np.int64) for j in range(n)])      # lower triangle, column-major
    hrow = np.concatenate([np.arange(j + 1, n + 1, dtype=np.int64) for j in range(n)])
    return DenseLayout(n, m, m, jrow, jcol, hrow, hcol, -np.ones(n), np.ones(n), P.b.copy(), P.b.copy(), np.zeros(n))
