"""Synthetic ACOPF-shaped NLP instances (input data for the hot path).

The reference never ships ACOPF data besides a 3-bus file
(/root/reference/examples/acopf/case3.m) and builds its ACOPF models through the
un-vendored PowerModels.jl (`ACPPowerModel` + `build_opf`,
/root/reference/test/opf.jl:5-9, /root/reference/examples/acopf/opf.jl:12-46).
This module generates networks of the IEEE-14/118/1354/9241 *shape* and lays the
NLP out the way `SqpSolver.Optimizer` would see it
(/root/reference/src/MOI_wrapper.jl:759-766 row ordering, :1083-1086 linear-row
count, :930-945 Jacobian COO, :1010-1025 Hessian COO with duplicates).

Pure numpy, deterministic per seed (PCG64).  This is *data synthesis*: both the
CPU oracle and the HIP evaluator consume the same arrays.

Variable order (PowerModels creation order):
    va[nb], vm[nb], pg[ng], qg[ng], p_f[nl], p_t[nl], q_f[nl], q_t[nl]
Row order (0-based blocks):
    [0, nl)            angle difference  va_f - va_t <= angmax      (linear <=)
    [nl, 2nl)          angle difference  va_f - va_t >= angmin      (linear >=)
    2nl                reference angle   va_ref == 0                (linear ==)
    2nl+1+2i, +1       bus i active / reactive balance              (linear ==)
    T0 + 2l, +1        thermal limit from / to end  p^2+q^2 <= s^2  (quadratic <=)
    O0 + 4l + {0..3}   Ohm's law p_f, q_f, p_t, q_t                 (NLP block, ==0)
with T0 = 2nl+1+2nb, O0 = T0+2nl;  n = 2nb+2ng+4nl,  m = 1+2nb+8nl.
"""
from __future__ import annotations

import dataclasses
import numpy as np

__all__ = ["Network", "acopf_synth", "acopf_synth_geo", "synth_case", "contingency", "renumber_buses", "NlpLayout", "acopf_layout", "acr_layout", "acwr_layout",
           "CASES"]

# nb, ng, nl per SURVEY.md section 8 table
CASES = {
    "case14": (14, 5, 20, 14),
    "case118": (118, 54, 186, 118),
    "case1354": (1354, 260, 1991, 1354),
    "case9241": (9241, 1445, 16049, 9241),
}


@dataclasses.dataclass
class Network:
    nb: int
    ng: int
    nl: int
    # buses
    pd: np.ndarray
    qd: np.ndarray
    vmin: np.ndarray
    vmax: np.ndarray
    ref_bus: int
    # generators
    gen_bus: np.ndarray      # int32 [ng]
    pmin: np.ndarray
    pmax: np.ndarray
    qmin: np.ndarray
    qmax: np.ndarray
    c2: np.ndarray           # cost in per-unit variables: c2*pg^2 + c1*pg
    c1: np.ndarray
    # branches
    f_bus: np.ndarray        # int32 [nl]
    t_bus: np.ndarray
    r: np.ndarray
    x: np.ndarray
    bc: np.ndarray           # total line charging susceptance
    rate_a: np.ndarray
    angmin: np.ndarray
    angmax: np.ndarray
    status: np.ndarray       # 1.0 in service, 0.0 outaged (admittance zeroed, pattern kept)
    tap: np.ndarray = None   # off-nominal turns ratio at the from end (None = 1)
    shift: np.ndarray = None # phase shift at the from end, radians (None = 0)
    gs: np.ndarray = None    # bus shunt conductance, p.u. (MW consumed at vm = 1 / baseMVA); None = 0
    bs: np.ndarray = None    # bus shunt susceptance, p.u. (MVAr injected at vm = 1 / baseMVA); None = 0
    # HVDC lines (MATPOWER mpc.dcline; PowerModels variable_dcline_power + constraint_dcline_power_losses, the
    # reference's build at examples/acopf/opf.jl:16,40-42): None = no dc lines.  Dict of arrays, one entry per line:
    #   f_bus, t_bus (int32), pminf, pmaxf (active power leaving the from bus into the line, p.u.),
    #   qminf, qmaxf, qmint, qmaxt (reactive power drawn at either end, p.u.), loss0 (p.u.), loss1
    dcline: dict = None

    @property
    def ndc(self):
        return 0 if self.dcline is None else len(self.dcline["f_bus"])

    def shunts(self):
        """(bus indices, gs, bs) of the buses with a shunt element; they add gs*vm^2 to the P balance and
        -bs*vm^2 to the Q balance of the bus (MATPOWER manual eq. 3.7; PowerModels constraint_power_balance)."""
        gs = np.zeros(self.nb) if self.gs is None else np.asarray(self.gs, dtype=np.float64)
        bs = np.zeros(self.nb) if self.bs is None else np.asarray(self.bs, dtype=np.float64)
        idx = np.flatnonzero((gs != 0.0) | (bs != 0.0)).astype(np.int32)
        return idx, gs[idx], bs[idx]

    def branch_coeffs(self):
        """Per-branch Ohm's-law coefficients of the pi model with an ideal transformer (ratio tau, shift phi) at the
        from end (MATPOWER / PowerModels convention; SURVEY.md App. B is the case tau = 1, phi = 0).

        Flow k of branch l (k = 0..3: p_f, q_f, p_t, q_t), th = va_f - va_t:
            F_k = A_k * v_self^2 + v_f v_t (Bc_k cos th + Bs_k sin th),   v_self = v_f (k < 2) or v_t
        with y = g + jb the series admittance and bc the total charging susceptance:
            p_f: A = g/tau^2          Bc = -(g cos phi - b sin phi)/tau   Bs = -(g sin phi + b cos phi)/tau
            q_f: A = -(b+bc/2)/tau^2  Bc =  (g sin phi + b cos phi)/tau   Bs = -(g cos phi - b sin phi)/tau
            p_t: A = g                Bc = -(g cos phi + b sin phi)/tau   Bs = -(g sin phi - b cos phi)/tau
            q_t: A = -(b+bc/2)        Bc =  (b cos phi - g sin phi)/tau   Bs =  (g cos phi + b sin phi)/tau
        Returned array [nl, 12] = (A, Bc, Bs) for k = 0..3, scaled by status (outaged branch: all zero).
        """
        z2 = self.r ** 2 + self.x ** 2
        g = self.r / z2
        b = -self.x / z2
        bh = 0.5 * self.bc
        tau = np.ones(self.nl) if self.tap is None else np.asarray(self.tap, dtype=np.float64)
        phi = np.zeros(self.nl) if self.shift is None else np.asarray(self.shift, dtype=np.float64)
        c, sn = np.cos(phi), np.sin(phi)
        co = np.stack([
            g / tau ** 2, -(g * c - b * sn) / tau, -(g * sn + b * c) / tau,
            -(b + bh) / tau ** 2, (g * sn + b * c) / tau, -(g * c - b * sn) / tau,
            g, -(g * c + b * sn) / tau, -(g * sn - b * c) / tau,
            -(b + bh), (b * c - g * sn) / tau, (g * c + b * sn) / tau], axis=1)
        return co * self.status[:, None]


def _bridges(nb, f, t):
    """Indices of branches whose removal disconnects the graph (iterative DFS)."""
    adj = [[] for _ in range(nb)]
    for k, (a, b) in enumerate(zip(f, t)):
        adj[a].append((b, k))
        adj[b].append((a, k))
    disc = [-1] * nb
    low = [0] * nb
    out = set()
    timer = 0
    for root in range(nb):
        if disc[root] != -1:
            continue
        stack = [(root, -1, 0)]
        disc[root] = low[root] = timer
        timer += 1
        while stack:
            u, pe, i = stack.pop()
            if i < len(adj[u]):
                stack.append((u, pe, i + 1))
                v, k = adj[u][i]
                if k == pe:
                    continue
                if disc[v] == -1:
                    disc[v] = low[v] = timer
                    timer += 1
                    stack.append((v, k, 0))
                else:
                    low[u] = min(low[u], disc[v])
            else:
                if stack:
                    par = stack[-1][0]
                    low[par] = min(low[par], low[u])
                    if low[u] > disc[par]:
                        out.add(pe)
    return out


def default_load_scale(nb: int) -> float:
    """Per-bus demand is U(0.1,1.0) p.u. times this factor.  1.0 is AC-feasible for the 14-bus shape;
    at 118 buses and beyond the generated graphs have a long electrical diameter and full demand cannot
    be served inside the 0.94-1.06 voltage band (SQP-TR then converges to an infeasible stationary
    point for every scenario, measured: prim_infeas stalls at 0.37), so larger shapes carry half of
    it -- 0.5 leaves 15 of the first 16 N-1 scenarios of case118 feasible, about the mix a real
    contingency screen sees."""
    return 1.0 if nb <= 30 else 0.5


def acopf_synth(nb: int, ng: int, nl: int, seed: int, load_scale: float | None = None) -> Network:
    """Connected synthetic transmission network."""
    assert nl >= nb - 1 and ng <= nb
    if load_scale is None:
        load_scale = default_load_scale(nb)
    rng = np.random.default_rng(seed)
    edges = set()
    f_bus, t_bus = [], []
    # random spanning tree with locally biased parents
    for i in range(1, nb):
        back = int(min(i - 1, np.floor(rng.exponential(2.0))))
        j = i - 1 - back
        edges.add((j, i))
        f_bus.append(j)
        t_bus.append(i)
    # locally biased chords
    while len(f_bus) < nl:
        a = int(rng.integers(0, nb - 1))
        b = a + 1 + int(np.floor(rng.exponential(4.0)))
        if b >= nb or (a, b) in edges:
            continue
        edges.add((a, b))
        f_bus.append(a)
        t_bus.append(b)
    f_bus = np.asarray(f_bus, dtype=np.int32)
    t_bus = np.asarray(t_bus, dtype=np.int32)
    r = rng.uniform(0.005, 0.05, nl)
    x = np.maximum(rng.uniform(0.05, 0.3, nl), 3.0 * r)
    bc = rng.uniform(0.0, 0.1, nl)
    pd = rng.uniform(0.1, 1.0, nb) * load_scale
    qd = 0.3 * pd
    gen_bus = np.sort(rng.choice(nb, size=ng, replace=False)).astype(np.int32)
    w = rng.uniform(0.5, 1.5, ng)
    pmax = 1.6 * pd.sum() * w / w.sum()
    pmin = np.zeros(ng)
    qmax = 0.6 * pmax
    qmin = -0.6 * pmax
    base = 100.0
    c2 = rng.uniform(0.01, 0.1, ng) * base * base
    c1 = rng.uniform(10.0, 40.0, ng) * base
    # thermal ratings from a DC power flow with proportional dispatch
    pg0 = pmax * (pd.sum() / pmax.sum())
    inj = -pd.copy()
    np.add.at(inj, gen_bus, pg0)
    B = np.zeros((nb, nb))
    for a, b, xx in zip(f_bus, t_bus, x):
        B[a, a] += 1 / xx
        B[b, b] += 1 / xx
        B[a, b] -= 1 / xx
        B[b, a] -= 1 / xx
    theta = np.zeros(nb)
    theta[1:] = np.linalg.solve(B[1:, 1:], inj[1:])
    flow = (theta[f_bus] - theta[t_bus]) / x
    rate_a = 1.5 * np.abs(flow) + 0.3 * max(1.0, float(np.median(np.abs(flow))))
    ang = np.full(nl, np.pi / 6)
    return Network(
        nb=nb, ng=ng, nl=nl, pd=pd, qd=qd,
        vmin=np.full(nb, 0.94), vmax=np.full(nb, 1.06), ref_bus=0,
        gen_bus=gen_bus, pmin=pmin, pmax=pmax, qmin=qmin, qmax=qmax, c2=c2, c1=c1,
        f_bus=f_bus, t_bus=t_bus, r=r, x=x, bc=bc, rate_a=rate_a,
        angmin=-ang, angmax=ang.copy(), status=np.ones(nl),
    )


def acopf_synth_geo(nb: int, ng: int, nl: int, seed: int, width: int | None = None, load_scale: float = 0.5,
                    qd_frac: float = 0.1, x_scale: float = 0.25, compensate: float = 0.0, regions: int = 1) -> Network:
    """Synthetic transmission network with a geography (round 3; VERDICT r2 "missing" #2): the large shapes of
    `acopf_synth` are chains hundreds of branches long that SQP-TR cannot bring to feasibility from a flat start inside
    any reasonable iteration budget.  Here the buses sit on a strip of a square lattice, `width`
    buses across (default 5: fronts of up to 134 rows at 1354 buses, 382 at 9241; 8 across gives 208 / 548), numbered column by column; branches are lattice edges -- a random
    spanning tree of the lattice plus random further lattice edges up to nl -- so every branch is short, the graph is
    planar and its separators are `width` buses wide (fronts of the multifrontal factorisation stay below ~200 rows);
    generators are spread evenly along the strip (every (nb / ng)-th bus, jittered) with capacity proportional to the
    demand of their neighbourhood, so power is consumed where it is produced; line charging is light and every
    generator can absorb its share of it.  Series impedances are a quarter of `acopf_synth`'s (x in 0.0125 .. 0.075 p.u.:
    short high-voltage lines) and the reactive demand is a tenth of the active one (power factor 0.995: compensated loads)
    -- reactive power cannot travel through a +-6 % voltage band over reactances of 0.3 p.u., which is what kept the
    large shapes of `acopf_synth` away from feasibility; measured with the oracle at 1354 buses: x_scale 1.0 / 0.4 /
    0.25 with 5 buses across -> no convergence in 40 iterations / none / converged in 14.  Costs, voltage band, angle
    limits and the thermal ratings from a DC power flow are those of `acopf_synth`.  Deterministic per seed."""
    assert nl >= nb - 1 and ng <= nb
    rng = np.random.default_rng(seed)
    W = int(width) if width else 5
    L = (nb + W - 1) // W

    def bus(r, c):
        return c * W + r

    # regions > 1: the strip is cut into that many regional strips (contiguous bus ranges), joined by two tie lines each
    # to the region (k - 1) // 2 -- a binary tree of regions, so the regions are parallel subtrees of the assembly tree
    # instead of one strip thousands of columns long
    R = max(1, int(regions))
    colcut = np.linspace(0, L, R + 1).astype(int)
    region_of_col = np.zeros(L, dtype=int)
    for k in range(R):
        region_of_col[colcut[k]:colcut[k + 1]] = k
    cand, ties = [], []                             # lattice edges among the nb buses (the last column may be short)
    for c in range(L):
        for r in range(W):
            a = bus(r, c)
            if a >= nb:
                continue
            if r + 1 < W and bus(r + 1, c) < nb:
                cand.append((a, bus(r + 1, c)))
            if c + 1 < L and bus(r, c + 1) < nb and region_of_col[c] == region_of_col[c + 1]:
                cand.append((a, bus(r, c + 1)))
    for k in range(1, R):
        par = (k - 1) // 2
        ca, cb = colcut[k], (colcut[par] + colcut[par + 1]) // 2          # first column of the region, middle of its parent
        for r in (0, W - 1):
            if bus(r, ca) < nb and bus(r, cb) < nb:
                ties.append((min(bus(r, ca), bus(r, cb)), max(bus(r, ca), bus(r, cb))))
    cand = ties + cand
    assert len(cand) >= nl, "the lattice has fewer edges than nl: choose a wider strip"
    order = np.concatenate([np.arange(len(ties)), len(ties) + rng.permutation(len(cand) - len(ties))]).astype(int)
    parent = list(range(nb))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    tree, rest = [], []
    for k in order:                                 # randomised Kruskal: a uniform-ish random spanning tree
        a, b = cand[k]
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
            tree.append((a, b))
        else:
            rest.append((a, b))
    assert len(tree) == nb - 1
    tie_set = set(ties)
    rest = [e for e in rest if e in tie_set] + [e for e in rest if e not in tie_set]      # every tie line is built
    edges = tree + rest[:nl - (nb - 1)]
    f_bus = np.asarray([min(a, b) for a, b in edges], dtype=np.int32)
    t_bus = np.asarray([max(a, b) for a, b in edges], dtype=np.int32)
    r = rng.uniform(0.005, 0.05, nl) * x_scale
    x = np.maximum(rng.uniform(0.05, 0.3, nl) * x_scale, 3.0 * r)
    bc = rng.uniform(0.0, 0.04, nl)
    pd = rng.uniform(0.1, 1.0, nb) * load_scale
    qd = qd_frac * pd
    # generators: one per stretch of nb / ng buses, at a random bus of the stretch; capacity 1.6 x the demand of the stretch
    cuts = np.linspace(0, nb, ng + 1).astype(int)
    gen_bus = np.asarray([int(rng.integers(cuts[g], max(cuts[g] + 1, cuts[g + 1]))) for g in range(ng)], dtype=np.int32)
    area = np.asarray([pd[cuts[g]:cuts[g + 1]].sum() for g in range(ng)])
    pmax = 1.6 * np.maximum(area, 0.2 * area.mean()) * rng.uniform(0.9, 1.1, ng)
    pmin = np.zeros(ng)
    qmax = 0.6 * pmax
    qmin = -0.6 * pmax
    base = 100.0
    c2 = rng.uniform(0.01, 0.1, ng) * base * base
    c1 = rng.uniform(10.0, 40.0, ng) * base
    # thermal ratings from a DC power flow with proportional dispatch (sparse: the lattice Laplacian)
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    pg0 = pmax * (pd.sum() / pmax.sum())
    inj = -pd.copy()
    np.add.at(inj, gen_bus, pg0)
    w = 1.0 / x
    Bm = sp.coo_matrix((np.concatenate([w, w, -w, -w]),
                        (np.concatenate([f_bus, t_bus, f_bus, t_bus]), np.concatenate([f_bus, t_bus, t_bus, f_bus]))),
                       shape=(nb, nb)).tocsc()
    theta = np.zeros(nb)
    theta[1:] = spla.spsolve(Bm[1:, 1:], inj[1:])
    flow = (theta[f_bus] - theta[t_bus]) / x
    rate_a = 1.5 * np.abs(flow) + 0.3 * max(1.0, float(np.median(np.abs(flow))))
    ang = np.full(nl, np.pi / 6)
    return Network(
        nb=nb, ng=ng, nl=nl, pd=pd, qd=qd,
        vmin=np.full(nb, 0.94), vmax=np.full(nb, 1.06), ref_bus=0,
        gen_bus=gen_bus, pmin=pmin, pmax=pmax, qmin=qmin, qmax=qmax, c2=c2, c1=c1,
        f_bus=f_bus, t_bus=t_bus, r=r, x=x, bc=bc, rate_a=rate_a,
        angmin=-ang, angmax=ang.copy(), status=np.ones(nl),
        bs=(compensate * qd if compensate > 0.0 else None),
    )


def synth_case(case: str, topology: str | None = None) -> Network:
    """The synthetic network of a BASELINE.json case shape.  topology "chain": the SURVEY.md section 8d recipe
    (`acopf_synth`; what the 14- and 118-bus workloads and every test pinned before round 3 use); "geo": `acopf_synth_geo`.
    Default: chain up to 118 buses, geo beyond -- the chain recipe does not give convergent NLPs at 1354 / 9241 buses."""
    nb, ng, nl, seed = CASES[case]
    if topology is None:
        topology = "chain" if nb <= 118 else "geo"
    if topology not in ("chain", "geo"):
        raise ValueError(f"unknown topology {topology!r}")
    if topology == "chain":
        return acopf_synth(nb, ng, nl, seed)
    # 9241 buses carry 1.74 branches per bus (1354: 1.47): 4 buses across and 32 regional strips keep the largest front
    # at 176 rows and the assembly tree at 74 levels (5 across, one strip: 382 rows; 4 across, one strip: 359 levels).
    # Oracle, flat start: 32 regions converge in 15 outer iterations, 16 regions of 5 across in 20; with 16 regions of 4
    # across the first trust-region QP stalls at 1.5e-6 (a nearly singular reduced Hessian under a permanent inertia
    # correction) and ends at the interior-point iteration limit.
    return acopf_synth_geo(nb, ng, nl, seed, width=4, regions=32) if nb > 5000 else acopf_synth_geo(nb, ng, nl, seed)


def contingency(net: Network, s: int, base_seed: int) -> Network:
    """Scenario `s` of a base network: outage of non-bridge branch ~ s mod nl with the
    admittance zeroed (sparsity pattern shared) and loads scaled by U(0.9,1.1)
    drawn from seed base_seed*1000+s."""
    br = _bridges(net.nb, net.f_bus, net.t_bus)
    k = s % net.nl
    while k in br:
        k = (k + 1) % net.nl
    rng = np.random.default_rng(base_seed * 1000 + s)
    scale = rng.uniform(0.9, 1.1)
    out = dataclasses.replace(
        net, pd=net.pd * scale, qd=net.qd * scale, status=net.status.copy())
    out.status[k] = 0.0
    return out


def renumber_buses(net: Network, seed: int) -> Network:
    """The same network with its buses renumbered by a random permutation (branch and generator order kept).
    `acopf_synth` numbers buses along its spanning tree, so neighbours in the numbering are neighbours in the grid;
    real case files carry no such order.  Used by the tests of the orderings: nothing may depend on the numbering."""
    rng = np.random.default_rng(seed)
    new = rng.permutation(net.nb)                 # old bus b becomes bus new[b]
    old = np.empty(net.nb, dtype=np.int64); old[new] = np.arange(net.nb)

    def bus(a):
        return None if a is None else np.asarray(a)[old]
    return dataclasses.replace(
        net, pd=bus(net.pd), qd=bus(net.qd), vmin=bus(net.vmin), vmax=bus(net.vmax), ref_bus=int(new[net.ref_bus]),
        gen_bus=new[net.gen_bus].astype(np.int32), f_bus=new[net.f_bus].astype(np.int32),
        t_bus=new[net.t_bus].astype(np.int32), gs=bus(net.gs), bs=bus(net.bs))


@dataclasses.dataclass
class NlpLayout:
    """What `SqpSolver.Model` holds (/root/reference/src/model.jl:3-35) for an ACOPF."""
    n: int
    m: int
    num_linear: int
    jrow: np.ndarray   # int64, 1-based COO (Julia-native)
    jcol: np.ndarray
    hrow: np.ndarray   # int64, 1-based lower-triangular COO with duplicates
    hcol: np.ndarray
    xL: np.ndarray
    xU: np.ndarray
    gL: np.ndarray
    gU: np.ndarray
    x0: np.ndarray
    # balance-row incidence in CSR form (bus -> list of (col, coef)) for the evaluators
    bal_ptr: np.ndarray   # int32 [nb+1]
    bal_colP: np.ndarray  # int32 column of the P-row entry
    bal_colQ: np.ndarray  # int32 column of the Q-row entry
    bal_coef: np.ndarray  # +1 arc, -1 generator
    # bus shunts (empty for a network without them): the balance rows of a network WITH shunts carry a vm^2 term,
    # so they are nonlinear rows (num_linear = 2 nl + 1), with two extra Jacobian entries (P row, Q row; column vm_i)
    # and one extra Hessian entry (vm_i, vm_i) per shunted bus appended to the COO lists
    sh_bus: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    sh_gs: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    sh_bs: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    dc_loss1: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))   # per dc line (empty: none)
    form: str = "polar"   # "polar" (ACPPowerModel, acopf_layout), "acr" (ACRPowerModel, acr_layout), "acwr" (acwr_layout)
    # W-space form only: bus pairs (i < j), the pair and orientation (+1: from = i) of every branch, tan of the pair's
    # angle limits (coefficients of the linear angle rows)
    bp_i: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    bp_j: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    br_bp: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, dtype=np.int32))
    br_sig: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    bp_tmin: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    bp_tmax: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))


def acopf_layout(net: Network) -> NlpLayout:
    nb, ng, nl, ndc = net.nb, net.ng, net.nl, net.ndc
    # dc lines: 4 variables each (p_f, p_t, q_f, q_t of the line) behind everything else, one loss row each behind
    # every other row (a linear row; sitting behind the nonlinear block it is simply re-linearised every iteration)
    n = 2 * nb + 2 * ng + 4 * nl + 4 * ndc
    m = 1 + 2 * nb + 8 * nl + ndc
    DC = 2 * nb + 2 * ng + 4 * nl              # p_dc_f[ndc], p_dc_t[ndc], q_dc_f[ndc], q_dc_t[ndc]
    VA, VM, PG, QG = 0, nb, 2 * nb, 2 * nb + ng
    PF = 2 * nb + 2 * ng
    PT, QF, QT = PF + nl, PF + 2 * nl, PF + 3 * nl
    T0 = 2 * nl + 1 + 2 * nb
    O0 = T0 + 2 * nl
    f, t = net.f_bus.astype(np.int64), net.t_bus.astype(np.int64)
    L = np.arange(nl, dtype=np.int64)

    jr, jc = [], []
    # angle <= and >= rows: (va_f, +1), (va_t, -1)
    for off in (0, nl):
        jr.append(np.repeat(off + L, 2))
        jc.append(np.stack([VA + f, VA + t], axis=1).ravel())
    # reference angle
    jr.append(np.array([2 * nl]))
    jc.append(np.array([VA + net.ref_bus]))
    # balance rows: arcs at the bus (branch order: from-end then to-end per branch), then gens
    inc = [[] for _ in range(nb)]
    for l in range(nl):
        inc[int(f[l])].append((PF + l, QF + l, 1.0))
        inc[int(t[l])].append((PT + l, QT + l, 1.0))
    for g in range(ng):
        inc[int(net.gen_bus[g])].append((PG + g, QG + g, -1.0))
    for d in range(ndc):                       # a dc line draws p_dc_f / q_dc_f at its from bus, p_dc_t / q_dc_t at its to bus
        inc[int(net.dcline["f_bus"][d])].append((DC + d, DC + 2 * ndc + d, 1.0))
        inc[int(net.dcline["t_bus"][d])].append((DC + ndc + d, DC + 3 * ndc + d, 1.0))
    bal_ptr = np.zeros(nb + 1, dtype=np.int32)
    colP, colQ, coef = [], [], []
    for i in range(nb):
        bal_ptr[i + 1] = bal_ptr[i] + len(inc[i])
        for cp, cq, cf in inc[i]:
            colP.append(cp)
            colQ.append(cq)
            coef.append(cf)
    colP = np.asarray(colP, dtype=np.int64)
    colQ = np.asarray(colQ, dtype=np.int64)
    for i in range(nb):
        s, e = bal_ptr[i], bal_ptr[i + 1]
        jr.append(np.full(e - s, 2 * nl + 1 + 2 * i))
        jc.append(colP[s:e])
        jr.append(np.full(e - s, 2 * nl + 2 + 2 * i))
        jc.append(colQ[s:e])
    # thermal rows: from (p_f, q_f), to (p_t, q_t)
    jr.append(np.repeat(T0 + 2 * L, 2)); jc.append(np.stack([PF + L, QF + L], 1).ravel())
    jr.append(np.repeat(T0 + 2 * L + 1, 2)); jc.append(np.stack([PT + L, QT + L], 1).ravel())
    # interleave thermal from/to per branch to keep creation order
    # (rows are T0+2l and T0+2l+1; COO order need not be sorted)
    # Ohm rows: own flow var, va_f, va_t, vm_f, vm_t
    own = [PF, QF, PT, QT]
    for k in range(4):
        jr.append(np.repeat(O0 + 4 * L + k, 5))
        jc.append(np.stack([own[k] + L, VA + f, VA + t, VM + f, VM + t], 1).ravel())
    sh_bus, sh_gs, sh_bs = net.shunts()
    if len(sh_bus):
        sb = sh_bus.astype(np.int64)
        jr.append(np.stack([2 * nl + 1 + 2 * sb, 2 * nl + 2 + 2 * sb], 1).ravel())
        jc.append(np.repeat(VM + sb, 2))
    if ndc:                                    # loss rows: (1 - loss1) p_dc_f + p_dc_t = loss0
        D = np.arange(ndc, dtype=np.int64)
        jr.append(np.repeat(1 + 2 * nb + 8 * nl + D, 2))
        jc.append(np.stack([DC + D, DC + ndc + D], 1).ravel())
    jrow = np.concatenate(jr).astype(np.int64) + 1
    jcol = np.concatenate(jc).astype(np.int64) + 1

    # Hessian COO (lower triangle, duplicates kept): objective, thermal, Ohm
    hr, hc = [], []
    G = np.arange(ng, dtype=np.int64)
    hr.append(PG + G); hc.append(PG + G)
    for base_p, base_q in ((PF, QF), (PT, QT)):
        hr.append(np.stack([base_p + L, base_q + L], 1).ravel())
        hc.append(np.stack([base_p + L, base_q + L], 1).ravel())
    # per Ohm row the 10 lower-triangular entries of the 4x4 block on (va_f, va_t, vm_f, vm_t)
    v4 = np.stack([VA + f, VA + t, VM + f, VM + t], 1)  # [nl,4]
    pairs = [(a, b) for a in range(4) for b in range(a + 1)]
    for k in range(4):
        for a, b in pairs:
            ia, ib = v4[:, a], v4[:, b]
            hr.append(np.maximum(ia, ib))
            hc.append(np.minimum(ia, ib))
    if len(sh_bus):
        hr.append(VM + sh_bus.astype(np.int64)); hc.append(VM + sh_bus.astype(np.int64))
    hrow = np.concatenate(hr).astype(np.int64) + 1
    hcol = np.concatenate(hc).astype(np.int64) + 1

    inf = np.inf
    xL = np.concatenate([np.full(nb, -inf), net.vmin, net.pmin, net.qmin,
                         -net.rate_a, -net.rate_a, -net.rate_a, -net.rate_a])
    xU = np.concatenate([np.full(nb, inf), net.vmax, net.pmax, net.qmax,
                         net.rate_a, net.rate_a, net.rate_a, net.rate_a])
    if ndc:
        dc = net.dcline
        # p_dc_t = loss0 - (1 - loss1) p_dc_f: its box is the image of the from-end box (implied by the loss row)
        a = dc["loss0"] - (1.0 - dc["loss1"]) * dc["pmaxf"]; b = dc["loss0"] - (1.0 - dc["loss1"]) * dc["pminf"]
        xL = np.concatenate([xL, dc["pminf"], np.minimum(a, b), dc["qminf"], dc["qmint"]])
        xU = np.concatenate([xU, dc["pmaxf"], np.maximum(a, b), dc["qmaxf"], dc["qmaxt"]])
    gL = np.empty(m); gU = np.empty(m)
    gL[0:nl] = -inf; gU[0:nl] = net.angmax
    gL[nl:2 * nl] = net.angmin; gU[nl:2 * nl] = inf
    gL[2 * nl] = gU[2 * nl] = 0.0
    gL[2 * nl + 1:T0:2] = -net.pd; gU[2 * nl + 1:T0:2] = -net.pd
    gL[2 * nl + 2:T0:2] = -net.qd; gU[2 * nl + 2:T0:2] = -net.qd
    gL[T0:O0] = -inf
    gU[T0:O0:2] = net.rate_a ** 2; gU[T0 + 1:O0:2] = net.rate_a ** 2
    gL[O0:] = 0.0; gU[O0:] = 0.0
    if ndc:
        gL[O0 + 4 * nl:] = net.dcline["loss0"]; gU[O0 + 4 * nl:] = net.dcline["loss0"]
    # start: midpoint of finite boxes, 0 otherwise
    # (/root/reference/examples/acopf/init_opf.jl:25-47)
    boxed = np.isfinite(xL) & np.isfinite(xU)
    x0 = np.zeros(n)
    x0[boxed] = 0.5 * (xL[boxed] + xU[boxed])
    x0[VM:VM + nb] = 1.0
    return NlpLayout(n=n, m=m, num_linear=T0 if len(sh_bus) == 0 else 2 * nl + 1,
                     jrow=jrow, jcol=jcol, hrow=hrow, hcol=hcol,
                     xL=xL, xU=xU, gL=gL, gU=gU, x0=x0,
                     bal_ptr=bal_ptr, bal_colP=colP.astype(np.int32),
                     bal_colQ=colQ.astype(np.int32), bal_coef=np.asarray(coef),
                     sh_bus=sh_bus, sh_gs=sh_gs, sh_bs=sh_bs,
                     dc_loss1=(np.zeros(0) if ndc == 0 else np.asarray(net.dcline["loss1"], dtype=np.float64)))


def acr_layout(net: Network) -> NlpLayout:
    """The same network in rectangular voltage coordinates: PowerModels' ACRPowerModel under the `build_opf` of
    /root/reference/examples/acopf/opf.jl:12-43 (the formulation `run_sqp_opf` instantiates, :46,:51).

    variables  vi[nb] (where the polar layout has va), vr[nb] (where it has vm), pg, qg, p_f, p_t, q_f, q_t, dc lines;
               vr, vi in [-vmax, vmax] (variable_bus_voltage_real / _imaginary, bounded), start vr = 1, vi = 0
    rows       0                vi[ref] = 0                                  (constraint_theta_ref, ACR form)
               1 .. 2 nb        power balance, P then Q per bus              (linear without bus shunts)
               V0 + 2 i, +1     vmin_i^2 <= vr_i^2 + vi_i^2,  vr_i^2 + vi_i^2 <= vmax_i^2
                                                                             (constraint_voltage_magnitude_bounds: two rows)
               T0 + 2 l, +1     p^2 + q^2 <= rate^2, from and to end         (constraint_thermal_limit_from / _to)
               O0 + 4 l + k     flow_k - F_k = 0, F_k = A (vr_s^2 + vi_s^2) + Bc (vr_f vr_t + vi_f vi_t)
                                                        + Bs (vi_f vr_t - vr_f vi_t)
                                (constraint_ohms_yt_from / _to: the polar form with v_f v_t cos th and v_f v_t sin th written
                                out; same twelve coefficients per branch, Network.branch_coeffs)
               D0 + d           dc-line loss rows
    The angle-difference rows are not part of this build (opf.jl:33 has them commented out)."""
    nb, ng, nl, ndc = net.nb, net.ng, net.nl, net.ndc
    n = 2 * nb + 2 * ng + 4 * nl + 4 * ndc
    m = 1 + 4 * nb + 6 * nl + ndc
    VI, VR, PG, QG = 0, nb, 2 * nb, 2 * nb + ng
    PF = 2 * nb + 2 * ng
    PT, QF, QT = PF + nl, PF + 2 * nl, PF + 3 * nl
    DC = PF + 4 * nl
    V0 = 1 + 2 * nb
    T0 = V0 + 2 * nb
    O0 = T0 + 2 * nl
    D0 = O0 + 4 * nl
    f, t = net.f_bus.astype(np.int64), net.t_bus.astype(np.int64)
    L = np.arange(nl, dtype=np.int64)
    I = np.arange(nb, dtype=np.int64)

    jr, jc = [np.array([0])], [np.array([VI + net.ref_bus])]
    inc = [[] for _ in range(nb)]
    for l in range(nl):
        inc[int(f[l])].append((PF + l, QF + l, 1.0))
        inc[int(t[l])].append((PT + l, QT + l, 1.0))
    for g in range(ng):
        inc[int(net.gen_bus[g])].append((PG + g, QG + g, -1.0))
    for d in range(ndc):
        inc[int(net.dcline["f_bus"][d])].append((DC + d, DC + 2 * ndc + d, 1.0))
        inc[int(net.dcline["t_bus"][d])].append((DC + ndc + d, DC + 3 * ndc + d, 1.0))
    bal_ptr = np.zeros(nb + 1, dtype=np.int32)
    colP, colQ, coef = [], [], []
    for i in range(nb):
        bal_ptr[i + 1] = bal_ptr[i] + len(inc[i])
        for cp, cq, cf in inc[i]:
            colP.append(cp); colQ.append(cq); coef.append(cf)
    colP = np.asarray(colP, dtype=np.int64)
    colQ = np.asarray(colQ, dtype=np.int64)
    for i in range(nb):
        s, e = bal_ptr[i], bal_ptr[i + 1]
        jr.append(np.full(e - s, 1 + 2 * i)); jc.append(colP[s:e])
        jr.append(np.full(e - s, 2 + 2 * i)); jc.append(colQ[s:e])
    # voltage-magnitude rows: (vr_i, vi_i) for the lower row, then for the upper row
    jr.append(np.repeat(V0 + 2 * I, 2)); jc.append(np.stack([VR + I, VI + I], 1).ravel())
    jr.append(np.repeat(V0 + 2 * I + 1, 2)); jc.append(np.stack([VR + I, VI + I], 1).ravel())
    jr.append(np.repeat(T0 + 2 * L, 2)); jc.append(np.stack([PF + L, QF + L], 1).ravel())
    jr.append(np.repeat(T0 + 2 * L + 1, 2)); jc.append(np.stack([PT + L, QT + L], 1).ravel())
    own = [PF, QF, PT, QT]
    for k in range(4):
        jr.append(np.repeat(O0 + 4 * L + k, 5))
        jc.append(np.stack([own[k] + L, VI + f, VI + t, VR + f, VR + t], 1).ravel())
    sh_bus, sh_gs, sh_bs = net.shunts()
    if len(sh_bus):                            # gs (vr^2 + vi^2) in the P row, -bs (vr^2 + vi^2) in the Q row
        sb = sh_bus.astype(np.int64)
        jr.append(np.stack([1 + 2 * sb, 1 + 2 * sb, 2 + 2 * sb, 2 + 2 * sb], 1).ravel())
        jc.append(np.stack([VR + sb, VI + sb, VR + sb, VI + sb], 1).ravel())
    if ndc:
        D = np.arange(ndc, dtype=np.int64)
        jr.append(np.repeat(D0 + D, 2)); jc.append(np.stack([DC + D, DC + ndc + D], 1).ravel())
    jrow = np.concatenate(jr).astype(np.int64) + 1
    jcol = np.concatenate(jc).astype(np.int64) + 1

    # Hessian COO, lower triangle (vr sits behind vi, so a (vr, vi) pair is (row, col)): every entry is a constant
    # times a multiplier -- the rows are quadratic
    hr, hc = [], []
    G = np.arange(ng, dtype=np.int64)
    hr.append(PG + G); hc.append(PG + G)
    for base_p, base_q in ((PF, QF), (PT, QT)):
        hr.append(np.stack([base_p + L, base_q + L], 1).ravel()); hc.append(np.stack([base_p + L, base_q + L], 1).ravel())
    hr.append(np.stack([VR + I, VI + I, VR + I, VI + I], 1).ravel()); hc.append(np.stack([VR + I, VI + I, VR + I, VI + I], 1).ravel())
    for k in range(4):
        s_ = f if k < 2 else t
        ents = [(VI + s_, VI + s_), (VR + s_, VR + s_),
                (np.maximum(VI + f, VI + t), np.minimum(VI + f, VI + t)),
                (np.maximum(VR + f, VR + t), np.minimum(VR + f, VR + t)),
                (VR + t, VI + f), (VR + f, VI + t)]
        for a, b in ents:                      # entry-major inside the block of row kind k, like the polar layout
            hr.append(a); hc.append(b)
    if len(sh_bus):
        sb = sh_bus.astype(np.int64)
        hr.append(np.stack([VR + sb, VI + sb], 1).ravel()); hc.append(np.stack([VR + sb, VI + sb], 1).ravel())
    hrow = np.concatenate(hr).astype(np.int64) + 1
    hcol = np.concatenate(hc).astype(np.int64) + 1

    inf = np.inf
    xL = np.concatenate([-net.vmax, -net.vmax, net.pmin, net.qmin, -net.rate_a, -net.rate_a, -net.rate_a, -net.rate_a])
    xU = np.concatenate([net.vmax, net.vmax, net.pmax, net.qmax, net.rate_a, net.rate_a, net.rate_a, net.rate_a])
    if ndc:
        dc = net.dcline
        a = dc["loss0"] - (1.0 - dc["loss1"]) * dc["pmaxf"]; b = dc["loss0"] - (1.0 - dc["loss1"]) * dc["pminf"]
        xL = np.concatenate([xL, dc["pminf"], np.minimum(a, b), dc["qminf"], dc["qmint"]])
        xU = np.concatenate([xU, dc["pmaxf"], np.maximum(a, b), dc["qmaxf"], dc["qmaxt"]])
    gL = np.empty(m); gU = np.empty(m)
    gL[0] = gU[0] = 0.0
    gL[1:V0:2] = -net.pd; gU[1:V0:2] = -net.pd
    gL[2:V0:2] = -net.qd; gU[2:V0:2] = -net.qd
    gL[V0:T0:2] = net.vmin ** 2; gU[V0:T0:2] = inf
    gL[V0 + 1:T0:2] = -inf; gU[V0 + 1:T0:2] = net.vmax ** 2
    gL[T0:O0] = -inf
    gU[T0:O0:2] = net.rate_a ** 2; gU[T0 + 1:O0:2] = net.rate_a ** 2
    gL[O0:D0] = 0.0; gU[O0:D0] = 0.0
    if ndc:
        gL[D0:] = net.dcline["loss0"]; gU[D0:] = net.dcline["loss0"]
    boxed = np.isfinite(xL) & np.isfinite(xU)
    x0 = np.zeros(n)
    x0[boxed] = 0.5 * (xL[boxed] + xU[boxed])
    x0[VI:VI + nb] = 0.0
    x0[VR:VR + nb] = 1.0
    return NlpLayout(n=n, m=m, num_linear=V0 if len(sh_bus) == 0 else 1,
                     jrow=jrow, jcol=jcol, hrow=hrow, hcol=hcol, xL=xL, xU=xU, gL=gL, gU=gU, x0=x0,
                     bal_ptr=bal_ptr, bal_colP=colP.astype(np.int32), bal_colQ=colQ.astype(np.int32),
                     bal_coef=np.asarray(coef), sh_bus=sh_bus, sh_gs=sh_gs, sh_bs=sh_bs,
                     dc_loss1=(np.zeros(0) if ndc == 0 else np.asarray(net.dcline["loss1"], dtype=np.float64)),
                     form="acr")


def acwr_layout(net: Network) -> NlpLayout:
    """The W-space model of /root/reference/examples/acopf/acwr.jl:1-37 (`ACWRPowerModel <: AbstractWRModel`, built by
    `build_acwr` with PowerModels' own `build_opf`): the lifted variables w_i = |v_i|^2, wr_ij, wi_ij of the W-R
    formulation carry every constraint linearly, and `constraint_model_voltage` ties them to rectangular voltages:

    variables  vi[nb], vr[nb] (free; start vr = 1), w[nb] in [vmin^2, vmax^2] (start 1.001), wr[nbp] (start 1), wi[nbp]
               per bus pair (i < j) within PowerModels' voltage-product bounds, pg, qg, p_f, p_t, q_f, q_t, dc lines
    rows       0                 vi[ref] = 0   (WR models have no reference constraint; kept so that the rotation of
                                 (vr, vi) -- which no other row sees -- does not leave the Newton matrix singular)
               1 .. 2 nb         power balance with the shunt terms gs w_i, -bs w_i            (linear)
               A0 + 2 k, +1      wi_k - tan(angmax_k) wr_k <= 0,  wi_k - tan(angmin_k) wr_k >= 0   (linear)
               O0 + 4 l + c      flow_c - (A w_self + Bc wr + sigma Bs wi) = 0                     (linear)
               V0 + i            w_i - vr_i^2 - vi_i^2 = 0
               V0 + nb + 2 k, +1 wr_k - (vr_i vr_j + vi_i vi_j) = 0,  wi_k - (vi_i vr_j - vr_i vi_j) = 0
               T0 + 2 l, +1      thermal limits
               D0 + d            dc-line loss rows
    sigma = +1 when the branch runs from i to j of its pair, -1 otherwise (wi changes sign with the orientation)."""
    nb, ng, nl, ndc = net.nb, net.ng, net.nl, net.ndc
    f, t = net.f_bus.astype(np.int64), net.t_bus.astype(np.int64)
    lo, hi = np.minimum(f, t), np.maximum(f, t)
    key = lo * nb + hi
    uniq, br_bp = np.unique(key, return_inverse=True)
    nbp = len(uniq)
    bp_i, bp_j = uniq // nb, uniq % nb
    sig = np.where(f == lo, 1.0, -1.0)
    # angle limits of a pair in its own orientation: the tightest over its branches
    amin = np.full(nbp, -np.inf); amax = np.full(nbp, np.inf)
    for l in range(nl):
        a, b = (net.angmin[l], net.angmax[l]) if sig[l] > 0 else (-net.angmax[l], -net.angmin[l])
        amin[br_bp[l]] = max(amin[br_bp[l]], a); amax[br_bp[l]] = min(amax[br_bp[l]], b)
    VI, VR, W, WR, WI = 0, nb, 2 * nb, 3 * nb, 3 * nb + nbp
    PG = 3 * nb + 2 * nbp
    QG, PF = PG + ng, PG + 2 * ng
    PT, QF, QT = PF + nl, PF + 2 * nl, PF + 3 * nl
    DC = PF + 4 * nl
    n = DC + 4 * ndc
    A0 = 1 + 2 * nb
    O0 = A0 + 2 * nbp
    V0 = O0 + 4 * nl
    T0 = V0 + nb + 2 * nbp
    D0 = T0 + 2 * nl
    m = D0 + ndc
    L = np.arange(nl, dtype=np.int64); I = np.arange(nb, dtype=np.int64); K = np.arange(nbp, dtype=np.int64)

    jr, jc = [np.array([0])], [np.array([VI + net.ref_bus])]
    inc = [[] for _ in range(nb)]
    for l in range(nl):
        inc[int(f[l])].append((PF + l, QF + l, 1.0))
        inc[int(t[l])].append((PT + l, QT + l, 1.0))
    for g in range(ng):
        inc[int(net.gen_bus[g])].append((PG + g, QG + g, -1.0))
    for d in range(ndc):
        inc[int(net.dcline["f_bus"][d])].append((DC + d, DC + 2 * ndc + d, 1.0))
        inc[int(net.dcline["t_bus"][d])].append((DC + ndc + d, DC + 3 * ndc + d, 1.0))
    bal_ptr = np.zeros(nb + 1, dtype=np.int32)
    colP, colQ, coef = [], [], []
    for i in range(nb):
        bal_ptr[i + 1] = bal_ptr[i] + len(inc[i])
        for cp, cq, cf in inc[i]:
            colP.append(cp); colQ.append(cq); coef.append(cf)
    colP = np.asarray(colP, dtype=np.int64); colQ = np.asarray(colQ, dtype=np.int64)
    for i in range(nb):
        s_, e = bal_ptr[i], bal_ptr[i + 1]
        jr.append(np.full(e - s_, 1 + 2 * i)); jc.append(colP[s_:e])
        jr.append(np.full(e - s_, 2 + 2 * i)); jc.append(colQ[s_:e])
    # shunt terms of the balance rows: (P row, w_i), (Q row, w_i) for every bus (coefficient 0 without a shunt)
    jr.append(np.stack([1 + 2 * I, 2 + 2 * I], 1).ravel()); jc.append(np.repeat(W + I, 2))
    # angle rows: (wi, wr) each
    jr.append(np.repeat(A0 + 2 * K, 2)); jc.append(np.stack([WI + K, WR + K], 1).ravel())
    jr.append(np.repeat(A0 + 2 * K + 1, 2)); jc.append(np.stack([WI + K, WR + K], 1).ravel())
    # Ohm rows, kind-major like the other layouts: (own, w_self, wr, wi)
    own = [PF, QF, PT, QT]
    for c in range(4):
        ws = W + (f if c < 2 else t)
        jr.append(np.repeat(O0 + 4 * L + c, 4)); jc.append(np.stack([own[c] + L, ws, WR + br_bp, WI + br_bp], 1).ravel())
    # model-voltage rows
    jr.append(np.repeat(V0 + I, 3)); jc.append(np.stack([W + I, VR + I, VI + I], 1).ravel())
    jr.append(np.repeat(V0 + nb + 2 * K, 5)); jc.append(np.stack([WR + K, VR + bp_i, VR + bp_j, VI + bp_i, VI + bp_j], 1).ravel())
    jr.append(np.repeat(V0 + nb + 2 * K + 1, 5)); jc.append(np.stack([WI + K, VI + bp_i, VR + bp_j, VR + bp_i, VI + bp_j], 1).ravel())
    jr.append(np.repeat(T0 + 2 * L, 2)); jc.append(np.stack([PF + L, QF + L], 1).ravel())
    jr.append(np.repeat(T0 + 2 * L + 1, 2)); jc.append(np.stack([PT + L, QT + L], 1).ravel())
    if ndc:
        D = np.arange(ndc, dtype=np.int64)
        jr.append(np.repeat(D0 + D, 2)); jc.append(np.stack([DC + D, DC + ndc + D], 1).ravel())
    jrow = np.concatenate(jr).astype(np.int64) + 1
    jcol = np.concatenate(jc).astype(np.int64) + 1

    hr, hc = [], []
    G = np.arange(ng, dtype=np.int64)
    hr.append(PG + G); hc.append(PG + G)
    for base_p, base_q in ((PF, QF), (PT, QT)):
        hr.append(np.stack([base_p + L, base_q + L], 1).ravel()); hc.append(np.stack([base_p + L, base_q + L], 1).ravel())
    hr.append(np.stack([VR + I, VI + I], 1).ravel()); hc.append(np.stack([VR + I, VI + I], 1).ravel())          # w rows
    hr.append(np.stack([VR + bp_j, VI + bp_j], 1).ravel()); hc.append(np.stack([VR + bp_i, VI + bp_i], 1).ravel())  # wr rows (j > i)
    hr.append(np.stack([VR + bp_j, VR + bp_i], 1).ravel()); hc.append(np.stack([VI + bp_i, VI + bp_j], 1).ravel())  # wi rows
    hrow = np.concatenate(hr).astype(np.int64) + 1
    hcol = np.concatenate(hc).astype(np.int64) + 1
    assert (hrow >= hcol).all()

    inf = np.inf
    vmin, vmax = net.vmin, net.vmax
    lo_ij, hi_ij = vmin[bp_i] * vmin[bp_j], vmax[bp_i] * vmax[bp_j]
    cmin = np.minimum(np.cos(amin), np.cos(amax))
    wr_min = np.where(amin >= 0, lo_ij * np.cos(amax), np.where(amax <= 0, lo_ij * np.cos(amin), lo_ij * cmin))
    wr_max = np.where(amin >= 0, hi_ij * np.cos(amin), np.where(amax <= 0, hi_ij * np.cos(amax), hi_ij))
    wi_min = np.where(amin >= 0, lo_ij * np.sin(amin), hi_ij * np.sin(amin))
    wi_max = np.where(amax <= 0, lo_ij * np.sin(amax), hi_ij * np.sin(amax))
    xL = np.concatenate([np.full(2 * nb, -inf), vmin ** 2, wr_min, wi_min, net.pmin, net.qmin,
                         -net.rate_a, -net.rate_a, -net.rate_a, -net.rate_a])
    xU = np.concatenate([np.full(2 * nb, inf), vmax ** 2, wr_max, wi_max, net.pmax, net.qmax,
                         net.rate_a, net.rate_a, net.rate_a, net.rate_a])
    if ndc:
        dc = net.dcline
        a = dc["loss0"] - (1.0 - dc["loss1"]) * dc["pmaxf"]; b = dc["loss0"] - (1.0 - dc["loss1"]) * dc["pminf"]
        xL = np.concatenate([xL, dc["pminf"], np.minimum(a, b), dc["qminf"], dc["qmint"]])
        xU = np.concatenate([xU, dc["pmaxf"], np.maximum(a, b), dc["qmaxf"], dc["qmaxt"]])
    gL = np.empty(m); gU = np.empty(m)
    gL[0] = gU[0] = 0.0
    gL[1:A0:2] = -net.pd; gU[1:A0:2] = -net.pd
    gL[2:A0:2] = -net.qd; gU[2:A0:2] = -net.qd
    gL[A0:O0:2] = -inf; gU[A0:O0:2] = 0.0
    gL[A0 + 1:O0:2] = 0.0; gU[A0 + 1:O0:2] = inf
    gL[O0:T0] = 0.0; gU[O0:T0] = 0.0
    gL[T0:D0] = -inf
    gU[T0:D0:2] = net.rate_a ** 2; gU[T0 + 1:D0:2] = net.rate_a ** 2
    if ndc:
        gL[D0:] = net.dcline["loss0"]; gU[D0:] = net.dcline["loss0"]
    boxed = np.isfinite(xL) & np.isfinite(xU)
    x0 = np.zeros(n)
    x0[boxed] = 0.5 * (xL[boxed] + xU[boxed])
    x0[VI:VI + nb] = 0.0; x0[VR:VR + nb] = 1.0; x0[W:W + nb] = 1.001; x0[WR:WR + nbp] = 1.0; x0[WI:WI + nbp] = 0.0
    sh_bus, sh_gs, sh_bs = net.shunts()
    return NlpLayout(n=n, m=m, num_linear=V0, jrow=jrow, jcol=jcol, hrow=hrow, hcol=hcol, xL=xL, xU=xU, gL=gL, gU=gU, x0=x0,
                     bal_ptr=bal_ptr, bal_colP=colP.astype(np.int32), bal_colQ=colQ.astype(np.int32),
                     bal_coef=np.asarray(coef), sh_bus=sh_bus, sh_gs=sh_gs, sh_bs=sh_bs,
                     dc_loss1=(np.zeros(0) if ndc == 0 else np.asarray(net.dcline["loss1"], dtype=np.float64)),
                     form="acwr", bp_i=bp_i.astype(np.int32), bp_j=bp_j.astype(np.int32), br_bp=br_bp.astype(np.int32),
                     br_sig=sig, bp_tmin=np.tan(amin), bp_tmax=np.tan(amax))
