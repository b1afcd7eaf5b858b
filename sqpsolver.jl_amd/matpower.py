"""MATPOWER case reader (the reference's on-disk input format).

The reference loads ACOPF data through PowerModels from MATPOWER `.m` files
(/root/reference/examples/acopf/opf.jl:12-16 `PowerModels.parse_file`; the shipped
example file is /root/reference/examples/acopf/case3.m:7-36 with `mpc.baseMVA`,
`mpc.bus`, `mpc.gen`, `mpc.gencost`, `mpc.branch`, `mpc.dcline`).  This module reads
that format (MATPOWER case format version 2) and converts it into the `Network` the
device evaluator and the oracle consume (`acopf_synth.Network`), in per-unit.

Supported subset = what the evaluator models (polar ACOPF, SURVEY.md App. B, with off-nominal
transformer taps and phase shifters at the from end and bus shunts Gs/Bs as MATPOWER defines them):
polynomial (model 2) generator costs of degree <= 2.  Anything else raises `UnsupportedCase` naming the offending rows, never a silent
approximation.  HVDC lines (`mpc.dcline`, modelled by the reference's custom build at
examples/acopf/opf.jl:16,40-42: `variable_dcline_power` + `constraint_dcline_power_losses`) become four variables
and one loss row each (`acopf_layout`); `dcline="drop"` ignores them.  A `mpc.dclinecost` table is rejected (the
shipped example has none).  [UNVERIFIED against PowerModels, which is not in the image: the sign convention of the
reactive limits QminF..QmaxT -- the shipped file's limits are symmetric, so it does not matter there.]

Column meaning (MATPOWER manual, Appendix B):
    bus    : bus_i type Pd Qd Gs Bs area Vm Va baseKV zone Vmax Vmin
    gen    : bus Pg Qg Qmax Qmin Vg mBase status Pmax Pmin ...
    gencost: model startup shutdown n c(n-1) ... c0
    branch : fbus tbus r x b rateA rateB rateC ratio angle status angmin angmax
"""
from __future__ import annotations

import re
from typing import Dict, Union

import numpy as np

from .acopf_synth import Network

__all__ = ["read_matpower", "network_from_matpower", "load_case", "write_matpower", "UnsupportedCase"]


class UnsupportedCase(ValueError):
    """The case uses a feature outside the modelled subset."""


_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|Inf|inf|NaN|nan)"


def _strip_comments(text: str) -> str:
    out = []
    for line in text.splitlines():
        # '%' starts a comment unless inside a quoted string
        buf, inq = [], False
        for ch in line:
            if ch == "'":
                inq = not inq
            if ch == "%" and not inq:
                break
            buf.append(ch)
        out.append("".join(buf))
    return "\n".join(out)


def read_matpower(src: str) -> Dict[str, Union[float, str, np.ndarray]]:
    """Parse MATPOWER case text (or a path to a `.m` file) into {field: scalar | string | 2-D array}.

    Matrices `mpc.name = [ ... ];` become float64 arrays (rows split on ';' or newlines, ragged rows
    are an error); scalars `mpc.name = value;` become float, quoted values str.  Cell arrays
    (`mpc.bus_name = { ... }`) and struct-array extensions are skipped."""
    if "\n" not in src and not src.lstrip().startswith(("function", "%", "mpc")):
        with open(src, "r") as fh:
            src = fh.read()
    text = _strip_comments(src)
    out: Dict[str, Union[float, str, np.ndarray]] = {}
    pos = 0
    pat = re.compile(r"mpc\.([A-Za-z_]\w*)\s*=\s*")
    while True:
        mt = pat.search(text, pos)
        if not mt:
            break
        name, i = mt.group(1), mt.end()
        if i < len(text) and text[i] == "[":
            j = text.find("]", i)
            if j < 0:
                raise ValueError(f"mpc.{name}: unterminated '['")
            body = text[i + 1:j]
            rows = []
            for raw in re.split(r"[;\n]", body):
                toks = re.findall(_NUM, raw.replace(",", " "))
                if toks:
                    rows.append([float(t) for t in toks])
            if rows:
                w = len(rows[0])
                if any(len(r) != w for r in rows):
                    raise ValueError(f"mpc.{name}: ragged matrix rows {[len(r) for r in rows]}")
                out[name] = np.asarray(rows, dtype=np.float64)
            else:
                out[name] = np.zeros((0, 0))
            pos = j + 1
        elif i < len(text) and text[i] == "{":
            j = text.find("}", i)
            pos = (j if j >= 0 else i) + 1
        else:
            j = i
            while j < len(text) and text[j] not in ";\n":
                j += 1
            val = text[i:j].strip()
            if val.startswith("'") and val.endswith("'"):
                out[name] = val[1:-1]
            else:
                try:
                    out[name] = float(val)
                except ValueError:
                    out[name] = val
            pos = j + 1
    if "bus" not in out or "branch" not in out or "gen" not in out:
        raise ValueError("not a MATPOWER case: mpc.bus / mpc.gen / mpc.branch missing")
    return out


def network_from_matpower(mpc: Dict, dcline: str = "error", unlimited_rate: float = 1e4) -> Network:
    """Convert parsed MATPOWER data to the per-unit `Network` of the evaluators.

    * buses are renumbered 0..nb-1 in file order; the reference bus is the (single) type-3 bus, or, when the
      file marks none, the bus of the largest in-service generator;
    * out-of-service generators (status <= 0) are dropped; out-of-service branches keep their slot
      with `status = 0` (admittance zeroed, sparsity pattern kept, like a contingency);
    * costs are converted to per-unit variables: c2*base^2 pg^2 + c1*base pg (the constant is dropped:
      it does not change the minimiser; `Network` has no field for it);
    * rateA = 0 means unlimited in MATPOWER -> `unlimited_rate` p.u.; angmin/angmax 0/0 or beyond
      +-60 degrees are clamped to +-60 degrees as PowerModels does."""
    base = float(mpc.get("baseMVA", 100.0))
    bus, gen, br = mpc["bus"], mpc["gen"], mpc["branch"]
    if bus.shape[1] < 13 or gen.shape[1] < 10 or br.shape[1] < 11:
        raise ValueError("MATPOWER v2 needs >= 13 bus, 10 gen and 11 branch columns")
    problems = []
    ids = bus[:, 0].astype(np.int64)
    if len(set(ids.tolist())) != len(ids):
        raise ValueError("duplicate bus numbers")
    idx = {int(b): k for k, b in enumerate(ids)}
    nb = len(ids)
    if np.any(bus[:, 1] == 4):
        problems.append(f"isolated buses (type 4) {ids[bus[:, 1] == 4].tolist()}")
    on = gen[:, 7] > 0
    g = gen[on]
    ref = np.flatnonzero(bus[:, 1] == 3)
    if len(ref) == 0 and len(g):
        # no type-3 bus (the reference's case3.m "tests reference bus detection"): PowerModels then takes the bus
        # of the in-service generator with the largest Pmax (correct_reference_buses!); first one on ties
        ref = np.asarray([idx[int(g[int(np.argmax(g[:, 8])), 0])]]) if int(g[int(np.argmax(g[:, 8])), 0]) in idx else ref
    if len(ref) != 1:
        problems.append(f"{len(ref)} reference buses (type 3); exactly one is modelled")
    if "gencost" in mpc and mpc["gencost"].size:
        gc = mpc["gencost"][: gen.shape[0]][on]      # rows beyond ng are reactive-power costs
        if np.any(gc[:, 0] != 2):
            problems.append("piecewise-linear generator costs (model 1)")
        ncoef = gc[:, 3].astype(int)
        if np.any(ncoef > 3):
            problems.append("generator cost polynomials of degree > 2")
        c2 = np.zeros(len(g)); c1 = np.zeros(len(g))
        for k in range(len(g)):
            co = gc[k, 4:4 + ncoef[k]]
            if ncoef[k] == 3:
                c2[k], c1[k] = co[0], co[1]
            elif ncoef[k] == 2:
                c1[k] = co[0]
    else:
        c2 = np.zeros(len(g)); c1 = np.zeros(len(g))
    ratio, shift = br[:, 8], br[:, 9]                # ratio 0 means "no transformer" = 1; shift in degrees
    if np.any(ratio < 0.0):
        problems.append(f"negative tap ratios on branches {np.flatnonzero(ratio < 0).tolist()}")
    dc = None
    if "dcline" in mpc and mpc["dcline"].size and dcline != "drop":
        if dcline not in ("error", "model"):
            raise ValueError("dcline must be 'model' (default), 'error' (same) or 'drop'")
        t = np.atleast_2d(mpc["dcline"])
        t = t[t[:, 2] > 0]                                   # in-service lines only
        if "dclinecost" in mpc and np.size(mpc["dclinecost"]):
            problems.append("HVDC line costs (mpc.dclinecost)")
        if t.shape[1] < 17:
            problems.append("mpc.dcline with fewer than 17 columns")
        else:
            missing = [int(b) for b in np.concatenate([t[:, 0], t[:, 1]]) if int(b) not in idx]
            if missing:
                raise ValueError(f"dcline rows reference unknown buses {sorted(set(missing))}")
            if np.any(t[:, 9] > t[:, 10]):
                problems.append("HVDC lines with Pmin > Pmax")
            dc = dict(f_bus=np.asarray([idx[int(b)] for b in t[:, 0]], dtype=np.int32),
                      t_bus=np.asarray([idx[int(b)] for b in t[:, 1]], dtype=np.int32),
                      pminf=t[:, 9] / base, pmaxf=t[:, 10] / base,
                      qminf=t[:, 11] / base, qmaxf=t[:, 12] / base, qmint=t[:, 13] / base, qmaxt=t[:, 14] / base,
                      loss0=t[:, 15] / base, loss1=t[:, 16].copy())
    for col, nm in ((0, "gen"),):
        missing = [int(b) for b in g[:, col] if int(b) not in idx]
        if missing:
            raise ValueError(f"{nm} rows reference unknown buses {missing}")
    missing = [int(b) for b in np.concatenate([br[:, 0], br[:, 1]]) if int(b) not in idx]
    if missing:
        raise ValueError(f"branch rows reference unknown buses {sorted(set(missing))}")
    if problems:
        raise UnsupportedCase("; ".join(problems))

    nl = br.shape[0]
    rate = br[:, 5] / base
    rate = np.where(rate <= 0.0, unlimited_rate, rate)
    if br.shape[1] >= 13:
        amin, amax = np.deg2rad(br[:, 11]), np.deg2rad(br[:, 12])
    else:
        amin, amax = np.full(nl, -np.pi / 3), np.full(nl, np.pi / 3)
    both0 = (amin == 0.0) & (amax == 0.0)
    amin = np.where(both0 | (amin < -np.pi / 3), -np.pi / 3, amin)
    amax = np.where(both0 | (amax > np.pi / 3), np.pi / 3, amax)
    return Network(
        nb=nb, ng=len(g), nl=nl,
        pd=bus[:, 2] / base, qd=bus[:, 3] / base,
        vmin=bus[:, 12].copy(), vmax=bus[:, 11].copy(), ref_bus=int(ref[0]),
        gen_bus=np.asarray([idx[int(b)] for b in g[:, 0]], dtype=np.int32),
        pmin=g[:, 9] / base, pmax=g[:, 8] / base, qmin=g[:, 4] / base, qmax=g[:, 3] / base,
        c2=c2 * base * base, c1=c1 * base,
        f_bus=np.asarray([idx[int(b)] for b in br[:, 0]], dtype=np.int32),
        t_bus=np.asarray([idx[int(b)] for b in br[:, 1]], dtype=np.int32),
        r=br[:, 2].copy(), x=br[:, 3].copy(), bc=br[:, 4].copy(), rate_a=rate,
        angmin=amin, angmax=amax, status=(br[:, 10] > 0).astype(np.float64),
        tap=np.where(ratio == 0.0, 1.0, ratio), shift=np.deg2rad(shift),
        gs=bus[:, 4] / base, bs=bus[:, 5] / base,        # MW / MVAr at vm = 1 -> per unit
        dcline=dc,
    )


def load_case(path_or_text: str, **kw) -> Network:
    """`read_matpower` + `network_from_matpower`."""
    return network_from_matpower(read_matpower(path_or_text), **kw)


def write_matpower(net: Network, name: str = "case_synth", base_mva: float = 100.0) -> str:
    """MATPOWER v2 text of a `Network` (inverse of `network_from_matpower` up to float formatting):
    lets the synthetic IEEE-shaped cases be handed to the Julia side (`PowerModels.parse_file`)."""
    f = lambda v: repr(float(v))
    gen_at = set(int(b) for b in net.gen_bus)
    L = [f"function mpc = {name}", "mpc.version = '2';", f"mpc.baseMVA = {f(base_mva)};", "mpc.bus = ["]
    for i in range(net.nb):
        typ = 3 if i == net.ref_bus else (2 if i in gen_at else 1)
        gsi = 0.0 if net.gs is None else net.gs[i] * base_mva
        bsi = 0.0 if net.bs is None else net.bs[i] * base_mva
        L.append("\t" + "\t".join([str(i + 1), str(typ), f(net.pd[i] * base_mva), f(net.qd[i] * base_mva), f(gsi), f(bsi), "1",
                                   "1.0", "0.0", "230.0", "1", f(net.vmax[i]), f(net.vmin[i])]) + ";")
    L += ["];", "mpc.gen = ["]
    for k in range(net.ng):
        L.append("\t" + "\t".join([str(int(net.gen_bus[k]) + 1), "0.0", "0.0", f(net.qmax[k] * base_mva), f(net.qmin[k] * base_mva),
                                   "1.0", f(base_mva), "1", f(net.pmax[k] * base_mva), f(net.pmin[k] * base_mva)]) + ";")
    L += ["];", "mpc.gencost = ["]
    for k in range(net.ng):
        L.append("\t" + "\t".join(["2", "0.0", "0.0", "3", f(net.c2[k] / base_mva ** 2), f(net.c1[k] / base_mva), "0.0"]) + ";")
    L += ["];", "mpc.branch = ["]
    for l in range(net.nl):
        L.append("\t" + "\t".join([str(int(net.f_bus[l]) + 1), str(int(net.t_bus[l]) + 1), f(net.r[l]), f(net.x[l]), f(net.bc[l]),
                                   f(net.rate_a[l] * base_mva), "0.0", "0.0",
                                   f(0.0 if net.tap is None or net.tap[l] == 1.0 else net.tap[l]),
                                   f(0.0 if net.shift is None else np.rad2deg(net.shift[l])), str(int(net.status[l] > 0)),
                                   f(np.rad2deg(net.angmin[l])), f(np.rad2deg(net.angmax[l]))]) + ";")
    L += ["];"]
    if net.ndc:
        dc = net.dcline
        L += ["mpc.dcline = ["]
        for d in range(net.ndc):
            pf = dc["pminf"][d] * base_mva
            L.append("\t" + "\t".join([str(int(dc["f_bus"][d]) + 1), str(int(dc["t_bus"][d]) + 1), "1", f(pf), f(pf), "0.0", "0.0",
                                       "1.0", "1.0", f(dc["pminf"][d] * base_mva), f(dc["pmaxf"][d] * base_mva),
                                       f(dc["qminf"][d] * base_mva), f(dc["qmaxf"][d] * base_mva),
                                       f(dc["qmint"][d] * base_mva), f(dc["qmaxt"][d] * base_mva),
                                       f(dc["loss0"][d] * base_mva), f(dc["loss1"][d])]) + ";")
        L += ["];"]
    L += [""]
    return "\n".join(L)
