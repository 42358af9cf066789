#!/usr/bin/env python3
"""Generates zopt_amd/csrc/quad_derivs_gen.h: closed-form first and second derivatives of the quadcopter's continuous dynamics
xd = inertialDynamics(x, u) as models.h states them (reference zopt/quadcopter.py:23-144, including its rotation-matrix quirk), by
symbolic differentiation (sympy) of that same expression tree:

    quad_jac_column<WIND>(j, a, o)   o[i] = d xd_i / d z_j                 z = [x (12) ; u (4)], column j = 0..15
    quad_hess_pair<WIND>(p, a, o)    o[i] = d2 xd_i / d z_a d z_b          for the model's declared pairs p = (a <= b)  (models.h, 28 pairs)

The kernels used to get these from the model evaluated on dual / hyper-dual numbers (jax.jacobian / jax.hessian in the reference,
pytrees.py:139-153, 180-186): one full model evaluation per column resp. pair.  The closed forms share the trigonometric values and
need a handful of operations per entry.  The generator also PROVES (symbolically) that every second derivative outside the declared
pairs is identically zero, wind included.

Run:  python3 tools/gen_quad_derivs.py            (rewrites the header; tests/test_quad_derivs.py checks it against the oracle on CPU)
"""
import os
import sys

import sympy as sp
from sympy.printing.c import C99CodePrinter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [0x00, 0x11, 0x22, 0x24, 0x15, 0x05, 0x23, 0x13, 0x04, 0x66, 0x67, 0x77, 0x68, 0x78, 0x88, 0x06, 0x07, 0x08, 0x16, 0x17, 0x18,
         0x26, 0x27, 0x28, 0x46, 0x47, 0x56, 0x57]      # models.h model_pair_table(ZM_MODEL_QUADCOPTER): a * 16 + b


def model():
    x = sp.symbols("x0:12", real=True)
    u = sp.symbols("u0:4", real=True)
    w = sp.symbols("w0:3", real=True)
    g, mass = sp.Float("9.807"), sp.Float("2.5")
    s6, c6, s7, c7, s8, c8 = sp.sin(x[6]), sp.cos(x[6]), sp.sin(x[7]), sp.cos(x[7]), sp.sin(x[8]), sp.cos(x[8])
    t7 = s7 / c7
    r00, r01, r02 = c7 * c8, s6 * s7 * c8 - c6 * s8, c6 * s7 * c8 - s6 * s8      # [0][2] as the reference writes it (quirk Q4)
    r10, r11, r12 = c7 * s8, s6 * s7 * s8 + c6 * c8, c6 * s7 * s8 - s6 * c8
    r20, r21, r22 = -s7, s6 * c7, c6 * c7
    wb = [r00 * w[0] + r10 * w[1] + r20 * w[2], r01 * w[0] + r11 * w[1] + r21 * w[2], r02 * w[0] + r12 * w[1] + r22 * w[2]]
    va = [x[i] - wb[i] for i in range(3)]
    half = sp.Rational(1, 2)
    fa = [-sp.Rational(1, 5) * va[0] - sp.Rational(1, 20) * va[0] ** 2, -sp.Rational(1, 5) * va[1] - sp.Rational(1, 20) * va[1] ** 2,
          -sp.Rational(3, 10) * va[2] - sp.Rational(1, 10) * va[2] ** 2]
    ft = [fa[0] + mass * g * (-s7), fa[1] + mass * g * (s6 * c7), mass * (-u[0]) + fa[2] + mass * g * (c6 * c7)]
    cr = [x[4] * x[2] - x[5] * x[1], x[5] * x[0] - x[3] * x[2], x[3] * x[1] - x[4] * x[0]]
    xd = [(ft[i] - cr[i]) / mass for i in range(3)]
    xd += [u[1] - sp.Rational(1, 10) * x[3], u[2] - sp.Rational(1, 10) * x[4], u[3] - sp.Rational(1, 20) * x[5]]
    xd += [x[3] + s6 * t7 * x[4] + c6 * t7 * x[5], c6 * x[4] - s6 * x[5], (s6 / c7) * x[4] + (c6 / c7) * x[5]]
    xd += [r00 * x[0] + r01 * x[1] + r02 * x[2], r10 * x[0] + r11 * x[1] + r12 * x[2], r20 * x[0] + r21 * x[1] + r22 * x[2]]
    del half
    return x, u, w, xd


ATOMS = {}


def to_atoms(e, x):
    """sin / cos of the angles -> the symbols the kernel passes in; negative powers of cos(theta) -> powers of ic7 = 1 / cos(theta)"""
    S = {n: sp.Symbol(n, real=True) for n in ("s6", "c6", "s7", "c7", "s8", "c8", "ic7")}
    ATOMS.update(S)
    e = e.subs({sp.sin(x[6]): S["s6"], sp.cos(x[6]): S["c6"], sp.sin(x[7]): S["s7"], sp.cos(x[7]): S["c7"], sp.sin(x[8]): S["s8"],
                sp.cos(x[8]): S["c8"]})
    e = e.replace(lambda t: t.is_Pow and t.base == S["c7"] and t.exp.is_negative, lambda t: S["ic7"] ** (-t.exp))
    return e


class Printer(C99CodePrinter):
    def _print_Pow(self, e):
        if e.exp.is_Integer and 2 <= int(e.exp) <= 4:
            b = self._factor(e.base)
            return "(" + " * ".join([b] * int(e.exp)) + ")"
        return super()._print_Pow(e)

    # Sums and products are spelled out operation by operation -- products left to right in sympy's canonical factor order, sums as a
    # chain of explicit fused multiply-adds -- and the generated functions switch the compiler's own contraction off (ZM_FP_STRICT):
    # the per-column / per-pair forms (behind a switch) and the straight-line forms then run the SAME sequence of roundings, whatever
    # code surrounds them, and agree bit for bit.
    def _factor(self, f):
        t = self._print(f)
        return t if (f.is_Atom or t.startswith("(") or t.startswith("__builtin_fma(")) else "(" + t + ")"

    def _product(self, factors):
        return "(" + " * ".join(self._factor(f) for f in factors) + ")" if len(factors) > 1 else self._factor(factors[0])

    def _print_Mul(self, e):
        c, rest = e.as_coeff_Mul()
        fs = list(rest.as_ordered_factors())
        if any(f.is_Pow and f.exp.is_negative for f in fs):
            raise ValueError(f"division left in {e}")
        if c == 1:
            return self._product(fs)
        if c == -1:
            return "(-" + self._product(fs) + ")"
        return "(" + repr(float(c)) + " * " + self._product(fs) + ")"

    def _print_Add(self, e):
        terms = e.as_ordered_terms()
        simple = [t for t in terms if not (t.is_Mul and len(t.as_coeff_Mul()[1].as_ordered_factors()) + (abs(t.as_coeff_Mul()[0]) != 1) >= 2)]
        prods = [t for t in terms if t not in simple]
        acc = None
        for t in simple:
            c, rest = t.as_coeff_Mul() if t.is_Mul else (sp.Integer(1), t)
            if t.is_Mul and c == -1:
                acc = ("(-" + self._factor(rest) + ")") if acc is None else "(" + acc + " - " + self._factor(rest) + ")"
            else:
                acc = self._factor(t) if acc is None else "(" + acc + " + " + self._factor(t) + ")"
        for t in prods:
            c, rest = t.as_coeff_Mul()
            fs = list(rest.as_ordered_factors())
            if abs(c) != 1:
                a, b = repr(float(c)), self._product(fs)
            else:
                a, b = ("-" if c == -1 else "") + self._factor(fs[0]), self._product(fs[1:])
            acc = "(" + a + " * " + b + ")" if acc is None else "__builtin_fma(" + a + ", " + b + ", " + acc + ")"
        return acc

    def _print_Symbol(self, s):
        n = s.name
        if n[0] == "x" and n[1:].isdigit():
            return f"a.x[{n[1:]}]"
        if n[0] == "w" and n[1:].isdigit():
            return f"a.w[{n[1:]}]"
        if n in ("u0", "s6", "c6", "s7", "c7", "s8", "c8", "ic7"):
            return "a." + n
        return n

    def _print_Float(self, f):
        return repr(float(f))

    def _print_Rational(self, r):
        return repr(float(r))


def emit_body(fh, table, pr, ind, packed=None, delta=True):
    """table: {case -> [(i, expr)]}.  One CSE over everything; a temporary that serves several cases is computed by every lane before
    the switch, one that serves a single case inside it.  packed = {(i, case) -> position}: instead of o[i] = expr the case writes
    t[position] = dt * expr + (i == case) -- the packed image of column `case` of [f_x | f_u] = I + dt d xd / d z."""
    keys = [(c, i) for c in sorted(table) for i, _ in table[c]]
    exprs = [e for c in sorted(table) for _, e in table[c]]
    temps, outs = sp.cse(exprs, symbols=sp.numbered_symbols("t"), optimizations="basic")
    td = dict(temps)

    def closure(e, acc):
        for sym in e.free_symbols:
            if sym in td and sym not in acc:
                acc.add(sym)
                closure(td[sym], acc)
        return acc

    use = {t: set() for t in td}
    for (c, _), o in zip(keys, outs):
        for t in closure(o, set()):
            use[t].add(c)
    for t, e in temps:
        if len(use[t]) > 1:
            fh.write(f"{ind}const double {t} = {pr.doprint(e)};\n")
    fh.write(f"{ind}switch (j) {{\n")
    for c in sorted(table):
        fh.write(f"{ind}    case {c}: {{ ZM_CASE_FENCE\n")
        for t, e in temps:
            if use[t] == {c}:
                fh.write(f"{ind}        const double {t} = {pr.doprint(e)};\n")
        for (cc, i), o in zip(keys, outs):
            if cc == c and packed is None:
                fh.write(f"{ind}        o[{i}] = {pr.doprint(o)};\n")
            elif cc == c:
                if delta:
                    fh.write(f"{ind}        t[{packed[(i, c)]}] = (dt == 0.0) ? ({pr.doprint(o)}) : __builtin_fma(dt, {pr.doprint(o)}, {1.0 if i == c else 0.0});\n")
                else:   # (the kernels' own form: o = (dt == 0) ? h : dt * h)
                    fh.write(f"{ind}        t[{packed[(i, c)]}] = (dt == 0.0) ? ({pr.doprint(o)}) : dt * ({pr.doprint(o)});\n")
        fh.write(f"{ind}    }} break;\n")
    fh.write(f"{ind}    default: break;\n{ind}}}\n")
    shared_ops = sum(sp.count_ops(e) for t, e in temps if len(use[t]) > 1)
    case_ops = sum(sp.count_ops(e) for t, e in temps if len(use[t]) <= 1) + sum(sp.count_ops(o) for o in outs)
    return shared_ops, case_ops


def emit_all(fh, table, pr, ind, packed, delta):
    """Straight-line form of emit_body's packed output: the same CSE (same expression list, same order -> same temporaries), every
    temporary and every entry evaluated by the calling lane -- one lane per trajectory point instead of one per column / pair."""
    keys = [(c, i) for c in sorted(table) for i, _ in table[c]]
    exprs = [e for c in sorted(table) for _, e in table[c]]
    temps, outs = sp.cse(exprs, symbols=sp.numbered_symbols("t"), optimizations="basic")
    for t, e in temps:
        fh.write(f"{ind}const double {t} = {pr.doprint(e)};\n")
    for (c, i), o in zip(keys, outs):
        if delta:
            fh.write(f"{ind}t[{packed[(i, c)]}] = (dt == 0.0) ? ({pr.doprint(o)}) : __builtin_fma(dt, {pr.doprint(o)}, {1.0 if i == c else 0.0});\n")
        else:
            fh.write(f"{ind}t[{packed[(i, c)]}] = (dt == 0.0) ? ({pr.doprint(o)}) : dt * ({pr.doprint(o)});\n")
    return sum(sp.count_ops(e) for _, e in temps) + sum(sp.count_ops(o) for o in outs)


def derivatives(wind):
    """-> (jac {column -> [(i, expr)]}, hes {pair -> [(i, expr)]}) in terms of the kernel's atoms; wind False: still air (w = 0)"""
    x, u, w, xd = model()
    if not wind:
        xd = [e.subs({w[0]: 0, w[1]: 0, w[2]: 0}) for e in xd]
    z = list(x) + list(u)
    jac = {}
    for j in range(16):
        jac[j] = [(i, to_atoms(d, x)) for i in range(12) for d in [sp.diff(xd[i], z[j])] if d != 0]
    declared = {(p >> 4, p & 15) for p in PAIRS}
    hes = {}
    for a in range(16):
        for b in range(a, 16):
            col = [sp.diff(xd[i], z[a], z[b]) for i in range(12)]
            if (a, b) in declared:
                hes[PAIRS.index(a * 16 + b)] = [(i, to_atoms(c, x)) for i, c in enumerate(col) if sp.simplify(c) != 0]
            else:
                assert all(sp.simplify(c) == 0 for c in col), f"pair ({a},{b}) is not declared but has a nonzero second derivative"
    return jac, hes


def main():
    pr = Printer()
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "zopt_amd", "csrc", "quad_derivs_gen.h")
    d = {wind: derivatives(wind) for wind in (False, True)}
    nnz1 = sum(len(v) for v in d[True][0].values())
    nnz2 = sum(len(v) for v in d[True][1].values())
    with open(out, "w") as fh:
        fh.write("// GENERATED by tools/gen_quad_derivs.py (sympy %s) -- do not edit; edit the generator.\n" % sp.__version__)
        fh.write("// Closed-form derivatives of the quadcopter's continuous dynamics xd = inertialDynamics(x, u) (models.h\n")
        fh.write("// quad_inertial_dynamics; reference zopt/quadcopter.py:23-144): %d nonzero first derivatives in 16 columns,\n" % nnz1)
        fh.write("// %d nonzero second derivatives in the %d declared pairs; every other second derivative is identically zero\n"
                 % (nnz2, len(PAIRS)))
        fh.write("// (checked symbolically by the generator, constant NED wind included).  WIND = false: still air (w = 0 substituted\n")
        fh.write("// before differentiating: less than half the operations).\n#pragma once\n\n")
        fh.write("#ifndef ZM_HD\n#if defined(__HIPCC__)\n#define ZM_HD __host__ __device__ __forceinline__\n#else\n#define ZM_HD inline\n#endif\n#endif\n\n")
        fh.write("// Each case is a real branch on the device: without the (empty, volatile) asm the compiler may turn the switch into selects --\n"
                 "// every lane evaluating every case.\n"
                 "#if defined(__HIP_DEVICE_COMPILE__)\n#define ZM_CASE_FENCE __asm__ volatile(\"\");\n#else\n#define ZM_CASE_FENCE\n#endif\n\n")
        fh.write("// The generated expressions spell their fused multiply-adds out; the compiler adds or removes none (see the generator's printer).\n"
                 "#if defined(__clang__)\n#define ZM_FP_STRICT _Pragma(\"clang fp contract(off)\")\n#else\n#define ZM_FP_STRICT\n#endif\n\n")
        fh.write("namespace zm {\n\n")
        fh.write("// what the closed forms read: the state, the thrust, the NED wind, sin / cos of phi (6), theta (7), psi (8), 1 / cos(theta)\n")
        fh.write("struct QuadAtoms {\n    double x[12], u0, w[3], s6, c6, s7, c7, s8, c8, ic7;\n};\n\n")
        stats = {}
        def two_pairs(hes):   # case j: pairs 2j (o[0..11]) and 2j+1 (o[12..23])
            return {j: [(12 * (p & 1) + i, e) for p in (2 * j, 2 * j + 1) if p in hes for i, e in hes[p]] for j in range((len(PAIRS) + 1) // 2)}

        for fn, table, nout, doc in (
                ("quad_jac_column", lambda w: d[w][0], 12, "o[i] = d xd_i / d z_j, z = [x ; u], column j"),
                ("quad_hess_pair", lambda w: d[w][1], 12, "o[i] = d2 xd_i / d z_a d z_b for declared pair j (models.h model_pair_table)"),
                ("quad_hess_pair2", lambda w: two_pairs(d[w][1]), 24,
                 "two declared pairs per case: o[i] = d2 xd_i for pair 2j, o[12 + i] for pair 2j+1 (14 cases: a 16-lane group per point)")):
            fh.write(f"// {doc}; entries not assigned are zero.  Temporaries shared by several cases are computed before the switch.\n")
            fh.write(f"template <bool WIND>\nZM_HD void {fn}(const int j, const QuadAtoms& a, double (&o)[{nout}]) {{\n    ZM_FP_STRICT\n")
            fh.write(f"    for (int i = 0; i < {nout}; ++i) o[i] = 0.0;\n    if constexpr (WIND) {{\n")
            stats[(fn, True)] = emit_body(fh, table(True), pr, "        ")
            fh.write("    } else {\n")
            stats[(fn, False)] = emit_body(fh, table(False), pr, "        ")
            fh.write("    }\n}\n\n")
        # packed Jacobians: the structurally nonzero entries of d xd / d z in (column, row) order; everything else of [f_x | f_u] is
        # the identity's 0 or 1.  The expansion kernel writes, the sweep kernels read only these.
        for wind in (False, True):
            jac = d[wind][0]
            pos, k = {}, 0
            for j in sorted(jac):
                for i, _ in jac[j]:
                    pos[(i, j)] = k
                    k += 1
            name = "WIND" if wind else "STILL"
            fh.write(f"constexpr int QUAD_NJ_{name} = {k};   // packed entries per point\n")
            fh.write(f"// packed position of d xd_i / d z_j, [i][j] row-major (255: structurally zero)\n")
            fh.write(f"constexpr unsigned char QUAD_JPOS_{name}[12 * 16] = {{\n")
            for i in range(12):
                fh.write("    " + ", ".join(str(pos.get((i, j), 255)) for j in range(16)) + ",\n")
            fh.write("};\n\n")
            d[wind] = d[wind] + (pos,)
        fh.write("// column j of the PACKED image: t[pos] = dt * d xd_i / d z_j + (i == j) for the column's nonzero entries (dt = 0: the derivative itself)\n")
        fh.write("template <bool WIND>\nZM_HD void quad_jac_column_packed(const int j, const QuadAtoms& a, const double dt, double* t) {\n    ZM_FP_STRICT\n")
        fh.write("    if constexpr (WIND) {\n")
        emit_body(fh, d[True][0], pr, "        ", packed=d[True][2])
        fh.write("    } else {\n")
        emit_body(fh, d[False][0], pr, "        ", packed=d[False][2])
        fh.write("    }\n}\n\n")
        fh.write("// the whole PACKED image by one lane (one lane per trajectory point): t[0 .. QUAD_NJ) as quad_jac_column_packed writes them, same\n"
                 "// temporaries and expressions\n")
        fh.write("template <bool WIND>\nZM_HD void quad_jac_all_packed(const QuadAtoms& a, const double dt, double* t) {\n    ZM_FP_STRICT\n")
        fh.write("    if constexpr (WIND) {\n")
        stats[("quad_jac_all_packed", True)] = (emit_all(fh, d[True][0], pr, "        ", d[True][2], True), 0)
        fh.write("    } else {\n")
        stats[("quad_jac_all_packed", False)] = (emit_all(fh, d[False][0], pr, "        ", d[False][2], True), 0)
        fh.write("    }\n}\n\n")
        # sparse second derivatives: of the 28 x 12 entries H[pair][i] only NH are structurally nonzero; the expansion writes those, the
        # DDP sweep scatters them into its dense LDS image (whose other entries stay zero)
        for wind in (False, True):
            hes = d[wind][1]
            hpos, k = {}, 0
            for pr_ in sorted(hes):
                for i, _ in hes[pr_]:
                    hpos[(i, pr_)] = k
                    k += 1
            name = "WIND" if wind else "STILL"
            fh.write(f"constexpr int QUAD_NH_{name} = {k};   // structurally nonzero second derivatives per point\n")
            fh.write(f"// dense index pair * 12 + i of packed entry k\n")
            fh.write(f"constexpr unsigned short QUAD_HDENSE_{name}[{k}] = {{\n    "
                     + ", ".join(str(pr_ * 12 + i) for (i, pr_), _ in sorted(hpos.items(), key=lambda kv: kv[1])) + ",\n};\n\n")
            d[wind] = d[wind] + (hpos,)

        def two_pairs_packed(hes, hpos):   # case j: pairs 2j and 2j+1; "row" index = packed position, "column" = the case
            tab, pk = {}, {}
            for j in range((len(PAIRS) + 1) // 2):
                tab[j] = []
                for pr_ in (2 * j, 2 * j + 1):
                    for i, e in hes.get(pr_, []):
                        tab[j].append((hpos[(i, pr_)], e))
                        pk[(hpos[(i, pr_)], j)] = hpos[(i, pr_)]
            return tab, pk

        fh.write("// the SPARSE image of the second derivatives: case j writes t[k] = dt * d2 xd_i (dt = 0: the derivative itself) for the nonzero\n"
                 "// entries of pairs 2j and 2j+1 at their packed positions k (QUAD_HDENSE_*)\n")
        fh.write("template <bool WIND>\nZM_HD void quad_hess_pair2_packed(const int j, const QuadAtoms& a, const double dt, double* t) {\n    ZM_FP_STRICT\n")
        fh.write("    if constexpr (WIND) {\n")
        tab, pk = two_pairs_packed(d[True][1], d[True][3])
        emit_body(fh, tab, pr, "        ", packed=pk, delta=False)
        fh.write("    } else {\n")
        tab, pk = two_pairs_packed(d[False][1], d[False][3])
        emit_body(fh, tab, pr, "        ", packed=pk, delta=False)
        fh.write("    }\n}\n\n")
        fh.write("// the whole SPARSE image by one lane: t[0 .. QUAD_NH) as quad_hess_pair2_packed writes them\n")
        fh.write("template <bool WIND>\nZM_HD void quad_hess_all_packed(const QuadAtoms& a, const double dt, double* t) {\n    ZM_FP_STRICT\n")
        fh.write("    if constexpr (WIND) {\n")
        tab, pk = two_pairs_packed(d[True][1], d[True][3])
        stats[("quad_hess_all_packed", True)] = (emit_all(fh, tab, pr, "        ", pk, False), 0)
        fh.write("    } else {\n")
        tab, pk = two_pairs_packed(d[False][1], d[False][3])
        stats[("quad_hess_all_packed", False)] = (emit_all(fh, tab, pr, "        ", pk, False), 0)
        fh.write("    }\n}\n\n")
        fh.write("}  // namespace zm\n")
    for k, v in stats.items():
        print(f"{k[0]} wind={k[1]}: {v[0]} operations before the switch, {v[1]} inside it", file=sys.stderr)
    print(f"wrote {out}: {nnz1} first derivatives, {nnz2} second derivatives in {len(PAIRS)} pairs", file=sys.stderr)


if __name__ == "__main__":
    main()
