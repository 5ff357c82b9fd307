#!/usr/bin/env python3
"""sympy -> C generator for the per-system dynamics headers.

Replaces ``sp.lambdify`` of f_disc / A_disc / B_disc
(/root/reference/python/symbolic_dynamics.py:23-25 and :88-90).

How parity with the lambdified numpy callables is kept:
the generator does NOT print the sympy tree with its own printer.  It asks
sympy for the exact python source ``lambdify`` would execute, parses that
source with ``ast`` and re-emits the same expression tree in C:

* same association of every ``+ - * /`` (python is left-associative, so is the
  emitted, fully parenthesised C),
* same literals (15-significant-digit decimals printed by sympy's StrPrinter;
  ``1/2`` stays a division of two exact constants),
* ``x**2`` becomes ``x*x`` (numpy's scalar pow is within 1 ulp of it),
* ``sin/cos`` calls are hoisted, one ``ILQR_DYN_SINCOS(arg, s, c)`` per distinct
  argument (the includer maps it to libm sin+cos in the oracle and to the shared
  range reduction of ``fast_trig.h`` on the device; an unused half is dead code),
  ``tan`` goes through ``ILQR_DYN_TAN``.  After the hoisted calls ONE branch
  ``if (ILQR_DYN_TRIG_BIG(a0) || ...) { ILQR_DYN_SINCOS_SLOW(...) }`` lets the device
  patch huge / non-finite arguments with the general ocml routines (never taken
  on sane trajectories; the oracle defines TRIG_BIG as 0),
* in ``f_sc`` divisors that are a state- and input-free multiple of another divisor share its reciprocal
  (``_proportional_divisors``)
* ``a / d`` becomes ``ILQR_DYN_DIV(a, d, rdK)`` with one hoisted ``rdK = ILQR_DYN_RCP(d)``
  per distinct non-constant divisor: the oracle defines DIV as the plain ``a / d``
  (so it stays bit-identical to numpy); the device shares the reciprocal among the
  many quotients with the same divisor and corrects each quotient with one fused
  residual step (q = a*r; q += (a - d*q)*r), which reproduces the correctly rounded
  IEEE quotient except in rare half-ulp cases.

Common sub-expression elimination is left to the C compiler, which is value
preserving under ``-ffp-contract=off`` (no re-association, no contraction).

Output: ``ilqr_amd/csrc/dynamics/<system>.h``.  The includer defines
``ILQR_DYN_FN`` (function qualifiers), ``ILQR_DYN_CONST`` (qualifier of the
sparsity tables), ``ILQR_DYN_SINCOS`` and ``ILQR_DYN_TAN`` -- hipcc for gfx950 in ``ilqr_hip.hip``, gcc in ``oracle/``.
The generated headers are committed: the GPU box never runs sympy.
"""
import ast
import re
import inspect
import os
import sys

import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import models  # noqa: E402

OUT_DIR = os.path.join(HERE, '..', 'ilqr_amd', 'csrc', 'dynamics')

_FUNCS = {'sin': 'sin', 'cos': 'cos', 'tan': 'tan', 'sqrt': 'sqrt', 'exp': 'exp', 'log': 'log'}


class _Emitter:
    """Translate one lambdify-generated function body to C statements."""

    def __init__(self, src):
        self.tree = ast.parse(src)
        fn = self.tree.body[0]
        assert isinstance(fn, ast.FunctionDef)
        argnames = [a.arg for a in fn.args.args]
        assert len(argnames) == 4, argnames
        carg = {argnames[0]: 'x', argnames[1]: 'u', argnames[3]: 'p'}
        self.names = {argnames[2]: 'dt'}
        self.ret = None
        for st in fn.body:
            if isinstance(st, ast.Assign):
                tgt, val = st.targets[0], st.value
                assert isinstance(val, ast.Name) and val.id in carg, ast.dump(st)
                for i, el in enumerate(tgt.elts):
                    self.names[el.id] = '%s[%d]' % (carg[val.id], i)
            elif isinstance(st, ast.Return):
                self.ret = st.value
        assert self.ret is not None
        self.trig = {}       # C text of a sin/cos argument -> index of its (sn_i, cs_i) pair
        self.trig_defs = []  # ordered definitions
        self.trig_args = []  # sin/cos arguments in order of first use
        self.tan_defs = []   # (index, argument) of tan calls
        self.rcp = {}        # C text of a divisor -> index of its hoisted reciprocal
        self.rcp_defs = []
        self.rcp_ast = {}    # index -> (C text, python ast) of the divisor

    def elements(self):
        call = self.ret
        assert isinstance(call, ast.Call) and call.func.id == 'array'
        rows = call.args[0].elts
        out = []
        for r in rows:
            out.append([self.expr(e) for e in r.elts])
        return out

    def expr(self, e):
        if isinstance(e, ast.BinOp):
            if isinstance(e.op, ast.Pow):
                return self.power(e.left, e.right)
            if isinstance(e.op, ast.Div):
                return self.divide(e.left, e.right)
            op = {ast.Add: '+', ast.Sub: '-', ast.Mult: '*'}[type(e.op)]
            return '(%s %s %s)' % (self.expr(e.left), op, self.expr(e.right))
        if isinstance(e, ast.UnaryOp):
            if isinstance(e.op, ast.USub):
                return '(-%s)' % self.expr(e.operand)
            if isinstance(e.op, ast.UAdd):
                return self.expr(e.operand)
        if isinstance(e, ast.Constant):
            v = e.value
            if isinstance(v, int):
                return '%d.0' % v
            return repr(float(v))
        if isinstance(e, ast.Name):
            return self.names[e.id]
        if isinstance(e, ast.Call):
            fname = e.func.id if isinstance(e.func, ast.Name) else e.func.attr
            assert fname in _FUNCS and len(e.args) == 1, ast.dump(e)
            arg = self.expr(e.args[0])
            if fname in ('sin', 'cos'):
                if arg not in self.trig:
                    i = len(self.trig)
                    self.trig[arg] = i
                    self.trig_defs.append('double sn%d, cs%d; ILQR_DYN_SINCOS(%s, sn%d, cs%d); (void)sn%d; (void)cs%d;' % (i, i, arg, i, i, i, i))
                    self.trig_args.append((arg, i))
                return ('sn%d' if fname == 'sin' else 'cs%d') % self.trig[arg]
            if fname == 'tan':
                key = 'tan:' + arg
                if key not in self.trig:
                    i = len(self.trig)
                    self.trig[key] = i
                    self.tan_defs.append((i, arg))
                return 'tn%d' % self.trig[key]
            return '%s(%s)' % (_FUNCS[fname], arg)
        raise NotImplementedError(ast.dump(e))

    def divide(self, num, den):
        """a / d -> ILQR_DYN_DIV(a, d, rdK) with one hoisted reciprocal per distinct divisor (see the header comment)."""
        n = self.expr(num)
        if isinstance(den, ast.Constant):
            return '(%s / %s)' % (n, self.expr(den))            # constant divisor: folded or a plain division
        d = self.expr(den)
        if d not in self.rcp:
            self.rcp[d] = len(self.rcp)
            self.rcp_ast[self.rcp[d]] = (d, den)
            self.rcp_defs.append('const double rd%d = ILQR_DYN_RCP(%s); (void)rd%d;' % (self.rcp[d], d, self.rcp[d]))
        return 'ILQR_DYN_DIV(%s, %s, rd%d)' % (n, d, self.rcp[d])

    def power(self, base, expo):
        b = self.expr(base)
        neg = False
        if isinstance(expo, ast.UnaryOp) and isinstance(expo.op, ast.USub):
            neg, expo = True, expo.operand
        assert isinstance(expo, ast.Constant), ast.dump(expo)
        v = expo.value
        if isinstance(v, int) and 1 <= v <= 4:
            r = {1: b, 2: 'ilqr_dyn_sq(%s)' % b, 3: 'ilqr_dyn_cube(%s)' % b,
                 4: 'ilqr_dyn_sq(ilqr_dyn_sq(%s))' % b}[v]
        elif float(v) == 0.5:
            r = 'sqrt(%s)' % b
        else:
            r = 'pow(%s, %r)' % (b, float(v))
        return '(1.0 / %s)' % r if neg else r


def _pattern(mat):
    """'0' structural zero, '1' structural one, 'x' anything else (row-major)."""
    out = []
    for i in range(mat.shape[0]):
        for j in range(mat.shape[1]):
            e = mat[i, j]
            out.append('0' if e == 0 else ('1' if e == 1 else 'x'))
    return ''.join(out)


def _dt_mask(mat, dt):
    """'1' where the entry is exactly the step dt (a uniform constant the solver need not store per time step), else '0'."""
    return ''.join('1' if mat[i, j] == dt else '0' for i in range(mat.shape[0]) for j in range(mat.shape[1]))


def _function(model, what, mat):
    args = (model.states, model.inputs, model.dt, model.parameters)
    src = inspect.getsource(sp.lambdify(args, mat))
    em = _Emitter(src)
    els = em.elements()
    lines = ['ILQR_DYN_FN void ilqr_dyn_%s_%s(const double* x, const double* u, const double dt, '
             'const double* p, double* out)' % (model.name, what), '{']
    if what == 'f':
        # the includer may allow fused multiply-adds inside f (and only there): ILQR_DYN_F_PROLOGUE, first in the body
        lines.append('    ILQR_DYN_F_PROLOGUE')
    lines.append('    (void)x; (void)u; (void)dt; (void)p;')
    # f only: sin/cos of (a + b) where a and b already have their pair -> angle addition (ILQR_DYN_SINCOS_SUM).  In f
    # every trig value is scaled by dt before it meets the state, so the ~2e-16 absolute error of the sum formula is far
    # below an ulp of x; the Jacobians (A, B entries are dt * expression, no large addend) keep the direct evaluation.
    sums = {}
    if what == 'f':
        for arg, i in em.trig_args:
            for a, ia in em.trig_args:
                for b, ib in em.trig_args:
                    if arg == '(%s + %s)' % (a, b) and i not in (ia, ib):
                        sums[i] = (a, ia, b, ib)
    for d in em.trig_defs:
        idx = re.match(r'double sn(\d+),', d)
        if idx is None or int(idx.group(1)) not in sums:
            lines.append('    ' + d)
    for i, (a, ia, b, ib) in sorted(sums.items()):
        lines.append('    double sn%d, cs%d; ILQR_DYN_SINCOS_SUM(%s, %s, sn%d, cs%d, sn%d, cs%d, sn%d, cs%d); (void)sn%d; (void)cs%d;'
                     % (i, i, a, b, ia, ia, ib, ib, i, i, i, i))
    if em.trig_args:
        # one rarely-taken branch for all arguments of this function (keeps the fast paths in one basic block); a sum
        # of two small angles is small enough for the sum formula, so only the plain arguments are tested
        lines.append('    if (%s) {' % ' || '.join('ILQR_DYN_TRIG_BIG(%s)' % a for a, i in em.trig_args if i not in sums))
        for a, i in em.trig_args:
            lines.append('        ILQR_DYN_SINCOS_SLOW(%s, sn%d, cs%d);' % (a, i, i))
        lines.append('    }')
    # tan: in f, an argument whose sin/cos pair exists anyway becomes one quotient (ILQR_DYN_TAN_SC; dt-scaled like the
    # sum formula above); elsewhere the library tan
    pair = dict(em.trig_args)
    for i, arg in em.tan_defs:
        if what == 'f' and arg in pair:
            lines.append('    const double tn%d = ILQR_DYN_TAN_SC(%s, sn%d, cs%d);' % (i, arg, pair[arg], pair[arg]))
        else:
            lines.append('    const double tn%d = ILQR_DYN_TAN(%s);' % (i, arg))
    for d in em.rcp_defs:
        lines.append('    ' + d)
    k = 0
    for row in els:
        for e in row:
            lines.append('    out[%d] = %s;' % (k, e))
            k += 1
    lines.append('}')
    if what == 'f':
        lines += _function_f_sc(model, em, els, sums)
    else:
        lines += _function_jac_sc(model, what, em, els)
    return '\n'.join(lines), src


def _function_jac_sc(model, what, em, els):
    """A or B once more for callers that carry the sin/cos pairs of the f_ang state components along a trajectory (the ltvlqr
    recursion walking a given trajectory backwards): those pairs come in through `sc` as in f_sc, the pair of a sum of two
    of them is plain angle addition, everything else -- other arguments, tan, every divisor -- is as in the plain function.
    The Jacobians' entries are dt * expression (plus a structural one): a trig value's last bits reach them scaled by dt."""
    f_ang = model._f_ang
    slot, sums, direct = {}, {}, []
    for arg, i in em.trig_args:
        m = re.fullmatch(r'x\[(\d+)\]', arg)
        if m and int(m.group(1)) in f_ang:
            slot[i] = f_ang.index(int(m.group(1)))
    for arg, i in em.trig_args:
        for a, ia in em.trig_args:
            for b, ib in em.trig_args:
                if arg == '(%s + %s)' % (a, b) and ia in slot and ib in slot and i not in slot:
                    sums[i] = (a, ia, b, ib)
    lines = ['', 'ILQR_DYN_FN void ilqr_dyn_%s_%s_sc(const double* x, const double* u, const double dt, const double* p, '
             'const double* sc, double* out)' % (model.name, what), '{', '    (void)x; (void)u; (void)dt; (void)p; (void)sc;']
    for arg, i in em.trig_args:
        if i in slot:
            lines.append('    const double sn%d = sc[%d], cs%d = sc[%d]; (void)sn%d; (void)cs%d;' % (i, 2 * slot[i], i, 2 * slot[i] + 1, i, i))
        elif i not in sums:
            lines.append('    double sn%d, cs%d; ILQR_DYN_SINCOS(%s, sn%d, cs%d); (void)sn%d; (void)cs%d;' % (i, i, arg, i, i, i, i))
            direct.append((arg, i))
    for i, (a, ia, b, ib) in sorted(sums.items()):
        lines.append('    double sn%d, cs%d; ILQR_DYN_SINCOS_SUM_SC(%s, %s, sn%d, cs%d, sn%d, cs%d, sn%d, cs%d); (void)sn%d; (void)cs%d;'
                     % (i, i, a, b, ia, ia, ib, ib, i, i, i, i))
    if direct:
        lines.append('    if (%s) {' % ' || '.join('ILQR_DYN_TRIG_BIG(%s)' % a for a, i in direct))
        for a, i in direct:
            lines.append('        ILQR_DYN_SINCOS_SLOW(%s, sn%d, cs%d);' % (a, i, i))
        lines.append('    }')
    for i, arg in em.tan_defs:
        lines.append('    const double tn%d = ILQR_DYN_TAN(%s);' % (i, arg))
    for d in em.rcp_defs:
        lines.append('    ' + d)
    k = 0
    for row in els:
        for e in row:
            lines.append('    out[%d] = %s;' % (k, e))
            k += 1
    lines.append('}')
    return lines


def _function_f_sc(model, em, els, sums):
    """f once more, for the time-step loops: sin/cos of the angles that are plain state components (x[i]) are NOT
    evaluated here but handed in by the caller in `sc` (pair k = sin, cos of state component f_ang[k]), who carries them
    from step to step by angle addition on the step's exact increment (kernels_common.h TrigCarry) instead of a fresh
    range reduction + two polynomial kernels per step.  Sums of two such angles keep the sum formula; any other argument
    is evaluated as in f.  Same expression tree otherwise."""
    up = model.name.upper()
    carried = []                       # (state index, pair index i) in order of first use
    for arg, i in em.trig_args:
        m = re.fullmatch(r'x\[(\d+)\]', arg)
        if m and i not in sums:
            carried.append((int(m.group(1)), i))
    slot = {i: k for k, (_, i) in enumerate(carried)}
    model._f_ang = [a for a, _ in carried]     # (the Jacobians' _sc variants, generated after f, take the same pairs)
    lines = ['', '#define ILQR_DYN_%s_F_NANG %d' % (up, len(carried)),
             '/* state components whose sin/cos pair the caller of f_sc carries */',
             'ILQR_DYN_CONST int ilqr_dyn_%s_f_ang[%d] = {%s};' % (model.name, max(len(carried), 1),
                                                                    ', '.join(str(a) for a, _ in carried) or '0'),
             'ILQR_DYN_FN void ilqr_dyn_%s_f_sc(const double* x, const double* u, const double dt, const double* p, '
             'const double* sc, double* out)' % model.name, '{', '    ILQR_DYN_F_PROLOGUE',
             '    (void)x; (void)u; (void)dt; (void)p; (void)sc;']
    direct = []
    for arg, i in em.trig_args:
        if i in slot:
            lines.append('    const double sn%d = sc[%d], cs%d = sc[%d]; (void)sn%d; (void)cs%d;' % (i, 2 * slot[i], i, 2 * slot[i] + 1, i, i))
        elif i not in sums:
            lines.append('    double sn%d, cs%d; ILQR_DYN_SINCOS(%s, sn%d, cs%d); (void)sn%d; (void)cs%d;' % (i, i, arg, i, i, i, i))
            direct.append((arg, i))
    for i, (a, ia, b, ib) in sorted(sums.items()):
        # both operands' pairs carried by the caller: ILQR_DYN_SINCOS_SUM_SC (the includer may drop the correction for the
        # rounding of a + b there: it is smaller than the carried pairs' own error)
        mac = 'ILQR_DYN_SINCOS_SUM_SC' if ia in slot and ib in slot else 'ILQR_DYN_SINCOS_SUM'
        lines.append('    double sn%d, cs%d; %s(%s, %s, sn%d, cs%d, sn%d, cs%d, sn%d, cs%d); (void)sn%d; (void)cs%d;'
                     % (i, i, mac, a, b, ia, ia, ib, ib, i, i, i, i))
    if direct:
        lines.append('    if (%s) {' % ' || '.join('ILQR_DYN_TRIG_BIG(%s)' % a for a, i in direct))
        for a, i in direct:
            lines.append('        ILQR_DYN_SINCOS_SLOW(%s, sn%d, cs%d);' % (a, i, i))
        lines.append('    }')
    pair = dict(em.trig_args)
    for i, arg in em.tan_defs:
        if arg in pair:
            lines.append('    const double tn%d = ILQR_DYN_TAN_SC(%s, sn%d, cs%d);' % (i, arg, pair[arg], pair[arg]))
        else:
            lines.append('    const double tn%d = ILQR_DYN_TAN(%s);' % (i, arg))
    # Divisors that differ by a factor free of states and inputs (acrobot: D, L_2 D, L_2^2 m_2 D) share ONE reciprocal
    # and one evaluation here: d_i = c_i * d_base, 1/d_i = (1/c_i) * (1/d_base), with c_i and 1/c_i invariant along a rollout
    # (the compiler hoists them out of the time loop).  The quotient keeps its residual correction against d_i, so it
    # stays within an ulp of a / d_i; d_i itself may differ from the reference's expression in the last bit, which
    # reaches the state scaled by dt (f only -- the plain f above and the Jacobians keep every divisor as lambdify wrote it).
    shared = _proportional_divisors(em)
    subst = {}
    for i in sorted(em.rcp_ast, key=lambda q: (q in shared, q)):          # the base before the divisors derived from it
        text = em.rcp_ast[i][0]
        if i in shared:
            base, ratio_c, inv_c = shared[i]
            lines.append('    const double dv%d = %s * dv%d;' % (i, ratio_c, base))
            lines.append('    const double rd%d = %s * rd%d; (void)rd%d;' % (i, inv_c, base, i))
            subst[i] = text
        elif any(b == i for b, _, _ in shared.values()):
            lines.append('    const double dv%d = %s;' % (i, text))
            lines.append('    const double rd%d = ILQR_DYN_RCP(dv%d); (void)rd%d;' % (i, i, i))
            subst[i] = text
        else:
            lines.append('    ' + em.rcp_defs[i])
    k = 0
    for row in els:
        for e in row:
            for i, text in subst.items():
                e = e.replace('%s, rd%d)' % (text, i), 'dv%d, rd%d)' % (i, i))
            lines.append('    out[%d] = %s;' % (k, e))
            k += 1
    lines.append('}')
    return lines


def _proportional_divisors(em):
    """{index: (base index, C text of d_i / d_base, C text of d_base / d_i)} for the divisors of one function that are a
    state- and input-free multiple of another one (the base: the divisor most others are multiples of, then the shortest)."""
    if len(em.rcp_ast) < 2:
        return {}
    loc = {name: sp.Symbol(name, real=True) for name in em.names}
    free = {sp.Symbol(name, real=True) for name, c in em.names.items() if c.startswith('p[') or c == 'dt'}
    exprs = {i: sp.sympify(ast.unparse(node), locals=loc) for i, (_, node) in em.rcp_ast.items()}
    best = {}
    for b in sorted(exprs, key=lambda i: (len(em.rcp_ast[i][0]), i)):
        found = {}
        if not (exprs[b].free_symbols - free):
            continue                                        # a divisor without states is no base (nothing to share per step)
        for i in exprs:
            if i == b:
                continue
            ratio = sp.simplify(exprs[i] / exprs[b])
            if ratio.free_symbols <= free and ratio != 0:
                found[i] = ratio
        if len(found) > len(best):
            best = {i: (b, r) for i, r in found.items()}
    def plain(node):
        """C text of a parameter-only expression: plain IEEE operations (evaluated once per rollout, not per step)"""
        if isinstance(node, ast.BinOp):
            if isinstance(node.op, ast.Pow):
                assert isinstance(node.right, ast.Constant) and isinstance(node.right.value, int) and 1 <= node.right.value <= 4, ast.dump(node)
                return '(' + ' * '.join([plain(node.left)] * node.right.value) + ')'
            op = {ast.Add: '+', ast.Sub: '-', ast.Mult: '*', ast.Div: '/'}[type(node.op)]
            return '(%s %s %s)' % (plain(node.left), op, plain(node.right))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return '(-%s)' % plain(node.operand)
        if isinstance(node, ast.Constant):
            return repr(float(node.value))
        if isinstance(node, ast.Name):
            return em.names[node.id]
        raise NotImplementedError(ast.dump(node))

    out = {}
    for i, (b, ratio) in best.items():
        num, den = sp.fraction(sp.together(ratio))
        c = lambda e: plain(ast.parse(sp.pycode(e), mode='eval').body)
        out[i] = (b, '(%s / %s)' % (c(num), c(den)) if den != 1 else c(num), '(%s / %s)' % (c(den), c(num)))
    return out


def render(name):
    """(text of the header, lambdify sources) of one system"""
    model = models.BUILDERS[name]()
    n, m, npar = model.n, model.m, model.n_params
    up = name.upper()
    parts = []
    parts.append('/* GENERATED by codegen/gen_dynamics.py from codegen/models.py:%s() -- do not edit.\n'
                 ' * Forward-Euler discretised dynamics and Jacobians of the %s system, emitted from the\n'
                 ' * python source sympy.lambdify would run (same expression tree, same literals).\n'
                 ' * Reference: /root/reference/python/symbolic_dynamics.py:23-25,88-90 (lambdify call sites).\n'
                 ' * Row-major: f out[n], A out[n*n] (A[i][j] = out[i*n+j]), B out[n*m].\n'
                 ' * The includer defines ILQR_DYN_FN, ILQR_DYN_CONST, ILQR_DYN_SINCOS(x, s, c), ILQR_DYN_SINCOS_SLOW(x, s, c),\n'
                 ' * ILQR_DYN_SINCOS_SUM(a, b, sa, ca, sb, cb, s, c), ILQR_DYN_SINCOS_SUM_SC(same), ILQR_DYN_TRIG_BIG(x), ILQR_DYN_TAN(x), ILQR_DYN_TAN_SC(x, s, c), ILQR_DYN_RCP(d), ILQR_DYN_DIV(a, d, rd), ILQR_DYN_F_PROLOGUE and must compile\n'
                 ' * with -ffp-contract=off. */'
                 % (name, name))
    parts.append('#ifndef ILQR_DYN_%s_H\n#define ILQR_DYN_%s_H' % (up, up))
    parts.append('#define ILQR_DYN_%s_ID %d\n#define ILQR_DYN_%s_N %d\n#define ILQR_DYN_%s_M %d\n#define ILQR_DYN_%s_NP %d'
                 % (up, models.SYSTEM_IDS[name], up, n, up, m, up, npar))
    parts.append('#ifndef ILQR_DYN_HELPERS\n#define ILQR_DYN_HELPERS\n'
                 'ILQR_DYN_FN double ilqr_dyn_sq(const double a) { return a * a; }\n'
                 'ILQR_DYN_FN double ilqr_dyn_cube(const double a) { return (a * a) * a; }\n#endif')
    parts.append('/* sparsity of the Jacobians: \'0\' structural zero, \'1\' structural one, \'x\' general */')
    parts.append('ILQR_DYN_CONST char ilqr_dyn_%s_A_pat[%d] = "%s";' % (name, n * n + 1, _pattern(model.A_disc)))
    parts.append('ILQR_DYN_CONST char ilqr_dyn_%s_B_pat[%d] = "%s";' % (name, n * m + 1, _pattern(model.B_disc)))
    parts.append('/* entries that are exactly dt (general for the algebra, but uniform: not stored per step) */')
    parts.append('ILQR_DYN_CONST char ilqr_dyn_%s_A_dt[%d] = "%s";' % (name, n * n + 1, _dt_mask(model.A_disc, model.dt)))
    parts.append('ILQR_DYN_CONST char ilqr_dyn_%s_B_dt[%d] = "%s";' % (name, n * m + 1, _dt_mask(model.B_disc, model.dt)))
    srcs = {}
    for what, mat in (('f', model.f_disc), ('A', model.A_disc), ('B', model.B_disc)):
        code, src = _function(model, what, mat)
        parts.append(code)
        srcs[what] = src
    parts.append('#endif')
    return '\n\n'.join(parts) + '\n', srcs


def generate(name):
    text, srcs = render(name)
    path = os.path.join(OUT_DIR, name + '.h')
    with open(path, 'w') as fh:
        fh.write(text)
    return path, srcs


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    names = sys.argv[1:] or list(models.BUILDERS)
    for name in names:
        path, _ = generate(name)
        print('wrote', os.path.relpath(path), os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
