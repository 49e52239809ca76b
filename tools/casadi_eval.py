"""An EVALUATING stand-in for the slice of casadi 3.6 that `misc/force_controller.py:12-225` uses -- build-container
fixture tooling (G12), never shipped, never imported by the product or by anything that runs on the GPU box.

Why: casadi / qpOASES are absent here (environment.yml:25) and cannot be installed, so the reference's convex-MPC force QP
cannot be SOLVED by the reference's solver.  What the reference owns, though, is the FORMULATION: the cost through the
linearised single-rigid-body dynamics (`force_controller.py:70-105`), the swing-zero equalities and the friction-pyramid
`bounded(...)` constraints (`:107-162`), and how `Kalman_Filter.predict_mpc` fills the parameters
(`kalman_filter/kalman_filter.py:140-152`).  With this module registered as `casadi`, `StanceController.__init__`,
`objective_function`, `control_constraints` and `predict_mpc` run as UNMODIFIED reference code; every casadi call they
make builds a lazy expression node here.  At `opti.solve()` the expression graph is evaluated ONCE in the ring of
polynomials of degree <= 2 in the decision variables (exact algebra -- no finite differences, no cancellation):

    cost(u)        = c + g.u + u.H.u          -> H (60x60, symmetrised), g (60), c
    each constraint = affine  A_r.u + b_r      (== 0 for `==`; >= 0 twice for `bounded(lo, e, hi)`: e - lo, hi - e,
                                               because lo / hi contain decision variables: -mu*fz <= fx <= mu*fz)

in Opti's own decision-variable order (variables in order of creation, each matrix column-major:
x = [vec f1; vec f2; vec f3; vec f4], `force_controller.py:52-55`).  Anything of degree > 2 in the cost or > 1 in a
constraint raises.  The QP (strictly convex: R = 1e-6 I, `kalman_filter.py:66-70`) is then solved by the KKT-certified
active-set routine of oracle/mpc_oracle.py (`solve`), whose certificate (stationarity, primal feasibility, multiplier
signs) is recorded next to the solution: "formulation = reference, solver = certified stand-in".  `sol.value(expr)`
evaluates any expression at the solution, as casadi's OptiSol does.

Semantics followed (casadi 3.6 as published): matrices are always 2-D; a 1x1 operand broadcasts in element-wise
operations; `*` is element-wise, `mtimes` the matrix product; `X[i]` with one index is column-major linear indexing;
NumPy 1-D arrays are column vectors; `if_else(c, a, b)` selects on a numeric 0/1 condition; `skew(v)` is the 3x3 cross-
product matrix; `inv`, `cos`, `sin`, `tan` and `/` by an expression are evaluated only where the operand is free of
decision variables (true for every use in the reference), else raise.
"""
import os
import sys
import types

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------------------
# values: matrix-valued polynomials of degree <= 2 in the n decision variables
# ---------------------------------------------------------------------------------------------------------------
class Poly:
    """c0 (r,c) + c1 (r,c,n).u + u.c2 (r,c,n,n).u ; c1 / c2 are None when identically zero."""
    __slots__ = ("c0", "c1", "c2")

    def __init__(self, c0, c1=None, c2=None):
        self.c0 = np.asarray(c0, dtype=np.float64)
        assert self.c0.ndim == 2, self.c0.shape
        self.c1, self.c2 = c1, c2

    @property
    def shape(self):
        return self.c0.shape

    def is_const(self):
        return self.c1 is None and self.c2 is None

    def const(self, what):
        if not self.is_const():
            raise ValueError(f"casadi_eval: {what} of an expression that depends on decision variables")
        return self.c0

    def map_parts(self, f):
        """Apply a linear re-arrangement of the (r,c) axes (index / transpose / concatenate helper) to all parts."""
        return Poly(f(self.c0), None if self.c1 is None else f(self.c1), None if self.c2 is None else f(self.c2))


def _bshape(a, b):
    if a.shape == b.shape or b.shape == (1, 1):
        return a.shape
    if a.shape == (1, 1):
        return b.shape
    raise ValueError(f"casadi_eval: element-wise op on shapes {a.shape} and {b.shape}")


def _addpart(x, y, shape, tail):
    if x is None and y is None:
        return None
    z = np.zeros(shape + tail)
    if x is not None:
        z = z + x
    if y is not None:
        z = z + y
    return z


def p_add(a, b, sign=1.0):
    shape = _bshape(a, b)
    n1 = (a.c1 if a.c1 is not None else b.c1)
    n = None if n1 is None else n1.shape[-1]
    n2 = (a.c2 if a.c2 is not None else b.c2)
    nn = None if n2 is None else n2.shape[-1]
    sb = lambda v: None if v is None else sign * v
    return Poly(np.zeros(shape) + a.c0 + sign * b.c0,
                _addpart(a.c1, sb(b.c1), shape, (n,)) if n is not None else None,
                _addpart(a.c2, sb(b.c2), shape, (nn, nn)) if nn is not None else None)


def p_neg(a):
    return Poly(-a.c0, None if a.c1 is None else -a.c1, None if a.c2 is None else -a.c2)


def _deg_guard(*high):
    for h in high:
        if h is not None and np.any(h != 0):
            raise ValueError("casadi_eval: expression of degree > 2 in the decision variables")


def p_mul(a, b):
    """element-wise (casadi `*`)"""
    shape = _bshape(a, b)
    if a.c2 is not None:
        _deg_guard(b.c1, b.c2)
    if b.c2 is not None:
        _deg_guard(a.c1, a.c2)
    c0 = np.zeros(shape) + a.c0 * b.c0
    c1 = None
    if a.c1 is not None:
        c1 = a.c1 * b.c0[..., None]
    if b.c1 is not None:
        t = a.c0[..., None] * b.c1
        c1 = t if c1 is None else c1 + t
    c2 = None
    for t in ((a.c2 * b.c0[..., None, None]) if a.c2 is not None else None,
              (a.c0[..., None, None] * b.c2) if b.c2 is not None else None,
              (a.c1[..., :, None] * b.c1[..., None, :]) if (a.c1 is not None and b.c1 is not None) else None):
        if t is not None:
            c2 = t if c2 is None else c2 + t
    if c1 is not None:
        c1 = np.zeros(shape + c1.shape[-1:]) + c1
    if c2 is not None:
        c2 = np.zeros(shape + c2.shape[-2:]) + c2
    return Poly(c0, c1, c2)


def p_mtimes(a, b):
    if a.shape == (1, 1) or b.shape == (1, 1):
        return p_mul(a, b)
    if a.shape[1] != b.shape[0]:
        raise ValueError(f"casadi_eval: mtimes of shapes {a.shape} and {b.shape}")
    if a.c2 is not None:
        _deg_guard(b.c1, b.c2)
    if b.c2 is not None:
        _deg_guard(a.c1, a.c2)
    c0 = a.c0 @ b.c0
    c1 = None
    if a.c1 is not None:
        c1 = np.einsum("ikn,kj->ijn", a.c1, b.c0)
    if b.c1 is not None:
        t = np.einsum("ik,kjn->ijn", a.c0, b.c1)
        c1 = t if c1 is None else c1 + t
    c2 = None
    for t in (np.einsum("ikmn,kj->ijmn", a.c2, b.c0) if a.c2 is not None else None,
              np.einsum("ik,kjmn->ijmn", a.c0, b.c2) if b.c2 is not None else None,
              np.einsum("ikm,kjn->ijmn", a.c1, b.c1) if (a.c1 is not None and b.c1 is not None) else None):
        if t is not None:
            c2 = t if c2 is None else c2 + t
    return Poly(c0, c1, c2)


# ---------------------------------------------------------------------------------------------------------------
# lazy expression nodes (what the reference's casadi calls build)
# ---------------------------------------------------------------------------------------------------------------
class Expr:
    __array_ufunc__ = None          # ndarray (op) Expr defers to Expr.__r*__  (np.eye(12) + A * dt, force_controller.py:93)
    __array_priority__ = 1000.0

    def __init__(self, op, args=(), meta=None):
        self.op, self.args, self.meta = op, tuple(args), meta

    # arithmetic ------------------------------------------------------------------------------------------
    def __add__(self, o): return Expr("add", (self, _E(o)))
    def __radd__(self, o): return Expr("add", (_E(o), self))
    def __sub__(self, o): return Expr("sub", (self, _E(o)))
    def __rsub__(self, o): return Expr("sub", (_E(o), self))
    def __mul__(self, o): return Expr("mul", (self, _E(o)))
    def __rmul__(self, o): return Expr("mul", (_E(o), self))
    def __truediv__(self, o): return Expr("div", (self, _E(o)))
    def __rtruediv__(self, o): return Expr("div", (_E(o), self))
    def __neg__(self): return Expr("neg", (self,))
    def __pos__(self): return self
    def __matmul__(self, o): return Expr("mtimes", (self, _E(o)))
    def __rmatmul__(self, o): return Expr("mtimes", (_E(o), self))
    def __eq__(self, o): return Expr("eq", (self, _E(o)))
    def __le__(self, o): return Expr("le", (self, _E(o)))
    def __ge__(self, o): return Expr("le", (_E(o), self))
    __hash__ = object.__hash__

    def __bool__(self):
        raise TypeError("casadi_eval: truth value of a symbolic expression")

    @property
    def T(self):
        return Expr("T", (self,))

    def __getitem__(self, key):
        return Expr("index", (self,), key)


def _E(x):
    if isinstance(x, Expr):
        return x
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)          # casadi: a 1-D NumPy array is a column vector
    elif a.ndim != 2:
        raise ValueError(f"casadi_eval: array of rank {a.ndim}")
    return Expr("const", (), a.copy())


def _norm_index(k, size):
    if isinstance(k, slice):
        return k
    k = int(k)
    if k < 0:
        k += size
    if not 0 <= k < size:
        raise IndexError(f"casadi_eval: index {k} out of range {size}")
    return slice(k, k + 1)


def _skew_parts(v):
    """v: (3,1,...) -> (3,3,...)  [[0,-z,y],[z,0,-x],[-y,x,0]]"""
    x, y, z = v[0, 0], v[1, 0], v[2, 0]
    o = np.zeros_like(x)
    return np.stack([np.stack([o, -z, y]), np.stack([z, o, -x]), np.stack([-y, x, o])])


class _Evaluator:
    def __init__(self, opti):
        self.opti = opti
        self.n = opti.nvar
        self.memo = {}

    def __call__(self, e):
        key = id(e)
        got = self.memo.get(key)
        if got is None:
            got = self._eval(e)
            self.memo[key] = got
        return got

    def _eval(self, e):
        op, n = e.op, self.n
        if op == "const":
            return Poly(e.meta)
        if op == "var":
            off, r, c = e.meta
            c1 = np.zeros((r, c, n))
            for j in range(c):
                for i in range(r):
                    c1[i, j, off + j * r + i] = 1.0          # column-major, Opti's order
            return Poly(np.zeros((r, c)), c1)
        if op == "param":
            pid, r, c = e.meta
            if pid not in self.opti.values:
                raise RuntimeError("casadi_eval: parameter without a value (opti.set_value missing)")
            return Poly(self.opti.values[pid])
        a = [self(x) for x in e.args]
        if op == "add":
            return p_add(a[0], a[1])
        if op == "sub":
            return p_add(a[0], a[1], -1.0)
        if op == "neg":
            return p_neg(a[0])
        if op == "mul":
            return p_mul(a[0], a[1])
        if op == "div":
            return p_mul(a[0], Poly(1.0 / a[1].const("division by")))
        if op == "mtimes":
            return p_mtimes(a[0], a[1])
        if op == "T":
            return a[0].map_parts(lambda v: np.swapaxes(v, 0, 1))
        if op in ("cos", "sin", "tan"):
            return Poly(getattr(np, op)(a[0].const(op)))
        if op == "inv":
            return Poly(np.linalg.inv(a[0].const("inv")))
        if op == "skew":
            if a[0].shape != (3, 1):
                raise ValueError(f"casadi_eval: skew of shape {a[0].shape}")
            return a[0].map_parts(_skew_parts)
        if op == "vertcat" or op == "horzcat":
            ax = 0 if op == "vertcat" else 1
            parts = [p for p in a if p.shape[ax] > 0 or True]
            has1 = any(p.c1 is not None for p in parts)
            has2 = any(p.c2 is not None for p in parts)
            c0 = np.concatenate([p.c0 for p in parts], axis=ax)
            c1 = np.concatenate([p.c1 if p.c1 is not None else np.zeros(p.shape + (n,)) for p in parts], axis=ax) if has1 else None
            c2 = np.concatenate([p.c2 if p.c2 is not None else np.zeros(p.shape + (n, n)) for p in parts], axis=ax) if has2 else None
            return Poly(c0, c1, c2)
        if op == "index":
            key = e.meta
            r, c = a[0].shape
            if isinstance(key, tuple):
                if len(key) != 2:
                    raise IndexError("casadi_eval: matrices take one or two indices")
                ki, kj = _norm_index(key[0], r), _norm_index(key[1], c)
                return a[0].map_parts(lambda v: v[ki, kj])
            # one index: column-major linear indexing -> column vector
            k = _norm_index(key, r * c)
            def lin(v):
                flat = np.swapaxes(v, 0, 1).reshape((r * c,) + v.shape[2:])
                return flat[k].reshape((-1, 1) + v.shape[2:])
            return a[0].map_parts(lin)
        if op == "eq":
            # as a VALUE (the condition of if_else): numeric comparison of variable-free operands
            return Poly((a[0].const("==") == a[1].const("==")).astype(np.float64) + np.zeros(_bshape(a[0], a[1])))
        if op == "if_else":
            c = a[0].const("if_else condition")
            if c.shape != (1, 1):
                raise ValueError("casadi_eval: if_else condition must be scalar")
            return a[1] if c[0, 0] != 0 else a[2]
        raise NotImplementedError(f"casadi_eval: op {op!r}")


# ---------------------------------------------------------------------------------------------------------------
# the casadi surface
# ---------------------------------------------------------------------------------------------------------------
def vertcat(*xs): return Expr("vertcat", [_E(x) for x in xs])
def horzcat(*xs): return Expr("horzcat", [_E(x) for x in xs])
def mtimes(a, b): return Expr("mtimes", (_E(a), _E(b)))
def if_else(c, a, b, *_): return Expr("if_else", (_E(c), _E(a), _E(b)))
def cos(x): return Expr("cos", (_E(x),)) if isinstance(x, Expr) else np.cos(x)
def sin(x): return Expr("sin", (_E(x),)) if isinstance(x, Expr) else np.sin(x)
def tan(x): return Expr("tan", (_E(x),)) if isinstance(x, Expr) else np.tan(x)
def transpose(x): return Expr("T", (_E(x),))
def inv(x): return Expr("inv", (_E(x),))
def skew(x): return Expr("skew", (_E(x),))


QP_LOG = []          # every solve() appends its extracted QP here (the fixture generator reads it)


class OptiSol:
    def __init__(self, opti, u):
        self._opti, self._u = opti, u

    def value(self, e):
        p = _Evaluator(self._opti)(_E(e))
        v = p.c0.copy()
        if p.c1 is not None:
            v += p.c1 @ self._u
        if p.c2 is not None:
            v += np.einsum("ijmn,m,n->ij", p.c2, self._u, self._u)
        return v


class Opti:
    def __init__(self, kind="nlp"):
        self.kind = kind
        self.nvar = 0
        self.nparam = 0
        self.values = {}
        self.constraints = []
        self.cost = None
        self.solver_name, self.solver_opts = None, None

    def variable(self, r=1, c=1):
        e = Expr("var", (), (self.nvar, int(r), int(c)))
        self.nvar += int(r) * int(c)
        return e

    def parameter(self, r=1, c=1):
        e = Expr("param", (), (self.nparam, int(r), int(c)))
        self.nparam += 1
        return e

    def set_value(self, par, val):
        if not (isinstance(par, Expr) and par.op == "param"):
            raise TypeError("casadi_eval: set_value on something that is not a parameter")
        pid, r, c = par.meta
        v = np.array(val, dtype=np.float64)
        if v.size != r * c:
            raise ValueError(f"casadi_eval: set_value shape {v.shape} for a ({r},{c}) parameter")
        self.values[pid] = v.reshape(r, c).copy()

    def bounded(self, lo, e, hi):
        return Expr("bounded", (_E(lo), _E(e), _E(hi)))

    def subject_to(self, c=None):
        if c is None:
            self.constraints = []
            return
        if not (isinstance(c, Expr) and c.op in ("eq", "bounded", "le")):
            raise TypeError("casadi_eval: subject_to takes ==, <= or bounded(...)")
        self.constraints.append(c)

    def minimize(self, e):
        self.cost = _E(e)

    def solver(self, name, opts=None, *_):
        self.solver_name, self.solver_opts = name, dict(opts or {})

    # -- extraction -----------------------------------------------------------------------------------------
    def extract_qp(self):
        """cost = c + g.u + u.H.u ; rows A u + b (== 0 where is_eq, else >= 0), in the order subject_to saw them."""
        ev = _Evaluator(self)
        n = self.nvar
        cp = ev(self.cost)
        if cp.shape != (1, 1):
            raise ValueError("casadi_eval: cost is not scalar")
        c = float(cp.c0[0, 0])
        g = cp.c1[0, 0].copy() if cp.c1 is not None else np.zeros(n)
        H = cp.c2[0, 0].copy() if cp.c2 is not None else np.zeros((n, n))
        H = 0.5 * (H + H.T)
        A, b, is_eq, src = [], [], [], []

        def rows(p, eq, tag):
            if p.c2 is not None and np.any(p.c2 != 0):
                raise ValueError("casadi_eval: constraint of degree 2")
            r, cc = p.shape
            for j in range(cc):                      # column-major, as casadi vectorises
                for i in range(r):
                    A.append(p.c1[i, j].copy() if p.c1 is not None else np.zeros(n))
                    b.append(p.c0[i, j]); is_eq.append(eq); src.append(tag)

        for k, con in enumerate(self.constraints):
            if con.op == "eq":
                rows(p_add(ev(con.args[0]), ev(con.args[1]), -1.0), True, k)
            elif con.op == "le":
                rows(p_add(ev(con.args[1]), ev(con.args[0]), -1.0), False, k)
            else:
                lo, e, hi = (ev(x) for x in con.args)
                rows(p_add(e, lo, -1.0), False, k)
                rows(p_add(hi, e, -1.0), False, k)
        return dict(H=H, g=g, c=c, A=np.array(A).reshape(-1, n), b=np.array(b), is_eq=np.array(is_eq, dtype=bool),
                    src=np.array(src, dtype=np.int32))

    def solve(self):
        if _ROOT not in sys.path:
            sys.path.insert(0, _ROOT)
        from oracle import mpc_oracle as mo
        qp = self.extract_qp()
        A, b, is_eq = qp["A"], qp["b"], qp["is_eq"]
        trivial = ~np.any(A != 0, axis=1)
        # rows without any variable (0 == 0 for stance legs in D, 0 <= 0 <= 150 for swing legs in F): must hold as they are
        if np.any(is_eq[trivial] & (b[trivial] != 0)) or np.any(~is_eq[trivial] & (b[trivial] < 0)):
            raise RuntimeError("casadi_eval: infeasible constant constraint")
        keep = np.where(~trivial)[0]
        # exact duplicates (bounded(-mu fz, fx, mu fz) and bounded(-mu fz, -fx, mu fz) state the same two rows)
        seen, uniq = {}, []
        for r in keep:
            key = (A[r].tobytes(), b[r].tobytes(), bool(is_eq[r]))
            if key not in seen:
                seen[key] = r; uniq.append(r)
        uniq = np.array(uniq, dtype=np.int64)
        C = A[uniq]
        lo = -b[uniq]
        hi = np.where(is_eq[uniq], -b[uniq], np.inf)
        u, info = mo.solve(qp["H"], 0.5 * qp["g"], C, lo, hi)
        qp.update(u=u.copy(), kkt=info, rows_used=uniq, cost_at_u=float(qp["c"] + qp["g"] @ u + u @ qp["H"] @ u),
                  params={k: v.copy() for k, v in self.values.items()})
        QP_LOG.append(qp)
        return OptiSol(self, u)


def install():
    """Register this module as `casadi` (before the reference is imported)."""
    m = types.ModuleType("casadi")
    names = ["Opti", "vertcat", "horzcat", "mtimes", "if_else", "cos", "sin", "tan", "transpose", "inv", "skew"]
    g = globals()
    for n in names:
        setattr(m, n, g[n])
    m.casadi = m
    m.__all__ = names + ["casadi"]
    m.__evaluating_standin__ = True
    sys.modules["casadi"] = m
    return m
