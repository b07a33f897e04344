"""Shim: the slice of `ortools.linear_solver.pywraplp` that pymht/tracker.py:1167-1210 uses,
backed by scipy.optimize.milp (HiGHS, exact branch and bound).  OR-Tools/CBC is not installed in
the development container.  Used ONLY by oracle/gen_golden.py to run the reference as an oracle.

The hook `RECORDER` (a list, or None) receives every solved instance so the golden-vector
generator can dump ILP fixtures."""
import time
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds
from scipy.sparse import csr_matrix

RECORDER = None


class _Expr:
    __array_priority__ = 1000.0

    def __init__(self, terms=None):
        self.terms = terms if terms is not None else {}

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        out = dict(self.terms)
        for k, v in _as_expr(other).terms.items():
            out[k] = out.get(k, 0.0) + v
        return _Expr(out)

    __radd__ = __add__

    def __mul__(self, c):
        c = float(c)
        return _Expr({k: v * c for k, v in self.terms.items()})

    __rmul__ = __mul__

    def __le__(self, rhs):
        return _Constraint(self, -np.inf, float(rhs))

    def __ge__(self, rhs):
        return _Constraint(self, float(rhs), np.inf)

    def __eq__(self, rhs):
        return _Constraint(self, float(rhs), float(rhs))

    __hash__ = None


class _Var(_Expr):
    def __init__(self, index, name):
        _Expr.__init__(self, {index: 1.0})
        self.index = index
        self.name = name
        self._value = 0.0

    def solution_value(self):
        return self._value

    def __hash__(self):
        return hash(self.index)


def _as_expr(x):
    if isinstance(x, _Expr):
        return x
    raise TypeError(type(x))


class _Constraint:
    def __init__(self, expr, lo, hi):
        self.expr, self.lo, self.hi = expr, lo, hi


class Solver:
    CBC_MIXED_INTEGER_PROGRAMMING = 1
    OPTIMAL = 0
    INFEASIBLE = 2
    ABNORMAL = 4

    def __init__(self, name, kind):
        self.vars = []
        self.cons = []
        self.obj = None
        self._wall = 0.0

    def BoolVar(self, name):
        v = _Var(len(self.vars), name)
        self.vars.append(v)
        return v

    def Sum(self, items):
        out = {}
        for it in items:
            for k, v in it.terms.items():
                out[k] = out.get(k, 0.0) + v
        return _Expr(out)

    def Minimize(self, expr):
        self.obj = expr

    def Add(self, constraint):
        self.cons.append(constraint)
        return constraint

    def WallTime(self):
        return self._wall * 1000.0

    def Solve(self):
        t0 = time.time()
        n = len(self.vars)
        c = np.zeros(n)
        for k, v in self.obj.terms.items():
            c[k] = v
        rows, cols, vals, lo, hi = [], [], [], [], []
        for r, con in enumerate(self.cons):
            for k, v in con.expr.terms.items():
                rows.append(r)
                cols.append(k)
                vals.append(v)
            lo.append(con.lo)
            hi.append(con.hi)
        A = csr_matrix((vals, (rows, cols)), shape=(len(self.cons), n))
        res = milp(c, constraints=LinearConstraint(A, lo, hi), integrality=np.ones(n),
                   bounds=Bounds(0, 1), options={"mip_rel_gap": 0.0})
        self._wall = time.time() - t0
        if res.status != 0 or res.x is None:
            return Solver.INFEASIBLE
        x = np.round(res.x)
        for v in self.vars:
            v._value = float(x[v.index])
        if RECORDER is not None:
            RECORDER.append(dict(c=c, A=A, lo=np.array(lo), hi=np.array(hi), x=x.copy(), fun=float(res.fun)))
        return Solver.OPTIMAL
