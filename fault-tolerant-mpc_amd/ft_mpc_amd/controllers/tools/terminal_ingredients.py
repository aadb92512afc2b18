"""Mirror of the LOADING half of ft_mpc/controllers/tools/terminal_ingredients.py (:451-474): reads
`config/terminal.yaml` -- the terminal cost expression and the 72 x 9 terminal polytope that the reference's
offline generator (same file, :21-426, out of scope) wrote -- WITHOUT `eval`, sympy or casadi.

The reference `eval`s the `cost` string (an `sp.lambdify((ep1..eo3), <expression>, modules=...)` call).  Here the
expression is parsed by a small recursive-descent parser for exactly the arithmetic the generator emits
(`+ - *`, `**`, parentheses, `Float('<digits>', precision=53)`, numbers, the nine error symbols) into
    cost(e) = sum_i c_i * prod_j e_j^a_ij                       (polynomial part; its degree-2 part is P)
            + sum_r c_r * (q_r(e) + eps_r)^p_r  + const         (the smoothed |.|^(4 p) terms, q_r polynomial)
so value, gradient and the quadratic weight P are available in closed form -- on the host (numpy) and, as plain
coefficient tables, to the kernels (ftmpc_config.term_* in include/ftmpc.h).  Anything outside that grammar raises.
"""
from __future__ import annotations

import json
import re
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

SYMBOLS = ("ep1", "ep2", "ep3", "ev1", "ev2", "ev3", "eo1", "eo2", "eo3")   # terminal_ingredients.py:461-469
DEFAULT_PATH = Path(__file__).resolve().parents[2] / "config" / "terminal.yaml"

_TOKEN = re.compile(r"\s*(?:(Float\(\s*'([^']+)'\s*,\s*precision\s*=\s*\d+\s*\))|(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+(?:[eE][-+]?\d+)?)"
                    r"|([A-Za-z_]\w*)|(\*\*|[-+*/()]))")


class _Poly(dict):
    """{exponent tuple (9 ints): coefficient}"""

    @staticmethod
    def const(c):
        return _Poly({(0,) * 9: float(c)}) if c != 0 else _Poly()

    @staticmethod
    def sym(i):
        e = [0] * 9
        e[i] = 1
        return _Poly({tuple(e): 1.0})

    def add(self, o, sign=1.0):
        r = _Poly(self)
        for k, v in o.items():
            r[k] = r.get(k, 0.0) + sign * v
            if r[k] == 0.0:
                del r[k]
        return r

    def mul(self, o):
        r = _Poly()
        for k1, v1 in self.items():
            for k2, v2 in o.items():
                k = tuple(a + b for a, b in zip(k1, k2))
                r[k] = r.get(k, 0.0) + v1 * v2
        return r

    def as_const(self):
        if not self:
            return 0.0
        if len(self) == 1 and (0,) * 9 in self:
            return self[(0,) * 9]
        return None


@dataclass
class _Expr:
    """poly + sum of coef * (inner poly)^power terms (power not a small non-negative integer)"""
    poly: _Poly = field(default_factory=_Poly)
    roots: list = field(default_factory=list)      # [(coef, inner _Poly, power)]

    def add(self, o, sign=1.0):
        return _Expr(self.poly.add(o.poly, sign), self.roots + [(sign * c, q, p) for c, q, p in o.roots])

    def mul(self, o):
        a, b = self, o
        if a.roots and b.roots:
            raise ValueError("terminal cost: product of two non-polynomial terms is outside the supported grammar")
        if b.roots:
            a, b = b, a
        if a.roots:
            s = b.poly.as_const()
            if s is None:
                raise ValueError("terminal cost: a non-polynomial term may only be scaled by a number")
            return _Expr(a.poly.mul(b.poly), [(c * s, q, p) for c, q, p in a.roots])
        return _Expr(a.poly.mul(b.poly), [])

    def power(self, o):
        p = o.poly.as_const() if not o.roots else None
        if p is None:
            raise ValueError("terminal cost: exponent must be a number")
        if self.roots:
            raise ValueError("terminal cost: power of a non-polynomial term is outside the supported grammar")
        if float(p).is_integer() and 0 <= p <= 8:
            r = _Poly.const(1.0)
            for _ in range(int(p)):
                r = r.mul(self.poly)
            return _Expr(r, [])
        return _Expr(_Poly(), [(1.0, self.poly, float(p))])


class _Parser:
    def __init__(self, text):
        self.toks = []
        pos = 0
        text = text.strip()
        while pos < len(text):
            m = _TOKEN.match(text, pos)
            if not m or m.end() == pos:
                raise ValueError(f"terminal cost: cannot tokenise at ...{text[pos:pos + 40]!r}")
            if m.group(1):
                self.toks.append(("num", float(m.group(2))))
            elif m.group(3):
                self.toks.append(("num", float(m.group(3))))
            elif m.group(4):
                self.toks.append(("id", m.group(4)))
            else:
                self.toks.append(("op", m.group(5)))
            pos = m.end()
        self.i = 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else ("end", None)

    def take(self):
        t = self.peek()
        self.i += 1
        return t

    def expr(self):
        sign = 1.0
        if self.peek() == ("op", "-"):
            self.take()
            sign = -1.0
        elif self.peek() == ("op", "+"):
            self.take()
        r = _Expr().add(self.term(), sign)
        while self.peek() in (("op", "+"), ("op", "-")):
            s = 1.0 if self.take()[1] == "+" else -1.0
            r = r.add(self.term(), s)
        return r

    def term(self):
        r = self.factor()
        while self.peek() in (("op", "*"), ("op", "/")):
            op = self.take()[1]
            f = self.factor()
            if op == "/":
                c = f.poly.as_const() if not f.roots else None
                if c is None or c == 0:
                    raise ValueError("terminal cost: division only by a non-zero number")
                f = _Expr(_Poly.const(1.0 / c))
            r = r.mul(f)
        return r

    def factor(self):
        if self.peek() == ("op", "-"):
            self.take()
            return _Expr(_Poly.const(-1.0)).mul(self.factor())
        b = self.base()
        if self.peek() == ("op", "**"):
            self.take()
            b = b.power(self.factor())
        return b

    def base(self):
        kind, val = self.take()
        if kind == "num":
            return _Expr(_Poly.const(val))
        if kind == "id":
            if val in SYMBOLS:
                return _Expr(_Poly.sym(SYMBOLS.index(val)))
            raise ValueError(f"terminal cost: unknown name {val!r} (only {SYMBOLS} and numbers are allowed; nothing is evaluated)")
        if (kind, val) == ("op", "("):
            r = self.expr()
            if self.take() != ("op", ")"):
                raise ValueError("terminal cost: missing ')'")
            return r
        raise ValueError(f"terminal cost: unexpected token {val!r}")


def _lambdify_body(src: str) -> str:
    """The expression inside `sp.lambdify((symbols...), <expression>[, modules=...])` (terminal_ingredients.py:454)."""
    m = re.match(r"\s*sp\.lambdify\(\s*\((.*?)\)\s*,(.*)\)\s*$", src, re.S)
    if not m:
        raise ValueError("terminal.yaml: `cost` is not an sp.lambdify((...), expr) string")
    names = tuple(s.strip() for s in m.group(1).split(","))
    if names != SYMBOLS:
        raise ValueError(f"terminal.yaml: cost arguments are {names}, expected {SYMBOLS}")
    body, depth, cut = m.group(2), 0, None
    for i, ch in enumerate(body):
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "," and depth == 0:
            cut = i
    if cut is not None and "modules" in body[cut:]:
        body = body[:cut]
    return body


class TerminalSet:
    """The two fields of the reference's MyPolytope the hot path reads (spiraling_mpc.py:199-202): A e <= b."""

    def __init__(self, A, b):
        self.A = np.asarray(A, float)
        self.b = np.asarray(b, float).reshape(-1, 1)       # the reference stores b as a column
        self.Nc = self.A.shape[0]

    def contains(self, e, tol=0.0):
        return bool(np.all(self.A @ np.asarray(e, float).reshape(-1) <= self.b.reshape(-1) + tol))


@dataclass
class TerminalIngredients:
    poly_coef: np.ndarray        # [K]      all polynomial terms (degree 2 included)
    poly_exp: np.ndarray         # [K, 9]
    root_coef: np.ndarray        # [R]
    root_exp: np.ndarray         # [R, 9]   inner monomial of each smoothed-absolute-value term
    root_eps: np.ndarray         # [R]
    root_pow: np.ndarray         # [R]
    const: float
    term_set: TerminalSet

    @property
    def P(self):
        """Quadratic weight: cost = e' P e + (terms of degree != 2)."""
        P = np.zeros((9, 9))
        for c, ex in zip(self.poly_coef, self.poly_exp):
            if ex.sum() != 2:
                continue
            idx = [i for i, a in enumerate(ex) for _ in range(int(a))]
            i, j = idx
            if i == j:
                P[i, i] += c
            else:
                P[i, j] += c / 2
                P[j, i] += c / 2
        return P

    def _nonquadratic_poly(self):
        keep = self.poly_exp.sum(axis=1) != 2
        return self.poly_coef[keep], self.poly_exp[keep]

    def cost(self, e, quadratic=True, nonquadratic=True):
        e = np.asarray(e, float).reshape(9)
        v = 0.0
        for c, ex in zip(self.poly_coef, self.poly_exp):
            if (ex.sum() == 2 and quadratic) or (ex.sum() != 2 and nonquadratic):
                v += c * np.prod(e ** ex)
        if nonquadratic:
            for c, ex, eps, p in zip(self.root_coef, self.root_exp, self.root_eps, self.root_pow):
                v += c * (np.prod(e ** ex) + eps) ** p
            v += self.const
        return float(v)

    def grad(self, e, quadratic=True, nonquadratic=True):
        e = np.asarray(e, float).reshape(9)
        g = np.zeros(9)

        def dmono(ex, i):      # d/de_i prod_j e_j^ex_j without dividing by e_i
            if ex[i] == 0:
                return 0.0
            r = ex[i] * e[i] ** (ex[i] - 1)
            for j in range(9):
                if j != i and ex[j]:
                    r *= e[j] ** ex[j]
            return r

        for c, ex in zip(self.poly_coef, self.poly_exp):
            if (ex.sum() == 2 and quadratic) or (ex.sum() != 2 and nonquadratic):
                for i in range(9):
                    g[i] += c * dmono(ex, i)
        if nonquadratic:
            for c, ex, eps, p in zip(self.root_coef, self.root_exp, self.root_eps, self.root_pow):
                base = np.prod(e ** ex) + eps
                for i in range(9):
                    g[i] += c * p * base ** (p - 1.0) * dmono(ex, i)
        return g

    def __call__(self, *e):
        """Same call shape as the reference's lambdified cost: cost(ep1, ..., eo3)  (spiraling_mpc.py:196)."""
        return self.cost(np.array(e, float).reshape(-1))

    def device_tables(self, max_poly=24, max_root=24):
        """Non-quadratic terms as fixed-size coefficient tables for ftmpc_config (the quadratic part travels as P)."""
        pc, pe = self._nonquadratic_poly()
        keep = pe.sum(axis=1) != 0
        const = self.const + float(pc[~keep].sum())
        pc, pe = pc[keep], pe[keep]
        if len(pc) > max_poly or len(self.root_coef) > max_root:
            raise ValueError("terminal cost has more non-quadratic terms than the device tables hold")
        return dict(poly_coef=pc, poly_exp=pe.astype(np.int32), root_coef=self.root_coef, root_exp=self.root_exp.astype(np.int32),
                    root_eps=self.root_eps, root_pow=self.root_pow, const=const)


def parse_terminal_cost(src: str):
    ex = _Parser(_lambdify_body(src))
    e = ex.expr()
    if ex.peek()[0] != "end":
        raise ValueError(f"terminal cost: trailing input at token {ex.peek()}")
    const = e.poly.pop((0,) * 9, 0.0)
    pc = np.array(list(e.poly.values()), float)
    pe = np.array(list(e.poly.keys()), int).reshape(-1, 9)
    rc, rx, reps, rp = [], [], [], []
    for c, q, p in e.roots:
        q = _Poly(q)
        eps = q.pop((0,) * 9, 0.0)
        if len(q) != 1 or list(q.values())[0] != 1.0:
            raise ValueError("terminal cost: a (q + eps)^p term must have a single monomial q")
        rc.append(c)
        rx.append(list(q.keys())[0])
        reps.append(eps)
        rp.append(p)
    return pc, pe, np.array(rc, float), np.array(rx, int).reshape(-1, 9), np.array(reps, float), np.array(rp, float), float(const)


def load_terminal(path=None) -> TerminalIngredients:
    import yaml
    doc = yaml.safe_load(open(path or DEFAULT_PATH))
    pc, pe, rc, rx, reps, rp, const = parse_terminal_cost(doc["cost"])
    ts = json.loads(doc["term_set"])
    return TerminalIngredients(pc, pe, rc, rx, reps, rp, const, TerminalSet(ts["A"], ts["b"]))


def load_terminal_ingredients(path=None):
    """Same return contract as the reference (terminal_ingredients.py:451-474): (cost callable of nine scalars,
    terminal set with .A [72, 9] / .b [72, 1] / .Nc)."""
    t = load_terminal(path)
    return t, t.term_set
