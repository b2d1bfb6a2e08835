"""Evaluator for the Lepton-style expression strings BLUES passes as
`alchemical_functions` (reference blues/simulation.py:654-659), e.g.
'min(1, (1/0.3)*abs(lambda-0.5))' and
'step(0.2-lambda) - 1/0.2*lambda*step(0.2-lambda) + 1/0.2*(lambda-0.8)*step(lambda-0.8)'.

OpenMM evaluates these inside the CustomIntegrator at every H step
(reference blues/integrators.py:226).  Here they are evaluated once on the
host into tables indexed by lambda_step (include/blues_engine.h,
BluesIntegratorDesc.lambda_sterics / lambda_electrostatics).
Only arithmetic and Lepton's function set is accepted (no attribute access,
no names other than the variables given).
"""
import ast
import math
import re

_FUNCS = {
    "min": min, "max": max, "abs": abs, "sqrt": math.sqrt, "exp": math.exp, "log": math.log,
    "sin": math.sin, "cos": math.cos, "tan": math.tan, "asin": math.asin, "acos": math.acos, "atan": math.atan,
    "sinh": math.sinh, "cosh": math.cosh, "tanh": math.tanh, "erf": math.erf, "erfc": math.erfc,
    "floor": math.floor, "ceil": math.ceil,
    "step": lambda x: 0.0 if x < 0 else 1.0,       # Lepton: 0 if x < 0, 1 otherwise
    "delta": lambda x: 1.0 if x == 0 else 0.0,
    "select": lambda x, y, z: z if x == 0 else y,
}
_BIN = {ast.Add: lambda a, b: a + b, ast.Sub: lambda a, b: a - b, ast.Mult: lambda a, b: a * b,
        ast.Div: lambda a, b: a / b, ast.Pow: lambda a, b: a ** b}


_COMPILED = {}   # expression -> evaluator (R chains of one protocol validate the same two strings: 0.13 ms of parsing each otherwise)


def compile_expression(expr):
    """Returns f(**variables) -> float for a Lepton-compatible expression string."""
    cached = _COMPILED.get(str(expr))
    if cached is not None:
        return cached
    f = _compile_expression(expr)
    if len(_COMPILED) < 256:
        _COMPILED[str(expr)] = f
    return f


def _compile_expression(expr):
    text = str(expr).replace("^", "**")
    text = re.sub(r"\blambda\b", "lambda_", text)  # `lambda` is a Python keyword
    tree = ast.parse(text, mode="eval")

    def ev(node, env):
        if isinstance(node, ast.Expression):
            return ev(node.body, env)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return float(node.value)
        if isinstance(node, ast.Name):
            if node.id in env:
                return env[node.id]
            raise ValueError("unknown variable %r in %r" % (node.id.rstrip("_"), expr))
        if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
            return _BIN[type(node.op)](ev(node.left, env), ev(node.right, env))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand, env)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _FUNCS and not node.keywords:
            return float(_FUNCS[node.func.id](*[ev(a, env) for a in node.args]))
        raise ValueError("unsupported construct in expression %r" % (expr,))

    def f(**variables):
        env = {("lambda_" if k == "lambda" else k): float(v) for k, v in variables.items()}
        return float(ev(tree, env))

    f(**{"lambda": 0.5})  # validate now, as OpenMM would at integrator construction
    return f


_TABLES = {}   # (expression, n) -> tuple of values: R chains of one protocol share the table (50 ms of tree walking each otherwise)
_TABLES_LOCK = __import__("threading").Lock()   # (chains are built from several host threads: bench.py --groups)


def tabulate(expr, n_lambda_steps):
    """[f(i / n_lambda_steps) for i in 0..n_lambda_steps]; lambda is formed exactly as the H step
    forms it, (lambda_step+1)/n_lambda_steps (reference blues/integrators.py:222)."""
    n = int(n_lambda_steps)
    key = (str(expr), n)
    with _TABLES_LOCK:
        table = _TABLES.get(key)
    if table is None:
        f = compile_expression(expr)
        table = (f(**{"lambda": 0.0}),) if n <= 0 else tuple(f(**{"lambda": i / n}) for i in range(n + 1))
        with _TABLES_LOCK:
            _TABLES[key] = table
            if len(_TABLES) > 64:      # (bounded: protocols are few)
                _TABLES.pop(next(iter(_TABLES)))
    return list(table)
