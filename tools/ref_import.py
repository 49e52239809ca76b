"""Import the reference's Kalman filter and GRU in THIS container (never on the GPU box).

The reference (`/root/reference`, read-only) is a script collection without packaging;
`kalman_filter/kalman_filter.py:3-5` imports `settings` and `misc.force_controller`, and the
latter does `from casadi import *` (`misc/force_controller.py:12`) only to build the convex-MPC
QP (`StanceController`, `misc/force_controller.py:15-225`), which is NOT on the hot path.
casadi is not installed here and there is no network, so an inert stand-in module is
registered under the name `casadi` *only to let the import and `Kalman_Filter.__init__`
succeed*.  Every function the golden vectors are generated from (`get_odom`,
`set_measurements`, `predict`, `update`, `next_state`, `rotation_matrix_body_world`)
is the reference's own unmodified code and touches no casadi symbol.

For the "next" row `estimate_state_mpc` the stand-in's
`Opti().solve().value(...)` can be primed with a preset (12, N) force matrix so that
`predict_mpc` runs end-to-end with externally supplied forces; only the QP solution
itself stays unpinned.
"""
import sys
import types

REF_ROOT = "/root/reference"


class _Inert:
    """Absorbs any use: callable, subscriptable, arithmetic, attribute access."""
    __array_ufunc__ = None          # ndarray (op) _Inert defers to _Inert.__r*__
    preset_forces = None            # class-level: what sol.value(controls) returns

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __getitem__(self, k):
        return _Inert()

    def __setitem__(self, k, v):
        pass

    def __iter__(self):
        return iter(())

    @property
    def T(self):
        return _Inert()

    def value(self, *_a, **_k):
        if _Inert.preset_forces is None:
            raise RuntimeError("casadi stand-in: no preset forces primed")
        return _Inert.preset_forces

    def solve(self, *a, **k):
        return self


def _binary(self, other):
    return _Inert()


for _n in ("add", "sub", "mul", "truediv", "matmul", "pow", "neg", "pos", "le", "ge", "lt", "gt", "eq"):
    setattr(_Inert, f"__{_n}__", _binary if _n not in ("neg", "pos") else (lambda self: _Inert()))
    if _n not in ("neg", "pos", "le", "ge", "lt", "gt", "eq"):
        setattr(_Inert, f"__r{_n}__", _binary)
_Inert.__hash__ = object.__hash__


def install_casadi_standin():
    if "casadi" in sys.modules:
        return sys.modules["casadi"]
    m = types.ModuleType("casadi")
    names = ["Opti", "vertcat", "horzcat", "mtimes", "if_else", "cos", "sin", "tan",
             "transpose", "inv", "skew", "MX", "SX", "DM", "sumsqr", "fabs", "sqrt"]
    for n in names:
        setattr(m, n, _Inert)
    m.casadi = m
    m.__all__ = names + ["casadi"]
    sys.modules["casadi"] = m
    return m


def prime_forces(f):
    """Set what the stand-in QP 'solution' returns (shape (12, N))."""
    _Inert.preset_forces = f


def import_reference():
    """Returns (Kalman_Filter, next_state, force_controller_module, INITIAL_PARAMS, RNN)."""
    install_casadi_standin()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
        sys.path.insert(0, REF_ROOT + "/gru")
    from kalman_filter.kalman_filter import Kalman_Filter
    import misc.force_controller as fc
    from settings import INITIAL_PARAMS
    from gru_model import RNN
    return Kalman_Filter, fc.next_state, fc, INITIAL_PARAMS, RNN
