"""optistate_amd -- MI355X-native implementation of the OptiState Kalman+GRU hot path.

`Kalman_Filter` and `RNN` mirror the reference classes (kalman_filter/kalman_filter.py, gru/gru_model.py);
`Engine` is the batched interface over the C-ABI (include/optistate_hip.h).  Importing this package does
not need a GPU; creating an Engine / calling forward does, and fails loudly without one.
"""
from . import synth  # noqa: F401


def __getattr__(name):
    if name in ("Engine", "default_engine", "flatten_state_dict"):
        from . import engine
        return getattr(engine, name)
    if name == "Kalman_Filter":
        from .kalman_filter import Kalman_Filter
        return Kalman_Filter
    if name == "RNN":
        from .gru_model import RNN
        return RNN
    raise AttributeError(name)
