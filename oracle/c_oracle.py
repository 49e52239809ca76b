"""ctypes bindings for oracle/liboracle.so (kf_oracle.c + gru_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DT = 0.01                      # settings.py:5
MASS = 8.8                     # settings.py:11
INERTIA = np.array([55303643.08 / 1e9, 60119440.34 / 1e9, 105304340.05 / 1e9])  # settings.py:20-23
GZ = -9.81                     # kalman_filter/kalman_filter.py:56


def build(force=False):
    if os.environ.get("ORACLE_LIB"):          # e.g. the sanitizer build (make -C oracle asan)
        return os.environ["ORACLE_LIB"]
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, s) for s in ("kf_oracle.c", "gru_oracle.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ok_gru_param_count.restype = C.c_size_t
        _LIB.ok_gru_train_loss.restype = C.c_double
        _LIB.ok_kf_run.restype = C.c_int
        _LIB.ok_update.restype = C.c_int
        _LIB.ok_max_threads.restype = C.c_int
        _LIB.ok_set_threads(1)            # default: the scalar port; the batch entry points can be split over threads
    return _LIB


def set_threads(n):
    """Threads for kf_run_batch / gru_forward (independent trajectories).  Returns the count actually set."""
    n = max(1, min(int(n), max_threads()))
    lib().ok_set_threads(n)
    return n


def max_threads():
    return int(lib().ok_max_threads())


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rotation(thx, thy, thz):
    R = np.zeros(9)
    lib().ok_rotation(C.c_double(thx), C.c_double(thy), C.c_double(thz), _d(R))
    return R.reshape(3, 3)


def get_odom(p, dp, contact, imu):
    p, dp, imu = _c64(p).ravel(), _c64(dp).ravel(), _c64(imu).ravel()
    c = np.ascontiguousarray(contact, dtype=np.uint8).ravel()
    od = np.zeros(4)
    lib().ok_get_odom(_d(p), _d(dp), c.ctypes.data_as(C.POINTER(C.c_uint8)), _d(imu), _d(od))
    return od


def next_state(x, p, f, dt=DT, mass=MASS, inertia=INERTIA, gz=GZ):
    """Returns (x_next, p_rotated)."""
    x, f = _c64(x).ravel(), _c64(f).ravel()
    p = _c64(p).ravel().copy()
    xn = np.zeros(12)
    ine = _c64(inertia)
    lib().ok_next_state(_d(x), _d(p), _d(f), C.c_double(dt), C.c_double(mass), _d(ine), C.c_double(gz), _d(xn))
    return xn, p


def update(x, P, z, R):
    x, P = _c64(x).ravel().copy(), _c64(P).copy()
    z, R = _c64(z).ravel(), _c64(R)
    K = np.zeros((12, 10))
    kg = C.c_double(0)
    st = lib().ok_update(_d(x), _d(P), _d(z), _d(R), _d(K), C.byref(kg))
    return x, P, K, kg.value, st


def kf_run_batch(p, f, dp, imu, contact, x0, P0, Q, R, body_ref=None, mode=0,
                 dt=DT, mass=MASS, inertia=INERTIA, gz=GZ, aux=True):
    """Inputs [B][T][field] (any float dtype; promoted to float64), contact uint8 [B][T][4].
    Returns dict with x [B][T][12], x_prior, p_rot, P_trace, K_gain [B][T], P_final [B][12][12],
    x_final [B][12], status [B]."""
    p, f, dp, imu = _c64(p), _c64(f), _c64(dp), _c64(imu)
    B, T = p.shape[0], p.shape[1]
    contact = np.ascontiguousarray(contact, dtype=np.uint8)
    x = _c64(x0).reshape(B, 12).copy()
    P = _c64(P0).reshape(B, 144).copy()
    Q, R, ine = _c64(Q), _c64(R), _c64(inertia)
    br = None if body_ref is None else _c64(body_ref)
    xo = np.zeros((B, T, 12)); xp = np.zeros((B, T, 12)) if aux else None
    pr = np.zeros((B, T, 12)) if aux else None
    pt = np.zeros((B, T)) if aux else None
    kg = np.zeros((B, T)) if aux else None
    st = np.zeros(B, dtype=np.int32)
    lib().ok_kf_run_batch(C.c_int(B), C.c_int(T), _d(p), _d(f), _d(dp), _d(imu),
                          contact.ctypes.data_as(C.POINTER(C.c_uint8)), _d(br), C.c_int(mode),
                          _d(x), _d(P), _d(Q), _d(R), C.c_double(dt), C.c_double(mass), _d(ine), C.c_double(gz),
                          _d(xo), _d(xp), _d(pr), _d(pt), _d(kg), st.ctypes.data_as(C.POINTER(C.c_int)))
    return dict(x=xo, x_prior=xp, p_rot=pr, P_trace=pt, K_gain=kg, P_final=P.reshape(B, 12, 12),
                x_final=x, status=st)


def gru_param_count(I, H, L, Cc):
    return int(lib().ok_gru_param_count(C.c_int(I), C.c_int(H), C.c_int(L), C.c_int(Cc)))


def gru_forward(x, w_flat, I, H, L, Cc, use_sigmoid=True, want_seq=False):
    """x [B][T][I]; w_flat in the flat layout of gru_oracle.c.  Returns (out [B][C], hlast [L][B][H], seq|None)."""
    x = _c64(x)
    B, T = x.shape[0], x.shape[1]
    w = _c64(w_flat).ravel()
    assert w.size == gru_param_count(I, H, L, Cc), (w.size, gru_param_count(I, H, L, Cc))
    out = np.zeros((B, Cc)); hl = np.zeros((L, B, H))
    seq = np.zeros((B, T, H)) if want_seq else None
    lib().ok_gru_forward(C.c_int(B), C.c_int(T), C.c_int(I), C.c_int(H), C.c_int(L), C.c_int(Cc), _d(x), _d(w),
                         C.c_int(1 if use_sigmoid else 0), _d(out), _d(hl), _d(seq))
    return out, hl, seq


def gru_train_loss(out, y):
    out, y = _c64(out), _c64(y)
    tgt = np.zeros_like(out)
    loss = lib().ok_gru_train_loss(C.c_int(out.shape[0]), _d(out), _d(y), _d(tgt))
    return float(loss), tgt


def feature_row(x_post, accel, f, p_world, dp, imu, minv=None, maxv=None):
    row = np.zeros(60)
    a = [_c64(v).ravel() for v in (x_post, accel, f, p_world, dp, imu)]
    mn = None if minv is None else _c64(minv); mx = None if maxv is None else _c64(maxv)
    lib().ok_feature_row(*[_d(v) for v in a], _d(mn), _d(mx), _d(row))
    return row


def flatten_state_dict(sd, L):
    """torch state_dict (reference keys gru.weight_ih_l{k}, ..., fc.weight, fc.bias) -> flat float64 vector."""
    parts = []
    for l in range(L):
        for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            parts.append(np.asarray(sd[f"gru.{k}_l{l}"].detach().cpu().numpy(), dtype=np.float64).ravel())
    parts.append(np.asarray(sd["fc.weight"].detach().cpu().numpy(), dtype=np.float64).ravel())
    parts.append(np.asarray(sd["fc.bias"].detach().cpu().numpy(), dtype=np.float64).ravel())
    return np.concatenate(parts)
