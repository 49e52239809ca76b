"""Drop-in `Kalman_Filter` with the reference's Python surface (kalman_filter/kalman_filter.py:7-193).

Same constructor, attributes (x, z, H, P, Q, R, P_trace, K_gain, K, x_model, f, dt) and methods
(get_odom, set_measurements, predict, update, estimate_state_mpc, rotation_matrix_body_world); arrays are
float64 column vectors as in the reference.  Every method runs the HIP kernels through the C-ABI with B = 1
(latency-bound by design: this class exists so that the reference's scripts keep working; throughput
comes from `Engine.kf_run`, which steps many trajectories per launch).  Values carry float32 precision.

Differences from the reference, all deliberate (SURVEY.md section 5 "race detection" and appendix):
  * no global state: x/P/Q/R are per-instance copies (the reference aliases class attributes of
    settings.INITIAL_PARAMS, kalman_filter/kalman_filter.py:10,27-29);
  * get_odom is defined for 0 and 4 stance legs (the reference raises ValueError there);
  * the convex-MPC QP inside predict_mpc (misc/force_controller.py:70-162, casadi/qpOASES in the reference) is solved
    by the library's own exact active-set kernel (os_mpc_solve); pass f= to replay logged forces instead.
"""
import numpy as np
import torch

from .engine import default_engine, _ptr, OS_KF_DENSE_FD, OS_KF_P_FLOAT64
from . import synth


class Kalman_Filter:
    def __init__(self, device=0):
        self._eng = default_engine(device)
        self._dev = self._eng.device
        self.x = synth.X0.reshape(12, 1).copy()                       # kalman_filter.py:10 / settings.py:25
        self.z = np.zeros((10, 1))
        self.H = np.zeros((10, 12), dtype=np.int64)
        for a, s in enumerate([0, 1, 2, 5, 6, 7, 8, 9, 10, 11]):     # kalman_filter.py:15-24
            self.H[a, s] = 1
        self.P = synth.Q_DEFAULT.copy()                               # settings.py:31  P = Q
        self.Q = synth.Q_DEFAULT.copy()
        self.R = synth.R_DEFAULT.copy()
        self.P_trace = float(np.trace(self.P))
        self.K_gain = 0.0
        self.K = np.zeros((12, 10))
        self.m = 8.8                                                   # settings.py:11
        self.inertia_rot = np.diag([55303643.08 / 1e9, 60119440.34 / 1e9, 105304340.05 / 1e9])   # settings.py:20-23
        self.g = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -9.81]).reshape(12, 1)               # kalman_filter.py:56
        self.dt = 0.01
        self.x_model = self.x.copy()
        self.f = np.zeros((12, 5))
        # predict_mpc leaves a covariance that float32 cannot carry to the next update within the 1e-4 bar (element-wise
        # exp(dt F), kalman_filter.py:157): the split predict_mpc() -> update() sequence therefore keeps P in float64 on
        # the device side too (OS_KF_P_FLOAT64), like the batched kernel does inside one launch
        self._p64_pending = False

    # -- helpers --
    # Every call stages ALL its inputs in one host array -> one host-to-device copy, and reads ALL its outputs back with
    # one device-to-host copy (B = 1 is pure latency: ~20 separate transfers per step cost more than the kernels).
    def _up(self, a, n):
        return torch.as_tensor(np.asarray(a, dtype=np.float32).reshape(n, 1)).to(self._dev)

    def _stage(self, *parts):
        host = np.concatenate([np.asarray(a, dtype=np.float32).reshape(-1) for a in parts])
        buf = torch.from_numpy(host).to(self._dev)
        offs, o = [], 0
        for a in parts:
            n = int(np.asarray(a).size)
            offs.append((o, n)); o += n
        return buf, offs

    @staticmethod
    def _seg(buf, off):
        return buf[off[0]:off[0] + off[1]]

    def _sync_noise(self):
        self._eng.set_noise(self.Q, self.R)

    def rotation_matrix_body_world(self, thx, thy, thz):
        # closed form of Rz Ry Rx (kalman_filter.py:184-193); host helper, not on the hot path
        cx, sx, cy, sy, cz, sz = np.cos(thx), np.sin(thx), np.cos(thy), np.sin(thy), np.cos(thz), np.sin(thz)
        return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                         [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                         [-sy, cy * sx, cy * cx]], dtype=np.float64).reshape(3, 3)

    @staticmethod
    def _pack_contact(contact_cur):
        c = np.asarray(contact_cur).reshape(4)
        packed = np.array([sum((int(c[k]) & 0xff) << (8 * k) for k in range(4))], dtype=np.int32)
        return packed.view(np.float32)                       # carried bit-for-bit inside the float staging buffer

    def get_odom(self, p_cur, dp_cur, contact_cur, imu):
        e = self._eng
        buf, o = self._stage(np.asarray(p_cur).reshape(-1)[:12], np.asarray(dp_cur).reshape(-1)[:12],
                             np.asarray(imu).reshape(-1)[:6], self._pack_contact(contact_cur), np.zeros(10))
        sg = lambda i: _ptr(self._seg(buf, o[i]))
        e._check(e.lib.os_kf_odom(e._h, 1, sg(0), sg(1), sg(3), sg(2), sg(4), e._stream()), "os_kf_odom")
        zz = self._seg(buf, o[4]).cpu().numpy().astype(np.float64)
        return np.array([zz[3], zz[7], zz[8], zz[9]]).reshape(4, 1)

    def set_measurements(self, imu, odom):
        # pure data movement (kalman_filter.py:108-117)
        imu = np.asarray(imu, dtype=np.float64).reshape(-1, 1)
        odom = np.asarray(odom, dtype=np.float64).reshape(4, 1)
        self.z[0:3] = imu[0:3]
        self.z[4:7] = imu[3:6]
        self.z[3] = odom[0]
        self.z[7:10] = odom[1:]

    @staticmethod
    def _f64_words(a, n):
        """float64 array carried bit-for-bit inside the float32 staging buffer (2 words per value; staged FIRST so that
        the device address is 8-byte aligned)."""
        return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(n)).view(np.float32)

    def _predict(self, p, f, body_ref=None):
        e = self._eng
        self._sync_noise()
        dense = body_ref is not None
        br = np.zeros(12) if body_ref is None else np.asarray(body_ref).reshape(-1)[:12]
        Pw = self._f64_words(self.P, 144) if dense else np.asarray(self.P, dtype=np.float64).reshape(144)
        buf, o = self._stage(Pw, np.asarray(p).reshape(-1)[:12], np.asarray(f).reshape(-1)[:12], self.x, np.zeros(1), br)
        sg = lambda i: _ptr(self._seg(buf, o[i]))
        flags = (OS_KF_DENSE_FD | OS_KF_P_FLOAT64) if dense else 0
        e._check(e.lib.os_kf_predict(e._h, 1, sg(1), sg(2), sg(5) if dense else None, sg(3), sg(0), sg(4),
                                     flags, e._stream()), "os_kf_predict")
        hb = buf.cpu().numpy()
        with np.errstate(invalid="ignore"):       # the float64 words of P, read as float32, may look like NaNs
            h = hb.astype(np.float64)
        self.x = h[o[3][0]:o[3][0] + 12].reshape(12, 1)
        if dense:
            self.P = hb[o[0][0]:o[0][0] + 288].view(np.float64).reshape(12, 12).copy()
        else:
            self.P = h[o[0][0]:o[0][0] + 144].reshape(12, 12)
        self._p64_pending = dense
        # the reference rotates the caller's p in place (misc/force_controller.py:274-277)
        if isinstance(p, np.ndarray) and p.flags.writeable:
            p[...] = h[o[1][0]:o[1][0] + 12].astype(p.dtype).reshape(p.shape)
        self.x_model = self.x.copy()
        self.P_trace = float(h[o[4][0]])

    def predict(self, p, f):
        """kalman_filter.py:119-138: p, f (12,1); p is rotated to the world frame in place."""
        self._predict(p, f)

    def update(self):
        """kalman_filter.py:164-174.  After predict_mpc the covariance stays in float64 through this call (see __init__)."""
        e = self._eng
        self._sync_noise()
        f64 = self._p64_pending
        self._p64_pending = False
        if f64:
            buf, o = self._stage(self._f64_words(self.P, 144), self._f64_words(np.zeros(120), 120), self.z, self.x, np.zeros(1),
                                 np.zeros(1), np.zeros(1))
        else:
            buf, o = self._stage(np.asarray(self.P, dtype=np.float64).reshape(144), np.zeros(120), self.z, self.x, np.zeros(1),
                                 np.zeros(1), np.zeros(1))
        sg = lambda i: _ptr(self._seg(buf, o[i]))
        st = self._seg(buf, o[6]).view(torch.int32)
        e._check(e.lib.os_kf_update(e._h, 1, sg(2), sg(3), sg(0), sg(1), sg(4), sg(5), _ptr(st), OS_KF_P_FLOAT64 if f64 else 0,
                                    e._stream()), "os_kf_update")
        hb = buf.cpu().numpy()
        status = int(hb[o[6][0]:o[6][0] + 1].view(np.int32)[0])
        if status & 1:
            # the reference's np.linalg.inv raises here (kalman_filter.py:168)
            raise np.linalg.LinAlgError("Singular matrix")
        with np.errstate(invalid="ignore"):
            h = hb.astype(np.float64)
        self.x = h[o[3][0]:o[3][0] + 12].reshape(12, 1)
        if f64:
            self.P = hb[o[0][0]:o[0][0] + 288].view(np.float64).reshape(12, 12).copy()
            self.K = hb[o[1][0]:o[1][0] + 240].view(np.float64).reshape(12, 10).copy()
        else:
            self.P = h[o[0][0]:o[0][0] + 144].reshape(12, 12)
            self.K = h[o[1][0]:o[1][0] + 120].reshape(12, 10)
        self.P_trace = float(h[o[4][0]])
        self.K_gain = float(h[o[5][0]])

    def skew(self, x):
        # kalman_filter.py:195-198
        return np.array([[0, -x[2][0], x[1][0]], [x[2][0], 0, -x[0][0]], [-x[1][0], x[0][0], 0]])

    def solve_mpc(self, p, body_ref, cur_contact):
        """The stance controller's QP (misc/force_controller.py:70-162, parameters as kalman_filter.py:141-149 sets them)
        solved on the GPU (os_mpc_solve); returns the (12, N) control matrix the reference keeps in self.f."""
        e = self._eng
        col = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64).reshape(-1)[:12].astype(np.float32)).reshape(12, 1).to(e.device)
        c = torch.as_tensor(self._pack_contact(cur_contact).view(np.int32)).to(e.device)
        r = e.mpc_solve(col(self.x), col(body_ref), col(p), c, want_all=True)
        if int(r["status"].cpu()[0]) & 4:
            raise RuntimeError("convex MPC: active-set iteration cap reached")       # qpOASES would report a failed solve
        return r["u"].cpu().numpy().astype(np.float64).reshape(5, 12).T.copy()

    def predict_mpc(self, p, body_ref, cur_contact, f=None):
        """kalman_filter.py:140-162.  f (12,) or (12,N), column 0 used: forces from a log; None: solve the QP here."""
        self.f = self.solve_mpc(p, body_ref, cur_contact) if f is None else np.asarray(f, dtype=np.float64).reshape(12, -1)
        self._predict(p, self.f[:, 0], body_ref=body_ref)

    def estimate_state_mpc(self, imu, p, dp, body_ref, contact, f=None):
        """kalman_filter.py:176-182.  Runs odometry + predict_mpc + update as ONE T = 1 launch of the batched kernel, which
        carries the predict_mpc covariance in float64 between the predict and the update (its element-wise exp(dt F) makes P
        ill-conditioned for float32 there); attributes x, P, z, x_model is not split out, P_trace, K_gain and f are set; `K`
        itself is only produced by update()."""
        e = self._eng
        self._sync_noise()
        # f = None: the convex MPC of kalman_filter.py:141-152 is solved on the GPU (os_mpc_solve); otherwise forces from a log
        self.f = self.solve_mpc(p, body_ref, contact) if f is None else np.asarray(f, dtype=np.float64).reshape(12, -1)
        odom = self.get_odom(p, dp, contact, imu)
        self.set_measurements(imu, odom)
        buf, o = self._stage(np.asarray(p).reshape(-1)[:12], self.f[:, 0], np.asarray(dp).reshape(-1)[:12],
                             np.asarray(imu).reshape(-1)[:6], self._pack_contact(contact), np.asarray(body_ref).reshape(-1)[:12],
                             self.x, np.asarray(self.P, dtype=np.float64).reshape(144), np.zeros(12), np.zeros(12), np.zeros(1),
                             np.zeros(1), np.zeros(1))
        sg = lambda i: _ptr(self._seg(buf, o[i]))
        st = self._seg(buf, o[12]).view(torch.int32)
        e._check(e.lib.os_kf_run(e._h, 1, 1, sg(0), sg(1), sg(2), sg(3), sg(4), sg(5), sg(6), sg(7), sg(8), sg(9), sg(10), sg(11),
                                 _ptr(st), OS_KF_DENSE_FD, e._stream()), "os_kf_run")
        hb = buf.cpu().numpy()
        if int(hb[o[12][0]:o[12][0] + 1].view(np.int32)[0]) & 1:
            raise np.linalg.LinAlgError("Singular matrix")
        h = hb.astype(np.float64)
        self.x = h[o[6][0]:o[6][0] + 12].reshape(12, 1)
        self.P = h[o[7][0]:o[7][0] + 144].reshape(12, 12)
        self.P_trace = float(h[o[10][0]])
        self.K_gain = float(h[o[11][0]])
        if isinstance(p, np.ndarray) and p.flags.writeable:            # next_state rotates the caller's p in place
            p[...] = h[o[9][0]:o[9][0] + 12].astype(p.dtype).reshape(p.shape)
        return self.x
