"""Drop-in `Kalman_Filter` with the reference's Python surface (kalman_filter/kalman_filter.py:7-193).

Same constructor, attributes (x, z, H, P, Q, R, P_trace, K_gain, K, x_model, f, dt) and methods
(get_odom, set_measurements, predict, update, estimate_state_mpc, rotation_matrix_body_world); arrays are
float64 column vectors as in the reference.  Every method is ONE `os_kf_step` call = one kernel launch + one stream
synchronise on host-resident float64 arrays (the kernel reads / writes a pinned, device-mapped staging block: no memcpy),
all arithmetic in float64 on one wavefront; `step()` runs the whole caller-loop body (odometry, measurement vector,
predict, update) in one launch.  Latency-bound by design: this class exists so that the reference's scripts keep working;
throughput comes from `Engine.kf_run`, which steps many trajectories per launch in float32.

Differences from the reference, all deliberate (SURVEY.md section 5 "race detection" and appendix):
  * no global state: x/P/Q/R are per-instance copies (the reference aliases class attributes of
    settings.INITIAL_PARAMS, kalman_filter/kalman_filter.py:10,27-29);
  * get_odom is defined for 0 and 4 stance legs (the reference raises ValueError there);
  * the convex-MPC QP inside predict_mpc (misc/force_controller.py:70-162, casadi/qpOASES in the reference) is solved
    by the library's own exact active-set kernel (os_mpc_solve); pass f= to replay logged forces instead;
  * update() raises numpy.linalg.LinAlgError when S = H P H^T + R is not POSITIVE DEFINITE (status bit 0 of os_kf_step: a
    LU pivot <= 0 or non-finite).  The reference's np.linalg.inv (kalman_filter.py:168) raises only for an exactly
    singular S and would carry an indefinite, invertible S through (reachable only after predict_mpc's element-wise
    exp(dt F) has destroyed P's definiteness): there this class raises where the reference continues with a meaningless
    gain.  A positive definite S -- every case a covariance filter is defined for -- takes the same path in both;
  * threads: the reference is single-threaded.  Here every instance stages its step through the engine context's one
    pinned block, so os_kf_step calls on one engine are serialised by a per-engine lock (different engines / GPUs run
    concurrently).
"""
import ctypes as C
import threading

import numpy as np
import torch

from .engine import default_engine
from ._capi import OS_STEP_ODOM, OS_STEP_PREDICT, OS_STEP_UPDATE, OS_STEP_DENSE_FD, OS_STEP_MPC
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
        self._buffers()

    # -- helpers --
    # Every method is ONE os_kf_step call: the library copies the float64 arrays into its pinned, device-mapped staging
    # block, one wavefront does the step in float64, one stream synchronise later the results are back in host memory
    # (no torch tensors, no hipMemcpy: B = 1 is pure latency).
    _MODEL = np.array([0.01, 8.8, 55303643.08 / 1e9, 60119440.34 / 1e9, 105304340.05 / 1e9, -9.81])   # settings.py:5-23, :56

    def _buffers(self):
        """Per-instance staging arrays and their ctypes pointers, made once: a step then costs a few small copies and ONE
        foreign call (at B = 1 the Python side is as much of the latency as the launch)."""
        dp = C.POINTER(C.c_double)
        b = {k: np.zeros(n) for k, n in (("x", 12), ("P", 144), ("z", 10), ("p", 12), ("f", 12), ("dp", 12), ("imu", 6), ("br", 12),
                                          ("Q", 144), ("R", 100), ("prot", 12), ("xm", 12), ("K", 120), ("pt", 1), ("kg", 1), ("fall", 60))}
        b["c"] = np.zeros(4, dtype=np.uint8)
        b["st"] = np.zeros(1, dtype=np.int32)
        b["it"] = np.zeros(1, dtype=np.int32)
        b["model"] = self._MODEL.copy()
        ptr = {k: v.ctypes.data_as(dp) for k, v in b.items() if v.dtype == np.float64}
        ptr["c"] = b["c"].ctypes.data_as(C.POINTER(C.c_uint8))
        ptr["st"] = b["st"].ctypes.data_as(C.POINTER(C.c_int32))
        ptr["it"] = b["it"].ctypes.data_as(C.POINTER(C.c_int32))
        self._b, self._ptr = b, ptr

    def _step(self, what, p=None, f=None, dp=None, imu=None, contact=None, body_ref=None, want_K=False):
        """One launch of the step kernel on self.x, self.P, self.z (read and written as `what` says); returns the
        world-rotated p (a view of a staging array: copy it if you keep it)."""
        e = self._eng
        b, q = self._b, self._ptr
        b["x"][:] = np.asarray(self.x, dtype=np.float64).reshape(-1)
        b["P"][:] = np.asarray(self.P, dtype=np.float64).reshape(-1)
        b["Q"][:] = np.asarray(self.Q, dtype=np.float64).reshape(-1)
        b["R"][:] = np.asarray(self.R, dtype=np.float64).reshape(-1)
        if not what & OS_STEP_ODOM:
            b["z"][:] = np.asarray(self.z, dtype=np.float64).reshape(-1)
        for key, v, n in (("p", p, 12), ("f", f, 12), ("dp", dp, 12), ("imu", imu, 6), ("br", body_ref, 12)):
            if v is not None:
                b[key][:] = np.asarray(v, dtype=np.float64).reshape(-1)[:n]
        if contact is not None:
            b["c"][:] = np.asarray(contact).reshape(-1)[:4]
        lock = e.__dict__.setdefault("_kf_step_lock", threading.Lock())     # the context's staging block is shared by every instance
        with lock:                                     # os_kf_step returns after its stream synchronise: results are in b[...] here
            if what & OS_STEP_MPC:                     # the QP in front of the step, one call (os_kf_step_mpc)
                rc = e.lib.os_kf_step_mpc(e._h, what, q["model"], q["p"], q["dp"] if dp is not None else None, q["imu"] if imu is not None else None,
                                          q["c"], q["br"], q["Q"], q["R"], q["x"], q["P"], q["z"], q["prot"], q["xm"],
                                          q["K"] if want_K else None, q["pt"], q["kg"], q["fall"], q["it"], q["st"], e._stream())
            else:
                rc = e.lib.os_kf_step(e._h, what, q["model"], q["p"] if p is not None else None, q["f"] if f is not None else None,
                                      q["dp"] if dp is not None else None, q["imu"] if imu is not None else None,
                                      q["c"] if contact is not None else None, q["br"] if body_ref is not None else None, q["Q"], q["R"],
                                      q["x"], q["P"], q["z"], q["prot"], q["xm"], q["K"] if want_K else None, q["pt"], q["kg"], q["st"],
                                      e._stream())
        if rc:
            e._check(rc, "os_kf_step")
        if what & OS_STEP_MPC:
            if b["st"][0] & 4:
                raise RuntimeError("convex MPC: active-set iteration cap reached")       # qpOASES would report a failed solve
            self.f = b["fall"].reshape(12, 5).copy()                  # the (12, N) control matrix of kalman_filter.py:150-152
        if what & OS_STEP_ODOM:
            self.z = b["z"].reshape(10, 1).copy()
        if what & OS_STEP_PREDICT:
            self.x = b["x"].reshape(12, 1).copy(); self.P = b["P"].reshape(12, 12).copy()
            self.x_model = b["xm"].reshape(12, 1).copy()
            if not (what & OS_STEP_DENSE_FD):
                self.P_trace = float(b["pt"][0])                      # predict_mpc leaves P_trace alone (kalman_filter.py:140-162)
        if what & OS_STEP_UPDATE:
            if b["st"][0] & 1:
                # the reference's np.linalg.inv raises here (kalman_filter.py:168)
                raise np.linalg.LinAlgError("Singular matrix")
            self.x = b["x"].reshape(12, 1).copy(); self.P = b["P"].reshape(12, 12).copy()
            if want_K:
                self.K = b["K"].reshape(12, 10).copy()
            self.P_trace = float(b["pt"][0]); self.K_gain = float(b["kg"][0])
        return b["prot"]

    def rotation_matrix_body_world(self, thx, thy, thz):
        # closed form of Rz Ry Rx (kalman_filter.py:184-193); host helper, not on the hot path
        cx, sx, cy, sy, cz, sz = np.cos(thx), np.sin(thx), np.cos(thy), np.sin(thy), np.cos(thz), np.sin(thz)
        return np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx],
                         [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                         [-sy, cy * sx, cy * cx]], dtype=np.float64).reshape(3, 3)

    @staticmethod
    def _pack_contact(contact_cur):
        c = np.asarray(contact_cur).reshape(4)
        return np.array([sum((int(c[k]) & 0xff) << (8 * k) for k in range(4))], dtype=np.int32)

    @staticmethod
    def _rotate_in_place(p, p_rot):
        # the reference rotates the caller's p in place (misc/force_controller.py:274-277)
        if isinstance(p, np.ndarray) and p.flags.writeable:
            p[...] = p_rot.astype(p.dtype).reshape(p.shape)

    def get_odom(self, p_cur, dp_cur, contact_cur, imu):
        """kalman_filter.py:79-105.  Returns odom (4,1) = [z, v_world]; self.z is NOT touched (set_measurements does that)."""
        keep = self.z
        self._step(OS_STEP_ODOM, p=p_cur, dp=dp_cur, imu=imu, contact=contact_cur)
        zz, self.z = self.z, keep
        return np.array([zz[3, 0], zz[7, 0], zz[8, 0], zz[9, 0]]).reshape(4, 1)

    def set_measurements(self, imu, odom):
        # pure data movement (kalman_filter.py:108-117)
        imu = np.asarray(imu, dtype=np.float64).reshape(-1, 1)
        odom = np.asarray(odom, dtype=np.float64).reshape(4, 1)
        self.z[0:3] = imu[0:3]
        self.z[4:7] = imu[3:6]
        self.z[3] = odom[0]
        self.z[7:10] = odom[1:]

    def predict(self, p, f):
        """kalman_filter.py:119-138: p, f (12,1); p is rotated to the world frame in place."""
        self._rotate_in_place(p, self._step(OS_STEP_PREDICT, p=p, f=f))

    def update(self):
        """kalman_filter.py:164-174."""
        self._step(OS_STEP_UPDATE, want_K=True)

    def step(self, p, f, dp, imu, contact, want_K=False):
        """The caller loop body `odom = get_odom(p, dp, contact, imu); set_measurements(imu, odom); predict(p, f); update()`
        (data_conversion_Kalman_to_Training.py:193-199) as ONE launch.  Returns self.x; p is rotated in place."""
        self._rotate_in_place(p, self._step(OS_STEP_ODOM | OS_STEP_PREDICT | OS_STEP_UPDATE, p=p, f=f, dp=dp, imu=imu, contact=contact,
                                            want_K=want_K))
        return self.x

    def skew(self, x):
        # kalman_filter.py:195-198
        return np.array([[0, -x[2][0], x[1][0]], [x[2][0], 0, -x[0][0]], [-x[1][0], x[0][0], 0]])

    def solve_mpc(self, p, body_ref, cur_contact):
        """The stance controller's QP (misc/force_controller.py:70-162, parameters as kalman_filter.py:141-149 sets them)
        solved on the GPU (os_mpc_solve); returns the (12, N) control matrix the reference keeps in self.f."""
        e = self._eng
        col = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64).reshape(-1)[:12].astype(np.float32)).reshape(12, 1).to(e.device)
        c = torch.as_tensor(self._pack_contact(cur_contact)).to(e.device)
        r = e.mpc_solve(col(self.x), col(body_ref), col(p), c, want_all=True)
        if int(r["status"].cpu()[0]) & 4:
            raise RuntimeError("convex MPC: active-set iteration cap reached")       # qpOASES would report a failed solve
        return r["u"].cpu().numpy().astype(np.float64).reshape(5, 12).T.copy()

    def predict_mpc(self, p, body_ref, cur_contact, f=None):
        """kalman_filter.py:140-162.  f (12,) or (12,N), column 0 used: forces from a log; None: the QP is solved on the GPU in
        front of the step, in the same call (os_kf_step_mpc)."""
        if f is None:
            self._rotate_in_place(p, self._step(OS_STEP_PREDICT | OS_STEP_DENSE_FD | OS_STEP_MPC, p=p, body_ref=body_ref, contact=cur_contact))
            return
        self.f = np.asarray(f, dtype=np.float64).reshape(12, -1)
        self._rotate_in_place(p, self._step(OS_STEP_PREDICT | OS_STEP_DENSE_FD, p=p, f=self.f[:, 0], body_ref=body_ref))

    def estimate_state_mpc(self, imu, p, dp, body_ref, contact, f=None):
        """kalman_filter.py:176-182: get_odom + set_measurements + predict_mpc + update as ONE launch (float64 throughout:
        predict_mpc's element-wise exp(dt F) makes P ill-conditioned for float32); x, P, z, x_model, K, P_trace, K_gain and f
        are set as the reference leaves them."""
        # f = None: the convex MPC of kalman_filter.py:141-152 is solved on the GPU in front of the step kernel, both launches behind
        # ONE call and ONE stream synchronise (os_kf_step_mpc; round 6: the separate solve_mpc call through torch tensors was 157 of
        # the loop's 194 us per step, profiles/r06_dropin_loops.json); otherwise forces from a log
        if f is None:
            self._rotate_in_place(p, self._step(OS_STEP_ODOM | OS_STEP_PREDICT | OS_STEP_DENSE_FD | OS_STEP_UPDATE | OS_STEP_MPC, p=p, dp=dp,
                                                imu=imu, contact=contact, body_ref=body_ref, want_K=True))
            return self.x
        self.f = np.asarray(f, dtype=np.float64).reshape(12, -1)
        self._rotate_in_place(p, self._step(OS_STEP_ODOM | OS_STEP_PREDICT | OS_STEP_DENSE_FD | OS_STEP_UPDATE, p=p, f=self.f[:, 0], dp=dp,
                                            imu=imu, contact=contact, body_ref=body_ref, want_K=True))
        return self.x
