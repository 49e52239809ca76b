"""Batched Kalman+GRU engine: the host side of the C-ABI (include/optistate_hip.h).

torch is used for device memory and streams only; all arithmetic on the hot path happens in the HIP
kernels of liboptistate_hip.so.  There is no CPU fallback: constructing an engine without a visible
MI355X raises.

Stream layout (see the header): structure of arrays, trajectory index fastest, e.g. p [T][12][B].
`pack()` converts the reference's per-trajectory row lists ([B][T][F]) on the device.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _capi
from ._capi import (OS_KF_DENSE_FD, OS_KF_SEQUENTIAL_UPDATE, OS_KF_SYMMETRIC_P, OS_FUSED_TWO_KERNEL,  # noqa: F401
                    OS_KF_LANE_PER_TRAJECTORY, OS_MPC_COLD_START, OS_FUSED_ONE_KERNEL, OS_KF_P_FLOAT64, OS_FUSED_SPLIT_BF16, OS_FUSED_SPLIT_BF16_2,
                    OS_FUSED_LATENT_IN_PLACE, OS_KF_WAVE_PER_TRAJECTORY, OS_STATUS_FAIL_MASK, OS_STATUS_TRUNC_EDGE)

# settings.py:5-23 and kalman_filter/kalman_filter.py:56
DT, MASS, GZ = 0.01, 8.8, -9.81
INERTIA = (55303643.08 / 1e9, 60119440.34 / 1e9, 105304340.05 / 1e9)


def _ptr(t):
    """Device pointer of a tensor handed to the C-ABI (None -> NULL).  The library sees only the address: a host tensor would fault on
    the device, a strided view or a float64 tensor would be read as contiguous float32 -- refuse them here (4-byte element types only:
    float32 streams, int32 / uint32 contact and status words)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise TypeError("optistate_amd: a device tensor is required (got a host tensor); the hot path has no CPU fallback")
    if not t.is_contiguous():
        raise ValueError("optistate_amd: a contiguous tensor is required (call .contiguous())")
    if t.element_size() != 4:
        raise TypeError(f"optistate_amd: float32 / int32 data expected, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


class StackLost(RuntimeError):
    """A layer-pipelined launch lost a producer (C-ABI return code -20, os_gru_set_stack)."""


class Engine:
    def __init__(self, device=0, dt=DT, mass=MASS, inertia=INERTIA, gz=GZ):
        self.lib = _capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("optistate_amd: no HIP device visible (the hot path has no CPU fallback)")
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index or 0)
        cfg = _capi.OsKfConfig(self.device.index, dt, mass, (C.c_float * 3)(*inertia), gz)
        h = C.c_void_p()
        rc = self.lib.os_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"os_create failed with code {rc} (is this an MI355X / gfx950?)")
        self._h = h
        self._stack_mode = int(self.lib.os_gru_get_stack(h))      # what os_create read from OS_GRU_STACK (default 1)
        self.stack_fallbacks = 0
        self._gru_dims = None
        self._gru_flat = None      # keeps the flat weight tensor alive (the library references it for the head)
        self._gru_owner = None     # who loaded the resident GRU weights (a context holds ONE model: see load_gru)
        self._diag_R = True
        self._sym_Q = True
        self._keyed_flats = {}
        self._next_key = 0
        self._inval_epoch = 0
        # a context holds ONE loaded model and ONE set of scratch buffers: whoever pairs load_gru with a forward on a context that
        # other threads may use holds this lock across the pair (RNN.forward, DataParallelTrainer.step, the ViT module)
        self.lock = threading.RLock()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.os_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == _capi.OS_ERR_STACK_LOST:
            raise StackLost(f"{what} failed ({rc}): {self.lib.os_last_error(self._h).decode()}")
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.os_last_error(self._h).decode()}")

    def set_stack_mode(self, mode):
        """0: a launch per layer; 1 (default): small batches run their layer stack as one pipelined launch and the call waits for it
        and checks its error word; 2: the same, asynchronous (os_gru_set_stack in include/optistate_hip.h)."""
        self._check(self.lib.os_gru_set_stack(self._h, int(mode)), "os_gru_set_stack")
        self._stack_mode = int(mode)

    def set_fused_tile(self, tile):
        """Pins the single fused kernel's tile shape (trajectories per workgroup: 256 | 128 | 64 | 32 | 16), 0 = chosen from the batch
        (os_fused_set_tile in include/optistate_hip.h)."""
        self._check(self.lib.os_fused_set_tile(self._h, int(tile)), "os_fused_set_tile")

    def set_gru_split_bf16(self, terms, any_batch=False, train=False):
        """OPT-IN reduced precision for the GRU layer GEMMs (never the default; the reference computes them in fp32,
        gru/gru_model.py:12): 0 = exact fp32; 3 / 2 = every operand of an H = 128 inference layer's gate GEMM split into that many bf16
        terms, products on the bf16 matrix instruction with fp32 accumulation (os_gru_set_split_bf16 in include/optistate_hip.h:
        batches of at least 128 x CUs trajectories; any_batch: every batch that is a multiple of 4)."""
        # train=True: also the training step's weight-gradient products (OS_GRU_SPLIT_TRAIN, round 6)
        self._check(self.lib.os_gru_set_split_bf16(self._h, int(terms) | (_capi.OS_GRU_SPLIT_ANY_BATCH if any_batch else 0) |
                                                   (_capi.OS_GRU_SPLIT_TRAIN if train else 0)), "os_gru_set_split_bf16")

    def _stack_guarded(self, call):
        """Runs call(); in VERIFIED mode (1) a StackLost belongs to the launch this very call made (the library waited for it: the
        producer workgroup never became resident, or another process held its CUs for seconds), so the call is run again with a launch
        per layer -- the GRU entry points are idempotent (outputs and the flat gradient are overwritten), the retry is exact.  In
        asynchronous mode (2) a StackLost reported at a call's entry refers to an EARLIER launch whose poisoned outputs this call
        would consume: it is not caught here -- whoever chose mode 2 redoes that work (DataParallelTrainer.step does, through
        stack_check)."""
        try:
            return call()
        except StackLost:
            if self._stack_mode != 1:
                raise
            self.lib.os_gru_set_stack(self._h, 0)
            self.stack_fallbacks += 1
            try:
                return call()
            finally:
                self.lib.os_gru_set_stack(self._h, 1)

    def stack_check(self):
        """Mode 2's verification point (os_stack_check): True when an asynchronous stacked launch since the last check lost a producer
        (the error word is cleared; what those launches wrote is NaN-poisoned and must be redone with set_stack_mode(0)).  Waits for the
        stream only if such a launch went out at all."""
        rc = self.lib.os_stack_check(self._h, self._stream())
        if rc == _capi.OS_ERR_STACK_LOST:
            return True
        self._check(rc, "os_stack_check")
        return False

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, a, shape=None):
        t = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32) if not torch.is_tensor(a) else a)
        t = t.to(self.device, dtype=torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)

    # ---- the per-trajectory status word ----
    @staticmethod
    def failed(status):
        """bool [B]: trajectories whose filter FAILED (bits 0-3 of the status word: S not positive definite, non-finite state,
        QP iteration cap, non-symmetric P0 under symmetric storage).  Bit 4 (OS_STATUS_TRUNC_EDGE) is informational and is not
        a failure: `status != 0` is the wrong test -- use this."""
        return (status & OS_STATUS_FAIL_MASK) != 0

    @staticmethod
    def trunc_edge(status):
        """bool [B]: trajectories that met the reference's int64-truncation knife edge at some step (status bit 4: an entry of
        the float64 rotation matrix within 2^-40 of +-1 away from the exact start; the 1e-4 state bar is not promised there --
        include/optistate_hip.h)."""
        return (status & OS_STATUS_TRUNC_EDGE) != 0

    # ---- per-kernel device timing ----
    def profile(self, enable=True):
        self._check(self.lib.os_profile_enable(self._h, 1 if enable else 0), "os_profile_enable")

    def profile_read(self):
        """Returns {phase: (ms_sum, launches)} since the last read; phases as _capi.PHASE_NAMES (header enum OS_PHASE_*)."""
        ms = (C.c_double * _capi.OS_PROF_PHASES)()
        n = (C.c_int32 * _capi.OS_PROF_PHASES)()
        self._check(self.lib.os_profile_read(self._h, ms, n), "os_profile_read")
        return {k: (ms[i], n[i]) for i, k in enumerate(_capi.PHASE_NAMES)}

    def kernel_name(self, phase):
        """The kernel variant most recently launched in `phase` (name from _capi.PHASE_NAMES or index)."""
        i = _capi.PHASE_NAMES.index(phase) if isinstance(phase, str) else int(phase)
        return self.lib.os_profile_kernel_name(self._h, i).decode()

    # ---- Kalman filter ----
    def set_noise(self, Q, R):
        """KF.Q = Q; KF.R = R (data_collection/data_conversion_Kalman_to_Training.py:139-143)."""
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(144)
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(100)
        self._diag_R = bool(np.count_nonzero(R.reshape(10, 10) - np.diag(np.diag(R.reshape(10, 10)))) == 0)
        # the symmetric-storage kernels replace Q by 0.5 (Q + Q^T) and read the upper triangle of P0; the reference never
        # symmetrises anything (kalman_filter.py:135,172), so a non-symmetric Q takes the full-P kernels by default
        self._sym_Q = bool(np.array_equal(Q.reshape(12, 12), Q.reshape(12, 12).T))
        self._check(self.lib.os_kf_set_noise(self._h, Q.ctypes.data_as(C.POINTER(C.c_float)),
                                             R.ctypes.data_as(C.POINTER(C.c_float))), "os_kf_set_noise")

    def pack(self, a_btf):
        """[B][T][F] -> [T][F][B] on the device."""
        a = self._f32(a_btf)
        B, T, F = a.shape
        out = torch.empty((T, F, B), dtype=torch.float32, device=self.device)
        self._check(self.lib.os_pack_stream(self._h, B, T, F, _ptr(a), _ptr(out), self._stream()), "os_pack_stream")
        return out

    def gru_input_with_latent(self, latent_btn):
        """The GRU input buffer [T][60 + NL][B] with the latent stream (B, T, NL) -- e.g. the ViT encoder's output -- packed
        straight into rows 60.. (os_pack_stream_rows); pass it as fused_run(..., gru_input=buf): the Kalman kernel fills rows
        0..59 in place and no copy of the latent is made (gru/gru_test.py:135-136 concatenates per window on the host)."""
        a = self._f32(latent_btn)
        B, T, NL = a.shape
        buf = torch.empty((T, 60 + NL, B), dtype=torch.float32, device=self.device)
        self._check(self.lib.os_pack_stream_rows(self._h, B, T, NL, _ptr(a), _ptr(buf), 60 + NL, 60, self._stream()), "os_pack_stream_rows")
        return buf

    def unpack(self, a_tfb):
        T, F, B = a_tfb.shape
        out = torch.empty((B, T, F), dtype=torch.float32, device=self.device)
        self._check(self.lib.os_unpack_stream(self._h, B, T, F, _ptr(a_tfb), _ptr(out), self._stream()),
                    "os_unpack_stream")
        return out

    def pack_contact(self, c_bt4):
        """uint8 [B][T][4] -> packed uint32-as-4-bytes [T][B]."""
        c = torch.as_tensor(c_bt4).to(self.device, dtype=torch.uint8).contiguous()
        B, T = c.shape[0], c.shape[1]
        return c.permute(1, 0, 2).reshape(T * B, 4).contiguous().view(torch.int32).reshape(T, B)

    @staticmethod
    def contact_soa_to_packed(c_t4b):
        """uint8 [T][4][B] -> packed [T][B] int32."""
        T, _, B = c_t4b.shape
        return c_t4b.permute(0, 2, 1).reshape(T * B, 4).contiguous().view(torch.int32).reshape(T, B)

    def kf_run(self, p, f, dp, imu, contact, x, P, body_ref=None, sequential=None, dense_fd=False,
               want_p_rot=False, want_trace=False, want_gain=False, symmetric=None, lane_per_trajectory=False,
               wave_per_trajectory=False):
        """Runs T filter steps for B trajectories.  All stream arguments are SoA device tensors; x [12][B] and
        P [144][B] are updated in place.  Returns dict(x_out [T][12][B], status [B], p_rot?, P_trace?, K_gain?).
        symmetric=None picks the symmetric-storage kernels (upper triangle of P in registers) when R is diagonal and Q is
        symmetric.  They read only the upper triangle of the caller's P0 and write the final P back mirrored: a P0 that is
        not symmetric sets status bit 3 (value 8) for that trajectory -- re-run it with symmetric=False, which keeps the
        full P like the reference (kalman_filter.py:172 never symmetrises).
        want_gain: K_gain [T][B].  The sequential / symmetric kernels (the default for a diagonal R) never form K: their
        K_gain is trace(P+ H^T R^-1) on the float32 POSTERIOR -- equal to the reference's trace(K) (kalman_filter.py:174) in
        exact arithmetic, to ~1e-3 relative in float32.  Callers that feed K_gain downstream and need the trace of the K the
        batch update actually forms pass sequential=False.  Test Engine.failed(status), not status != 0 (bit 4 is informational)."""
        T, _, B = p.shape
        if sequential is None:
            sequential = self._diag_R           # K_gain no longer forces the batch form: trace(P+ H^T R^-1) on the posterior
        if symmetric is None:
            symmetric = sequential and not dense_fd and self._sym_Q
        flags = (OS_KF_SEQUENTIAL_UPDATE if sequential else 0) | (OS_KF_DENSE_FD if dense_fd else 0) | \
                (OS_KF_SYMMETRIC_P if symmetric else 0) | (OS_KF_LANE_PER_TRAJECTORY if lane_per_trajectory else 0) | \
                (OS_KF_WAVE_PER_TRAJECTORY if wave_per_trajectory else 0)   # one wavefront per trajectory, P in LDS: measurement only
        dev = self.device
        x_out = torch.empty((T, 12, B), dtype=torch.float32, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        p_rot = torch.empty((T, 12, B), dtype=torch.float32, device=dev) if want_p_rot else None
        ptr = torch.empty((T, B), dtype=torch.float32, device=dev) if want_trace else None
        kg = torch.empty((T, B), dtype=torch.float32, device=dev) if want_gain else None
        self._check(self.lib.os_kf_run(self._h, B, T, _ptr(p), _ptr(f), _ptr(dp), _ptr(imu), _ptr(contact),
                                       _ptr(body_ref), _ptr(x), _ptr(P), _ptr(x_out), _ptr(p_rot), _ptr(ptr), _ptr(kg),
                                       _ptr(status), flags, self._stream()), "os_kf_run")
        return dict(x_out=x_out, status=status, p_rot=p_rot, P_trace=ptr, K_gain=kg)

    def kf_run_noise(self, p, f, dp, imu, contact, x, P, q_diag, r_diag, want_p_rot=False, want_trace=False):
        """kf_run with per-trajectory diagonal noise: q_diag [12][B], r_diag [10][B] device tensors (each reference filter
        instance carries its own Q and R: data_conversion_Kalman_to_Training.py:138-144)."""
        T, _, B = p.shape
        dev = self.device
        q = q_diag.to(dev, dtype=torch.float32).contiguous(); r = r_diag.to(dev, dtype=torch.float32).contiguous()
        if q.shape != (12, B) or r.shape != (10, B):
            raise ValueError("q_diag must be [12][B] and r_diag [10][B]")
        x_out = torch.empty((T, 12, B), dtype=torch.float32, device=dev)
        status = torch.empty((B,), dtype=torch.int32, device=dev)
        p_rot = torch.empty((T, 12, B), dtype=torch.float32, device=dev) if want_p_rot else None
        ptr = torch.empty((T, B), dtype=torch.float32, device=dev) if want_trace else None
        self._check(self.lib.os_kf_run_noise(self._h, B, T, _ptr(p), _ptr(f), _ptr(dp), _ptr(imu), _ptr(contact), _ptr(x), _ptr(P),
                                             _ptr(q), _ptr(r), _ptr(x_out), _ptr(p_rot), _ptr(ptr), _ptr(status), 0,
                                             self._stream()), "os_kf_run_noise")
        return dict(x_out=x_out, status=status, p_rot=p_rot, P_trace=ptr)

    # ---- GRU ----
    def load_gru(self, flat, input_size, hidden_size, num_layers, num_classes, use_sigmoid=True, owner=None, key=0):
        """flat: 1-D float32 device tensor in the flat layout of the header (see flatten_state_dict).
        A context runs ONE model at a time; `owner` records who selected it so that several weight containers sharing this
        engine (two RNN modules on one GPU, a trainer and its model) can tell whether their weights are the active ones.
        key != 0 (a name for these weights in this state, new_gru_key()): the library keeps the packed images of the four
        most recently used keys, so alternating models are re-selected without re-packing (os_gru_load_keyed)."""
        d = _capi.OsGruDims(input_size, hidden_size, num_layers, num_classes, 1 if use_sigmoid else 0)
        n = self.lib.os_gru_param_count(C.byref(d))
        flat = flat.to(self.device, dtype=torch.float32).contiguous()
        if flat.numel() != n:
            raise ValueError(f"flat weight vector has {flat.numel()} floats, expected {n}")
        self._check(self.lib.os_gru_load_keyed(self._h, C.byref(d), _ptr(flat), int(key), self._stream()), "os_gru_load")
        self._gru_flat, self._gru_dims, self._gru_owner = flat, d, owner
        if key:
            self._keyed_flats[int(key)] = flat                 # keep every cached image's flat weights alive (head / biases read them)
            while len(self._keyed_flats) > 8:
                self._keyed_flats.pop(next(iter(self._keyed_flats)))

    def new_gru_key(self):
        self._next_key += 1
        return self._next_key

    def gru_generation(self):
        return int(self.lib.os_gru_generation(self._h))

    def invalidate_gru(self):
        """The flat weights were modified in place (optimiser step; torch's version counters did not move): whoever runs
        next must flatten and pack again, cached images included."""
        self._gru_owner = None
        self._inval_epoch += 1

    def gru_forward(self, x_bti, want_h_last=False):
        """RNN.forward on x (B, T, I) -> (B, C)  (gru/gru_model.py:25-49)."""
        x = x_bti.to(self.device, dtype=torch.float32).contiguous()
        B, T, I = x.shape
        d = self._gru_dims
        if d is None or I != d.input_size:
            raise ValueError("load_gru first / input width mismatch")
        out = torch.empty((B, d.num_classes), dtype=torch.float32, device=self.device)
        hl = torch.empty((d.num_layers, B, d.hidden_size), dtype=torch.float32, device=self.device) if want_h_last else None
        self._stack_guarded(lambda: self._check(self.lib.os_gru_forward(self._h, B, T, _ptr(x), _ptr(out), _ptr(hl), self._stream()), "os_gru_forward"))
        return (out, hl) if want_h_last else out

    def gru_forward_windows(self, rows_ni, window):
        """The reference's inference mode without materialised windows (gru/gru_test.py:138-140,174-191): rows (N, I), one
        time-ordered row stream -> (N - window + 1, C), output i = RNN.forward(rows[i : i + window]).  The input half of the first
        layer's gate GEMM is computed once per row and shared by the `window` windows that contain it (os_gru_forward_windows).
        hidden_size 128 / 64 and input_size <= 192; other shapes raise (materialise the windows and call gru_forward)."""
        rows = rows_ni.to(self.device, dtype=torch.float32).contiguous()
        N, I = rows.shape
        d = self._gru_dims
        if d is None or I != d.input_size:
            raise ValueError("load_gru first / input width mismatch")
        if not 1 <= window <= N:
            raise ValueError("1 <= window <= number of rows")
        out = torch.empty((N - window + 1, d.num_classes), dtype=torch.float32, device=self.device)
        self._stack_guarded(lambda: self._check(self.lib.os_gru_forward_windows(self._h, N, int(window), _ptr(rows), _ptr(out), self._stream()),
                                                "os_gru_forward_windows"))
        return out

    def gru_bands(self, out, min_v, max_v):
        """(pred, band_above, band_below), de-normalised, from the model's output [B][2 n] = [prediction | error] (gru/gru_test.py:
        184-189,208-213) in one launch (os_gru_bands)."""
        out = out.to(self.device, dtype=torch.float32).contiguous()
        B, C2 = out.shape
        n = C2 // 2
        mn = min_v.to(self.device, dtype=torch.float32).contiguous(); mx = max_v.to(self.device, dtype=torch.float32).contiguous()
        if C2 != 2 * n or mn.numel() != n or mx.numel() != n:
            raise ValueError("out must be [B][2 n] and min_v / max_v [n]")
        pred, above, below = (torch.empty((B, n), dtype=torch.float32, device=self.device) for _ in range(3))
        self._check(self.lib.os_gru_bands(self._h, B, n, _ptr(out), _ptr(mn), _ptr(mx), _ptr(pred), _ptr(above), _ptr(below), self._stream()),
                    "os_gru_bands")
        return pred, above, below

    def gru_windows_supported(self):
        """Shapes os_gru_forward_windows takes: hidden 128 with up to 192 inputs, hidden 64 with up to 190 (the eight-wave split body's
        LDS: gru_kernels.hip split_lds_bytes)."""
        d = self._gru_dims
        return d is not None and ((d.hidden_size == 128 and d.input_size <= 192) or (d.hidden_size == 64 and d.input_size <= 190))

    def gru_forward_soa(self, xs_tib):
        T, I, B = xs_tib.shape
        d = self._gru_dims
        if d is None:
            raise RuntimeError("gru_forward_soa: load_gru first")
        out = torch.empty((B, d.num_classes), dtype=torch.float32, device=self.device)
        self._stack_guarded(lambda: self._check(self.lib.os_gru_forward_soa(self._h, B, T, _ptr(xs_tib), _ptr(out), None, self._stream()),
                                                "os_gru_forward_soa"))
        return out

    # ---- training step (gru/gru_train.py:232-249) ----
    def gru_forward_train(self, x_bti):
        x = x_bti.to(self.device, dtype=torch.float32).contiguous()
        B, T, I = x.shape
        d = self._gru_dims
        if d is None or I != d.input_size:
            raise ValueError("load_gru first / input width mismatch")
        out = torch.empty((B, d.num_classes), dtype=torch.float32, device=self.device)
        self._stack_guarded(lambda: self._check(self.lib.os_gru_forward_train(self._h, B, T, _ptr(x), _ptr(out), self._stream()), "os_gru_forward_train"))
        return out

    def gru_forward_train_ws(self, x_bti):
        """Training forward with the saved activations in a workspace of its own (returned): any number of forwards may be
        outstanding before their backwards run.  Returns (out, ws, flat, dims): everything gru_backward_ws needs."""
        x = x_bti.to(self.device, dtype=torch.float32).contiguous()
        B, T, I = x.shape
        d = self._gru_dims
        if d is None or I != d.input_size:
            raise ValueError("load_gru first / input width mismatch")
        n = self.lib.os_gru_train_ws_floats(C.byref(d), B, T)
        ws = torch.empty((n,), dtype=torch.float32, device=self.device)
        out = torch.empty((B, d.num_classes), dtype=torch.float32, device=self.device)
        self._stack_guarded(lambda: self._check(self.lib.os_gru_forward_train_ws(self._h, B, T, _ptr(x), _ptr(out), _ptr(ws), self._stream()),
                                                "os_gru_forward_train_ws"))
        return out, ws, self._gru_flat, _capi.OsGruDims(d.input_size, d.hidden_size, d.num_layers, d.num_classes, d.use_sigmoid)

    def gru_backward_ws(self, dims, flat, ws, x_bti, out, dout, grad_flat=None, want_dx=False):
        x = x_bti.to(self.device, dtype=torch.float32).contiguous()
        B, T, I = x.shape
        n = self.lib.os_gru_param_count(C.byref(dims))
        if grad_flat is None:
            grad_flat = torch.empty((n,), dtype=torch.float32, device=self.device)
        dx = torch.empty_like(x) if want_dx else None
        dout = dout.to(self.device, dtype=torch.float32).contiguous()
        self._stack_guarded(lambda: self._check(self.lib.os_gru_backward_ws(self._h, C.byref(dims), _ptr(flat), B, T, _ptr(x), _ptr(out), _ptr(dout), _ptr(ws),
                                                                            _ptr(grad_flat), _ptr(dx), self._stream()), "os_gru_backward_ws"))
        return (grad_flat, dx) if want_dx else grad_flat

    def gru_loss(self, out, y, want_target=False):
        """target = [y | |out[:, :C/2] - y|] (detached), MSE; returns (loss (1,) device tensor, dout, target|None)."""
        B, Cc = out.shape
        y = y.to(self.device, dtype=torch.float32).contiguous()
        dout = torch.empty_like(out)
        loss = torch.empty((1,), dtype=torch.float32, device=self.device)
        tgt = torch.empty_like(out) if want_target else None
        self._check(self.lib.os_gru_loss(self._h, B, _ptr(out), _ptr(y), _ptr(tgt), _ptr(dout), _ptr(loss), self._stream()),
                    "os_gru_loss")
        return loss, dout, tgt

    def gru_backward(self, x_bti, out, dout, grad_flat=None, want_dx=False):
        x = x_bti.to(self.device, dtype=torch.float32).contiguous()
        B, T, I = x.shape
        n = self.lib.os_gru_param_count(C.byref(self._gru_dims))
        if grad_flat is None:
            grad_flat = torch.empty((n,), dtype=torch.float32, device=self.device)
        dx = torch.empty_like(x) if want_dx else None
        dout = dout.to(self.device, dtype=torch.float32).contiguous()
        self._stack_guarded(lambda: self._check(self.lib.os_gru_backward(self._h, B, T, _ptr(x), _ptr(out), _ptr(dout), _ptr(grad_flat), _ptr(dx),
                                                                         self._stream()), "os_gru_backward"))
        return (grad_flat, dx) if want_dx else grad_flat

    def gru_backward_mark(self, layer, event):
        """event: a recorded torch.cuda.Event (or None to switch off): see os_gru_backward_mark."""
        self._check(self.lib.os_gru_backward_mark(self._h, int(layer), None if event is None else C.c_void_p(event.cuda_event)),
                    "os_gru_backward_mark")

    def adam_step(self, w, g, m, v, lr, beta1, beta2, eps, step):
        self._check(self.lib.os_adam_step(self._h, w.numel(), _ptr(w), _ptr(g), _ptr(m), _ptr(v), lr, beta1, beta2, eps, step,
                                          self._stream()), "os_adam_step")

    # ---- fused ----
    def fused_run(self, p, f, dp, imu, contact, accel, minmax, x, P, body_ref=None, latent=None, sequential=None,
                  dense_fd=False, symmetric=None, two_kernel=None, split_bf16=False, gru_input=None):
        """KF + feature pack + normalise + GRU.  Returns dict(out [B][C], x_out [T][12][B], status [B]).
        split_bf16: False (exact fp32 MFMA, the default) | True or 3 (three bf16 terms per operand) | 2 (two terms): opt-in.
        two_kernel: None = the library picks (single fused kernel from about a third of a chip of trajectories up: B > 80 per CU), True / False force
        the two-kernel / the single-kernel path (the latter only where the shapes allow it)."""
        T, _, B = p.shape
        if sequential is None:
            sequential = self._diag_R
        if symmetric is None:
            symmetric = sequential and not dense_fd and self._sym_Q
        flags = (OS_KF_SEQUENTIAL_UPDATE if sequential else 0) | (OS_KF_DENSE_FD if dense_fd else 0) | \
                (OS_KF_SYMMETRIC_P if symmetric else 0) | (OS_FUSED_TWO_KERNEL if two_kernel else 0) | \
                (OS_FUSED_ONE_KERNEL if two_kernel is False else 0) | \
                (OS_FUSED_SPLIT_BF16_2 if split_bf16 == 2 else (OS_FUSED_SPLIT_BF16 if split_bf16 else 0))
        d = self._gru_dims
        if gru_input is not None:          # [T][60 + NL][B] with the latent rows in place (gru_input_with_latent): written by this call
            if latent is not None or gru_input.shape[1] <= 60 or gru_input.shape[0] != T or gru_input.shape[2] != B:
                raise ValueError("gru_input must be [T][60 + NL][B] and excludes latent=")
            latent, flags = gru_input, flags | OS_FUSED_LATENT_IN_PLACE
            nl = gru_input.shape[1] - 60
        else:
            nl = 0 if latent is None else latent.shape[1]
        out = torch.empty((B, d.num_classes), dtype=torch.float32, device=self.device)
        x_out = torch.empty((T, 12, B), dtype=torch.float32, device=self.device)
        status = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._check(self.lib.os_fused_run(self._h, B, T, _ptr(p), _ptr(f), _ptr(dp), _ptr(imu), _ptr(contact),
                                          _ptr(accel), _ptr(body_ref), _ptr(latent), nl, _ptr(minmax), _ptr(x), _ptr(P),
                                          _ptr(x_out), _ptr(out), _ptr(status), flags, self._stream()), "os_fused_run")
        return dict(out=out, x_out=x_out, status=status)


    # ---- batched single-step pieces (the B = 1 drop-in class uses the same entry points) ----
    def kf_odom(self, p, dp, contact, imu):
        """get_odom + set_measurements for B trajectories (kalman_filter.py:79-117): p, dp [12][B], imu [6][B],
        contact [B] packed -> z [10][B]."""
        B = p.shape[1]
        z = torch.empty((10, B), dtype=torch.float32, device=self.device)
        self._check(self.lib.os_kf_odom(self._h, B, _ptr(p), _ptr(dp), _ptr(contact), _ptr(imu), _ptr(z), self._stream()), "os_kf_odom")
        return z

    def kf_predict(self, p, f, x, P, body_ref=None):
        """predict (kalman_filter.py:119-138) or, with body_ref, predict_mpc's covariance (:153-161) for B trajectories.
        p [12][B] is rotated to the world frame in place, x [12][B] and P [144][B] are updated in place."""
        B = p.shape[1]
        self._check(self.lib.os_kf_predict(self._h, B, _ptr(p), _ptr(f), _ptr(body_ref), _ptr(x), _ptr(P), None,
                                           OS_KF_DENSE_FD if body_ref is not None else 0, self._stream()), "os_kf_predict")

    def kf_update(self, z, x, P, sequential=False, want_K=False):
        """update() (kalman_filter.py:164-174) for B trajectories: z [10][B]; x [12][B], P [144][B] in place.
        Returns dict(status [B], P_trace [B], K_gain [B], K [120][B] (12 x 10 row-major) if want_K).  sequential=True (diagonal R)
        never forms K during the update; K and K_gain then come from the posterior, K = P+ H^T R^-1."""
        B = x.shape[1]
        mk = lambda n: torch.empty((n, B) if n > 1 else (B,), dtype=torch.float32, device=self.device)
        K = mk(120) if want_K else None
        ptr, kg = mk(1), mk(1)
        status = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._check(self.lib.os_kf_update(self._h, B, _ptr(z), _ptr(x), _ptr(P), _ptr(K), _ptr(ptr), _ptr(kg), _ptr(status),
                                          OS_KF_SEQUENTIAL_UPDATE if sequential else 0, self._stream()), "os_kf_update")
        return dict(status=status, P_trace=ptr, K_gain=kg, K=K)

    # ---- convex-MPC ground-reaction forces (misc/force_controller.py:70-162, kalman_filter.py:141-152) ----
    def mpc_set_weights(self, q_weights, r_weight=1e-6, mu=0.6, fz_max=150.0):
        """diag(Q) of the stance controller (kalman_filter.py:64), R scalar (:66), friction and force cap (force_controller.py:147-149)."""
        w = (C.c_double * 12)(*[float(v) for v in q_weights])
        self._check(self.lib.os_mpc_set_weights(self._h, w, float(r_weight), float(mu), float(fz_max)), "os_mpc_set_weights")

    def mpc_solve(self, x, body_ref, p, contact, want_all=False, max_iter=0):
        """x, body_ref, p: [12][B] float32 device tensors; contact: [B] packed uint32 (int32 storage).
        Returns dict(f [12][B] = column 0 of the optimal controls, status [B], iters [B], u [60][B] if want_all)."""
        B = x.shape[1]
        f = torch.empty((12, B), dtype=torch.float32, device=self.device)
        u = torch.empty((60, B), dtype=torch.float32, device=self.device) if want_all else None
        iters = torch.empty((B,), dtype=torch.int32, device=self.device)
        status = torch.zeros((B,), dtype=torch.int32, device=self.device)
        self._check(self.lib.os_mpc_solve(self._h, B, _ptr(x), _ptr(body_ref), _ptr(p), _ptr(contact), _ptr(f), _ptr(u),
                                          _ptr(iters), _ptr(status), int(max_iter), self._stream()), "os_mpc_solve")
        return dict(f=f, u=u, iters=iters, status=status)


    def kf_mpc_run(self, p, dp, imu, contact, body_ref, x, P, sequential=False, want_p_rot=False, want_trace=False,
                   want_gain=False, want_iters=False, cold_start=False):
        """estimate_state_mpc for B trajectories x T steps (kalman_filter.py:176-182): QP forces + dense-F_d filter step.
        Streams [T][.][B] as kf_run; x [12][B], P [144][B] in/out.  Returns dict(x_out, f [T][12][B], status, ...).
        Up to 32 trajectories per CU one persistent kernel; above, a launch sequence per step whose plain form (no sequential update, no
        P_trace / K_gain output) runs the filter step inside the QP launch and the batch as two concurrent halves (include/optistate_hip.h;
        OS_MPC_FUSE_KF=0 / OS_MPC_SHARDS=1 switch those off): same results either way, the call stays asynchronous on the current stream."""
        T, _, B = p.shape
        mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)
        x_out, f_out = mk(T, 12, B), mk(T, 12, B)
        p_rot = mk(T, 12, B) if want_p_rot else None
        ptrace = mk(T, B) if want_trace else None
        kgain = mk(T, B) if want_gain else None
        iters = torch.empty((T, B), dtype=torch.int32, device=self.device) if want_iters else None
        status = torch.empty((B,), dtype=torch.int32, device=self.device)
        flags = (OS_KF_SEQUENTIAL_UPDATE if sequential else 0) | (OS_MPC_COLD_START if cold_start else 0)
        self._check(self.lib.os_kf_mpc_run(self._h, B, T, _ptr(p), _ptr(dp), _ptr(imu), _ptr(contact), _ptr(body_ref),
                                           _ptr(x), _ptr(P), _ptr(x_out), _ptr(f_out), _ptr(p_rot), _ptr(ptrace),
                                           _ptr(kgain), _ptr(iters), _ptr(status), flags, self._stream()), "os_kf_mpc_run")
        return dict(x_out=x_out, f=f_out, p_rot=p_rot, ptrace=ptrace, kgain=kgain, iters=iters, status=status)


def flatten_state_dict(sd, num_layers, device=None):
    """state_dict with the reference's keys (gru.weight_ih_l{k}, gru.weight_hh_l{k}, gru.bias_ih_l{k},
    gru.bias_hh_l{k}, fc.weight, fc.bias) -> flat float32 tensor."""
    parts = []
    for l in range(num_layers):
        for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            parts.append(sd[f"gru.{k}_l{l}"].detach().reshape(-1).float())
    parts += [sd["fc.weight"].detach().reshape(-1).float(), sd["fc.bias"].detach().reshape(-1).float()]
    flat = torch.cat(parts)
    return flat.to(device) if device is not None else flat


_default_engines = threading.local()


def reset_default_engines():
    """Forgets the calling thread's default contexts (the next default_engine() creates a fresh one, re-reading the OS_* tuning
    variables): for tools that compare tuning knobs inside one process."""
    _default_engines.__dict__.pop("engines", None)


def default_engine(device=0):
    """The calling THREAD's context on `device` (one context per (thread, GPU); calls on a context are stream-ordered
    and a context holds ONE loaded model and one set of scratch buffers, so two threads must not share one).  The weight
    containers (RNN, Transformer_Autoencoder), the drop-in Kalman_Filter and DataParallelTrainer take theirs from here; a
    container that is handed to another thread keeps the engine it was first used with and serialises its load + forward pairs on
    that engine's lock (Engine.lock)."""
    idx = device if isinstance(device, int) else (torch.device(device).index or 0)
    engines = _default_engines.__dict__.setdefault("engines", {})
    if idx not in engines:
        engines[idx] = Engine(idx)
    return engines[idx]
