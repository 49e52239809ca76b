"""ctypes binding of liboptistate_hip.so (include/optistate_hip.h).  No CPU fallback: if the HIP library is
missing or no MI355X is visible, loading/creating a context raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liboptistate_hip.so")

OS_KF_SEQUENTIAL_UPDATE = 1
OS_KF_DENSE_FD = 2
OS_KF_SYMMETRIC_P = 4
OS_FUSED_TWO_KERNEL = 8
OS_KF_LANE_PER_TRAJECTORY = 32
OS_MPC_COLD_START = 64
OS_FUSED_ONE_KERNEL = 128
OS_KF_P_FLOAT64 = 256
OS_FUSED_SPLIT_BF16 = 512
OS_FUSED_SPLIT_BF16_2 = 1024
OS_FUSED_LATENT_IN_PLACE = 2048
OS_KF_WAVE_PER_TRAJECTORY = 4096
# status word (include/optistate_hip.h): bits 0-3 are failures, bit 4 is informational
OS_STATUS_S_NOT_PD, OS_STATUS_NONFINITE, OS_STATUS_QP_ITER, OS_STATUS_P0_ASYM, OS_STATUS_TRUNC_EDGE = 1, 2, 4, 8, 16
OS_STATUS_FAIL_MASK = 15
OS_GRU_SPLIT_ANY_BATCH = 0x100
OS_GRU_SPLIT_TRAIN = 0x200
OS_ERR_STACK_LOST = -20     # a layer-pipelined launch (gru_stack_kernel / bwd_sweep_stack_kernel) lost a producer: see os_gru_set_stack
OS_STEP_ODOM, OS_STEP_PREDICT, OS_STEP_UPDATE, OS_STEP_DENSE_FD, OS_STEP_MPC = 1, 2, 4, 8, 16
OS_PROF_PHASES = 12         # include/optistate_hip.h
PHASE_NAMES = ("kf", "gru_layer", "gru_head", "fused", "mpc", "train_sweep", "train_dw", "train_misc", "vit_gemm",
               "vit_attn", "vit_misc", "pack")

# every symbol include/optistate_hip.h declares
EXPORTS = [
    "os_create", "os_destroy", "os_last_error", "os_version", "os_build_arch", "os_kf_set_noise", "os_kf_run",
    "os_kf_odom", "os_kf_predict", "os_kf_update", "os_gru_param_count", "os_gru_load", "os_gru_forward",
    "os_gru_forward_soa", "os_fused_run", "os_pack_stream", "os_unpack_stream", "os_profile_enable", "os_profile_read",
    "os_gru_forward_train", "os_gru_loss", "os_gru_backward", "os_adam_step",
    "os_vit_param_count", "os_vit_load", "os_vit_encode", "os_mpc_set_weights", "os_mpc_solve", "os_kf_mpc_run",
    "os_kf_run_noise", "os_gru_generation", "os_gru_train_ws_floats", "os_gru_forward_train_ws", "os_gru_backward_ws",
    "os_profile_kernel_name", "os_build_id", "os_kf_step", "os_gru_load_keyed", "os_pack_stream_rows", "os_gru_backward_mark",
    "os_gru_forward_windows", "os_gru_bands", "os_gru_set_stack", "os_gru_set_split_bf16",
    "os_gru_get_stack", "os_fused_set_tile", "os_stack_check", "os_kf_step_mpc",
]


class OsKfConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("dt", C.c_float), ("mass", C.c_float), ("inertia", C.c_float * 3),
                ("gz", C.c_float)]


class OsVitDims(C.Structure):
    _fields_ = [("img_size", C.c_int32), ("patch_size", C.c_int32), ("in_chans", C.c_int32), ("embed_dim", C.c_int32),
                ("depth", C.c_int32), ("num_heads", C.c_int32), ("mlp_hidden", C.c_int32)]


class OsGruDims(C.Structure):
    _fields_ = [("input_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32),
                ("num_classes", C.c_int32), ("use_sigmoid", C.c_int32)]


_lib = None


def load():
    """Loads the shared library (built by optistate_amd.build) and declares prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own HIP runtime, and the library must bind to that same copy (loading the system
    # libamdhip64 before torch's leaves the process with two runtimes and os_create then sees no device)
    import torch  # noqa: F401
    path = os.environ.get("OPTISTATE_HIP_LIB", LIB_PATH)      # A/B builds of the same C-ABI (development aid)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -m optistate_amd.build` "
                           "(there is no CPU fallback for the OptiState hot path)")
    lib = C.CDLL(path)
    vp, i32, u32, f32p = C.c_void_p, C.c_int32, C.c_uint32, C.c_void_p
    lib.os_build_id.restype = C.c_char_p
    if "OPTISTATE_HIP_LIB" not in os.environ:
        # the .so is a git-ignored artefact that travels as a file: make sure it was built from THESE sources and flags
        from . import build as _build
        have, want = lib.os_build_id().decode().split("-")[0], _build.source_id()
        if have != want:
            raise RuntimeError(f"{path} is stale: built from sources {have}, the tree is {want}; run `python -m optistate_amd.build`")
    lib.os_create.argtypes = [C.POINTER(OsKfConfig), C.POINTER(vp)]
    lib.os_create.restype = C.c_int
    lib.os_destroy.argtypes = [vp]
    lib.os_destroy.restype = None
    lib.os_last_error.argtypes = [vp]
    lib.os_last_error.restype = C.c_char_p
    lib.os_version.restype = C.c_int
    lib.os_build_arch.restype = C.c_char_p
    lib.os_kf_set_noise.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.os_kf_run.argtypes = [vp, i32, i32] + [f32p] * 6 + [f32p] * 2 + [f32p] * 4 + [vp, u32, vp]
    lib.os_kf_odom.argtypes = [vp, i32, f32p, f32p, vp, f32p, f32p, vp]
    lib.os_kf_predict.argtypes = [vp, i32, f32p, f32p, f32p, f32p, f32p, f32p, u32, vp]
    lib.os_kf_update.argtypes = [vp, i32, f32p, f32p, f32p, f32p, f32p, f32p, vp, u32, vp]
    lib.os_gru_param_count.argtypes = [C.POINTER(OsGruDims)]
    lib.os_gru_param_count.restype = C.c_size_t
    lib.os_gru_load.argtypes = [vp, C.POINTER(OsGruDims), f32p, vp]
    lib.os_gru_backward_mark.argtypes = [vp, i32, vp]
    lib.os_gru_backward_mark.restype = C.c_int
    lib.os_gru_load_keyed.argtypes = [vp, C.POINTER(OsGruDims), f32p, C.c_uint64, vp]
    lib.os_gru_load_keyed.restype = C.c_int
    lib.os_gru_forward.argtypes = [vp, i32, i32, f32p, f32p, f32p, vp]
    lib.os_gru_forward_soa.argtypes = [vp, i32, i32, f32p, f32p, f32p, vp]
    lib.os_gru_forward_windows.argtypes = [vp, i32, i32, f32p, f32p, vp]
    lib.os_gru_forward_windows.restype = C.c_int
    lib.os_gru_bands.argtypes = [vp, i32, i32, f32p, f32p, f32p, f32p, f32p, f32p, vp]
    lib.os_gru_bands.restype = C.c_int
    lib.os_gru_set_stack.argtypes = [vp, i32]
    lib.os_gru_set_stack.restype = C.c_int
    lib.os_gru_get_stack.argtypes = [vp]
    lib.os_gru_get_stack.restype = C.c_int
    lib.os_stack_check.argtypes = [vp, vp]
    lib.os_stack_check.restype = C.c_int
    lib.os_fused_set_tile.argtypes = [vp, i32]
    lib.os_fused_set_tile.restype = C.c_int
    lib.os_gru_set_split_bf16.argtypes = [vp, i32]
    lib.os_gru_set_split_bf16.restype = C.c_int
    lib.os_fused_run.argtypes = [vp, i32, i32] + [f32p] * 8 + [i32, f32p] + [f32p] * 4 + [vp, u32, vp]
    lib.os_pack_stream.argtypes = [vp, i32, i32, i32, f32p, f32p, vp]
    lib.os_unpack_stream.argtypes = [vp, i32, i32, i32, f32p, f32p, vp]
    lib.os_pack_stream_rows.argtypes = [vp, i32, i32, i32, f32p, f32p, i32, i32, vp]
    lib.os_pack_stream_rows.restype = C.c_int
    lib.os_gru_forward_train.argtypes = [vp, i32, i32, f32p, f32p, vp]
    lib.os_gru_loss.argtypes = [vp, i32, f32p, f32p, f32p, f32p, f32p, vp]
    lib.os_gru_backward.argtypes = [vp, i32, i32, f32p, f32p, f32p, f32p, f32p, vp]
    lib.os_adam_step.argtypes = [vp, C.c_size_t, f32p, f32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]
    for n in ("os_gru_forward_train", "os_gru_loss", "os_gru_backward", "os_adam_step"):
        getattr(lib, n).restype = C.c_int
    lib.os_vit_param_count.argtypes = [C.POINTER(OsVitDims)]
    lib.os_vit_param_count.restype = C.c_size_t
    lib.os_vit_load.argtypes = [vp, C.POINTER(OsVitDims), f32p]
    lib.os_vit_encode.argtypes = [vp, i32, f32p, f32p, vp]
    lib.os_vit_load.restype = C.c_int
    lib.os_vit_encode.restype = C.c_int
    lib.os_mpc_set_weights.argtypes = [vp, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double]
    lib.os_mpc_solve.argtypes = [vp, i32, f32p, f32p, f32p, vp, f32p, f32p, vp, vp, i32, vp]
    lib.os_kf_mpc_run.argtypes = [vp, i32, i32, f32p, f32p, f32p, vp, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, vp, vp, u32, vp]
    lib.os_kf_mpc_run.restype = C.c_int
    lib.os_mpc_set_weights.restype = C.c_int
    lib.os_mpc_solve.restype = C.c_int
    lib.os_profile_enable.argtypes = [vp, C.c_int]
    lib.os_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.os_profile_enable.restype = C.c_int
    lib.os_profile_read.restype = C.c_int
    lib.os_profile_kernel_name.argtypes = [vp, C.c_int]
    lib.os_profile_kernel_name.restype = C.c_char_p
    lib.os_kf_run_noise.argtypes = [vp, i32, i32] + [f32p] * 5 + [f32p] * 4 + [f32p] * 3 + [vp, u32, vp]
    lib.os_kf_run_noise.restype = C.c_int
    lib.os_kf_step.argtypes = [vp, u32] + [vp] * 19
    lib.os_kf_step.restype = C.c_int
    lib.os_kf_step_mpc.argtypes = [vp, u32] + [vp] * 20
    lib.os_kf_step_mpc.restype = C.c_int
    lib.os_gru_generation.argtypes = [vp]
    lib.os_gru_generation.restype = C.c_uint64
    lib.os_gru_train_ws_floats.argtypes = [C.POINTER(OsGruDims), i32, i32]
    lib.os_gru_train_ws_floats.restype = C.c_size_t
    lib.os_gru_forward_train_ws.argtypes = [vp, i32, i32, f32p, f32p, f32p, vp]
    lib.os_gru_forward_train_ws.restype = C.c_int
    lib.os_gru_backward_ws.argtypes = [vp, C.POINTER(OsGruDims), f32p, i32, i32, f32p, f32p, f32p, f32p, f32p, f32p, vp]
    lib.os_gru_backward_ws.restype = C.c_int
    for n in ("os_kf_set_noise", "os_kf_run", "os_kf_odom", "os_kf_predict", "os_kf_update", "os_gru_load",
              "os_gru_forward", "os_gru_forward_soa", "os_fused_run", "os_pack_stream", "os_unpack_stream"):
        getattr(lib, n).restype = C.c_int
    _lib = lib
    return lib
