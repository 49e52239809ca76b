#!/usr/bin/env python3
"""G13: the reference's two data-conversion SCRIPTS executed unmodified in this container on a synthetic raw log.

Run:  python tools/gen_golden_etl.py        (needs /root/reference; never runs on the GPU box)

  step 1  data_collection/data_conversion_raw_to_Kalman.py   (ETL: `.mat` -> saved_trajectories.pkl, :39-447)
  step 2  data_collection/data_conversion_Kalman_to_Training.py  (Q/R fit :31-109, then the estimate_state_mpc loop and the
          60-column rows :115-336 -> Q_R.pkl, rnn_data.pkl, p_trace_data.mat)

Both are top-level scripts; their TEXT is read at generation time and exec'd (never stored) with `__file__` pointed into a
scratch `<tmp>/OptiState/data_collection/` tree, so every path they derive from it (`:19-21`, `:20`) lands in scratch.
What is substituted, and only because the image lacks it:
  cv2         -> `imwrite` records the frame instead of encoding a PNG (the script uses nothing else of cv2, :427-436)
  scaler_kin  -> Leg.leg_jacobian_3DoF = tests/golden_recipes.g13_leg_jacobian (the script uses nothing else, :397);
                 io.load_mat_trajectory takes the same callable as a parameter
  casadi      -> tools/casadi_eval.py (the evaluating stand-in of G12: formulation = reference, solver = certified stand-in)
  matplotlib  -> the real library on the Agg backend (plt.show() is a no-op there)
What is set, as a user of the scripts would:
  settings.INITIAL_PARAMS.DATA_CUTOFF_END = 640 (shipped value 4494; DATA_CUTOFF_START = 430 as shipped) for the committed
  arrays -- a second pass at the shipped 430 / 4494 on a 4,500-row log commits list lengths, column sums and sampled entries;
  step 2's hand switch `load_Q_R = True` (:19) flipped to False for the run, which is how its author produces Q_R.pkl
  (the only edit of the text, applied in memory; with True the script needs a Q_R.pkl that only the False branch writes).

tests/golden/etl_g13.npz:
  per trajectory k in (1, 2):  k{k}_{p_list_est,p_list_ref,dp_list,imu_list,contact_list,t265_list,mocap_list,ref_list}
      (len, width) float64, k{k}_time_list (len,), k{k}_depth_u8 (len_p, 4, 4) uint8 = the frames handed to cv2.imwrite
  fit_Q (12,12), fit_R (10,10)        Q_R.pkl as step 2 wrote it (fitted on the LAST trajectory; R from aliased KF.z, :74)
  k{k}_state_INPUT (len_p, 60), k{k}_state_MOCAP (len_p, 12), k{k}_state_T265 (len_p, 12), k{k}_p_trace (len_p,)
  run_R (10,10)                       the R the filter ran with (R[0:3] = 1e-4, :142-144)
  full_len_*, full_sum_*, full_pick_* the 430 / 4494 pass: list lengths, per-column sums, entries [0, 1, 2031, -2, -1]
The raw logs themselves are NOT stored: tests rebuild them from tests/golden_recipes.g13_raw (arithmetic-only recipe).
"""
import os
import pickle
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

os.environ["MPLBACKEND"] = "Agg"
import matplotlib                                     # noqa: E402
matplotlib.use("Agg")
import scipy.io                                       # noqa: E402

import casadi_eval                                    # noqa: E402
from golden_recipes import g13_raw, g13_leg_jacobian, G13_SEED, G13_ROWS, G13_CUTOFF, G13_END   # noqa: E402

IMWRITE_LOG = []


def install_standins():
    casadi_eval.install()
    cv2 = types.ModuleType("cv2")
    cv2.imwrite = lambda name, img: IMWRITE_LOG.append((name, np.array(img))) or True
    sys.modules["cv2"] = cv2
    sk = types.ModuleType("scaler_kin"); v3 = types.ModuleType("scaler_kin.v3")
    leg = types.ModuleType("scaler_kin.v3.SCALER_v2_Leg_6DOF_gripper")

    class Leg:
        leg_jacobian_3DoF = staticmethod(g13_leg_jacobian)
    leg.Leg = Leg
    sk.v3 = v3; v3.SCALER_v2_Leg_6DOF_gripper = leg
    sys.modules.update({"scaler_kin": sk, "scaler_kin.v3": v3, "scaler_kin.v3.SCALER_v2_Leg_6DOF_gripper": leg})
    if REF not in sys.path:
        sys.path.insert(0, REF)


def run_script(rel, scratch, edit=None):
    """exec the reference script `rel` with __file__ = <scratch>/OptiState/<rel>."""
    with open(os.path.join(REF, rel)) as fh:
        text = fh.read()
    if edit is not None:
        old, new = edit
        assert text.count(old) == 1, f"{rel}: expected exactly one {old!r}"
        text = text.replace(old, new)
    fake = os.path.join(scratch, "OptiState", rel)
    os.makedirs(os.path.dirname(fake), exist_ok=True)
    import matplotlib.pyplot as plt
    g = {"__file__": fake, "__name__": "__main__"}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(compile(text, fake, "exec"), g)
    plt.close("all")
    return g


def stack(lst):
    return np.stack([np.asarray(v, dtype=np.float64).reshape(-1) for v in lst])


LISTS = ("p_list_est", "p_list_ref", "dp_list", "imu_list", "contact_list", "t265_list", "mocap_list", "ref_list")


def main():
    install_standins()
    from settings import INITIAL_PARAMS
    assert (INITIAL_PARAMS.DATA_CUTOFF_START, INITIAL_PARAMS.DATA_CUTOFF_END) == (430, 4494)
    out = {}
    # ------------------------------------------------------------------ pass A: committed arrays, cutoffs 430 / 640
    scratch = tempfile.mkdtemp(prefix="g13_")
    try:
        traj_dir = os.path.join(scratch, "OptiState", "data_collection", "trajectories")
        os.makedirs(traj_dir); os.makedirs(os.path.join(scratch, "OptiState", "data_results"))
        scipy.io.savemat(os.path.join(traj_dir, "traj_a.mat"), g13_raw(G13_SEED, G13_ROWS))
        scipy.io.savemat(os.path.join(traj_dir, "traj_b.mat"), g13_raw(G13_SEED + 1, G13_ROWS, drop_rows=()))
        assert INITIAL_PARAMS.DATA_CUTOFF_START == G13_CUTOFF
        INITIAL_PARAMS.DATA_CUTOFF_END = G13_END
        run_script("data_collection/data_conversion_raw_to_Kalman.py", scratch)
        with open(os.path.join(traj_dir, "saved_trajectories.pkl"), "rb") as fh:
            saved = pickle.load(fh)
        assert sorted(saved.keys()) == [1, 2]
        for k in (1, 2):
            for name in LISTS:
                out[f"k{k}_{name}"] = stack(saved[k][name])
            out[f"k{k}_time_list"] = np.asarray(saved[k]["time_list"], dtype=np.float64)
            frames = [img for name, img in IMWRITE_LOG if f"saved_images_traj_{k}/" in name]
            out[f"k{k}_depth_u8"] = np.stack(frames)
            print(f"traj {k}:", {n: out[f'k{k}_{n}'].shape for n in LISTS}, "frames", len(frames))
        n_comb = sum(1 for name, _ in IMWRITE_LOG if "saved_images_combined/" in name)
        assert n_comb == sum(out[f"k{k}_depth_u8"].shape[0] for k in (1, 2))
        # step 2: fit + filter loop + rows
        g2 = run_script("data_collection/data_conversion_Kalman_to_Training.py", scratch, edit=("load_Q_R = True", "load_Q_R = False"))
        with open(os.path.join(traj_dir, "Q_R.pkl"), "rb") as fh:
            Qf, Rf = pickle.load(fh)
        with open(os.path.join(traj_dir, "rnn_data.pkl"), "rb") as fh:
            rnn = pickle.load(fh)
        # Q_R.pkl was dumped before the run loop set R[0:3] = 1e-4 on the live array (:142-144): the pickle holds the fitted R
        out["fit_Q"], out["fit_R"] = np.array(Qf), np.array(Rf)
        out["run_R"] = np.array(g2["R"])
        for k in (1, 2):
            for name in ("state_INPUT", "state_MOCAP", "state_T265"):
                out[f"k{k}_{name}"] = np.asarray(rnn[k][name], dtype=np.float64)
            out[f"k{k}_p_trace"] = np.asarray(g2["p_trace_list"][k - 1], dtype=np.float64)
            print(f"traj {k}: rows", out[f"k{k}_state_INPUT"].shape, "|x| max", np.abs(out[f"k{k}_state_INPUT"][:, :12]).max(),
                  "|f| max", np.abs(out[f"k{k}_state_INPUT"][:, 18:30]).max(), "p_trace max", out[f"k{k}_p_trace"].max())
        print("fit Q diag", np.diag(out["fit_Q"])); print("fit R diag", np.diag(out["fit_R"]))
        print("QPs solved:", len(casadi_eval.QP_LOG), "kkt max", max(q["kkt"]["stationarity"] for q in casadi_eval.QP_LOG))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    # ------------------------------------------------------------------ pass B: the shipped cutoffs 430 / 4494
    IMWRITE_LOG.clear()
    scratch = tempfile.mkdtemp(prefix="g13_")
    try:
        traj_dir = os.path.join(scratch, "OptiState", "data_collection", "trajectories")
        os.makedirs(traj_dir)
        scipy.io.savemat(os.path.join(traj_dir, "traj_full.mat"), g13_raw(G13_SEED + 2, 4500, drop_rows=(450, 3000)))
        INITIAL_PARAMS.DATA_CUTOFF_END = 4494
        run_script("data_collection/data_conversion_raw_to_Kalman.py", scratch)
        with open(os.path.join(traj_dir, "saved_trajectories.pkl"), "rb") as fh:
            saved = pickle.load(fh)
        for name in LISTS:
            a = stack(saved[1][name])
            out[f"full_len_{name}"] = np.array([a.shape[0]]); out[f"full_sum_{name}"] = a.sum(0)
            out[f"full_pick_{name}"] = a[[0, 1, 2031, -2, -1]]
        print("full pass lengths:", {n: int(out[f'full_len_{n}'][0]) for n in LISTS})
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    out["label"] = np.array(["scripts = reference, unmodified text (load_Q_R switch flipped); cv2 / scaler_kin = inert stand-ins; "
                             "QP formulation = reference, solver = certified stand-in (tools/casadi_eval.py)"])
    fn = os.path.join(ROOT, "tests", "golden", "etl_g13.npz")
    np.savez_compressed(fn, **out)
    print(f"{os.path.basename(fn)} {os.path.getsize(fn) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
