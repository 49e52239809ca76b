"""CPU: the register / scratch budget of the shipped library, read from its gfx950 code objects (tools/codeobj_report.py:
`.hip_fatbin` -> offload bundles -> NT_AMDGPU_METADATA via llvm-readelf).  A kernel that a reference-shaped call reaches must
not touch scratch memory: a spill is an HBM round trip per lane inside a loop that is otherwise register / LDS resident.
The list below is the contract; everything else with scratch is printed (run with -s) and tracked in profiles/r05_scratch.md.
"""
import os
import re
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj_report as cr   # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(cr.LIB) or not os.path.exists(os.path.join(cr.LLVM, "llvm-readelf")),
                                reason="needs the built library and llvm-readelf")

# kernel (demangled, as tools/codeobj_report.py prints it) -> which reference-shaped call reaches it
SCRATCH_FREE = {
    # BASELINE configs[2]: the headline kernel and its full-Q twin; GRU(60,64,L>1) writes layer 0's sequence (SEQOUT)
    "osf::fused_kf_gru_kernel_v2<true, false, 2>": "fused KF+GRU headline (diagonal Q), 64 trajectories per wavefront",
    "osf::fused_kf_gru_kernel_v2<false, false, 2>": "fused KF+GRU, full-matrix Q",
    "osf::fused_kf_gru_kernel_v2<true, true, 2>": "fused KF+GRU(60,64,L>1): layer 0 with its sequence written",
    # round 6: the tile shapes of the shards of a fixed 65,536 batch (os_fused_set_tile)
    "osf::fused_kf_gru_kernel_v2<true, false, 1>": "32 trajectories per wavefront (a 32,768 shard)",
    "osf::fused_kf_gru_kernel_v2<false, false, 1>": "the same with a full-matrix Q",
    "osf::fused_kf_gru_kernel_v2<true, true, 1>": "the same, layer 0 of GRU(60,64,L>1)",
    "osf::fused_kf_gru_kernel_v2<false, true, 1>": "the same, full-matrix Q",
    "osf::fused_kf_gru_kernel_v3<true, 1>": "16-trajectory tiles on the 16x16x4 MFMA (a 16,384 shard)",
    "osf::fused_kf_gru_kernel_v3<true, 2>": "a tile's hidden units over two wavefronts (an 8,192 shard)",
    "osf::fused_kf_gru_kernel_v3<true, 4>": "a tile's hidden units over four wavefronts (4,096 trajectories and below)",
    "osf::fused_kf_gru_kernel_v3<false, 1>": "16-trajectory tiles, full-matrix Q",
    "osf::fused_kf_gru_kernel_v3<false, 2>": "two wavefronts per tile, full-matrix Q",
    "osf::fused_kf_gru_kernel_v3<false, 4>": "four wavefronts per tile, full-matrix Q",
    # BASELINE configs[1] and the KF-only large batch
    "osk::kf_run_sym_kernel<0, true, false>": "KF only, B = 65,536",
    "osk::kf_run_sym_kernel<0, true, true>": "KF only with p_rot",
    "osk::kf_run_rows2_kernel<false, false, false>": "KF only, B = 4,096 x 1,000 (configs[1])",
    "osk::kf_run_rows2_kernel<true, false, true>": "small-batch KF with aux outputs",
    # the GRU layer kernels of the reference's model shapes
    "osg::gru_layer_stage_kernel<4>": "RNN(188,128,4) at the headline batch",
    "osg::gru_layer_stage_kernel<2>": "GRU(60,64,4) layers 1..3 at the headline batch",
    "osg::gru_layer_ahead_kernel": "training / windows forward, H = 128, one tile per CU",
    "osf::fused_kf_gru_kernel_v2<false, true, 2>": "the same with a full-matrix Q",
    "osg::gru_layer_split_kernel<4, false, false>": "H = 128 small batches",
    "osg::gru_layer_split_kernel<4, true, false>": "H = 128 small batches, training forward",
    "osg::gru_layer_split_kernel<2, false, false>": "H = 64 small batches",
    "osg::gru_layer_split_kernel<2, true, false>": "H = 64 small batches, training forward",
    "osg::gru_layer_split_kernel<4, false, true>": "window-stream inference, layer 0 of RNN(188,128,4) (os_gru_forward_windows)",
    "osg::gru_layer_split_kernel<2, false, true>": "window-stream inference, layer 0 of an H = 64 model",
    "osg::gru_layer_bf16_kernel<3>": "opt-in split-bf16 layer GEMMs (os_gru_set_split_bf16), three terms",
    "osg::gru_layer_bf16_kernel<2>": "the same, two terms",
    "osg::gru_gi_kernel<1>": "window-stream inference: rows . W_ih^T once per row",
    "osg::gru_gi_kernel<2>": "the same, 64-row tiles",
    "osg::gru_wide_kernel<false>": "the reference's own windows as a batch (2 <= layers, H = 128, up to 512 windows): four CUs per (layer, tile)",
    "osg::gru_wide_kernel<true>": "batch-64 training forward (gru/gru_train.py:36)",
    "osg::gru_stack_kernel<4, false>": "H = 128 batches of 513 .. 2,048 windows (and OS_GRU_WIDE=0)",
    "osg::gru_stack_kernel<4, true>": "batch-64 training forward (gru/gru_train.py:36)",
    "osg::gru_stack_kernel<2, false>": "layer-pipelined stack, H = 64",
    "osg::gru_stack_kernel<2, true>": "layer-pipelined stack, H = 64, training forward",
    "osg::gru_vec_kernel<128>": "one window per call (gru/gru_test.py:157-177)",
    "osg::gru_vec_kernel<64>": "one window per call, H = 64",
    "osg::gru_layer_kernel<2, 2>": "H = 128 fallback when the stage kernel steps aside",
    # BASELINE configs[3]: the training step
    "ost::bwd_sweep_kernel<1, 8, 0>": "training backward sweep",
    "ost::bwd_sweep_wide_kernel": "batch-64 training backward (gru/gru_train.py:36): all layers in one launch, four CUs per (layer, tile)",
    "ost::bwd_sweep_stack_kernel": "training backward of 513 .. 2,048 windows, all layers in one launch (and OS_GRU_WIDE=0)",
    "ost::dw2_kernel<6>": "training weight gradients, the 188-wide first layer",
    "ost::dw2_kernel<4>": "training weight gradients, the 128-wide layers",
    "ost::dw2_kernel<2>": "training weight gradients, 60 / 64-wide layers",
    "ost::dw_kernel<false, 4>": "weight gradients of an epoch's last, partial batch (B % 32 != 0 with the batch_first input): two passes for 188 columns",
    "ost::dw_kernel<false, 2>": "the same, 60 / 64-wide layers",
    # estimate_state_mpc -- the reference's real loop
    # NOT HELD: "osm::kf_mpc_persistent_kernel<1>" (estimate_state_mpc at the reference's shape): 344 B = the callee-saved VGPRs of the QP
    # call's ABI, written at call entry and read back at its exit, nothing inside a loop body; the two ways around the call that were
    # built and measured (all four solver instances inlined in a QP wave: 460 registers, 44 us per step against 33; one QP wave per
    # leg count: 5 waves, 256-register cap, spills) are slower -- DESIGN.md section 4.5
    "osk::kf_predict_rows_kernel<true, float>": "Engine.kf_predict with body_ref: the Q / R fitter's predict_mpc batch (pipeline.fit_noise_covariances)",
    "osk::kf_predict_rows_kernel<false, float>": "Engine.kf_predict (os_kf_predict)",
    "osk::kf_predict_rows_kernel<true, double>": "the split predict_mpc -> update sequence with a float64 P (OS_KF_P_FLOAT64)",
    "osk::kf_predict_rows_kernel<false, double>": "os_kf_predict, float64 P",
    "osk::kf_update_rows_kernel<false, float>": "Engine.kf_update, batch form (os_kf_update)",
    "osk::kf_update_rows_kernel<true, float>": "Engine.kf_update, sequential form",
    "osk::kf_update_rows_kernel<false, double>": "os_kf_update, float64 P, batch form",
    "osk::kf_update_rows_kernel<true, double>": "os_kf_update, float64 P, sequential form",
    # BASELINE configs[5]: the depth encoder at the reference's shape (197 tokens, 4 heads x 32)
    "osv::attention_mfma_dma_kernel<7, 8>": "ViT attention, persistent workgroups with LDS-DMA K / V (transformer/transformer_model.py:113-135)",
    "osv::attention_mfma_kernel<7, 8>": "the same, one workgroup per head (OS_VIT_ATT_DMA=0)",
    "osv::attention_kernel<32>": "ViT attention for other token counts (non-default img_size)",
    "osv::attention_kernel<64>": "ViT attention with a head dimension of 64 (non-default num_heads / embed_dim)",
    "osv::vit_mlp_kernel": "ViT MLP block",
    "osv::vit_gemm_kernel<1, 3>": "ViT patch embedding / qkv / projection GEMMs",
    "osm::mpc_solve_kernel<1>": "the force QP (os_mpc_solve, the launch sequence of os_kf_mpc_run), one leg on the ground",
    "osm::mpc_solve_kernel<2>": "the same, trot (two legs)",
    "osm::mpc_solve_kernel<3>": "three legs",
    "osm::mpc_solve_kernel<4>": "four legs",
    "osk::kf_dense_rows_kernel<false, false, false, true>": "predict_mpc covariance + batch update, float64, 16 lanes per trajectory",
    "osk::kf_dense_rows_kernel<true, false, false, true>": "the same with the sequential update (diagonal R)",
    "osk::kf_dense_rows_kernel<false, true, false, true>": "batch update with P_trace / K_gain outputs",
    "osk::kf_dense_rows_kernel<true, true, false, true>": "sequential update with P_trace / K_gain outputs",
    "osk::kf_dense_rows_kernel<false, false, false, false>": "predict(p,f) covariance + BATCH update (non-diagonal R / sequential=False), float64",
    "osk::kf_dense_rows_kernel<false, true, false, false>": "the same with P_trace / K_gain outputs",
    "osk::kf_dense_rows_kernel<false, false, true, false>": "the same as the first kernel of the two-kernel fused path",
    "osk::kf_dense_rows_kernel<true, false, false, false>": "predict(p,f) covariance + sequential update with the full P (non-symmetric Q / symmetric=False), float64",
    "osk::kf_dense_rows_kernel<true, true, false, false>": "the same with P_trace / K_gain outputs",
    "osk::kf_dense_rows_kernel<true, false, true, false>": "the same as the first kernel of the two-kernel fused path",
    "osk::kf_dense_rows_kernel<false, false, true, true>": "feature rows for the two-kernel fused path (dense F_d)",
}


@pytest.fixture(scope="module")
def rows():
    return {r["name"]: r for r in cr.report()}


def test_metadata_is_readable_and_covers_the_library(rows):
    assert len(rows) > 100
    hk = rows["osf::fused_kf_gru_kernel_v2<true, false, 2>"]
    assert hk["vgpr_count"] > 256 and hk["agpr_count"] > 0          # the headline kernel uses both register halves


@pytest.mark.parametrize("name", sorted(SCRATCH_FREE))
def test_reference_shaped_kernels_use_no_scratch(rows, name):
    assert name in rows, f"{name} is not in the library (renamed? update the list): {[k for k in rows if name.split('<')[0] in k]}"
    r = rows[name]
    assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, \
        f"{name} ({SCRATCH_FREE[name]}): scratch {r['private_segment_fixed_size']} B/lane, {r['vgpr_spill_count']} VGPR spills"


def test_report_the_rest(rows, capsys):
    rest = [r for n, r in sorted(rows.items()) if r["private_segment_fixed_size"] and n not in SCRATCH_FREE]
    with capsys.disabled():
        print("\nkernels outside the scratch-free contract that use scratch (B/lane):")
        for r in rest:
            print(f"  {r['name']}: {r['private_segment_fixed_size']} (VGPR spills {r['vgpr_spill_count']})")
    # nothing outside the list may grow unnoticed past a page of spills
    assert all(r["private_segment_fixed_size"] <= 6000 for r in rest)
