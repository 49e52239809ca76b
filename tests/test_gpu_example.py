"""GPU: the end-to-end example (KF rows -> scaling -> windows -> training with the reference's loop -> banded predictions)."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--mpc"]], ids=["force-log", "mpc-forces"])
def test_pipeline_demo_trains_and_predicts(extra):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import pipeline_demo
    losses, mae = pipeline_demo.main(["--traj", "16", "--steps", "120", "--epochs", "4", "--hidden", "64", "--layers", "1"] + extra)
    assert losses[-1] < losses[0]            # the self-referential target of gru_train.py:237-244 is being fitted
    assert mae < 1.0
