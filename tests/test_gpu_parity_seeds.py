"""GPU: the full-size parity of the default (mixed-arithmetic) training step over SEVERAL seeds of weights, images and annotations.
Round 3 moved FPN + heads to the f16c8 arithmetic, which took the margin under the 1e-3 bar from ~10x to ~3.5x; one seed is not
evidence at that distance.  Default: three more seeds of BASELINE configs[1] beside tests/test_gpu_parity.py's; PP_PARITY_SEEDS=n runs
n (the round's sweep of 8 is profiles/r03_parity_seed_sweep.txt).  Run with -s for the margins."""
import os

import pytest

from tests.test_gpu_parity import _train_step_vs_oracle

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("PP_PARITY_SEEDS", "3"))


@pytest.fixture(scope="module")
def ctx():
    from pyrapose_amd.runtime import default_context
    return default_context()


@pytest.mark.parametrize("k", list(range(N_SEEDS)))
def test_config1_train_step_other_seeds(ctx, k):
    import bench
    from pyrapose_amd import arch
    from pyrapose_amd.utils import anchors as UA
    # (four images each -- tests/test_gpu_parity.py has the bench's batch of 8; the float64 oracle on the host is what these tests
    # cost, and the annotation density, not the batch, is what varies the gradient's support)
    B, H, W, C = 4, 480, 640, 13
    x, images, anns = bench.synth_batch(B, H, W, C, seed=2000 + 17 * k, boxes=(None, 1, 8, 15)[k % 4])
    tg = UA.anchor_targets_bbox_device(UA.anchors_for_shape_device((H, W)), images, anns, C)
    w, worst, total, _ = _train_step_vs_oracle(ctx, B, H, W, C, x, tg, arch.init_weights(C, seed=100 + k))
    assert max(w.values()) <= 1e-3 and worst[1] <= 1e-3
