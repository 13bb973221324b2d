import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "anchors_targets.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(GOLDEN)


def unpack_annotations(z, prefix, n):
    anns = []
    for i in range(n):
        a = {k: z[f"{prefix}_ann{i}_{k}"] for k in
             ("labels", "bboxes", "poses", "segmentations", "cam_params", "mask_ids")}
        a["mask"] = [z[f"{prefix}_ann{i}_mask"]]
        anns.append(a)
    return anns
