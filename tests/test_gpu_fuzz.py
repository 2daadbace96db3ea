"""Fixed-seed slices of the random-input tools under tools/ (every entry point of the C ABI
against the CPU oracle on drawn shapes, parameters and data): the same generators the long
developer runs use, a few hundred cases each, so that what those runs find stays under test."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))


def run(tool, *argv):
    saved = os.environ.get("SFM_PAIR_KERNEL")
    try:
        return importlib.import_module(tool).main(list(argv))
    finally:   # some tools walk through the kernel-selection switches
        if saved is None:
            os.environ.pop("SFM_PAIR_KERNEL", None)
        else:
            os.environ["SFM_PAIR_KERNEL"] = saved


@pytest.mark.parametrize("tool,argv", [
    ("hamming_fuzz", (300,)),               # knn2 + robust matches, every Hamming kernel, D = 1..64
    ("hamming_fuzz", (6, "big")),           # train sets around and above one 8192-row range
    ("l2_fuzz", (300,)),                     # L2 / L2SQR, every candidate pass
    ("lk_fuzz", (300,)),                     # windows 3..33, 0-4 levels, all criteria
    ("lk_fuzz", (200, 21)),                  # the 21 x 21 kernel (derivatives formed in the tracker)
    ("lk_fuzz", (80, 21, "batch")),         # several image pairs per call, padded rows
    ("corners_fuzz", (300,)),
    ("match_pairs_fuzz", (100,)),            # windows of pairs, both output formats
    ("triangulate_fuzz", (150,)),
    ("geometry_fuzz", (60,)),               # epipolar filter + essential matrix masks
    ("pose_fuzz", (150,)),                   # recoverPose
    ("epipolar_pairs_fuzz", (200,)),         # batched in-place epipolar filter
    ("corners_batch_fuzz", (150,)),          # batched corner detector, ragged content
    ("flow_chain_fuzz", (40,)),             # corners -> LK -> filter -> pose on the device
])
def test_fuzz_slice(tool, argv):
    assert run(tool, *argv) == 0
