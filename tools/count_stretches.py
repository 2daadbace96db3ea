"""How many queries of the 20 000 x 20 000 x 128 sets the narrow candidate pass leaves to the
row-by-row scoring of one stretch (SFM_L2_COUNT_STRETCHES=1 makes the exhaustive-pass counter
include them); run on a GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sfm_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
for integer in (False, True):
    q, t = synth.float_descriptors(20000, 20000, dim=128, seed=99, integer=integer)
    for flag in ("0", "1"):
        os.environ["SFM_L2_COUNT_STRETCHES"] = flag
        ctx.knn2(capi.NORM_L2SQR, q, t)
        print("integer-valued" if integer else "unit-norm", "rows; stretches counted:", flag, "->",
              ctx.l2_exhaustive_count(), flush=True)
