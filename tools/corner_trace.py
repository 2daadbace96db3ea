import sys
sys.path.insert(0, '.')
import numpy as np
from sfm_amd import capi, synth
ctx = capi.Context(0)
img = synth.to_u8(synth.texture(720, 1280, seed=42))
for args in [(2000, 0.01, 10.0), (0, 0.01, 10.0), (500, 0.01, 30.0), (5000, 0.001, 3.0)]:
    print(args, len(ctx.good_features(img, *args)), flush=True)
