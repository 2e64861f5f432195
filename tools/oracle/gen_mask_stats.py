"""Statistics of the REAL reference's block-wise MIM mask generator (models/masking_generator.py:27-105) -> tests/golden/mim_mask_stats.npz:
per-patch mask frequency, the histogram of how many new patches each accepted block added, the histogram of the number of grid rows a
mask touches and the 2-point co-occurrence at offsets (0,1), (1,0), (1,1) -- what the device sampler (csrc/elementwise.hip
mim_masks_kernel) is pinned against.  Runs in the build container only; the reference file is loaded where it lies (it imports
random / math / numpy only)."""
import importlib.util
import json
import os
import random
import sys

import numpy as np

REF = "/root/reference/models/masking_generator.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "mim_mask_stats.npz")


def main(n=50000, grid=14, num=75, min_num=16):
    spec = importlib.util.spec_from_file_location("ref_masking_generator", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    deltas = np.zeros(grid * grid + 1, dtype=np.int64)

    class Recording(mod.MaskingGenerator):
        def _mask(self, mask, max_mask_patches):
            d = super()._mask(mask, max_mask_patches)
            if d > 0:
                deltas[d] += 1
            return d

    random.seed(20260101)
    np.random.seed(20260101)
    g = Recording(input_size=grid, num_masking_patches=num, min_num_patches=min_num)
    freq = np.zeros((grid, grid), dtype=np.int64)
    rows = np.zeros(grid + 1, dtype=np.int64)
    pair = np.zeros(3, dtype=np.float64)
    for _ in range(n):
        m = g()
        assert m.sum() == num
        freq += m
        rows[int((m.sum(1) > 0).sum())] += 1
        pair[0] += (m[:, :-1] * m[:, 1:]).mean()
        pair[1] += (m[:-1, :] * m[1:, :]).mean()
        pair[2] += (m[:-1, :-1] * m[1:, 1:]).mean()
    meta = {"n": n, "grid": grid, "num": num, "min_num": min_num, "source": "models/masking_generator.py:27-105, random.seed / np.random.seed 20260101"}
    np.savez_compressed(OUT, freq=freq.reshape(-1), deltas=deltas, rows=rows, pair=pair / n, meta=json.dumps(meta))
    print("wrote", OUT, "mean freq", freq.mean() / n, "blocks per mask", deltas.sum() / n)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50000)
