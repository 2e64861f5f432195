"""Fixture for the resolution change of BEiT relative-position tables (beit2.py:753-849): the REFERENCE's own
`interpolate_pos_embed` run in the build container on a formula table, 224 px -> 384 px and -> 480 px.

The reference calls `scipy.interpolate.interp2d(x, y, z, kind='cubic')`, which SciPy >= 1.14 no longer has (the name survives as a stub
that raises).  For this run `scipy.interpolate.interp2d` is bound to oracle/fitpack_interp2d.interp2d_cubic -- a from-source restatement
of what interp2d did on a regular grid (FITPACK regrid with s = 0 + bispev), itself held to the installed SciPy's FITPACK
(RectBivariateSpline) at 1e-10 by tests/test_oracle_relpos_interp.py.  Everything else -- the geometric-progression source coordinates,
the target offsets, the orientation of z, the handling of the three extra (cls) entries, the per-head loop -- is the reference's code,
unmodified.  Writes tests/golden/relpos_interp.npz (inputs are regenerated from the formula generator; outputs are stored whole for
two heads and as moments for all twelve)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "oracle"))

import ref_shim  # noqa: E402
from oracle.fitpack_interp2d import interp2d_cubic  # noqa: E402
from xfm_amd import synthetic as syn  # noqa: E402


def main():
    ref_shim.install()
    import scipy.interpolate
    scipy.interpolate.interp2d = interp2d_cubic
    import models.beit2 as rb
    assert rb.interpolate.interp2d is interp2d_cubic
    heads = 12
    src = syn.symmetric("relpos_interp.table", (27 * 27 + 3, heads), 0.5)
    out = {}
    for res, grid in ((384, 24), (480, 30)):
        dst_n = (2 * grid - 1) ** 2 + 3

        class _Model:   # what interpolate_pos_embed reads from the model (beit2.py:768-771, 827)
            pos_embed = None
            patch_embed = type("P", (), {"patch_shape": (grid, grid), "num_patches": grid * grid})()

            def state_dict(self):
                return {"blocks.0.attn.relative_position_bias_table": torch.zeros(dst_n, heads)}

        ck = {"blocks.0.attn.relative_position_bias_table": src.clone(), "blocks.0.attn.relative_position_index": torch.zeros(1)}
        got = rb.interpolate_pos_embed(_Model(), ck)
        assert "blocks.0.attn.relative_position_index" not in got
        t = got["blocks.0.attn.relative_position_bias_table"]
        assert tuple(t.shape) == (dst_n, heads) and t.dtype == torch.float32
        out[f"table_{res}_heads_0_7"] = t[:, [0, 7]].numpy()
        out[f"table_{res}_sum"] = t.double().sum(0).numpy()
        out[f"table_{res}_sq"] = (t.double() ** 2).sum(0).numpy()
    meta = {"heads": heads, "src_entries": 27 * 27 + 3, "tag": "relpos_interp.table", "scale": 0.5,
            "generator": "tools/oracle/gen_relpos_interp.py (reference interpolate_pos_embed + FITPACK restatement for the removed interp2d)"}
    path = os.path.join(ROOT, "tests", "golden", "relpos_interp.npz")
    np.savez_compressed(path, meta=json.dumps(meta), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
