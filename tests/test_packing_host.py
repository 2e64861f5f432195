"""Host-side logic of the packed rows (no GPU): the one-upload fusion layout against the device-op formulation it replaces."""
import numpy as np
import pytest
import torch

from xfm_amd.packing import Pack, image_major_fusion_layout, image_major_layout


@pytest.mark.parametrize("seed,B,T", [(0, 8, 30), (1, 64, 30), (2, 5, 40)])
def test_host_fusion_layout_equals_device_ops(seed, B, T):
    """image_major_fusion_layout (numpy + one upload) gives the same pack, meta rows, cross-attention ranges, row-gather index and
    sequence start rows as image_major_layout + Pack.gather_index + index_select -- the 4B sequences of the pre-training fusion pass
    (positives | negative images | negative texts | MLM inputs) over 2B source sequences, drawn negatives and ragged lengths."""
    rng = np.random.default_rng(seed)
    lh = rng.integers(3, T + 1, size=B).tolist()
    lh[0], lh[-1] = T, 3
    im = rng.integers(0, B, size=B).tolist()
    tn = rng.integers(0, B, size=B).tolist()
    src = Pack.from_lens(lh + lh, T, torch.device("cpu"))        # the text tower's pack: clean | masked
    seq_len = lh + lh + [lh[j] for j in tn] + lh
    seq_img = list(range(B)) + im + list(range(B)) + list(range(B))
    seq_txt = list(range(B)) + list(range(B)) + tn + [B + j for j in range(B)]
    M = 7
    sel_off = list(range(3 * B)) + [3 * B + j * M for j in range(B)]
    sel_len = [1] * (3 * B) + [M] * B
    extra = (seq_txt, seq_img, sel_off, sel_len)
    fpack, _, _, meta, ranges = image_major_layout(seq_len, seq_img, B, T, "cpu", extra=extra)
    gidx = fpack.gather_index(src, meta[1].long())
    start_of = fpack.start.index_select(0, meta[0].long())
    src_start = np.concatenate([[0], np.cumsum(src.lens_host)[:-1]]).tolist()
    hp, hmeta, hranges, hgidx, hstart = image_major_fusion_layout(seq_len, seq_img, B, T, torch.device("cpu"), src_start, seq_txt, extra=extra)
    assert hp.cap == fpack.cap and hp.lens_host == fpack.lens_host and hp.exact
    assert torch.equal(hp.start, fpack.start) and torch.equal(hp.lens, fpack.lens)
    assert torch.equal(hmeta, meta)
    assert torch.equal(hranges[0], ranges[0]) and torch.equal(hranges[1], ranges[1]) and hranges[2] == ranges[2]
    assert torch.equal(hgidx, gidx) and int(hgidx.min()) >= 0      # exact packing: every row copies a source row
    assert torch.equal(hstart, start_of)
