"""Unpadded ("packed") token rows for the text and fusion towers.

The reference pads every caption to the batch's longest one (dataset/pretrain_dataset.py:264-312, max_tokens = 30) and pushes the
padding through every Linear / LayerNorm / attention of the text tower (2B sequences) and the fusion tower (4B sequences); nothing
ever reads those rows (ITC / ITM take the [CLS] row, MLM the masked positions, padded keys are masked out).  With caption lengths
U[8, 30] that is 37 % of the token rows.  Here the towers can run on the real tokens only: a `Pack` describes where each sequence's
rows live inside a [cap, D] row buffer, the GEMM / LayerNorm kernels simply see fewer rows, and the attention kernels take the
(start, length) arrays (xfm_attn_args.q_start ...).  Values on every real token are what the padded computation gives: a padded
key contributes probability exp(-10000 + ...) == 0 in fp32, exactly like a key that is not there.

Lengths must be known on the HOST (they size buffers and grids) -- callers pass `text_lens`, computed from the CPU batch before it
is uploaded (xfm_amd.pretrain_loop) -- except for sequences picked by a device-side index (the hard-negative texts of ITM,
xfm.py:717-746): those get worst-case room (`cap` > rows in use) and their offsets are computed on the device, no sync.  Slack
rows hold zeros, stay finite through the towers, receive zero gradient and so add nothing to any weight gradient.
Attention masks are assumed to be prefix masks (ones then zeros), which is what tokenisers produce.
"""
import numpy as np
import torch

from . import functional as Fx

BF16, F32 = torch.bfloat16, torch.float32


class Pack:
    def __init__(self, start, lens, T, cap, lens_host=None):
        self.start, self.lens = start.contiguous(), lens.contiguous()   # int32 [B] on the device
        self.B, self.T, self.cap = int(start.numel()), int(T), int(cap)
        self.lens_host = None if lens_host is None else [int(x) for x in lens_host]
        self._row_map = None

    @property
    def pair(self):
        return (self.start, self.lens)

    @property
    def exact(self):
        """Every row of the buffer belongs to a sequence (no slack): kernels that only write real-token rows cover the buffer."""
        return self.lens_host is not None and sum(self.lens_host) == self.cap

    @classmethod
    def from_lens(cls, lens_host, T, device):
        """Exact packing of sequences whose lengths the host knows."""
        lh = np.asarray(lens_host, dtype=np.int64).reshape(-1)
        if lh.size == 0 or lh.min() < 1 or lh.max() > T:
            raise ValueError(f"packed sequences need 1 <= length <= {T}")
        start = np.concatenate([[0], np.cumsum(lh)[:-1]]).astype(np.int32)
        both = torch.from_numpy(np.stack([start, lh.astype(np.int32)])).to(device, non_blocking=True)
        return cls(both[0], both[1], T, int(lh.sum()), lens_host=lh.tolist())

    @classmethod
    def concat(cls, blocks, T):
        """blocks: list of (lens int32 device [n], cap int, lens_host or None).  Each block starts at the sum of the caps before it;
        inside a block the sequences are contiguous (device-side cumulative sum: no sync when lens is device data)."""
        starts, lens, host, base = [], [], [], 0
        for ln, cap, lh in blocks:
            ln = ln.to(torch.int32)
            starts.append((torch.cumsum(ln, 0, dtype=torch.int32) - ln) + base)
            lens.append(ln)
            host = None if (host is None or lh is None) else host + [int(x) for x in lh]
            base += int(cap)
        return cls(torch.cat(starts), torch.cat(lens), T, base, lens_host=host)

    def head(self, n):
        """The first n sequences (their rows are a prefix of the buffer)."""
        rows = sum(self.lens_host[:n]) if self.lens_host is not None else None
        p = Pack(self.start[:n], self.lens[:n], self.T, rows if rows is not None else self.cap,
                 None if self.lens_host is None else self.lens_host[:n])
        return p

    def rows_of_head(self, n):
        if self.lens_host is None:
            raise ValueError("row count of a sequence prefix needs host-known lengths")
        return sum(self.lens_host[:n])

    def row_map(self):
        """int32 [B*T]: packed row of token (b, t), -1 for padding."""
        if self._row_map is None:
            ar = torch.arange(self.T, device=self.start.device, dtype=torch.int32)
            m = self.start[:, None] + ar[None, :]
            self._row_map = torch.where(ar[None, :] < self.lens[:, None], m, torch.full_like(m, -1)).reshape(-1).contiguous()
        return self._row_map

    def gather_index(self, src, seq_src=None):
        """int32 [cap]: for every row of THIS layout the row of layout `src` it copies (sequence j <- src sequence seq_src[j], same
        token offset), -1 on slack rows.  Device-side, no sync."""
        dev = self.start.device
        ar = torch.arange(self.T, device=dev, dtype=torch.int32)
        s_start = src.start if seq_src is None else src.start.index_select(0, seq_src.long())
        vals = (s_start[:, None] + ar[None, :])
        dest = (self.start[:, None] + ar[None, :]).long()
        valid = ar[None, :] < self.lens[:, None]
        dest = torch.where(valid, dest, torch.full_like(dest, self.cap))          # invalid tokens land in a dummy slot
        out = torch.full((self.cap + 1,), -1, dtype=torch.int32, device=dev)
        out.scatter_(0, dest.reshape(-1), vals.reshape(-1).to(torch.int32))
        return out[:self.cap].contiguous()


def image_major_layout(seq_len, seq_img, n_images, T, device, extra=()):
    """Lay sequences out IMAGE BY IMAGE.  seq_len / seq_img: host lists (length and image of every sequence, caller's order).
    Returns (pack over the re-ordered sequences, order [new position -> caller index], pos_of [caller index -> new position],
    meta int32 device tensor with rows [pos_of, *extra re-ordered..., image starts (padded), image counts (padded)],
    ranges = (start int32 [U], count int32 [U], max count) of each image's contiguous query rows).  `extra`: host lists in the
    caller's order that the caller wants on the device in the NEW order (one upload for everything)."""
    n = len(seq_len)
    order = sorted(range(n), key=seq_img.__getitem__)      # stable
    pos_of = [0] * n
    for k, j in enumerate(order):
        pos_of[j] = k
    pack = Pack.from_lens([seq_len[j] for j in order], T, device)
    counts = [0] * n_images
    for j in range(n):
        counts[seq_img[j]] += seq_len[j]
    starts = [0] * n_images
    for u in range(1, n_images):
        starts[u] = starts[u - 1] + counts[u - 1]
    pad = [0] * (n - n_images)
    rows = [pos_of] + [[e[j] for j in order] for e in extra] + [starts + pad, counts + pad]
    meta = torch.tensor(rows, dtype=torch.int32).to(device, non_blocking=True)
    ranges = (meta[-2, :n_images].contiguous(), meta[-1, :n_images].contiguous(), max(counts))
    return pack, order, pos_of, meta, ranges


_PINNED = {}


def _upload_i32(arr, device):
    """One int32 host array -> device through a ring of pinned staging buffers (a pageable upload blocks the host and goes through a
    bounce buffer; this one is a plain asynchronous copy on the current stream).  The ring is 8 deep; every slot carries the event of
    its last copy and the host waits for it before rewriting the slot (normally long past: eight uploads ago) -- a path without a
    per-step host sync (caller-supplied negatives, XFM_PACK_SYNC=0) can run ahead of the stream, and a rewritten staging buffer would
    upload the NEXT step's layout indices silently."""
    if device.type != "cuda":
        return torch.from_numpy(arr).to(device)
    n = int(arr.size)
    key = (device.index, max(1024, 1 << (n - 1).bit_length()))
    ring = _PINNED.get(key)
    if ring is None:
        ring = _PINNED[key] = [[torch.empty(key[1], dtype=torch.int32).pin_memory() for _ in range(8)], 0, [None] * 8]
    slot = ring[1]
    buf = ring[0][slot]
    ring[1] = (slot + 1) % 8
    if ring[2][slot] is not None:
        ring[2][slot].synchronize()
    buf[:n].numpy()[:] = arr.reshape(-1)
    out = buf[:n].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ring[2][slot] = ev
    return out


def image_major_fusion_layout(seq_len, seq_img, n_images, T, device, src_start, seq_src, extra=()):
    """image_major_layout + everything the packed fusion pass derives from it, computed on the HOST and uploaded ONCE (the step's
    launch stream is empty while this runs -- the host has just read the drawn negatives back -- so every small device op and every
    pageable upload here is GPU idle time: 0.44 ms -> ~0.15 ms of the pre-training step).
    src_start: host start rows of the SOURCE layout (the text tower's pack); seq_src[j]: source sequence the caller's sequence j copies.
    Returns (pack, meta [rows as image_major_layout], ranges, gather index int32 [cap] (row of the source layout, -1 = none),
    start_of int32 [n] (first row of every sequence, caller's order))."""
    n = len(seq_len)
    sl, si = np.asarray(seq_len, dtype=np.int64), np.asarray(seq_img, dtype=np.int64)
    order = np.argsort(si, kind="stable")
    pos_of = np.empty(n, dtype=np.int64)
    pos_of[order] = np.arange(n)
    lens_new = sl[order]
    if lens_new.min() < 1 or lens_new.max() > T:
        raise ValueError(f"packed sequences need 1 <= length <= {T}")
    start_new = np.concatenate([[0], np.cumsum(lens_new)[:-1]])
    cap = int(lens_new.sum())
    counts = np.bincount(si, weights=sl, minlength=n_images).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    pad = np.zeros(n - n_images, dtype=np.int64)
    rows = [pos_of] + [np.asarray(e, dtype=np.int64)[order] for e in extra] + [np.concatenate([starts, pad]), np.concatenate([counts, pad])]
    # gather index: row r of sequence k (new order) copies row src_start[seq_src[order[k]]] + (r - start_new[k]) of the source layout
    src0 = np.asarray(src_start, dtype=np.int64)[np.asarray(seq_src, dtype=np.int64)[order]]
    gidx = np.repeat(src0 - start_new, lens_new) + np.arange(cap)
    start_of = start_new[pos_of]
    nm = len(rows)
    blob = np.concatenate([start_new, lens_new] + rows + [start_of, gidx]).astype(np.int32)
    dev = _upload_i32(blob, device)
    pack = Pack(dev[:n], dev[n:2 * n], T, cap, lens_host=lens_new.tolist())
    meta = dev[2 * n:(2 + nm) * n].view(nm, n)
    ranges = (meta[-2, :n_images], meta[-1, :n_images], int(counts.max()))
    return pack, meta, ranges, dev[(3 + nm) * n:], dev[(2 + nm) * n:(3 + nm) * n]


class _RowsGatherFn(torch.autograd.Function):
    """out[r] = rows[index[r]] (zeros where index < 0); backward = scatter-add in fp32."""

    @staticmethod
    def forward(ctx, rows, index):
        rows = rows.contiguous()
        ctx.n, ctx.dtype = rows.shape[0], rows.dtype
        ctx.index = index
        if rows.dtype == F32:   # (the fp32 twin of a tower output: gathered as it is)
            return Fx.rows_gather(rows, index)
        return Fx.rows_gather(rows if rows.dtype == BF16 else rows.to(BF16), index)

    @staticmethod
    def backward(ctx, dy):
        if ctx.dtype == F32:
            keep = ctx.index >= 0
            acc = torch.zeros((ctx.n, dy.shape[1]), dtype=F32, device=dy.device)
            return acc.index_add_(0, ctx.index.clamp_min(0).long(), dy.float() * keep[:, None]), None
        dy = (dy if dy.dtype == BF16 else dy.to(BF16)).contiguous()
        acc = torch.zeros((ctx.n, dy.shape[1]), dtype=F32, device=dy.device)
        Fx.rows_scatter_add(dy, ctx.index, acc)
        return acc.to(ctx.dtype), None


def rows_gather(rows, index):
    """Differentiable row gather: rows bf16 [n, D], index int32 [R] (device) -> [R, D]."""
    return _RowsGatherFn.apply(rows, index.to(torch.int32).contiguous())


def unpack(rows, pack):
    """Packed rows [cap, D] -> the reference's padded layout [B, T, D] (zeros at padding)."""
    return rows_gather(rows, pack.row_map()).view(pack.B, pack.T, rows.shape[-1])


def pack_rows(x, pack):
    """Padded [B, T, D] -> packed rows [cap, D] (slack rows zero)."""
    dense = Pack(torch.arange(pack.B, device=x.device, dtype=torch.int32) * pack.T, pack.lens, pack.T, pack.B * pack.T)
    return rows_gather(x.reshape(pack.B * pack.T, x.shape[-1]), pack.gather_index(dense))


def lens_from_mask(text_atts):
    """Host-side lengths of a CPU prefix mask [B, T] (raises when the mask is not a prefix mask)."""
    m = text_atts if not text_atts.is_cuda else None
    if m is None:
        raise ValueError("lens_from_mask wants the CPU copy of the mask (a device mask would cost a sync)")
    m = m.to(torch.int64)
    lens = m.sum(1)
    T = m.shape[1]
    if not bool((m == (torch.arange(T)[None, :] < lens[:, None]).to(torch.int64)).all()):
        raise ValueError("attention mask is not a prefix mask: packed rows need ones followed by zeros")
    return lens
