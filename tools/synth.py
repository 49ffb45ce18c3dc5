"""Synthetic inputs of bench.py and the GPU tests: integer-only frames, numpy.  Neutral module: it belongs neither to the
product nor to the oracle (oracle/frontend.c holds the same generator in C for its own use; tests/test_oracle_frontend.py holds the
two bit-identical).

    I(x, y) = clamp(ramp + tex + blobs),  ramp = x*96/(W-1) + y*64/(H-1),
    tex     = xorshift32(seed ^ bx*73856093 ^ by*19349663) & tex_mask   per 16x16 block of the (shifted) texture plane,
    blobs   = n_blobs squares (3x3 for odd k, 5x5 for even k) of +-blob_amp at positions from one xorshift32 stream.

A sequence translates texture and blobs by (shift_x, shift_y) pixels per frame, so consecutive frames really match.
The defaults (tex_mask 63, 4096 blobs) give 7-9 % of the pixels a FAST response -- several times denser than camera images;
`sparse=True` (tex_mask 15: block steps stay under the FAST threshold, 1024 blobs) gives the 1-2 % of real imagery.
"""
import numpy as np

_M32 = 0xFFFFFFFF


def _xs32(s):
    s ^= (s << 13) & _M32
    s ^= s >> 17
    s ^= (s << 5) & _M32
    return s


def _xs32_np(s):
    s = s.astype(np.uint32)
    s ^= s << np.uint32(13)
    s ^= s >> np.uint32(17)
    s ^= s << np.uint32(5)
    return s


class SequenceSynth:
    """All frames of one sequence (one seed): the texture and blob planes are built once on a canvas that covers every shift up to
    (max_shift_x, max_shift_y); frame (sx, sy) is a window of them plus the (unshifted) ramp, clipped to 8 bits."""

    def __init__(self, w, h, seed, max_shift_x=0, max_shift_y=0, tex_mask=63, n_blobs=4096, blob_amp=80, sparse=False):
        if sparse:
            tex_mask, n_blobs = 15, 1024
        self.w, self.h, self.msx, self.msy = w, h, max_shift_x, max_shift_y
        W, H = w + max_shift_x, h + max_shift_y
        x = np.arange(w, dtype=np.int64)[None, :]
        y = np.arange(h, dtype=np.int64)[:, None]
        self.ramp = ((x * 96) // (w - 1) + (y * 64) // (h - 1)).astype(np.int16)
        bx = (np.arange(W, dtype=np.int64)[None, :] >> 4).astype(np.uint32)
        by = (np.arange(H, dtype=np.int64)[:, None] >> 4).astype(np.uint32)
        s = np.uint32(seed & _M32) ^ (bx * np.uint32(73856093)) ^ (by * np.uint32(19349663))
        s = np.where(s == 0, np.uint32(0x9E3779B9), s)
        self.tex = (_xs32_np(s) & np.uint32(tex_mask)).astype(np.int16)
        st = (seed * 2654435761 + 12345) & _M32
        if st == 0:
            st = 1
        cx = np.empty(n_blobs, np.int64); cy = np.empty(n_blobs, np.int64); sg = np.empty(n_blobs, np.int64)
        for k in range(n_blobs):
            st = _xs32(st); cx[k] = st % w
            st = _xs32(st); cy[k] = st % h
            st = _xs32(st); sg[k] = blob_amp if (st & 1) else -blob_amp
        # blob k covers canvas columns cx-r .. cx+r (canvas X = x + shift_x); a frame clips it to its own window, so the
        # canvas keeps a margin of 2 on the low side
        self.blob = np.zeros((H + 4, W + 4), np.int32)
        for r, sel in ((2, slice(0, None, 2)), (1, slice(1, None, 2))):      # even k: 5x5, odd k: 3x3 (the sums commute)
            d = np.arange(-r, r + 1)
            xs = cx[sel][:, None, None] + d[None, None, :] + 0 * d[None, :, None] + 2
            ys = cy[sel][:, None, None] + d[None, :, None] + 0 * d[None, None, :] + 2
            v = np.broadcast_to(sg[sel][:, None, None], xs.shape)
            np.add.at(self.blob, (ys.ravel(), xs.ravel()), v.ravel().astype(np.int32))

    def frame(self, sx=0, sy=0):
        assert 0 <= sx <= self.msx and 0 <= sy <= self.msy
        w, h = self.w, self.h
        acc = self.ramp + self.tex[sy:sy + h, sx:sx + w] + self.blob[2 + sy:2 + sy + h, 2 + sx:2 + sx + w].astype(np.int16)
        return np.clip(acc, 0, 255).astype(np.uint8)


def synth_frame(w, h, seed, shift_x=0, shift_y=0, **kw):
    return SequenceSynth(w, h, seed, shift_x, shift_y, **kw).frame(shift_x, shift_y)


def synth_sequences(n, w, h, base_seed, n_seq=8, sparse=False):
    """n frames as n_seq sequences of n/n_seq frames: frame i of sequence q = synth(base_seed + q, shift (2i, i))."""
    per = max(n // n_seq, 1)
    out = np.empty((n, h, w), np.uint8)
    for q in range((n + per - 1) // per):
        g = SequenceSynth(w, h, base_seed + q, 2 * (per - 1), per - 1, sparse=sparse)
        for i in range(per):
            if q * per + i < n:
                out[q * per + i] = g.frame(2 * i, i)
    return out


def synth_descriptor_pair(p, nq=2000, nt=2000, n_inlier=1400, flip=0.08):
    """C3: query set = nq x 256 random bits (seed p); target set = n_inlier rows of it, permuted, each bit flipped with
    probability `flip` (mean distance ~20), then nt - n_inlier random rows (mean distance 128)."""
    rng = np.random.Generator(np.random.Philox(key=p))
    q = rng.integers(0, 2 ** 32, (nq, 8), dtype=np.uint64).astype(np.uint32)
    perm = rng.permutation(nq)[:n_inlier]
    flips = np.packbits(rng.random((n_inlier, 256)) < flip, axis=1, bitorder="little").view(np.uint32)
    t = np.concatenate([q[perm] ^ flips, rng.integers(0, 2 ** 32, (nt - n_inlier, 8), dtype=np.uint64).astype(np.uint32)])
    return q, t, perm
