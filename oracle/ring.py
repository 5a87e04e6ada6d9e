"""TEST INFRASTRUCTURE (oracle): ring-latent window gather / scatter on CPU.

Restates utils/shift_window_utils.py:14-206 of the reference:
  * get_dimension_slices_and_sizes  (:14-38)  -> ring_segments
  * RingLatent.get_window_latent    (:48-114) -> ring_gather
  * RingLatent.set_window_latent    (:116-206)-> ring_scatter
The reference builds the window from python slices + torch.cat; the restatement uses the
closed form `out[..., f, y, x] = P[..., (f0+f) % F, (y0+y) % H, (x0+x) % W]`, which is
what those slices enumerate (checked bit-exactly against the reference in
tests/test_oracle_golden.py).
"""
import numpy as np
import torch


def ring_segments(begin, end, size):
    """shift_window_utils.py:14-38 -- split [begin, end) on a ring of length `size`.

    Returns [(start, stop), ...] in traversal order (multi-wrap allowed)."""
    segs = []
    pos = begin
    while pos < end:
        start = pos % size
        nxt = (pos // size + 1) * size
        stop_pos = min(end, nxt)
        length = stop_pos - pos
        segs.append((start, start + length))
        pos = stop_pos
    return segs


def _check_window(lo, hi, size, what):
    # shift_window_utils.py:73-75 / :141-143
    assert 0 <= lo < hi <= size * 2, f"Invalid {what} [{lo}, {hi}) on ring of {size}"


def _defaults(shape, pos_left, pos_right, pos_top, pos_down, frame_begin, frame_end):
    F, H, W = shape[2], shape[3], shape[4]
    pos_left = 0 if pos_left is None else pos_left
    pos_right = W if pos_right is None else pos_right
    pos_top = 0 if pos_top is None else pos_top
    pos_down = H if pos_down is None else pos_down
    frame_begin = 0 if frame_begin is None else frame_begin
    frame_end = F if frame_end is None else frame_end
    _check_window(pos_left, pos_right, W, "pos_left/pos_right")
    _check_window(pos_top, pos_down, H, "pos_top/pos_down")
    _check_window(frame_begin, frame_end, F, "frame_begin/frame_end")
    return pos_left, pos_right, pos_top, pos_down, frame_begin, frame_end


def ring_gather(pano, pos_left=None, pos_right=None, pos_top=None, pos_down=None,
                frame_begin=None, frame_end=None):
    """RingLatent.get_window_latent (shift_window_utils.py:48-114). pano: [B,C,F,H,W]."""
    x0, x1, y0, y1, f0, f1 = _defaults(pano.shape, pos_left, pos_right, pos_top, pos_down,
                                       frame_begin, frame_end)
    F, H, W = pano.shape[2:]
    fi = torch.arange(f0, f1) % F
    yi = torch.arange(y0, y1) % H
    xi = torch.arange(x0, x1) % W
    return pano[:, :, fi][:, :, :, yi][:, :, :, :, xi].clone()


def ring_scatter(pano, tile, pos_left=None, pos_right=None, pos_top=None, pos_down=None,
                 frame_begin=None, frame_end=None):
    """RingLatent.set_window_latent (shift_window_utils.py:116-206): overwrite in place."""
    x0, x1, y0, y1, f0, f1 = _defaults(pano.shape, pos_left, pos_right, pos_top, pos_down,
                                       frame_begin, frame_end)
    F, H, W = pano.shape[2:]
    # :145-147 -- a window may not overlap itself
    assert x1 - x0 <= W and y1 - y0 <= H and f1 - f0 <= F, "warp should not occur"
    # :190
    assert tuple(tile.shape[2:]) == (f1 - f0, y1 - y0, x1 - x0), \
        f"Input latent shape {tuple(tile.shape[2:])} does not match window {(f1 - f0, y1 - y0, x1 - x0)}"
    fi = (torch.arange(f0, f1) % F)[:, None, None]
    yi = (torch.arange(y0, y1) % H)[None, :, None]
    xi = (torch.arange(x0, x1) % W)[None, None, :]
    pano[:, :, fi, yi, xi] = tile
    return pano


def ring_index_map(shape, x0, y0, f0, tw, th, tf):
    """Flat int64 index (into one [F,H,W] channel plane) of every window element, row-major
    [tf, th, tw].  Used by tests to state 'bit-exact tile index maps'."""
    F, H, W = shape[2:]
    fi = ((np.arange(f0, f0 + tf) % F)[:, None, None]).astype(np.int64)
    yi = ((np.arange(y0, y0 + th) % H)[None, :, None]).astype(np.int64)
    xi = ((np.arange(x0, x0 + tw) % W)[None, None, :]).astype(np.int64)
    return (fi * H + yi) * W + xi
