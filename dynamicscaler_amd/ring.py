"""Wrap-around panorama latent on the GPU (drop-in for utils/shift_window_utils.py:RingLatent, :40-206) and
the host-side window arithmetic of the ring pipelines.

Gather / scatter are single HIP launches (ds_ring_gather / ds_ring_scatter3) with `(idx % size)` addressing
instead of the reference's slice lists + torch.cat; `ring_segments` keeps the reference's slice enumeration
(:14-38) as a host mirror for tests and diagnostics.
"""
import torch

from . import ops

VAE_SCALE_FACTOR = 8  # pipeline/t2v_normal_pipeline.py:48


def get_dimension_slices_and_sizes(begin, end, size):
    """Host mirror of utils/shift_window_utils.py:14-38 (same return type: list of slices, list of sizes)."""
    slices, sizes = [], []
    pos = begin
    while pos < end:
        start = pos % size
        stop_pos = min(end, (pos // size + 1) * size)
        n = stop_pos - pos
        slices.append(slice(start, start + n))
        sizes.append(n)
        pos = stop_pos
    return slices, sizes


class RingLatent:
    def __init__(self, init_latent):
        assert len(init_latent.shape) == 5, f"[RingLatent.__init__] init_latent shape {init_latent.shape} not legal"
        if not init_latent.is_cuda:
            raise RuntimeError("RingLatent lives on the GPU in this build (no CPU path); pass a HIP tensor")
        self.torch_latent = init_latent.clone().contiguous()

    def get_shape(self):
        return self.torch_latent.shape

    def _window(self, pos_left, pos_right, pos_top, pos_down, frame_begin, frame_end):
        shape = self.get_shape()
        depth, height, width = shape[2], shape[-2], shape[-1]
        pos_left = 0 if pos_left is None else pos_left
        pos_right = width if pos_right is None else pos_right
        pos_top = 0 if pos_top is None else pos_top
        pos_down = height if pos_down is None else pos_down
        frame_begin = 0 if frame_begin is None else frame_begin
        frame_end = depth if frame_end is None else frame_end
        assert 0 <= pos_left < pos_right <= width * 2, f"Invalid pos_left {pos_left} and pos_right {pos_right}"
        assert 0 <= pos_top < pos_down <= height * 2, f"Invalid pos_top {pos_top} and pos_down {pos_down}"
        assert 0 <= frame_begin < frame_end <= depth * 2, f"Invalid frame_begin {frame_begin} and frame_end {frame_end}"
        return pos_left, pos_right, pos_top, pos_down, frame_begin, frame_end

    def get_window_latent(self, pos_left=None, pos_right=None, pos_top=None, pos_down=None, frame_begin=None,
                          frame_end=None):
        l, r, t, d, fb, fe = self._window(pos_left, pos_right, pos_top, pos_down, frame_begin, frame_end)
        assert self.torch_latent.shape[0] == 1, "batch 1 only (the reference's init repeat is only valid for batch 1)"
        tiles, _ = ops.ring_gather(self.torch_latent, [(fb, t, l)], (fe - fb, d - t, r - l))
        return tiles  # [1, C, tf, th, tw]

    def set_window_latent(self, input_latent, pos_left=None, pos_right=None, pos_top=None, pos_down=None,
                          frame_begin=None, frame_end=None):
        l, r, t, d, fb, fe = self._window(pos_left, pos_right, pos_top, pos_down, frame_begin, frame_end)
        depth, height, width = self.get_shape()[2], self.get_shape()[-2], self.get_shape()[-1]
        assert r - l <= width, "warp should not occur"
        assert d - t <= height, "warp should not occur"
        assert fe - fb <= depth, "warp should not occur"
        assert tuple(input_latent.shape[2:]) == (fe - fb, d - t, r - l), \
            f"Input latent shape {tuple(input_latent.shape[2:])} does not match target window shape {(fe - fb, d - t, r - l)}"
        src = input_latent.to(self.torch_latent.dtype).contiguous()
        ops.ring_scatter3(self.torch_latent, None, None, src, None, [(fb, t, l)])


def ring_axis_steps(total, tile, num_windows, loop_step):
    """One axis of the overlapped-ring grid, pixels in / latent units out
    (pipeline/t2v_sphere_panorama_pipeline.py:437-476).  Returns (overlap_ratio, window_step, offset_step)."""
    overlap = 1 - (total / tile - 1) / (num_windows - 1)
    window_step = int(tile * (1 - overlap)) // VAE_SCALE_FACTOR
    offset_step = int((1 - overlap) * tile / loop_step) // VAE_SCALE_FACTOR
    if num_windows == 1:
        offset_step = 0
    return overlap, window_step, offset_step


def t2v_ring_windows(i, *, latent_h, latent_w, frames, total_latent_h, step_w, step_h, off_w, off_h, step_f,
                     num_windows_w, num_windows_h, num_windows_f, loop_step, dock_at_h=None):
    """Windows of DDIM step i in the reference's loop order f -> w -> h
    (pipeline/t2v_sphere_panorama_pipeline.py:483-532).  Entries: (left, right, top, down, f_begin, f_end)."""
    k = i % loop_step
    left0, top0, fr0 = k * off_w, k * off_h, k * step_f
    wins = []
    for fi in range(num_windows_f):
        for wi in range(num_windows_w):
            h_ids = list(range(num_windows_h))
            if dock_at_h:
                h_ids = [-100] + h_ids + [-101]
            for hi in h_ids:
                left = left0 + wi * step_w
                top = top0 + hi * step_h
                fb = fr0 + fi * frames
                if dock_at_h:
                    if hi in (-100, -101) and k == 0:
                        continue  # no shift this step: docking windows are skipped (:517-519, :524-526)
                    if hi == -100:
                        top = 0
                    elif hi == -101:
                        top = total_latent_h - latent_h
                    if top + latent_h > total_latent_h:
                        continue  # regular window crossing the bottom edge is skipped when docking (:530-532)
                wins.append((left, left + latent_w, top, top + latent_h, fb, fb + frames))
    return wins
