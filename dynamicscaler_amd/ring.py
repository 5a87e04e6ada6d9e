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


def t2v_grid_windows(i, *, latent_h, latent_w, frames, num_windows_w, num_windows_h, num_windows_f, loop_step,
                     shift_jump_odd_w=False, shift_jump_odd_h=False, shift_jump_odd_f=False, docking_w=False,
                     docking_h=False, docking_f=False, docking_step_range=None):
    """Windows of step i of the NON-overlapping shifted grid, reference order f -> w -> h
    (pipeline/t2v_normal_pipeline.py:419-432, 441-443, 471-522).  NB the reference crosses the two jump flags:
    shift_jump_odd_h moves the LEFT start, shift_jump_odd_w the TOP (:471-474) -- kept."""
    vs = VAE_SCALE_FACTOR
    step_w = 0 if num_windows_w == 1 else (latent_w * vs // loop_step) // vs
    step_h = 0 if num_windows_h == 1 else (latent_h * vs // loop_step) // vs
    step_f = 0 if num_windows_f == 1 else frames // loop_step
    k = i % loop_step
    left0, top0, fr0 = k * step_w, k * step_h, k * step_f
    if i % 2 == 1 and shift_jump_odd_h and num_windows_h > 1:
        left0 += latent_w * num_windows_w // 2
    if i % 2 == 1 and shift_jump_odd_w and num_windows_w > 1:
        top0 += latent_h * num_windows_h // 2
    if i % 2 == 1 and shift_jump_odd_f and num_windows_f > 1:
        fr0 += frames * num_windows_f // 2
    in_dock = docking_step_range is not None and i in docking_step_range
    wins = []
    for fi in (range(-1, num_windows_f) if docking_f else range(num_windows_f)):
        for wi in (range(-1, num_windows_w) if docking_w else range(num_windows_w)):
            for hi in (range(-1, num_windows_h) if docking_h else range(num_windows_h)):
                left, top, fb = left0 + wi * latent_w, top0 + hi * latent_h, fr0 + fi * frames
                if docking_w and in_dock:
                    if wi == -1:
                        left = 0
                    if wi == num_windows_w - 1:
                        left = latent_w * (num_windows_w - 1)
                elif wi == -1:
                    continue
                if docking_h and in_dock:
                    if hi == -1:
                        top = 0
                    if hi == num_windows_h - 1:
                        top = latent_h * (num_windows_h - 1)
                elif hi == -1:
                    continue
                if docking_f and in_dock:
                    if fi == -1:
                        fb = 0
                    if fi == num_windows_f - 1:
                        fb = frames * (num_windows_f - 1)
                elif fi == -1:
                    continue
                wins.append((left, left + latent_w, top, top + latent_h, fb, fb + frames))
    return wins


I2V_DOCK_START_INDEX = -101
I2V_DOCK_END_INDEX = -111


def i2v_ring_windows(i, *, latent_h, latent_w, frames, total_f, step_w, step_h, off_w, off_h, num_windows_w,
                     num_windows_h, loop_step, overlap_ratio_f, loop_step_frame=None, dock_at_f=None,
                     begin_index_offset=0):
    """Windows of step i of the i2v overlapped ring, reference order f -> w -> h
    (pipeline/i2v_sphere_panorama_pipeline.py:779-854).  step_w / step_h are FLOAT latent strides rounded per window
    (:818-820); frame windows wrap modulo total_f (:828-830) and are docked / skipped when dock_at_f (:832-854)."""
    import math
    k = (i + begin_index_offset) % loop_step
    left0, top0 = k * off_w, k * off_h
    n_f = math.ceil((total_f // frames - 1) / (1 - overlap_ratio_f)) + 1
    if total_f > frames:
        fr0 = (i % loop_step_frame) * max(int(overlap_ratio_f * frames / loop_step_frame), 1)
        f_ids = list(range(n_f))
        if dock_at_f:
            f_ids = [I2V_DOCK_START_INDEX] + f_ids + [I2V_DOCK_END_INDEX]
    elif total_f == frames:
        fr0, f_ids = 0, [0]
    else:
        raise ValueError(f"total_f {total_f} should >= frames {frames} !")
    wins = []
    for fi in f_ids:
        for wi in range(num_windows_w):
            for hi in range(num_windows_h):
                left = left0 + round(wi * step_w)
                top = top0 + round(hi * step_h)
                fb = (fr0 + fi * int(frames * (1 - overlap_ratio_f))) % total_f
                fe = fb + frames
                if dock_at_f:
                    if fi == I2V_DOCK_START_INDEX:
                        if fr0 == 0:
                            continue
                        fb, fe = 0, frames
                    if fi == I2V_DOCK_END_INDEX:
                        if fr0 == 0:
                            continue
                        fb, fe = total_f - frames, total_f
                    if fe > total_f:
                        continue
                wins.append((left, left + latent_w, top, top + latent_h, fb, fe))
    return wins


def i2v_frame_windows(i, *, frames, total_f, overlap_ratio_f, loop_step_frame=None, dock_at_f=None):
    """Frame windows [(f_begin, f_end)] of step i of the i2v SPHERE loop, reference order
    (pipeline/i2v_sphere_panorama_pipeline.py:256-315): f_begin wraps modulo total_f, f_end = f_begin + frames may run
    past total_f (the window then wraps); with dock_at_f two docking windows are added when the grid is shifted and
    windows running past total_f are skipped (:302-315)."""
    import math
    n_f = math.ceil((total_f // frames - 1) / (1 - overlap_ratio_f)) + 1
    if total_f > frames:
        fr0 = (i % loop_step_frame) * max(int(overlap_ratio_f * frames / loop_step_frame), 1)
        f_ids = list(range(n_f))
        if dock_at_f:
            f_ids = [I2V_DOCK_START_INDEX] + f_ids + [I2V_DOCK_END_INDEX]
    elif total_f == frames:
        fr0, f_ids = 0, [0]
    else:
        raise ValueError(f"total_f {total_f} should >= frames {frames} !")
    out = []
    for fi in f_ids:
        fb = (fr0 + fi * int(frames * (1 - overlap_ratio_f))) % total_f
        fe = fb + frames
        if dock_at_f:
            if fi == I2V_DOCK_START_INDEX:
                if fr0 == 0:
                    continue
                fb, fe = 0, frames
            if fi == I2V_DOCK_END_INDEX:
                if fr0 == 0:
                    continue
                fb, fe = total_f - frames, total_f
            if fe > total_f:
                continue
        out.append((fb, fe))
    return out


def i2v_grid_windows(i, *, height, width, frames, num_windows_h, num_windows_w, num_windows_f, loop_step, dock_at_h=None):
    """Windows of step i of VC2_Pipeline_I2V.basic_sample_shift_multi_windows (pipeline/i2v_normal_pipeline.py:214-290):
    non-overlapping tiles shifted by (i % loop_step) * tile/loop_step, order f -> w -> h with the two docking windows
    (-100 top edge, -101 bottom edge) FIRST in the h list (:233-235).  Returns (latent windows (left, right, top, down,
    f_begin, f_end), image crop origins (img_left, img_top) in pixels -- computed from the pixel step like the reference)."""
    vs = VAE_SCALE_FACTOR
    lh, lw = height // vs, width // vs
    img_sw = width // loop_step
    lat_sw = 0 if num_windows_w == 1 else img_sw // vs
    img_sh = height // loop_step
    lat_sh = 0 if num_windows_h == 1 else img_sh // vs
    lat_sf = 0 if num_windows_f == 1 else frames // loop_step
    k = i % loop_step
    total_lh = height * num_windows_h // vs
    wins, crops = [], []
    for fi in range(num_windows_f):
        for wi in range(num_windows_w):
            h_ids = list(range(num_windows_h))
            if dock_at_h:
                h_ids = [-100, -101] + h_ids
            for hi in h_ids:
                img_left, img_top = k * img_sw + wi * width, k * img_sh + hi * height
                left, top = k * lat_sw + wi * lw, k * lat_sh + hi * lh
                fb = k * lat_sf + fi * frames
                if dock_at_h:
                    if hi in (-100, -101) and k == 0:
                        continue
                    if hi == -100:
                        top, img_top = 0, 0
                    elif hi == -101:
                        top, img_top = total_lh - lh, height * num_windows_h - height
                    if top + lh > total_lh:
                        continue
                wins.append((left, left + lw, top, top + lh, fb, fb + frames))
                crops.append((img_left, img_top))
    return wins, crops
