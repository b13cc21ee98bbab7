"""Constant tables of the render path, built on the host exactly as the reference builds them.

``direction_table``: output2env.__init__ (models.py:353-363) and renderingLayer.__init__
(models.py:437-452).  ``view_vectors``: renderingLayer.__init__ (models.py:415-430).
Both are computed with numpy in float64 and stored as float32, like the reference, and then
packed into the device layout the kernels read (see include/sgrender.h).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def direction_table(env_height: int, env_width: int) -> Tuple[np.ndarray, np.ndarray]:
    """``(ls[J,3], omega[J])`` float32; ``j = e*env_width + a``."""
    az = ((np.arange(env_width) + 0.5) / env_width - 0.5) * 2 * np.pi
    el = ((np.arange(env_height) + 0.5) / env_height) * np.pi / 2.0
    az, el = np.meshgrid(az, el)
    az = az.reshape(-1)
    el = el.reshape(-1)
    ls = np.stack([np.sin(el) * np.cos(az), np.sin(el) * np.sin(az), np.cos(el)], axis=1)
    omega = np.sin(el) * np.pi * np.pi / env_width / env_height
    return ls.astype(np.float32), omega.astype(np.float32)


def packed_direction_table(env_height: int, env_width: int) -> np.ndarray:
    """Device layout (flat float32, see include/sgrender.h): ``[Jpad,4] = (lx, ly, lz, omega)`` with zero
    rows up to a multiple of 32, then the separable form ``rows[ehp,8] = (s, c, omega, s^2, 2sc, c^2, 0, 0)``
    and the azimuth factors of the first half row (``(ca, sa)`` pairs, then ``(ca^2, 2 ca sa, sa^2, 0)``, then
    -- at float offset ``4*ew`` of the column block -- ``(ca_a, ca_a+1, sa_a, sa_a+1)`` per azimuth pair)."""
    ls, omega = direction_table(env_height, env_width)
    J = ls.shape[0]
    jpad = (J + 31) // 32 * 32
    gen = np.zeros((jpad, 4), dtype=np.float32)
    gen[:J, :3] = ls
    gen[:J, 3] = omega
    az = ((np.arange(env_width) + 0.5) / env_width - 0.5) * 2 * np.pi
    el = ((np.arange(env_height) + 0.5) / env_height) * np.pi / 2.0
    ehp = (env_height + 1) // 2 * 2
    rows = np.zeros((ehp, 8), dtype=np.float64)
    s, c = np.sin(el), np.cos(el)
    rows[:env_height, 0], rows[:env_height, 1] = s, c
    rows[:env_height, 2] = s * np.pi * np.pi / env_width / env_height
    rows[:env_height, 3], rows[:env_height, 4], rows[:env_height, 5] = s * s, 2 * s * c, c * c
    # cols (8*ew floats reserved): first half row only (the second half is its negation):
    #   [ew/2][2] = (ca, sa), then at float offset ew: [ew/2][4] = (ca^2, 2 ca sa, sa^2, 0)
    half = env_width // 2
    ca, sa = np.cos(az[:half]), np.sin(az[:half])
    cols = np.zeros(8 * env_width, dtype=np.float64)
    cols[0:2 * half:2], cols[1:2 * half:2] = ca, sa
    ext = np.zeros((half, 4))
    ext[:, 0], ext[:, 1], ext[:, 2] = ca * ca, 2 * ca * sa, sa * sa
    cols[env_width:env_width + 4 * half] = ext.reshape(-1)
    #   at float offset 4*ew: [ew/4][4] = (ca_a, ca_a+1, sa_a, sa_a+1) per azimuth pair (packed-math kernels)
    #   (an odd half row leaves the last pair half-filled with zeros, exactly like sgr_fill_direction_table; the packed
    #   kernels only run on envWidth 16 / 32)
    pairs = np.zeros(((half + 1) // 2, 4))
    pairs[np.arange(half) // 2, np.arange(half) % 2] = ca
    pairs[np.arange(half) // 2, 2 + np.arange(half) % 2] = sa
    cols[4 * env_width:4 * env_width + pairs.size] = pairs.reshape(-1)
    return np.concatenate([gen.reshape(-1), rows.astype(np.float32).reshape(-1), cols.astype(np.float32)])


def generic_table_view(packed: np.ndarray, env_height: int, env_width: int) -> np.ndarray:
    """The ``[Jpad,4]`` generic part of :func:`packed_direction_table`."""
    jpad = (env_height * env_width + 31) // 32 * 32
    return packed[:4 * jpad].reshape(jpad, 4)


def view_vectors(im_width: int, im_height: int, fov_deg: float = 57.0,
                 camera_pos: Sequence[float] = (0, 0, 0)) -> np.ndarray:
    """Unit view vectors ``v[3,imHeight,imWidth]`` float32 (imHeight x imWidth is the env grid)."""
    fov = fov_deg / 180.0 * np.pi
    x_range = 1 * np.tan(fov / 2)
    y_range = float(im_height) / float(im_width) * x_range
    x, y = np.meshgrid(np.linspace(-x_range, x_range, im_width), np.linspace(-y_range, y_range, im_height))
    y = np.flip(y, axis=0)
    z = -np.ones((im_height, im_width), dtype=np.float32)
    p = np.stack([x, y, z]).astype(np.float32)
    cam = np.array(camera_pos, dtype=np.float32).reshape(3, 1, 1)
    v = cam - p
    v = v / np.sqrt(np.maximum(np.sum(v * v, axis=0), 1e-12))[np.newaxis]
    return np.ascontiguousarray(v.astype(np.float32))
