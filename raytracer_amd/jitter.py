"""Soft-shadow jitter sources.

The reference draws its jitter from the global NumPy stream (`np.random.random(M)` twice per
shadow-grid cell, reference ray_tracer.py:221-222).  The renderer takes the jitter source
explicitly instead (BASELINE.json north_star: "the jitter is seeded and passed in explicitly"):

* ``PixelKeyedJitter(seed)`` — "mode B" of SURVEY.md App. A.8.  The two uniforms of a cell are a
  pure function of ``(seed, pixel, depth, light, cell)``, so pixels are independent and an image
  can be tiled over GPUs.  The integer mixer below is implemented three times with identical
  results: here (NumPy uint32), in ``oracle/rt_oracle.c`` and in ``csrc/rt_hip.hip``.
* ``NumpyStreamJitter(seed)`` — "mode A": the legacy MT19937 stream that
  ``np.random.seed(seed); render_vectorized(...)`` consumes, element m of each draw belonging to
  the m-th hit of the batch.  The library regenerates that stream itself (same generator, same
  consumption order), which forces the reference's depth-first batch order on the device.

Uniforms are fp32 values in [0, 1] (a uint32 converted to float32, times 2**-32), so the fp64 and
fp32 kernels and the CPU oracle all see exactly the same jitter values.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

JITTER_PIXEL_KEYED = 0   # mode B
JITTER_NUMPY_STREAM = 1  # mode A



def lowbias32(x):
    """32-bit integer finaliser (xorshift-multiply, 2 rounds); x: uint32 array or scalar."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def light_key(seed, pixels, depth, light_index):
    """Per-(pixel, depth, light) 32-bit key; `pixels` are global flat indices y*W + x."""
    with np.errstate(over="ignore"):
        h = lowbias32(np.asarray(pixels, dtype=np.uint32) ^ np.uint32(seed & 0xFFFFFFFF))
        h = lowbias32(h + np.uint32(depth))
        h = lowbias32(h + np.uint32(light_index))
    return h


_PHI_A, _PHI_B, _KB_XOR = np.uint32(0x9E3779B9), np.uint32(0x7F4A7C15), np.uint32(0x68E31DA4)


def cell_uniforms(key, cell):
    """(u_i, u_j) float64 for shadow-grid cell `cell = i*n + j` from the per-light key.

    Per-pixel rotated lattice: u = f32(k + (cell+1)*PHI) * 2**-32 with uint32 wrap-around, where
    ka = key, kb = lowbias32(key ^ const) and f32() is the round-to-nearest uint32 -> float32
    conversion (so every platform gets the same fp32 value; it can equal 1.0)."""
    ka = np.asarray(key, dtype=np.uint32)
    with np.errstate(over="ignore"):
        kb = lowbias32(ka ^ _KB_XOR)
        xa = ka + np.uint32(cell + 1) * _PHI_A
        xb = kb + np.uint32(cell + 1) * _PHI_B
    u_i = xa.astype(np.float32).astype(np.float64) * 2.0 ** -32
    u_j = xb.astype(np.float32).astype(np.float64) * 2.0 ** -32
    return u_i, u_j


@dataclass(frozen=True)
class PixelKeyedJitter:
    seed: int = 1234
    mode: int = JITTER_PIXEL_KEYED

    def uniforms(self, pixels, depth, light_index, cell):
        return cell_uniforms(light_key(self.seed, pixels, depth, light_index), cell)


@dataclass(frozen=True)
class NumpyStreamJitter:
    """Reproduces `np.random.seed(seed)` followed by the reference's draws."""
    seed: int = 1234
    mode: int = JITTER_NUMPY_STREAM


def as_jitter(spec):
    """Accept a jitter object, an int seed (pixel-keyed), or ("numpy"|"pixel", seed)."""
    if spec is None:
        return PixelKeyedJitter()
    if isinstance(spec, (PixelKeyedJitter, NumpyStreamJitter)):
        return spec
    if isinstance(spec, (int, np.integer)):
        return PixelKeyedJitter(int(spec))
    kind, seed = spec
    if kind in ("numpy", "stream", "A"):
        return NumpyStreamJitter(int(seed))
    if kind in ("pixel", "keyed", "B"):
        return PixelKeyedJitter(int(seed))
    raise ValueError(f"unknown jitter spec {spec!r}")
