"""Integer-only synthetic stereo / mono sequences.

No libm and no floats anywhere, so the same bytes come out on every host: the
golden fixtures, the CPU oracle, the reference build and the GPU bench all see
identical frames.

  canvas(seed, W, H)   (H+64) x (W+1024) texture: a 32-bit integer hash of
                       (seed, y, x) -> byte noise, 5x5 integer box blur, contrast
                       clamp((b-128)*4+128).  (A stateless hash instead of a
                       sequential xorshift32 stream so numpy can vectorise it.)
  stereo_frame(c,f,..) left  = crop at (x0, 32),  x0 = 64 + (3 f mod 900)
                       right = crop at (x0 + disparity, 32)  -> u_left - u_right = +disparity
  mono_frame(c,f,..)   crop at (x0, 32 + f mod 7)  (vertical jitter, config 3)
  ramp variant         right image gets a row-dependent disparity
                       disparity + (v * ramp_num) // ramp_den, which widens the
                       prior boxes of pass 2 (more SADs per findMatch).
"""
import numpy as np

CANVAS_PAD_W = 1024
CANVAS_PAD_H = 64


def _hash32(seed, y, x):
    """murmur3-style finaliser over (seed, y, x); all arithmetic in uint32."""
    h = (np.uint32(seed) + y.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (x.astype(np.uint32) * np.uint32(0x85EBCA77))
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def canvas(seed, W, H, blur=5):
    cw, ch = W + CANVAS_PAD_W, H + CANVAS_PAD_H
    r = blur // 2
    yy, xx = np.meshgrid(np.arange(ch + 2 * r, dtype=np.uint32), np.arange(cw + 2 * r, dtype=np.uint32), indexing="ij")
    with np.errstate(over="ignore"):
        noise = (_hash32(seed, yy, xx) >> np.uint32(24)).astype(np.int32)
    # (2r+1)^2 box sum via integral image, integer division
    ii = np.zeros((ch + 2 * r + 1, cw + 2 * r + 1), dtype=np.int64)
    ii[1:, 1:] = noise.cumsum(0).cumsum(1)
    k = 2 * r + 1
    box = ii[k:, k:] - ii[:-k, k:] - ii[k:, :-k] + ii[:-k, :-k]
    b = (box // (k * k)).astype(np.int32)
    out = np.clip((b - 128) * 4 + 128, 0, 255).astype(np.uint8)
    assert out.shape == (ch, cw)
    return out


def _x0(f):
    return 64 + (3 * f) % 900


def stereo_frame(cv, f, W, H, disparity=20, ramp=(0, 1)):
    """returns (left, right) uint8 arrays of shape (H, W), C-contiguous."""
    x0 = _x0(f)
    left = np.ascontiguousarray(cv[32:32 + H, x0:x0 + W])
    if ramp[0] == 0:
        right = np.ascontiguousarray(cv[32:32 + H, x0 + disparity:x0 + disparity + W])
    else:
        right = np.empty((H, W), dtype=np.uint8)
        for v in range(H):
            d = disparity + (v * ramp[0]) // ramp[1]
            right[v] = cv[32 + v, x0 + d:x0 + d + W]
    return left, right


def mono_frame(cv, f, W, H):
    x0 = _x0(f)
    y0 = 32 + f % 7
    return np.ascontiguousarray(cv[y0:y0 + H, x0:x0 + W])


def stereo_sequence(seed, W, H, n_frames, disparity=20, ramp=(0, 1), blur=5):
    cv = canvas(seed, W, H, blur=blur)
    return [stereo_frame(cv, f, W, H, disparity, ramp) for f in range(n_frames)]


def mono_sequence(seed, W, H, n_frames, blur=5):
    cv = canvas(seed, W, H, blur=blur)
    return [mono_frame(cv, f, W, H) for f in range(n_frames)]
