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


# ---------------------------------------------------------------------------------------------
# Street scene with real depth structure (integer arithmetic only): a ground plane 1.65 m below
# the camera and two facades at x = +-wall_mm, textured from a mip pyramid of one canvas, seen by
# a camera (or a rectified stereo pair) that drives straight ahead.  Unlike the planar crops
# above it has depth-dependent disparities and flow, and the monocular egomotion (which needs
# a ground plane to fix the scale) succeeds on it.
#   pixel (u, v), camera at (cam_x, 0, cam_z), focal f_px, principal point (cu, cv) [integers]:
#     ground:  Z = f*h/(v-cv) for v > cv;   facade: Z = f*(wall -+ cam_x)/|u-cu|;   nearer one wins
#     texel size 2^level * cell_mm with level = floor(log2(Z / z0)): the footprint of a pixel stays
#     within a factor of two of a texel at every depth
# ---------------------------------------------------------------------------------------------
ROAD_F, ROAD_H_MM, ROAD_WALL_MM, ROAD_CELL_MM = 720, 1650, 6000, 4


def road_pyramid(seed, levels=10, size=2048):
    """levels of one size x size texture, each a 2x2 box reduction of the previous (integer mean)
    with the contrast restored, tiled back to size x size (the textures are used periodically)"""
    base = canvas(seed, size - CANVAS_PAD_W, size - CANVAS_PAD_H)[:size, :size]
    pyr = [base]
    for _ in range(1, levels):
        b = pyr[-1].astype(np.int32)
        b = (b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2]) // 4
        b = np.clip((b - 128) * 2 + 128, 0, 255)
        pyr.append(np.tile(b, (2, 2)).astype(np.uint8))
    return pyr


def road_view(pyr, W, H, cam_x_mm, cam_z_mm, cu=None, cv=None):
    """one uint8 image of the street from a camera at lateral offset cam_x_mm, advanced by cam_z_mm.
    The mip level follows the footprint of a pixel ALONG the surface (Z^2 / (f * distance to the
    surface plane)), the direction in which a grazing view compresses the texture."""
    cu = W // 2 if cu is None else cu
    cv = (H * 2) // 5 if cv is None else cv
    size = pyr[0].shape[0]
    u = np.arange(W, dtype=np.int64)[None, :] - cu
    v = np.arange(H, dtype=np.int64)[:, None] - cv
    big = np.int64(1) << 40
    zg = np.where(v > 0, (ROAD_F * ROAD_H_MM) // np.maximum(v, 1), big) + 0 * u              # ground hit
    wall_off = np.where(u > 0, ROAD_WALL_MM - cam_x_mm, ROAD_WALL_MM + cam_x_mm) + 0 * v      # distance to the facade
    zw = np.where(u != 0, (ROAD_F * wall_off) // np.maximum(np.abs(u), 1), big) + 0 * v       # facade hit
    on_ground = zg <= zw
    z = np.minimum(zg, zw)
    far = z >= 120000                                                                         # sky beyond 120 m
    z = np.minimum(z, 120000)
    x = (u * z) // ROAD_F + cam_x_mm
    y = (v * z) // ROAD_F
    zz = z + cam_z_mm
    ta = np.where(on_ground, x + 50000, y + 70000 + np.where(u > 0, 0, 33331))                # across the surface
    tb = zz                                                                                   # along the street
    foot = (z * z) // (ROAD_F * np.where(on_ground, ROAD_H_MM, wall_off))                     # mm per pixel along the street
    lvl = np.zeros_like(z)
    for k in range(1, len(pyr)):
        lvl += (foot > (ROAD_CELL_MM << (k - 1))).astype(np.int64)
    out = np.zeros((H, W), dtype=np.uint8)
    for k in range(len(pyr)):
        sel = lvl == k
        if sel.any():
            cell = ROAD_CELL_MM << k
            out[sel] = pyr[k][((tb // cell) % size)[sel], ((ta // cell) % size)[sel]]
    out[far | (foot > (ROAD_CELL_MM << len(pyr)))] = 96                                       # too far to resolve
    return out


def road_stereo_frame(pyr, f, W, H, step_mm=600, base_mm=540):
    """rectified pair after f steps of step_mm forward motion; disparity = f*base/Z"""
    return road_view(pyr, W, H, 0, f * step_mm), road_view(pyr, W, H, base_mm, f * step_mm)


def road_mono_frame(pyr, f, W, H, step_mm=600):
    return road_view(pyr, W, H, 0, f * step_mm)
