"""Synthetic stereo keyframes for the feature-extraction tests: a textured left image,
corners with right-image positions from planted depths, and the awkward ones -- corners inside the BRIEF border,
failed stereo correspondences, zero / negative disparity, depths outside the accepted range."""
import numpy as np

from multi_robot_slam_separators_amd import _abi


def brief_tests(seed, nbytes):
    """A test table in the format of sf_brief_set_pattern: int8 [8 * bytes, 4], offsets within +-24."""
    rng = np.random.default_rng(seed)
    t = np.clip(np.rint(rng.normal(0.0, 48 / 5.0, size=(8 * nbytes, 4))), -24, 24).astype(np.int8)
    return t


def make_case(seed, n=600, width=752, height=480, pad=8, min_depth=0.0, max_depth=0.0, identity=False, no_stereo=False):
    rng = np.random.default_rng(seed)
    # texture: blocky noise + gradient, so that box sums differ and a few ties exist too
    base = rng.integers(0, 256, size=(height // 4 + 1, width // 4 + 1))
    img = np.kron(base, np.ones((4, 4)))[:height, :width] * 0.7 + np.linspace(0, 70, width)[None, :]
    img = np.clip(img + rng.normal(0, 3, size=img.shape), 0, 255).astype(np.uint8)
    buf = np.zeros((height, width + pad), np.uint8)      # pitch > width
    buf[:, :width] = img
    image = buf[:, :width]
    kp = np.zeros(n, _abi.KEYPOINT_DTYPE)
    kp["x"] = rng.uniform(0, width, n).astype(np.float32)
    kp["y"] = rng.uniform(0, height, n).astype(np.float32)
    edge = rng.random(n) < 0.1                            # exactly on / next to the border limits
    kp["x"][edge] = rng.choice(np.array([27.5, 28.0, width - 28.0, width - 28.5], np.float32), int(edge.sum()))
    half = rng.random(n) < 0.2                            # .5 positions exercise the rounding
    kp["x"][half] = np.floor(kp["x"][half]) + 0.5
    kp["size"] = 7.0
    kp["angle"] = -1.0
    kp["response"] = rng.random(n).astype(np.float32)
    kp["octave"] = rng.integers(0, 4, n)
    kp["class_id"] = -1
    fx, baseline = 460.0, 0.11
    depth = rng.uniform(0.6, 25.0, n).astype(np.float32)
    disparity = (np.float32(fx) * np.float32(baseline) / depth).astype(np.float32)
    right_x = (kp["x"] - disparity).astype(np.float32)
    bad = rng.random(n)
    right_x[bad < 0.05] = kp["x"][bad < 0.05]             # zero disparity
    right_x[(bad >= 0.05) & (bad < 0.10)] += 40.0         # negative disparity
    status = (rng.random(n) > 0.08).astype(np.uint8)
    lt = None
    if not identity:
        lt = np.array([[0, 0, 1, 0.1], [-1, 0, 0, 0.05], [0, -1, 0, 0.3]], np.float32)   # optical -> base
    cam = _abi.stereo_camera(fx, 458.0, 367.2, 248.4, baseline, cx_right=0.0 if seed % 2 else 379.9,
                             local_transform=lt, min_depth=min_depth, max_depth=max_depth)
    if no_stereo:
        right_x, status = None, None
    return image, kp, right_x, status, cam


def numpy_extract(image, kp, right_x, status, cam, tests):
    """Independent restatement with numpy (vectorised; float32 arithmetic in the documented order)."""
    h, w = image.shape
    S = np.zeros((h + 1, w + 1), np.int64)
    S[1:, 1:] = np.cumsum(np.cumsum(image.astype(np.int64), axis=0), axis=1)
    x, y = kp["x"], kp["y"]
    border = 28
    inside = (x >= np.float32(border)) & (x < np.float32(w - border)) & (y >= np.float32(border)) & (y < np.float32(h - border))
    px = (x + np.float32(0.5)).astype(np.int64)
    py = (y + np.float32(0.5)).astype(np.int64)
    tests = np.asarray(tests, np.int64).reshape(-1, 4)
    nbytes = tests.shape[0] // 8
    idx = np.nonzero(inside)[0]

    def box(dx, dy):
        yy = py[idx, None] + dy[None, :]
        xx = px[idx, None] + dx[None, :]
        y1, x1 = np.minimum(yy + 5, h), np.minimum(xx + 5, w)      # (the last row / column repeats)
        return S[y1, x1] - S[y1, xx - 4] - S[yy - 4, x1] + S[yy - 4, xx - 4]
    bits = (box(tests[:, 0], tests[:, 1]) < box(tests[:, 2], tests[:, 3])).astype(np.uint8)
    desc = np.packbits(bits.reshape(len(idx), nbytes, 8), axis=2, bitorder="big").reshape(len(idx), nbytes)
    f32 = np.float32
    p = np.full((len(idx), 3), np.nan, f32)
    if right_x is not None:
        xi, yi = x[idx], y[idx]
        disp = xi - right_x[idx]
        ok = (disp > 0) if status is None else ((disp > 0) & (status[idx] != 0))
        c = f32(cam.cx_right - cam.cx) if (cam.cx_right > 0 and cam.cx > 0) else f32(0)
        with np.errstate(divide="ignore", invalid="ignore"):
            W = f32(cam.baseline) / (disp + c)
            X, Y, Z = (xi - f32(cam.cx)) * W, (yi - f32(cam.cy)) * W, f32(cam.fx) * W
        ok &= np.isfinite(X) & np.isfinite(Y) & np.isfinite(Z)
        ok &= (cam.min_depth < 0) | (Z > f32(cam.min_depth))
        ok &= (cam.max_depth <= 0) | (Z <= f32(cam.max_depth))
        L = np.array(list(cam.local_transform), f32).reshape(3, 4)
        if np.array_equal(L, np.eye(4, dtype=f32)[:3]):
            q = np.stack([X, Y, Z], axis=1)
        else:
            with np.errstate(invalid="ignore"):
                q = np.stack([((L[r, 0] * X + L[r, 1] * Y) + L[r, 2] * Z) + L[r, 3] for r in range(3)], axis=1)
        p[ok] = q[ok].astype(f32)
    keep = np.ones(len(idx), bool)
    if cam.min_depth > 0 or cam.max_depth > 0:
        keep = np.isfinite(p).all(axis=1)
    return desc[keep], p[keep], kp[idx][keep]


def make_stereo_pair(seed, width=752, height=480, pad=8, max_disp=40.0, noise=1.5):
    """A rectified pair for the stereo-correspondence tests: multi-scale texture on the left, the right image is the
    left one resampled along the rows by a smooth disparity field (2 .. max_disp px) plus sensor noise.  Both images
    share one pitch (width + pad).  Returns left, right (views with pitch > width) and the disparity field."""
    rng = np.random.default_rng(seed)
    tex = np.zeros((height, width + 160))
    for cell, amp in ((32, 60.0), (12, 50.0), (5, 35.0), (2, 20.0)):
        g = rng.normal(size=(height // cell + 3, (width + 160) // cell + 3))
        ys = np.arange(height) / cell
        xs = np.arange(width + 160) / cell
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        tex += amp * ((1 - fy) * (1 - fx) * g[y0][:, x0] + (1 - fy) * fx * g[y0][:, x0 + 1]
                      + fy * (1 - fx) * g[y0 + 1][:, x0] + fy * fx * g[y0 + 1][:, x0 + 1])
    tex = 128 + tex * 0.6
    yy, xx = np.mgrid[0:height, 0:width]
    disp = 2.0 + (max_disp - 2.0) * (0.5 + 0.5 * np.sin(xx / width * 2.1 + 0.4) * np.cos(yy / height * 1.7))
    left = tex[:, 80:80 + width]
    # right(x) = left(x + d): a point at x in the left image appears at x - d in the right one
    src = xx + 80 + disp
    s0 = np.floor(src).astype(int)
    f = src - s0
    right = (1 - f) * tex[yy, np.clip(s0, 0, width + 159)] + f * tex[yy, np.clip(s0 + 1, 0, width + 159)]
    bl = np.zeros((height, width + pad), np.uint8)
    br = np.zeros((height, width + pad), np.uint8)
    bl[:, :width] = np.clip(left + rng.normal(0, noise, left.shape), 0, 255).astype(np.uint8)
    br[:, :width] = np.clip(right + rng.normal(0, noise, right.shape), 0, 255).astype(np.uint8)
    return bl[:, :width], br[:, :width], disp
