"""CPU: the C restatement of cv::goodFeaturesToTrack (oracle/sf_oracle_gftt.c, SURVEY section 8 row f3, the detector)
against an independent numpy restatement: response map to the last bit, corner list and order exactly."""
import numpy as np
import pytest

from oracle import pyoracle
from tests import extract_cases as ec


def numpy_gftt(image, max_corners, quality, min_distance):
    f32 = np.float32
    img = image.astype(f32)
    h, w = img.shape
    p = np.pad(img, 1, mode="reflect")                       # BORDER_REFLECT_101
    s1, s2 = f32(1.0 / 3060.0), f32(2.0 / 3060.0)
    r = p[:, 2:] - p[:, :-2]                                 # rows -1..h, columns 0..w-1
    dx = s2 * r[1:-1] + s1 * (r[:-2] + r[2:])
    c = s2 * p[:, 1:-1] + s1 * (p[:, :-2] + p[:, 2:])
    dy = c[2:] - c[:-2]
    def box(a):
        q = np.pad(a, 1, mode="reflect")
        rows = (q[:, :-2] + q[:, 1:-1]) + q[:, 2:]
        return (rows[:-2] + rows[1:-1]) + rows[2:]
    a, b, cc = box(dx * dx) * f32(0.5), box(dx * dy), box(dy * dy) * f32(0.5)
    eig = (a + cc) - np.sqrt((a - cc) * (a - cc) + b * b)
    thr = f32(float(eig.max()) * quality) if eig.max() > 0 else f32(0)
    t = np.where(eig > thr, eig, f32(0))
    q = np.pad(t, 1, mode="constant", constant_values=0)
    dil = np.max(np.stack([q[1 + dy_: h + 1 + dy_, 1 + dx_: w + 1 + dx_] for dy_ in (-1, 0, 1) for dx_ in (-1, 0, 1)]), axis=0)
    mask = (t != 0) & (t == dil)
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    ys, xs = np.nonzero(mask)
    idx = ys * w + xs
    order = np.lexsort((-idx, -eig[ys, xs].astype(np.float64)))   # value descending, then index descending
    out = []
    cell = int(round(min_distance)) if min_distance >= 1 else 0
    grid = {}
    md2 = f32(min_distance * min_distance)
    for o in order:
        if max_corners > 0 and len(out) >= max_corners:
            break
        x, y = int(xs[o]), int(ys[o])
        if cell:
            cx, cy = x // cell, y // cell
            good = True
            for yy in range(cy - 1, cy + 2):
                for xx in range(cx - 1, cx + 2):
                    for (px, py) in grid.get((xx, yy), ()):
                        if f32(x - px) * f32(x - px) + f32(y - py) * f32(y - py) < md2:
                            good = False
            if not good:
                continue
            grid.setdefault((cx, cy), []).append((x, y))
        out.append((x, y))
    return np.array(out, np.int64).reshape(-1, 2), eig


@pytest.mark.parametrize("seed,shape,maxc,q,md", [
    (1, (120, 160), 1000, 0.001, 3.0), (2, (97, 131), 200, 0.01, 7.0), (3, (64, 64), 0, 0.001, 1.0),
    (4, (480, 752), 1000, 0.001, 3.0), (5, (50, 70), 50, 0.05, 0.0), (6, (33, 35), 1000, 0.001, 2.4),
])
def test_gftt_matches_numpy(seed, shape, maxc, q, md):
    image = ec.make_case(seed, n=1, width=shape[1], height=shape[0])[0]
    kp, eig = pyoracle.detect_corners(image, maxc, q, md, want_eig=True)
    ref, eig2 = numpy_gftt(np.ascontiguousarray(image), maxc, q, md)
    assert eig.tobytes() == eig2.astype(np.float32).tobytes()
    assert len(kp) == len(ref) and len(kp) > 0
    assert np.array_equal(np.stack([kp["x"], kp["y"]], axis=1).astype(np.int64), ref)
    assert (kp["size"] == 3).all() and (kp["angle"] == -1).all() and (kp["response"] == 0).all()
    assert (kp["octave"] == 0).all() and (kp["class_id"] == -1).all()
    if maxc > 0:
        assert len(kp) <= maxc


def test_gftt_ties_and_flat_image():
    flat = np.full((40, 50), 77, np.uint8)
    assert len(pyoracle.detect_corners(flat)) == 0           # no response anywhere
    # two identical blobs: equal responses, the one with the HIGHER address comes first
    img = np.zeros((60, 80), np.uint8)
    img[10:14, 10:14] = 255
    img[40:44, 50:54] = 255
    kp = pyoracle.detect_corners(img, 4, 0.5, 3.0)
    assert len(kp) >= 2
    first = (int(kp["y"][0]), int(kp["x"][0]))
    assert first[0] >= 38                                    # a corner of the lower blob leads the list
