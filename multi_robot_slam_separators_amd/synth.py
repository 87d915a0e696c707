"""Seeded synthetic workload for the separator-finder hot path.

The reference ships no rosbag, fixtures or tests, so every input is generated here:
  * NetVLAD descriptors: unit-norm rows (what nets.vgg16NetvladPca emits, data_handler.py:63)
    with planted revisits;
  * keyframe geometric features with the layouts of PKG/msg/{Descriptors,KeyPoint3DVec,
    KeyPointVec}.msg: K keypoints inside a 640x480 pin-hole image (fx=fy=600, cx=320, cy=240),
    depth 1..20 m, 3D points in the robot BASE frame (x forward), random binary descriptors;
  * true pairs: 40 % of B's features are A's features seen from a pose T_gt away
    (p_A = T_gt * p_B), 2 cm Gaussian noise, 5 % descriptor bit flips; false pairs independent.
"""
import numpy as np

from . import _abi

FX = FY = 600.0
CX, CY = 320.0, 240.0
WIDTH, HEIGHT = 640, 480
# pose of the optical frame in the base frame: p_base = L * p_cam  (z_cam forward = x_base)
LOCAL_TRANSFORM = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 0]], dtype=np.float32)


def camera_params(p=None):
    """Fill the camera block of Params with the synthetic pin-hole model."""
    if p is None:
        p = _abi.default_params()
    p.fx, p.fy, p.cx, p.cy = FX, FY, CX, CY
    p.image_width, p.image_height = WIDTH, HEIGHT
    for i, v in enumerate(LOCAL_TRANSFORM.reshape(-1)):
        p.local_transform[i] = float(v)
    return p


def random_rotation(rng, max_angle_deg):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rng.uniform(0.0, max_angle_deg))
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def random_transform(rng, max_angle_deg=30.0, max_trans=2.0):
    """4x4 T_gt: rotation <= max_angle about a random axis, translation <= max_trans metres."""
    T = np.eye(4)
    T[:3, :3] = random_rotation(rng, max_angle_deg)
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    T[:3, 3] = d * rng.uniform(0.0, max_trans)
    return T


def project_base_points(xyz_base):
    """Pixel coordinates of base-frame points through the synthetic camera (float32)."""
    xyz_base = np.asarray(xyz_base, dtype=np.float64)
    xc, yc, zc = -xyz_base[..., 1], -xyz_base[..., 2], xyz_base[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = FX * xc / zc + CX
        v = FY * yc / zc + CY
    return u.astype(np.float32), v.astype(np.float32)


def _keypoints(u, v, octave=0):
    k = np.zeros(u.shape, dtype=_abi.KEYPOINT_DTYPE)
    k["x"], k["y"] = u, v
    k["size"] = 31.0
    k["angle"] = -1.0
    k["response"] = 0.01
    k["octave"] = octave
    k["class_id"] = -1
    return k


def make_points(rng, shape):
    """Base-frame 3D points whose projections are uniform inside the image, depth 1..20 m."""
    u = rng.uniform(0.0, WIDTH - 1.0, size=shape)
    v = rng.uniform(0.0, HEIGHT - 1.0, size=shape)
    z = rng.uniform(1.0, 20.0, size=shape)
    xc, yc = (u - CX) / FX * z, (v - CY) / FY * z
    xyz = np.stack([z, -xc, -yc], axis=-1).astype(np.float32)  # base: x fwd, y left, z up
    return xyz


def make_keyframe(rng, k, cols=32):
    xyz = make_points(rng, (k,))
    u, v = project_base_points(xyz)
    desc = rng.integers(0, 256, size=(k, cols), dtype=np.uint8)
    return _abi.FeatureArrays(desc, xyz, _keypoints(u, v))


def flip_bits(rng, desc, p):
    if p <= 0:
        return desc.copy()
    bits = np.unpackbits(desc, axis=-1)
    bits ^= (rng.random(bits.shape) < p).astype(np.uint8)
    return np.packbits(bits, axis=-1)


def make_true_partner(rng, fa, T_gt, overlap=0.4, noise=0.02, flip=0.05):
    """Keyframe B observing `overlap` of keyframe A's features from pose T_gt (p_A = T_gt p_B).
    Returns (FeatureArrays B, gt) where gt[j] = index in A of B's feature j, or -1."""
    k, cols = fa.desc.shape
    n_ov = int(round(overlap * k))
    sel = rng.permutation(k)[:n_ov]
    Tinv = np.linalg.inv(T_gt)
    pa = fa.xyz[sel].astype(np.float64)
    pb = pa @ Tinv[:3, :3].T + Tinv[:3, 3] + rng.normal(scale=noise, size=pa.shape)
    xyz_new = make_points(rng, (k - n_ov,))
    desc_ov = flip_bits(rng, fa.desc[sel], flip)
    desc_new = rng.integers(0, 256, size=(k - n_ov, cols), dtype=np.uint8)
    xyz = np.concatenate([pb.astype(np.float32), xyz_new], axis=0)
    desc = np.concatenate([desc_ov, desc_new], axis=0)
    gt = np.concatenate([sel, -np.ones(k - n_ov, dtype=np.int64)])
    perm = rng.permutation(k)
    xyz, desc, gt = xyz[perm], desc[perm], gt[perm]
    u, v = project_base_points(xyz)
    return _abi.FeatureArrays(desc, xyz, _keypoints(u, v)), gt


def without_3d(fa):
    """The same keyframe without 3D points (keypoints and descriptors only)."""
    return _abi.FeatureArrays(fa.desc, np.zeros((0, 3), np.float32), fa.kpts)


def float_descriptors(fa, dims, rng=None, jitter=0.0):
    """The float32-descriptor twin of a keyframe with binary descriptors (sf_params.desc_type 1): dimension d of a row is
    +-1 by bit d of the binary descriptor (+ Gaussian jitter), so a true partner's rows stay close to their originals
    (4 per flipped bit) and unrelated rows are ~4 * dims / 2 apart -- and, without jitter, distances tie often."""
    bits = np.unpackbits(fa.desc, axis=1)[:, :dims].astype(np.float32) * 2.0 - 1.0
    if jitter > 0:
        bits = bits + rng.normal(scale=jitter, size=bits.shape).astype(np.float32)
    return _abi.FeatureArrays(np.ascontiguousarray(bits, dtype=np.float32), fa.xyz, fa.kpts)


def make_pairs(seed, n, k=500, cols=32, true_frac=0.2, overlap=0.4, noise=0.02, flip=0.05):
    """n candidate pairs (from=A, to=B).  Returns (list A, list B, is_true[n], T_gt list)."""
    rng = np.random.default_rng(seed)
    A, B, is_true, Ts = [], [], np.zeros(n, dtype=bool), []
    for i in range(n):
        a = make_keyframe(rng, k, cols)
        if rng.random() < true_frac:
            T = random_transform(rng)
            b, _ = make_true_partner(rng, a, T, overlap, noise, flip)
            is_true[i] = True
            Ts.append(T)
        else:
            b = make_keyframe(rng, k, cols)
            Ts.append(None)
        A.append(a)
        B.append(b)
    return A, B, is_true, Ts


def make_netvlad(seed, n_local, n_other, dim, planted_frac=0.05, planted_noise=0.05):
    """Unit-norm float32 NetVLAD rows for two robots.  A fraction of the OTHER robot's rows
    are planted revisits normalize(local[j] + planted_noise * g / sqrt(dim)) of distinct local
    rows j (distance ~ planted_noise).  Returns (local, other, planted_local_idx[n_other])."""
    rng = np.random.default_rng(seed)
    local = rng.normal(size=(n_local, dim)).astype(np.float32)
    local /= np.linalg.norm(local, axis=1, keepdims=True)
    other = rng.normal(size=(n_other, dim)).astype(np.float32)
    other /= np.linalg.norm(other, axis=1, keepdims=True)
    n_pl = min(int(round(planted_frac * n_other)), n_local)
    planted = -np.ones(n_other, dtype=np.int64)
    if n_pl > 0:
        rows = rng.permutation(n_other)[:n_pl]
        src = rng.permutation(n_local)[:n_pl]
        g = rng.normal(size=(n_pl, dim)).astype(np.float32) * (planted_noise / np.sqrt(dim))
        rev = local[src] + g
        rev /= np.linalg.norm(rev, axis=1, keepdims=True)
        other[rows] = rev
        planted[rows] = src
    return local, other, planted


def make_store_batch(seed, n_kf, k=500, cols=32, true_frac=0.2, overlap=0.4, noise=0.02, flip=0.05):
    """Vectorised generation of two robots' keyframe stores for the benchmark: robot A has
    n_kf keyframes, robot B has n_kf keyframes of which `true_frac` are revisits of the
    same-index A keyframe.  Returns dict of arrays:
      desc_a/desc_b [n][k][cols] u8, xyz_a/xyz_b [n][k][3] f32, kp_a/kp_b [n][k] KEYPOINT_DTYPE,
      is_true [n] bool, T_gt [n][4][4] f64 (identity for false pairs)."""
    rng = np.random.default_rng(seed)
    xyz_a = make_points(rng, (n_kf, k))
    desc_a = rng.integers(0, 256, size=(n_kf, k, cols), dtype=np.uint8)
    xyz_b = make_points(rng, (n_kf, k))
    desc_b = rng.integers(0, 256, size=(n_kf, k, cols), dtype=np.uint8)
    is_true = rng.random(n_kf) < true_frac
    T_gt = np.tile(np.eye(4), (n_kf, 1, 1))
    n_ov = int(round(overlap * k))
    for i in np.nonzero(is_true)[0]:
        T = random_transform(rng)
        T_gt[i] = T
        Tinv = np.linalg.inv(T)
        sel = rng.permutation(k)[:n_ov]
        dst = rng.permutation(k)[:n_ov]
        pa = xyz_a[i, sel].astype(np.float64)
        pb = pa @ Tinv[:3, :3].T + Tinv[:3, 3] + rng.normal(scale=noise, size=pa.shape)
        xyz_b[i, dst] = pb.astype(np.float32)
        desc_b[i, dst] = flip_bits(rng, desc_a[i, sel], flip)
    ua, va = project_base_points(xyz_a)
    ub, vb = project_base_points(xyz_b)
    return dict(desc_a=desc_a, desc_b=desc_b, xyz_a=xyz_a, xyz_b=xyz_b,
                kp_a=_keypoints(ua, va), kp_b=_keypoints(ub, vb), is_true=is_true, T_gt=T_gt)


def pose_error(result, T_gt):
    """(translation error [m], rotation error [rad]) of an sf_result record against T_gt."""
    x, y, z, w = result["orientation"]
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    dt = np.linalg.norm(np.asarray(result["position"]) - T_gt[:3, 3])
    c = (np.trace(R.T @ T_gt[:3, :3]) - 1.0) / 2.0
    return dt, float(np.arccos(np.clip(c, -1.0, 1.0)))
