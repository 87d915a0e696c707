#!/usr/bin/env python3
"""Generate the golden vectors that PIN the NN stage of the oracle (tests/golden/nn_*.npz).

The reference's DataHandler.find_matches (PKG/scripts/data_handler.py:166-209) is Python 2 and
imports rospy/cv2/tensorflow, so the module itself cannot be imported here (SyntaxError /
ModuleNotFoundError -- ordinary Python errors, nothing was denied).  Its ARITHMETIC however is
scipy.spatial.distance.cdist + numpy.argsort, both importable.  This script drives those two
library calls through the same statement sequence as data_handler.py:168-205 on seeded inputs
and stores inputs + outputs.  Inputs avoid exact ties (numpy's quicksort leaves their order
unspecified) except for the all-inf rows that masking creates, which can never pass the
threshold at :202.

Run from the repo root:  python oracle/gen_golden.py
"""
import os
import sys

import numpy as np
from scipy.spatial.distance import cdist

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def drive_find_matches(local_descriptors, received_descriptors, local_used, other_used,
                       pairs_ignored, netvlad_distance, max_matches_nb):
    """scipy/numpy driven in the order of data_handler.py:168-205."""
    a = np.array(local_descriptors)                       # :168
    b = np.array(received_descriptors)                    # :169
    dist = cdist(a, b)                                    # :170
    if len(local_used) > 0:                               # :178-179
        dist[np.array(local_used)] = np.inf
    if len(other_used) > 0:                               # :180-181
        dist[:, np.array(other_used)] = np.inf
    for pr in pairs_ignored:                              # :183-184
        dist[pr[0], pr[1]] = np.inf
    arg_each = np.argsort(dist, axis=1)[:, 0]             # :187
    val_each = dist[np.arange(len(dist)), arg_each]       # :188-189
    order = np.argsort(val_each)                          # :191
    matches = []
    for i in range(min(len(order), max_matches_nb)):      # :194
        il = order[i]
        io = arg_each[order[i]]
        if io in [m[1] for m in matches]:                 # :199-200
            continue
        if dist[il, io] < netvlad_distance:               # :202-203
            matches.append((int(il), int(io)))
        else:                                             # :204-205
            break
    return matches, val_each, arg_each


def make_case(seed, n_l, n_r, dim, planted, thr, max_nb, n_lu, n_ou, n_ig, noise=0.05):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n_l, dim))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n_r, dim))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    n_pl = min(planted, n_l, n_r)
    rows = rng.permutation(n_r)[:n_pl]
    src = rng.permutation(n_l)[:n_pl]
    for r, s in zip(rows, src):
        v = a[s] + rng.normal(size=dim) * (noise * rng.uniform(0.3, 1.5) / np.sqrt(dim))
        b[r] = v / np.linalg.norm(v)
    # two received rows close to the SAME local row -> exercises the "idx_other already taken"
    # rule from the other side: two local rows whose nearest is the same received row
    if n_l >= 4 and n_r >= 2:
        v = b[rows[0]] if n_pl else b[0]
        tgt = int(rows[0]) if n_pl else 0
        extra = [i for i in range(n_l) if i not in set(src.tolist())][:2]
        for j, e in enumerate(extra):
            w = v + rng.normal(size=dim) * (noise * (0.4 + 0.3 * j) / np.sqrt(dim))
            a[e] = w / np.linalg.norm(w)
        _ = tgt
    # float32-representable inputs: the HIP path ingests float32 rows
    a = a.astype(np.float32).astype(np.float64)
    b = b.astype(np.float32).astype(np.float64)
    lu = rng.permutation(n_l)[:n_lu].tolist()
    ou = rng.permutation(n_r)[:n_ou].tolist()
    ig = []
    # ignore some true nearest pairs so the second-nearest column has to be found
    m0, v0, a0 = drive_find_matches(a, b, [], [], [], 10.0, n_l)
    for (il, io) in m0[: n_ig // 2]:
        ig.append([il, io])
    while len(ig) < n_ig:
        ig.append([int(rng.integers(n_l)), int(rng.integers(n_r))])
    matches, val_each, arg_each = drive_find_matches(a, b, lu, ou, ig, thr, max_nb)
    return dict(local=a.astype(np.float32), received=b.astype(np.float32), local_used=np.array(lu, dtype=np.int32),
                other_used=np.array(ou, dtype=np.int32),
                ignored=np.array(ig, dtype=np.int32).reshape(-1, 2),
                netvlad_distance=np.float64(thr), max_matches_nb=np.int32(max_nb),
                matches=np.array(matches, dtype=np.int32).reshape(-1, 2),
                row_min=val_each, row_arg=arg_each.astype(np.int32))


CASES = {
    # name: (seed, n_l, n_r, dim, planted, thr, max_nb, n_local_used, n_other_used, n_ignored)
    "nn_default_128": (101, 60, 80, 128, 12, 0.13, 20, 0, 0, 0),
    "nn_masks_128": (102, 90, 70, 128, 25, 0.13, 20, 5, 4, 6),
    "nn_loose_thr": (103, 40, 50, 128, 10, 2.5, 20, 3, 3, 4),
    "nn_maxnb_small": (104, 64, 64, 64, 30, 0.13, 5, 0, 2, 2),
    "nn_dim4096": (105, 48, 56, 4096, 10, 0.13, 20, 2, 2, 2),
    "nn_ragged_1xN": (106, 1, 37, 128, 1, 0.13, 20, 0, 0, 0),
    "nn_ragged_Nx1": (107, 33, 1, 128, 1, 0.13, 20, 0, 0, 0),
    "nn_all_rows_used": (108, 8, 9, 32, 4, 0.5, 20, 8, 0, 0),
    "nn_unaligned": (109, 131, 67, 100, 20, 0.13, 50, 4, 5, 8),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, cfg in CASES.items():
        case = make_case(*cfg)
        # reject cases with (near-)ties among finite row minima or inside a row's top two
        fin = np.sort(case["row_min"][np.isfinite(case["row_min"])])
        if fin.size > 1 and np.min(np.diff(fin)) < 1e-9:
            print("tie in", name, file=sys.stderr)
            sys.exit(1)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **case)
        print("%-20s n_l=%d n_r=%d dim=%d matches=%d" % (
            name, case["local"].shape[0], case["received"].shape[0], case["local"].shape[1],
            case["matches"].shape[0]))


if __name__ == "__main__":
    main()
