#!/usr/bin/env python3
"""Generate the golden vectors that PIN the NN stage (tests/golden/nn_*.npz, tests/golden/refsession_*.npz) by
EXECUTING the reference's own functions (oracle/ref_exec.py: the text of PKG/scripts/data_handler.py:166-209,
297-328, 373-408, 421-422, 437-441 is read from /root/reference at run time and exec'd with a logging-only rospy;
numpy / scipy are the reference's arithmetic libraries).  Runs in the build container only; the fixtures are data.

  nn_*.npz          one find_matches call on a given state: inputs, masks, the returned matches, and the per-row
                    minima of the masked distance matrix the reference itself logs (data_handler.py:206)
  refsession_*.npz  a multi-tick session of ONE computing robot: every tick it gains keyframes, serves a
                    find_matches_service request carrying the querying robot's new descriptors (flat float64[],
                    FindMatches.srv:1) and then a receive_separators_service request with seeded success flags,
                    i.e. the mask bookkeeping feeds the next tick's search (a1 + a2 + a3 of SURVEY.md section 8)

Exact ties between finite distances are avoided (numpy's argsort leaves their order unspecified); NEAR-ties (rows
whose two best columns differ by 1e-7 .. 1e-5 relative, below the fp32 expansion's cancellation error) are planted on
purpose in nn_near_ties_*.

Run from the repo root:  python oracle/gen_golden.py
"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ref_exec  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def run_reference(a, b, lu, ou, ig, thr, max_nb):
    matches, dist = ref_exec.find_matches(a, b, lu, ou, ig, thr, max_nb)
    arg_each = np.argsort(dist, axis=1)[:, 0]
    val_each = dist[np.arange(len(dist)), arg_each]
    return matches, val_each, arg_each


def unit_rows(rng, n, dim):
    a = rng.normal(size=(n, dim))
    return a / np.linalg.norm(a, axis=1, keepdims=True)


def make_case(seed, n_l, n_r, dim, planted, thr, max_nb, n_lu, n_ou, n_ig, noise=0.05):
    rng = np.random.default_rng(seed)
    a = unit_rows(rng, n_l, dim)
    b = unit_rows(rng, n_r, dim)
    n_pl = min(planted, n_l, n_r)
    rows = rng.permutation(n_r)[:n_pl]
    src = rng.permutation(n_l)[:n_pl]
    for r, s in zip(rows, src):
        v = a[s] + rng.normal(size=dim) * (noise * rng.uniform(0.3, 1.5) / np.sqrt(dim))
        b[r] = v / np.linalg.norm(v)
    # two local rows whose nearest is the SAME received row ("idx_other already taken", :199-200)
    if n_l >= 4 and n_r >= 2:
        v = b[rows[0]] if n_pl else b[0]
        extra = [i for i in range(n_l) if i not in set(src.tolist())][:2]
        for j, e in enumerate(extra):
            w = v + rng.normal(size=dim) * (noise * (0.4 + 0.3 * j) / np.sqrt(dim))
            a[e] = w / np.linalg.norm(w)
    # float32-representable inputs: the HIP path ingests float32 rows
    a = a.astype(np.float32).astype(np.float64)
    b = b.astype(np.float32).astype(np.float64)
    lu = rng.permutation(n_l)[:n_lu].tolist()
    ou = rng.permutation(n_r)[:n_ou].tolist()
    ig = []
    # ignore some true nearest pairs so the second-nearest column has to be found
    m0, _, _ = run_reference(a, b, [], [], [], 10.0, n_l)
    for (il, io) in m0[: n_ig // 2]:
        ig.append([il, io])
    while len(ig) < n_ig:
        ig.append([int(rng.integers(n_l)), int(rng.integers(n_r))])
    matches, val_each, arg_each = run_reference(a, b, lu, ou, ig, thr, max_nb)
    return pack(a, b, lu, ou, ig, thr, max_nb, matches, val_each, arg_each)


def pack(a, b, lu, ou, ig, thr, max_nb, matches, val_each, arg_each):
    return dict(local=a.astype(np.float32), received=b.astype(np.float32), local_used=np.array(lu, dtype=np.int32),
                other_used=np.array(ou, dtype=np.int32),
                ignored=np.array(ig, dtype=np.int32).reshape(-1, 2),
                netvlad_distance=np.float64(thr), max_matches_nb=np.int32(max_nb),
                matches=np.array(matches, dtype=np.int32).reshape(-1, 2),
                row_min=val_each, row_arg=arg_each.astype(np.int32))


def make_near_tie_case(seed, n_l, n_r, dim, rel_gaps):
    """Every planted local row has TWO received rows at almost the same distance: column j1 at distance d and
    column j2 at d * (1 + gap), gap cycling through rel_gaps (1e-7 .. 1e-5: below / around the cancellation error of
    the fp32 expansion |a|^2 + |b|^2 - 2ab, far above float64 resolution).  The lower distance is alternately the
    lower and the higher column index.  Values are float32-representable, the gap is checked in float64."""
    rng = np.random.default_rng(seed)
    a = unit_rows(rng, n_l, dim).astype(np.float32).astype(np.float64)
    b = unit_rows(rng, n_r, dim).astype(np.float32).astype(np.float64)
    n_pl = min(n_l, n_r // 2)
    cols = rng.permutation(n_r)[: 2 * n_pl].reshape(n_pl, 2)
    rows = rng.permutation(n_l)[:n_pl]
    achieved = []
    for t, (r, (j1, j2)) in enumerate(zip(rows, cols)):
        if t % 2:
            j1, j2 = j2, j1
        gap = rel_gaps[t % len(rel_gaps)]
        u = rng.normal(size=dim); u /= np.linalg.norm(u)
        w = rng.normal(size=dim); w /= np.linalg.norm(w)
        d = 0.02 + 0.06 * rng.random()
        b[j1] = (a[r] + d * u).astype(np.float32)
        # search the scale of the second offset so that, AFTER float32 rounding, dist2 / dist1 - 1 ~ gap
        d1 = np.linalg.norm(a[r] - b[j1])
        lo, hi = d * 0.9, d * 1.1
        best = None
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            cand = (a[r] + mid * w).astype(np.float32).astype(np.float64)
            d2 = np.linalg.norm(a[r] - cand)
            if d2 > d1 * (1 + gap):
                hi = mid
                best = cand if best is None or d2 < np.linalg.norm(a[r] - best) else best
            else:
                lo = mid
        if best is None:
            continue
        b[j2] = best
        achieved.append(np.linalg.norm(a[r] - b[j2]) / d1 - 1.0)
    matches, val_each, arg_each = run_reference(a, b, [], [], [], 0.13, n_l)
    case = pack(a, b, [], [], [], 0.13, n_l, matches, val_each, arg_each)
    case["near_tie_rel_gap"] = np.array(achieved)
    return case


def make_session(seed, dim, ticks, per_tick_local, per_tick_recv, thr, max_nb, p_success):
    """Multi-tick session of the computing robot, executed by the reference's find_matches_service /
    receive_separators_service.  Returns the flat fixture dict."""
    rng = np.random.default_rng(seed)
    h = ref_exec.RefDataHandler(thr, dim, max_nb)
    out = dict(netvlad_distance=np.float64(thr), max_matches_nb=np.int32(max_nb), dim=np.int32(dim),
               ticks=np.int32(ticks))
    pool = unit_rows(rng, 4096, dim)          # places; both robots observe noisy copies of some of them
    next_kf = 0
    for t in range(ticks):
        # the computing robot's new keyframes (data_handler.py:157-158 appends .tolist() rows; kf ids :287)
        nl = int(per_tick_local[t])
        places_l = rng.integers(0, 160, size=nl)
        new_local = pool[places_l] + rng.normal(size=(nl, dim)) * (0.03 / np.sqrt(dim))
        new_local = new_local.astype(np.float32).astype(np.float64)
        kf_ids = []
        for row in new_local:
            next_kf += int(rng.integers(1, 4))            # keyframe ids are not consecutive (frames are skipped)
            kf_ids.append(next_kf)
            h.local_descriptors.append(row.tolist())
            h.kf_ids_of_frames_kept.append(next_kf)
            h.geometric_feats.append(types.SimpleNamespace(descriptors=None, kpts3D=None, kpts=None))
        # the querying robot's new descriptors, flat on the wire
        nr = int(per_tick_recv[t])
        places_r = rng.integers(0, 160, size=nr)
        new_recv = pool[places_r] + rng.normal(size=(nr, dim)) * (0.03 / np.sqrt(dim))
        new_recv = new_recv.astype(np.float32).astype(np.float64)
        req = types.SimpleNamespace(new_netvlad_descriptors=new_recv.reshape(-1).tolist())
        resp = h.find_matches_service(req)
        if len(resp) == 7:       # the early return (:311): FindMatchesResponse of seven empty deques
            kf_matched, comp, quer = [], [], []
        else:
            kf_matched, comp, quer = [list(map(int, x)) for x in resp]
        # geometric verification outcome (seeded), then the feedback request (ReceiveSeparators.srv)
        success = (rng.random(len(comp)) < p_success)
        fb = types.SimpleNamespace(
            robot_from_id=1, robot_to_id=0,
            kf_ids_from=[1000 + q for q in quer], kf_ids_to=list(kf_matched),
            frames_kepts_ids_from=list(quer), frames_kepts_ids_to=list(comp),
            transform_est_success=[bool(s) for s in success], separators=[None] * len(comp),
            pose_estimates_from=[], pose_estimates_to=[])
        h.receive_separators_service(fb)
        out["t%d_new_local" % t] = new_local.astype(np.float32)
        out["t%d_new_local_kf_ids" % t] = np.array(kf_ids, dtype=np.int32)
        out["t%d_new_received" % t] = new_recv.astype(np.float32)
        out["t%d_frames_computing" % t] = np.array(comp, dtype=np.int32)
        out["t%d_frames_querying" % t] = np.array(quer, dtype=np.int32)
        out["t%d_kf_ids_computing" % t] = np.array(kf_matched, dtype=np.int32)
        out["t%d_success" % t] = success.astype(np.uint8)
        out["t%d_local_used_after" % t] = np.array(h.local_kf_already_used, dtype=np.int32)
        out["t%d_other_used_after" % t] = np.array(h.other_kf_already_used, dtype=np.int32)
        out["t%d_ignored_after" % t] = np.array(h.frames_kept_pairs_ignored, dtype=np.int32).reshape(-1, 2)
    out["separators_found_kf"] = np.array([(a, b) for a, b, _ in h.separators_found], dtype=np.int32).reshape(-1, 2)
    return out


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

NEAR_TIES = {
    "nn_near_ties_128": (201, 96, 200, 128, (1e-7, 3e-7, 1e-6, 1e-5)),
    "nn_near_ties_4096": (202, 40, 90, 4096, (1e-7, 3e-7, 1e-6, 1e-5)),
}

SESSIONS = {
    # name: (seed, dim, ticks, new local keyframes per tick, new received per tick, thr, max_nb, P(success))
    "refsession_default": (301, 128, 6, (60, 25, 0, 40, 15, 30), (50, 35, 20, 0, 25, 30), 0.13, 20, 0.6),
    "refsession_first_tick_empty": (302, 128, 4, (0, 40, 25, 12), (30, 0, 25, 20), 0.13, 5, 0.5),
    "refsession_dim64_loose": (303, 64, 5, (30, 30, 30, 30, 30), (30, 30, 30, 30, 30), 0.2, 50, 0.3),
}


def main():
    if not ref_exec.available():
        sys.exit("needs /root/reference (build container only); the committed fixtures are the product of this script")
    os.makedirs(OUT, exist_ok=True)
    for name, cfg in CASES.items():
        case = make_case(*cfg)
        # reject cases with (near-)ties among finite row minima: those have their own fixtures below
        fin = np.sort(case["row_min"][np.isfinite(case["row_min"])])
        if fin.size > 1 and np.min(np.diff(fin)) < 1e-9:
            sys.exit("tie in " + name)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **case)
        print("%-28s n_l=%d n_r=%d dim=%d matches=%d" % (
            name, case["local"].shape[0], case["received"].shape[0], case["local"].shape[1],
            case["matches"].shape[0]))
    for name, cfg in NEAR_TIES.items():
        case = make_near_tie_case(*cfg)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **case)
        g = case["near_tie_rel_gap"]
        print("%-28s n_l=%d n_r=%d dim=%d matches=%d near-tie rows=%d (relative gaps %.1e .. %.1e)" % (
            name, case["local"].shape[0], case["received"].shape[0], case["local"].shape[1],
            case["matches"].shape[0], g.size, g.min(), g.max()))
    for name, cfg in SESSIONS.items():
        sess = make_session(*cfg)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **sess)
        print("%-28s ticks=%d matches per tick=%s separators=%d" % (
            name, int(sess["ticks"]), [int(sess["t%d_frames_computing" % t].size) for t in range(int(sess["ticks"]))],
            sess["separators_found_kf"].shape[0]))


if __name__ == "__main__":
    main()
