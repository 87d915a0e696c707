#!/usr/bin/env python3
"""NN-stage soak: random database sizes, dimensions, thresholds, masks, ignored pairs and incremental
appends; the matches of both GPU paths (fp16 filter and fp32 ranking) must equal the oracle's.
usage: tools/soak_nn.py [rounds=50] [n_max=500]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_robot_slam_separators_amd import lib, _abi
from oracle import pyoracle
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_max = int(sys.argv[2]) if len(sys.argv) > 2 else 500      # database sizes are drawn from [1, n_max)
t0 = time.time(); nq = 0
for rd in range(rounds):
    rng = np.random.default_rng(7000 + rd)
    dim = int(rng.choice([8, 33, 100, 128, 512, 1100, 2048, 4096]))
    n_l, n_r = int(rng.integers(1, n_max)), int(rng.integers(1, n_max))
    scale = float(rng.choice([1.0, 0.2, 30.0]))
    a = rng.normal(size=(n_l, dim)); a /= np.linalg.norm(a, axis=1, keepdims=True); a *= scale
    b = rng.normal(size=(n_r, dim)); b /= np.linalg.norm(b, axis=1, keepdims=True); b *= scale
    npl = int(rng.integers(0, min(n_l, n_r) + 1))
    rows = rng.permutation(n_r)[:npl]; src = rng.permutation(n_l)[:npl]
    b[rows] = a[src] + rng.normal(size=(npl, dim)) * scale * rng.uniform(0.01, 0.2, size=(npl, 1)) / np.sqrt(dim)
    a = a.astype(np.float32); b = b.astype(np.float32)
    thr = float(rng.choice([0.05, 0.13, 0.3, 1.0])) * scale
    max_nb = int(rng.choice([1, 5, 20, 1000]))
    for precision in (1, 0):
        p = _abi.default_params(); p.netvlad_distance = thr; p.netvlad_max_matches_nb = max_nb
        p.netvlad_dimensions = dim; p.nn_precision = precision
        lu, ou, ig = [], [], []
        with lib.SeparatorFinder(p) as f:
            cut_l, cut_r = int(rng.integers(0, n_l + 1)), int(rng.integers(0, n_r + 1))
            sent_l = sent_r = 0
            for phase, (el, er) in enumerate([(max(cut_l, 1), max(cut_r, 1)), (n_l, n_r)]):
                if el > sent_l: f.nn_append_local(a[sent_l:el]); sent_l = el
                if er > sent_r: f.nn_append_received(b[sent_r:er]); sent_r = er
                m = f.nn_find_matches()
                mo, _, _ = pyoracle.find_matches(a[:sent_l].astype(np.float64), b[:sent_r].astype(np.float64), lu, ou, ig, thr, max_nb)
                assert np.array_equal(m["idx_local"], mo["idx_local"]) and np.array_equal(m["idx_other"], mo["idx_other"]), (rd, precision, phase)
                assert np.allclose(m["distance"], mo["distance"], rtol=1e-12, atol=0)
                nq += 1
                # feed results back like receive_separators_service: accept some, ignore the others
                for r in m:
                    if rng.random() < 0.5:
                        f.nn_mark_local_used(r["idx_local"]); f.nn_mark_other_used(r["idx_other"])
                        lu.append(int(r["idx_local"])); ou.append(int(r["idx_other"]))
                    else:
                        f.nn_ignore_pair(r["idx_local"], r["idx_other"]); ig.append((int(r["idx_local"]), int(r["idx_other"])))
                m2 = f.nn_find_matches()
                mo2, _, _ = pyoracle.find_matches(a[:sent_l].astype(np.float64), b[:sent_r].astype(np.float64), lu, ou, ig, thr, max_nb)
                assert np.array_equal(m2["idx_local"], mo2["idx_local"]) and np.array_equal(m2["idx_other"], mo2["idx_other"]), (rd, precision, phase, "masked")
                nq += 1
print("NN SOAK DONE: %d rounds, %d queries identical to the oracle, %.0f s" % (rounds, nq, time.time() - t0))
