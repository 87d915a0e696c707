"""The oracle's float32-descriptor matching (sf_params.desc_type 1; oracle/sf_oracle.c sfo_match_global with desc_type 1)
against a numpy restatement: float32 accumulation of squared differences in dimension order, kNN-2 with strict comparisons
(ties keep the lower id), NNDR on the squared distances, ids matched exactly once on each side
(PKG/src/myRegistrationVis.cpp:839-894 with float rows behind VWDictionary::addNewWords [upstream])."""
import numpy as np


def test_float_global_matching_equals_numpy(oracle):
    """The oracle's float32 scan against a numpy restatement (float32 accumulation in dimension order) of the kNN-2 +
    NNDR + uniqueness rule, with the ties integer-valued rows produce."""
    rng = np.random.default_rng(9)
    for kf, kt, dims in ((50, 70, 64), (130, 90, 128), (2, 5, 64), (1, 5, 64)):
        df = rng.integers(0, 2, size=(kf, dims)).astype(np.float32) * 2 - 1
        dt = df[rng.integers(0, kf, size=kt)].copy()
        dt[::3] += rng.normal(scale=0.3, size=dt[::3].shape).astype(np.float32)
        dt[1::5] = rng.normal(size=dt[1::5].shape).astype(np.float32)
        cf, ct, wf, wt, wt2 = oracle.match_global(df, dt, 0.6, desc_type=1)
        match = -np.ones(kt, dtype=np.int64)
        for t in range(kt):
            dist = np.zeros(kf, dtype=np.float32)
            for k in range(dims):
                dd = (dt[t, k] - df[:, k]).astype(np.float32)
                dist = (dist + dd * dd).astype(np.float32)
            order = np.argsort(dist, kind="stable")
            if kf >= 2 and not dist[order[0]] > np.float32(0.6) * dist[order[1]]:
                match[t] = order[0]
        cnt = np.bincount(match[match >= 0], minlength=kf)
        exp = [(f_, int(np.nonzero(match == f_)[0][0])) for f_ in range(kf) if cnt[f_] == 1]
        assert list(zip(cf.tolist(), ct.tolist())) == exp
