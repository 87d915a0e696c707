"""Soak of the accepted-result stream (sf_accept_stream_*): random database sizes, feature counts, estimators, threshold
and duplicate rows through sf_find_matches_and_verify_device with the stream on; every streamed record must be the
gathered result of its slot, every accepted match must have been streamed, and the NN matches must equal a handle's
that never streams.  usage: python tools/soak_stream.py [rounds=60]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from multi_robot_slam_separators_amd import _abi, lib, synth  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    bad = 0
    total = streamed_calls = 0
    for rd in range(rounds):
        rng = np.random.default_rng(4000 + rd)
        n_kf = int(rng.integers(20, 400))
        k = int(rng.choice([64, 200, 500]))
        dim = int(rng.choice([128, 512, 4096]))
        est = int(rng.integers(0, 2))
        feats = synth.make_store_batch(500 + rd, n_kf, k=k, cols=32, true_frac=float(rng.uniform(0.1, 0.9)))
        nv_a = rng.normal(size=(n_kf, dim)); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
        nv_b = nv_a + 0.002 * rng.normal(size=(n_kf, dim)); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
        for _ in range(int(rng.integers(0, 4))):                    # rows sharing a nearest column
            i, j = rng.integers(0, n_kf, 2)
            nv_b[i] = nv_b[j]
        p = synth.camera_params()
        p.estimation_type = est
        p.iterations = int(rng.choice([50, 200]))
        p.netvlad_dimensions = dim
        p.netvlad_max_matches_nb = n_kf
        p.netvlad_distance = 0.13
        p.max_features = k

        def up(x):
            x = np.ascontiguousarray(x)
            return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
        T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}
        outs = []
        for stream in (True, False):
            with lib.SeparatorFinder(p) as f:
                f.set_stream(torch.cuda.current_stream().cuda_stream)
                sa = f.store_add_keyframes_device(n_kf, k, 32, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
                sb = f.store_add_keyframes_device(n_kf, k, 32, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
                f.nn_append_received(nv_a); f.nn_append_local(nv_b)
                cap = n_kf + n_kf // 8 + 320
                rec = torch.zeros((cap, 368), dtype=torch.uint8).pin_memory()
                idx = torch.full((cap,), -1, dtype=torch.int32).pin_memory()
                if stream:
                    f.accept_stream_set(0, rec.data_ptr(), idx.data_ptr(), None, cap)
                    f.accept_stream_select(0)
                d_res = torch.zeros((n_kf, 368), dtype=torch.uint8, device=dev)
                for rep in range(3):                                 # (counter blocks alternate between queries)
                    rec.zero_(); idx.fill_(-1)
                    m = f.find_matches_and_verify_device(sa, sb, d_res.data_ptr(), cap=n_kf)
                    torch.cuda.synchronize()
                    res = np.frombuffer(d_res.cpu().numpy()[:len(m)].tobytes(), dtype=_abi.RESULT_DTYPE)
                    ok = True
                    if stream:
                        st, pairs = f.accept_stream_status()
                        r, ix, n = f.last_match_results()
                        if st:
                            streamed_calls += 1
                            slot_of_match = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_int32)), shape=(n,)).copy() \
                                if ix else np.arange(n, dtype=np.int32)
                            got = idx.numpy()
                            n_all = int((got >= 0).sum())
                            ok = ok and (got[:n_all] >= 0).all() and (got[n_all:] == -1).all() and len(set(got[:n_all].tolist())) == n_all
                            recs = np.frombuffer(rec.numpy()[:n_all].tobytes(), dtype=_abi.RESULT_DTYPE)
                            by = {int(j): recs[i] for i, j in enumerate(got[:n_all])}
                            ok = ok and bool(recs["success"].all())
                            for i in range(n):
                                j = int(slot_of_match[i])
                                if res["success"][i]:
                                    ok = ok and j in by and by[j].tobytes() == res[i].tobytes()
                                else:
                                    ok = ok and j not in by
                        else:
                            ok = ok and (idx.numpy() == -1).all()
                    total += 1
                    if not ok:
                        bad += 1
                        print("round %d rep %d: MISMATCH (n_kf %d, k %d, dim %d, est %d)" % (rd, rep, n_kf, k, dim, est), flush=True)
                outs.append((m.tobytes(), res.tobytes()))
        if outs[0] != outs[1]:
            bad += 1
            print("round %d: streaming handle and plain handle disagree" % rd, flush=True)
        if rd % 10 == 9:
            print("round %d: %d queries checked (%d streamed), %d bad" % (rd, total, streamed_calls, bad), flush=True)
    print("STREAM SOAK DONE: %d rounds, %d queries (%d streamed), %d mismatching" % (rounds, total, streamed_calls, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
