"""Soak of the step pair (sf_step_issue / sf_step_retire) in the forms it takes: random database sizes, feature counts,
thresholds, duplicate rows and max_matches (every row / the reference's 20); per round the reference is the two separate
calls, and the steps of five more handles -- the device-resident step in its speculative and its serial form, the split
form (k_match_split + k_chain, forced down to these sizes with SF_STEP_SPLIT_MIN=1), round 3's host walk, a device mirror
pair -- each with a random ring depth (1 .. 9) and lane count (1 .. 4), nine steps kept `depth` in flight, must deliver the
same matches, the same accepted / rejected decision per match and, for every accepted match, the same 368 bytes.
3D-3D and PnP.  usage: python tools/soak_steps.py [rounds=40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SF_STEP_SPLIT_MIN"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

from multi_robot_slam_separators_amd import _abi, lib, synth  # noqa: E402

RB = _abi.RESULT_DTYPE.itemsize


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    bad = steps = accepted = 0
    for rd in range(rounds):
        rng = np.random.default_rng(9000 + rd)
        n_kf = int(rng.integers(20, 600))
        k = int(rng.choice([64, 200, 500]))
        dim = int(rng.choice([128, 512]))
        est = int(rng.integers(0, 2))
        feats = synth.make_store_batch(700 + rd, n_kf, k=k, cols=32, true_frac=float(rng.uniform(0.1, 0.9)))
        nv_a = rng.normal(size=(n_kf, dim)); nv_a /= np.linalg.norm(nv_a, axis=1, keepdims=True)
        nv_b = nv_a + 0.002 * rng.normal(size=(n_kf, dim)); nv_b /= np.linalg.norm(nv_b, axis=1, keepdims=True)
        for _ in range(int(rng.integers(0, 4))):                    # rows sharing a nearest column
            i, j = rng.integers(0, n_kf, 2)
            nv_b[i] = nv_b[j]
        p = synth.camera_params()
        p.estimation_type = est
        p.iterations = int(rng.choice([50, 200]))
        p.netvlad_dimensions = dim
        p.netvlad_max_matches_nb = n_kf if rng.random() < 0.7 else 20
        p.netvlad_distance = 0.13
        p.max_features = k

        def up(x):
            x = np.ascontiguousarray(x)
            return torch.from_numpy(x.view(np.uint8) if x.dtype.fields else x).to(dev)
        T = {key: up(feats[key]) for key in ("desc_a", "xyz_a", "kp_a", "desc_b", "xyz_b", "kp_b")}

        def fill(f):
            f.set_stream(torch.cuda.current_stream().cuda_stream)
            sa = f.store_add_keyframes_device(n_kf, k, 32, T["desc_a"].data_ptr(), T["xyz_a"].data_ptr(), T["kp_a"].data_ptr())
            sb = f.store_add_keyframes_device(n_kf, k, 32, T["desc_b"].data_ptr(), T["xyz_b"].data_ptr(), T["kp_b"].data_ptr())
            f.nn_append_received(nv_a); f.nn_append_local(nv_b)
            return sa, sb

        with lib.SeparatorFinder(p) as f:
            f.set_option(_abi.SF_OPT_STEP_OVERLAP, 0)
            sa, sb = fill(f)
            m_ref = f.nn_find_matches(cap=n_kf)       # (at most netvlad_max_matches_nb rows are walked)
            d = torch.zeros((max(len(m_ref), 1), RB), dtype=torch.uint8, device=dev)
            f.verify_matches_device(m_ref, sa, sb, d.data_ptr())
            torch.cuda.synchronize()
            res_ref = np.frombuffer(d[: len(m_ref)].cpu().numpy().tobytes(), dtype=_abi.RESULT_DTYPE).copy()
        ok_ref = res_ref["success"].astype(bool)
        accepted += int(ok_ref.sum())

        def check(out, what):
            m, rom, recs, info = out
            good = m.tobytes() == m_ref.tobytes() and np.array_equal(rom >= 0, ok_ref) and bool(recs["success"].all())
            if good:
                for i in np.nonzero(ok_ref)[0]:
                    good = good and recs[rom[i]].tobytes() == res_ref[i].tobytes()
            if not good:
                print("round %d %s: MISMATCH (n_kf %d, k %d, dim %d, est %d)" % (rd, what, n_kf, k, dim, est), flush=True)
            return good

        for form in ("speculative", "serial", "split", "hostwalk", "mirror"):
            with lib.SeparatorFinder(p) as f:
                sa, sb = fill(f)
                depth = int(rng.choice([1, 2, 4, 6, 9]))
                n_lanes = int(rng.choice([1, 2, 3, 4]))
                f.set_option(_abi.SF_OPT_STEP_DEPTH, depth)
                f.set_option(_abi.SF_OPT_STEP_LANES, n_lanes)
                if form == "serial":
                    f.set_option(_abi.SF_OPT_STEP_SPECULATE, 0)
                if form == "split":
                    f.set_option(_abi.SF_OPT_STEP_SPLIT, 1)
                if form == "hostwalk":
                    f.set_option(_abi.SF_OPT_STEP_DEVICE_WALK, 0)
                lanes = [torch.cuda.current_stream()] * 2
                send = None
                if form == "mirror":
                    cap = n_kf + n_kf // 8 + 256
                    send = [torch.zeros((cap + 1, RB), dtype=torch.uint8, device=dev) for _ in range(2)]
                    f.step_mirror_pair((send[0][1:].data_ptr(), send[0].data_ptr()), (send[1][1:].data_ptr(), send[1].data_ptr()), cap)
                    s_even, s_odd = f.step_mirror_streams()
                    lanes = [torch.cuda.current_stream(), torch.cuda.ExternalStream(s_odd)]
                    depth = min(depth, 2)                     # (a mirror's buffer is rewritten two steps later)
                outs, infl, n_steps = [], 0, 9
                for s_i in range(n_steps):
                    if infl >= depth:
                        outs.append(f.step_retire(copy=True))
                        infl -= 1
                    with torch.cuda.stream(lanes[s_i & 1]):
                        if send is not None:
                            send[s_i & 1].zero_()
                        f.step_issue(sa, sb)
                    infl += 1
                while infl:
                    outs.append(f.step_retire(copy=True))
                    infl -= 1
                torch.cuda.synchronize()
                for o in outs:
                    steps += 1
                    bad += 0 if check(o, "%s depth %d lanes %d" % (form, depth, n_lanes)) else 1
                if send is not None:
                    for b, o in ((0, outs[-1]), (1, outs[-2])):       # steps 8 (even) and 7 (odd) wrote last
                        cnt = int(send[b][0, :4].view(torch.int32).item())
                        got = send[b][1: 1 + cnt].cpu().numpy().tobytes()
                        if cnt != o[3]["n_records"] or got != o[2][:cnt].tobytes():
                            bad += 1
                            print("round %d mirror buffer %d: MISMATCH" % (rd, b), flush=True)
                    f.step_mirror(None, None, 0)
        if rd % 10 == 9:
            print("round %d: %d steps checked, %d accepted separators per reference pass so far, %d bad" % (rd, steps, accepted, bad), flush=True)
    print("STEP SOAK DONE: %d rounds, %d steps in five forms, %d mismatching" % (rounds, steps, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
