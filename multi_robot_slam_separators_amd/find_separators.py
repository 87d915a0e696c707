"""One tick of the reference's main loop (PKG/scripts/find_separators.py:45-137), ROS-free: the
CALLER of the hot path.  Service proxies become direct calls on the peer's objects; the relay
node (communication.cpp) and the 0.3 Hz rate limiter are transport and are not mirrored."""
import numpy as np

from .messages import EstTransformRequest, FindMatchesRequest, ReceiveSeparatorsRequest


def find_separators_tick(dataHandler, geometry, peer_dataHandler):
    """Robot `dataHandler` (querying) against `peer_dataHandler` (computing).  Returns the
    ReceiveSeparatorsRequest sent to the peer (None when nothing was exchanged)."""
    if len(dataHandler.local_descriptors) == 0:                                   # :55
        return None
    # :59-63 only send descriptors which have not been sent yet
    to_send = dataHandler.local_descriptors[dataHandler.nb_descriptors_already_sent:]
    flat = np.asarray(to_send, dtype=np.float64).reshape(-1)
    res_matches = peer_dataHandler.find_matches_service(FindMatchesRequest(flat))
    dataHandler.nb_descriptors_already_sent += len(to_send)                        # :68
    if not res_matches.kf_ids_computing_robot:                                     # :71-72
        return None
    frames_from, frames_to, kf_to, success, separators = [], [], [], [], []
    for i in range(len(res_matches.kf_ids_computing_robot)):                       # :83
        local = dataHandler.get_geom_features(res_matches.frames_kept_ids_querying_robot[i])   # :85-86
        # transform FROM the local (querying) frame TO the other robot's frame, :89-91
        res = geometry.estimateTransformation(EstTransformRequest(
            local.descriptors, res_matches.descriptors_vec[i], local.kpts3D, res_matches.kpts3D_vec[i],
            local.kpts, res_matches.kpts_vec[i]))
        success.append(bool(res.success))                                          # :98-101
        frames_from.append(res_matches.frames_kept_ids_querying_robot[i])          # :103-105
        frames_to.append(res_matches.frames_kept_ids_computing_robot[i])
        kf_to.append(res_matches.kf_ids_computing_robot[i])                        # :107-108
        separators.append(res.poseWithCov)                                         # :110
    kf_from = dataHandler.get_kf_ids_from_frames_kept_ids(frames_from)             # :115-116
    dataHandler.found_separators_local(kf_from, kf_to, frames_from, frames_to, [], [], success, separators)  # :128
    req = ReceiveSeparatorsRequest(dataHandler.local_robot_id, dataHandler.other_robot_id, kf_from, kf_to,
                                   frames_from, frames_to, [], [], success, separators)
    peer_dataHandler.receive_separators_service(req)                               # :132-133
    return req
