"""CPU restatement (numpy) of landmark triangulation — TEST INFRASTRUCTURE, the checker for bsgpu_triangulate.

Follows the reference's call sites
  VisualOdometry::TriangulateLandmark      bs_models/src/visual_odometry.cpp:532-610
  SLAMInitialization::TriangulateLandmark  bs_models/src/slam_initialization.cpp:699-701
  VisualMap::GetCameraPose                 bs_models/src/lib/vision/visual_map.cpp:43-54  (T_world_camera = T_world_baselink T_cam_baselink^-1)
and the [EXT] beam_cv::Triangulation::TriangulatePoint they call.  libbeam is not under /root/reference (un-vendored,
version unpinned — SURVEY.md §8c), so the DLT below is its published algorithm recalled, not checked against a build:
PARITY UNPINNED for the triangulated coordinates; what the reference's own sources pin is the call contract
(>= 2 views, integer-truncated pixels, the two thresholds of vo_params.json:2-3, "no point" on rejection).
"""
import numpy as np

OK, TOO_FEW_VIEWS, BEHIND_CAMERA, TOO_FAR, REPROJECTION, AT_INFINITY = range(6)


def _rot(q):
    """Eigen::Quaterniond(w, x, y, z).toRotationMatrix(), no normalisation (visual_map.cpp:61-64)."""
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def camera_from_world(q, p, R_cb, t_cb):
    """T_camera_world (3x4) = T_cam_baselink * T_world_baselink^-1 (visual_odometry.cpp:545-546)."""
    R = R_cb @ _rot(q).T
    return np.hstack([R, (t_cb - R @ p)[:, None]])


def triangulate_point(T_cam_world, pixels, K, max_dist=-1.0, max_reproj=-1.0):
    """[EXT] beam_cv::Triangulation::TriangulatePoint -> (point[3], status)."""
    fx, fy, cx, cy = K
    if len(T_cam_world) < 2:                                   # visual_odometry.cpp:572
        return np.zeros(3), TOO_FEW_VIEWS
    A = np.zeros((2 * len(T_cam_world), 4))
    for i, (T, z) in enumerate(zip(T_cam_world, pixels)):
        m = np.array([(z[0] - cx) / fx, (z[1] - cy) / fy, 1.0])
        m /= np.linalg.norm(m)                                 # BackProject returns a unit bearing
        A[2 * i] = m[0] * T[2] - m[2] * T[0]
        A[2 * i + 1] = m[1] * T[2] - m[2] * T[1]
    h = np.linalg.svd(A)[2][-1]                                # JacobiSVD(A, ComputeFullV).matrixV().col(3)
    if h[3] == 0.0:
        return np.zeros(3), AT_INFINITY
    P = h[:3] / h[3]                                           # .hnormalized()
    for T, z in zip(T_cam_world, pixels):
        pc = T[:, :3] @ P + T[:, 3]
        if pc[2] < 0.0:
            return P, BEHIND_CAMERA
        if max_dist > 0.0 and np.linalg.norm(pc) > max_dist:
            return P, TOO_FAR
        if max_reproj > 0.0:
            proj = np.array([fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy])
            if not np.linalg.norm(np.asarray(z, float) - proj) <= max_reproj:
                return P, REPROJECTION
    return P, OK


def triangulate_tracks(values, offsets, track_start, q_block, p_block, pixels, camera, truncate_pixels=True, max_dist=-1.0,
                       max_reproj=-1.0):
    """Batch form with the argument meaning of bsgpu_triangulate (include/bsgpu.h).  camera = (fx, fy, cx, cy, R_cb[9], t_cb[3])."""
    cam = np.asarray(camera, float)
    K, R_cb, t_cb = cam[:4], cam[4:13].reshape(3, 3), cam[13:16]
    pixels = np.asarray(pixels, float).reshape(-1, 2)
    if truncate_pixels:
        pixels = np.trunc(pixels)                              # m.value.cast<int>()  (visual_odometry.cpp:547)
    n = len(track_start) - 1
    pts, st = np.zeros((n, 3)), np.zeros(n, np.int32)
    for i in range(n):
        obs = range(track_start[i], track_start[i + 1])
        Ts = [camera_from_world(values[offsets[q_block[o]]:offsets[q_block[o]] + 4], values[offsets[p_block[o]]:offsets[p_block[o]] + 3],
                                R_cb, t_cb) for o in obs]
        pts[i], st[i] = triangulate_point(Ts, [pixels[o] for o in obs], K, max_dist, max_reproj)
    return pts, st
