"""Landmark triangulation — the factor producer of SURVEY.md §8f rank 4.
CPU: the numpy restatement (oracle/triangulation.py) recovers known points and honours the call contract of
bs_models/src/visual_odometry.cpp:532-610.  GPU: bsgpu_triangulate against that restatement on the tracks of a VIO window.
Tolerance: 1e-7 relative on the coordinates — the device takes the smallest eigenvector of A^T A (4x4, in registers), the
restatement the singular vector of A; the two agree to ~cond(A) * 1e-16."""
import os
import sys

import numpy as np
import pytest

from beam_slam_amd import capi, synthetic

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import triangulation as tri  # noqa: E402


def _tracks_of(pr):
    """(track_start, q_block, p_block, pixels, landmark block per track) from the window's reprojection factors."""
    idx = np.concatenate([f[0] for f in pr.factors[capi.F_REPROJ]])
    consts = np.concatenate([f[1] for f in pr.factors[capi.F_REPROJ]])
    order = np.argsort(idx[:, 2], kind="stable")
    idx, consts = idx[order], consts[order]
    lms, counts = np.unique(idx[:, 2], return_counts=True)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return start, idx[:, 0].astype(np.int32), idx[:, 1].astype(np.int32), consts[:, :2].copy(), lms


def _camera_row(pr, cam=0):
    c = pr.cameras[cam]
    return np.array([c.fx, c.fy, c.cx, c.cy, *c.R_cam_baselink, *c.t_cam_baselink])


def test_restatement_recovers_points_from_exact_pixels():
    pr = synthetic.vio_window(n_kf=12, n_lm=80, seed=5, pixel_sigma=0.0, sigma_rot=0.0, sigma_pos=0.0, sigma_lm=0.0)
    start, qb, pb, px, lms = _tracks_of(pr)
    cam = _camera_row(pr)
    # exact (un-rounded) projections of the true points
    K, R_cb, t_cb = cam[:4], cam[4:13].reshape(3, 3), cam[13:]
    for i in range(len(lms)):
        for o in range(start[i], start[i + 1]):
            T = tri.camera_from_world(pr.block(int(qb[o])), pr.block(int(pb[o])), R_cb, t_cb)
            pc = T[:, :3] @ pr.block(int(lms[i])) + T[:, 3]
            px[o] = [K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]]
    pts, st = tri.triangulate_tracks(pr.values, pr.offset, start, qb, pb, px, cam, truncate_pixels=False, max_dist=30.0, max_reproj=20.0)
    assert (st == tri.OK).all()
    truth = np.stack([pr.block(int(b)) for b in lms])
    assert np.abs(pts - truth).max() <= 1e-8
    # the reference truncates the pixel to integers before back-projecting (visual_odometry.cpp:547): sub-pixel error only
    pts_t, st_t = tri.triangulate_tracks(pr.values, pr.offset, start, qb, pb, px, cam, truncate_pixels=True, max_dist=30.0, max_reproj=20.0)
    assert (st_t == tri.OK).all() and 1e-6 < np.abs(pts_t - truth).max() < 1.0


def test_restatement_call_contract():
    cam = np.array([458.654, 457.296, 367.215, 248.375, *np.eye(3).ravel(), 0, 0, 0])
    q = np.array([1.0, 0, 0, 0])
    values = np.concatenate([q, [0, 0, 0], q, [1.0, 0, 0]])
    offsets = [0, 4, 7, 11]
    P = np.array([0.3, -0.2, 6.0])

    def pix(p):
        return [cam[0] * p[0] / p[2] + cam[2], cam[1] * p[1] / p[2] + cam[3]]

    z = np.array([pix(P), pix(P - [1.0, 0, 0])])
    args = (values, offsets)
    pts, st = tri.triangulate_tracks(*args, [0, 2], [0, 2], [1, 3], z, cam, False)
    assert st[0] == tri.OK and np.abs(pts[0] - P).max() < 1e-9
    # one view only: no point (visual_odometry.cpp:572)
    assert tri.triangulate_tracks(*args, [0, 1], [0], [1], z[:1], cam, False)[1][0] == tri.TOO_FEW_VIEWS
    # farther than max_triangulation_distance
    assert tri.triangulate_tracks(*args, [0, 2], [0, 2], [1, 3], z, cam, False, max_dist=5.0)[1][0] == tri.TOO_FAR
    # diverging rays intersect behind the cameras
    zb = np.array([pix(P - [1.0, 0, 0]), pix(P)])
    assert tri.triangulate_tracks(*args, [0, 2], [0, 2], [1, 3], zb, cam, False)[1][0] == tri.BEHIND_CAMERA
    # an outlier view breaks the re-projection bound: a third camera looking at a different point
    values3 = np.concatenate([values, q, [2.0, 0, 0]])
    z3 = np.array([pix(P), pix(P - [1.0, 0, 0]), pix(P + [0.0, 2.0, 0])])
    st3 = tri.triangulate_tracks(values3, [0, 4, 7, 11, 14, 18], [0, 3], [0, 2, 4], [1, 3, 5], z3, cam, False, max_reproj=5.0)[1]
    assert st3[0] == tri.REPROJECTION


@pytest.mark.gpu
@pytest.mark.parametrize("truncate,max_dist,max_reproj", [(True, 30.0, 20.0), (False, -1.0, -1.0), (True, 9.0, 1.5)])
def test_device_triangulation_matches_restatement(gpu_solver_cls, truncate, max_dist, max_reproj):
    pr = synthetic.vio_window(n_kf=30, n_lm=700, seed=11)
    start, qb, pb, px, lms = _tracks_of(pr)
    # ragged edge cases: an empty track and a single-view track in front, a very long one at the end
    start = np.concatenate([[0, 0, 1], start[1:] + 0]).astype(np.int32)
    start[2] = 1
    g = gpu_solver_cls(0)
    pr.load(g)
    pts, st = g.triangulate(start, qb, pb, px, 0, truncate, max_dist, max_reproj)
    ref, st_ref = tri.triangulate_tracks(pr.values, pr.offset, start, qb, pb, px, _camera_row(pr), truncate, max_dist, max_reproj)
    assert st_ref[0] == tri.TOO_FEW_VIEWS and st_ref[1] == tri.TOO_FEW_VIEWS
    assert np.array_equal(st, st_ref)
    if max_dist > 0 and max_dist < 20:
        assert (st_ref == tri.TOO_FAR).any() or (st_ref == tri.REPROJECTION).any()
    ok = st_ref != tri.TOO_FEW_VIEWS
    assert np.abs(pts[ok] - ref[ok]).max() <= 1e-7 * np.abs(ref[ok]).max()
    assert (pts[~ok] == 0).all()


@pytest.mark.gpu
def test_device_triangulation_uses_the_values_the_solve_left(gpu_solver_cls):
    """After a solve the keyframe poses the kernel reads are the optimised ones on the device (no re-upload)."""
    pr = synthetic.vio_window(n_kf=10, n_lm=120, seed=3)
    start, qb, pb, px, lms = _tracks_of(pr)
    g = gpu_solver_cls(0)
    pr.load(g)
    g.solve()
    x = g.get_blocks()
    pts, st = g.triangulate(start, qb, pb, px, 0, True, 30.0, 20.0)
    ref, st_ref = tri.triangulate_tracks(x, pr.offset, start, qb, pb, px, _camera_row(pr), True, 30.0, 20.0)
    assert np.array_equal(st, st_ref)
    assert np.abs(pts - ref).max() <= 1e-7 * np.abs(ref).max()
    # and the triangulated points sit near the optimised landmark blocks
    good = st == 0
    lm = np.stack([pr.block(int(b), x) for b in lms])
    assert np.median(np.linalg.norm(pts[good] - lm[good], axis=1)) < 0.2


@pytest.mark.gpu
def test_device_triangulation_rejects_malformed_input(gpu_solver_cls):
    pr = synthetic.vio_window(n_kf=6, n_lm=40, seed=2)
    start, qb, pb, px, _ = _tracks_of(pr)
    g = gpu_solver_cls(0)
    pr.load(g)
    with pytest.raises(capi.SolverError):
        g.triangulate(start, pb, qb, px)               # position block where an orientation is expected
    with pytest.raises(capi.SolverError):
        g.triangulate(start, qb, pb, px, camera=3)     # no such camera
    bad = start.copy(); bad[2] = bad[1] - 1
    with pytest.raises(capi.SolverError):
        g.triangulate(bad, qb, pb, px)
