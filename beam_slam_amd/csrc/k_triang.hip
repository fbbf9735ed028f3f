// Landmark triangulation for a batch of feature tracks — the factor producer that creates the Point3DLandmark
// blocks the reprojection factors attach to.  Follows the reference's call sites
//   VisualOdometry::TriangulateLandmark      bs_models/src/visual_odometry.cpp:532-610
//   SLAMInitialization::TriangulateLandmark  bs_models/src/slam_initialization.cpp:699-701
// which collect, per track, T_camera_world = (T_world_baselink * T_cam_baselink^-1)^-1 of every keyframe of the
// track that is in the graph (VisualMap::GetCameraPose, bs_models/src/lib/vision/visual_map.cpp:43-54) and the
// measured pixel truncated to integers (`m.value.cast<int>()`, visual_odometry.cpp:547), demand >= 2 views (:572)
// and call [EXT] beam_cv::Triangulation::TriangulatePoint(cam, T_cam_world, pixels, max_dist, max_reprojection).
// libbeam is not under /root/reference (un-vendored, version unpinned); its published algorithm,
// restated here: back-project each pixel to a unit bearing m, stack the two DLT rows
//      m.x * T.row(2) - m.z * T.row(0),   m.y * T.row(2) - m.z * T.row(1)
// per view into A (2V x 4), take the right singular vector of the smallest singular value, de-homogenise, and
// reject the point if in any view it is behind the camera, farther than max_dist (> 0) or re-projects more than
// max_reprojection (> 0) pixels from the measurement.  The camera is the skew-free pinhole (K of the camera table)
// the reprojection factors themselves use.
//
// One lane per track: a track has 2..~12 views, the 4x4 Gram matrix A^T A and its eigenvectors (cyclic Jacobi,
// all indices compile-time => registers) live in the lane; the smallest eigenvector of A^T A is the wanted singular
// vector.  Streamed per view: 8 B of offsets + 16 B pixel; the keyframe poses are gathered (L2-resident: a window has
// a few hundred of them).  HBM-bound, ~24 B/view + 28 B/track out.
#include "bsgpu_device.h"

namespace bsg {

namespace {

// rows 0..2 of T_camera_world = T_cam_baselink * T_world_baselink^-1, row-major 3x4
BSG_DEV void camera_from_world(const double* __restrict__ x, int xq, int xp, const DevCamera& cam, double T[12]) {
  const double q[4] = {x[xq], x[xq + 1], x[xq + 2], x[xq + 3]};
  const double t[3] = {x[xp], x[xp + 1], x[xp + 2]};
  double Rwb[9];
  quat_to_rot(q, Rwb);
  // R = Rcb * Rwb^T ;  tt = tcb - R t
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      T[4 * i + j] = cam.R[3 * i] * Rwb[3 * j] + cam.R[3 * i + 1] * Rwb[3 * j + 1] + cam.R[3 * i + 2] * Rwb[3 * j + 2];
    T[4 * i + 3] = cam.t[i] - (T[4 * i] * t[0] + T[4 * i + 1] * t[1] + T[4 * i + 2] * t[2]);
  }
}

template <int P, int Q>
BSG_DEV void jacobi_rotate(double a[4][4], double v[4][4]) {
  const double apq = a[P][Q];
  if (apq == 0.0) return;
  const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // A <- A G
    const double akp = a[k][P], akq = a[k][Q];
    a[k][P] = c * akp - s * akq;
    a[k][Q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // A <- G^T A
    const double apk = a[P][k], aqk = a[Q][k];
    a[P][k] = c * apk - s * aqk;
    a[Q][k] = s * apk + c * aqk;
  }
  a[P][Q] = a[Q][P] = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double vkp = v[k][P], vkq = v[k][Q];
    v[k][P] = c * vkp - s * vkq;
    v[k][Q] = s * vkp + c * vkq;
  }
}

}  // namespace

// status: 0 = triangulated; 1 = fewer than 2 views; 2 = behind a camera; 3 = farther than max_dist;
//         4 = re-projection above max_reproj; 5 = point at infinity (homogeneous w == 0)
__global__ __launch_bounds__(256) void triangulate_kernel(int n_tracks, const int* __restrict__ track_start,
                                                          const int2* __restrict__ pose_off, const double2* __restrict__ pix,
                                                          const double* __restrict__ x, DevCamera cam, int truncate,
                                                          double max_dist, double max_reproj, double* __restrict__ points,
                                                          int* __restrict__ status) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n_tracks) return;
  const int beg = track_start[l], end = track_start[l + 1];
  double P[3] = {0.0, 0.0, 0.0};
  int st = 0;
  if (end - beg < 2) {
    st = 1;
  } else {
    double a[4][4], v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[i][j] = 0.0; v[i][j] = i == j ? 1.0 : 0.0; }
    for (int o = beg; o < end; ++o) {
      const int2 po = pose_off[o];
      double2 z = pix[o];
      if (truncate) { z.x = trunc(z.x); z.y = trunc(z.y); }
      double T[12];
      camera_from_world(x, po.x, po.y, cam, T);
      double m[3] = {(z.x - cam.cx) / cam.fx, (z.y - cam.cy) / cam.fy, 1.0};
      const double inv = 1.0 / sqrt(m[0] * m[0] + m[1] * m[1] + 1.0);
      m[0] *= inv; m[1] *= inv; m[2] *= inv;
      double r0[4], r1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r0[k] = m[0] * T[8 + k] - m[2] * T[k];
        r1[k] = m[1] * T[8 + k] - m[2] * T[4 + k];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] += r0[i] * r0[j] + r1[i] * r1[j];
    }
    // cyclic Jacobi on the 4x4 Gram matrix; converges quadratically, 8 sweeps are far past double precision
    for (int sweep = 0; sweep < 8; ++sweep) {
      const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[0][3]) + fabs(a[1][2]) + fabs(a[1][3]) + fabs(a[2][3]);
      if (off == 0.0) break;
      jacobi_rotate<0, 1>(a, v); jacobi_rotate<0, 2>(a, v); jacobi_rotate<0, 3>(a, v);
      jacobi_rotate<1, 2>(a, v); jacobi_rotate<1, 3>(a, v); jacobi_rotate<2, 3>(a, v);
    }
    // eigenvector of the smallest eigenvalue (selects instead of dynamic indexing keep v in registers)
    double best = a[0][0];
    double h[4] = {v[0][0], v[1][0], v[2][0], v[3][0]};
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const bool lt = a[j][j] < best;
      best = lt ? a[j][j] : best;
#pragma unroll
      for (int k = 0; k < 4; ++k) h[k] = lt ? v[k][j] : h[k];
    }
    if (h[3] == 0.0) {
      st = 5;
    } else {
      P[0] = h[0] / h[3]; P[1] = h[1] / h[3]; P[2] = h[2] / h[3];
      for (int o = beg; o < end && st == 0; ++o) {
        const int2 po = pose_off[o];
        double2 z = pix[o];
        if (truncate) { z.x = trunc(z.x); z.y = trunc(z.y); }
        double T[12];
        camera_from_world(x, po.x, po.y, cam, T);
        double pc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pc[i] = T[4 * i] * P[0] + T[4 * i + 1] * P[1] + T[4 * i + 2] * P[2] + T[4 * i + 3];
        if (pc[2] < 0.0) { st = 2; break; }
        if (max_dist > 0.0 && sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]) > max_dist) { st = 3; break; }
        if (max_reproj > 0.0) {
          const double du = z.x - (cam.fx * pc[0] / pc[2] + cam.cx), dv = z.y - (cam.fy * pc[1] / pc[2] + cam.cy);
          if (!(sqrt(du * du + dv * dv) <= max_reproj)) { st = 4; break; }
        }
      }
    }
  }
  points[3 * (size_t)l] = P[0]; points[3 * (size_t)l + 1] = P[1]; points[3 * (size_t)l + 2] = P[2];
  status[l] = st;
}

void launch_triangulate(hipStream_t s, int n_tracks, const int* track_start, const int2* pose_off, const double2* pix, const double* x,
                        const DevCamera& cam, bool truncate, double max_dist, double max_reproj, double* points, int* status) {
  if (n_tracks <= 0) return;
  hipLaunchKernelGGL(triangulate_kernel, dim3((n_tracks + 255) / 256), dim3(256), 0, s, n_tracks, track_start, pose_off, pix, x, cam,
                     truncate ? 1 : 0, max_dist, max_reproj, points, status);
}

}  // namespace bsg
