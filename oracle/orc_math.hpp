// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU oracle for the small_gicp registration hot path: fixed-size double-precision math.
//
// The reference (koide3/small_gicp v1.0.1) does all of its fixed-size algebra with Eigen 3.4.0, which is an
// external dependency that is ABSENT from /root/reference and from this image (CMakeLists.txt:42-56 fetches it
// from the network).  The handful of Eigen operations the hot path uses are restated here from Eigen 3.4.0's
// published algorithms:
//   * Matrix3d::inverse()                        -> cofactor / determinant closed form          (gicp_factor.hpp:60)
//   * Matrix<double,6,6>::ldlt().solve()         -> LDL^T with diagonal pivoting                (optimizer.hpp:46,109)
//   * SelfAdjointEigenSolver<Matrix3d>::computeDirect() -> trigonometric closed form + kernel extraction
//                                                                                              (normal_estimation.hpp:88-89)
//   * Quaterniond::toRotationMatrix()            -> standard (unnormalised) formula             (lie.hpp:84)
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>

namespace orc {

struct Vec3 {
  double v[3];
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 operator*(double s, const Vec3& a) { return {s * a[0], s * a[1], s * a[2]}; }
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 cross(const Vec3& a, const Vec3& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline double sqnorm(const Vec3& a) { return dot(a, a); }
inline double norm(const Vec3& a) { return std::sqrt(dot(a, a)); }

// Row-major 3x3.
struct Mat3 {
  double m[3][3];
  double& operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
  static Mat3 zero() { return Mat3{{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}}; }
  static Mat3 identity() { return Mat3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
  Vec3 col(int c) const { return {m[0][c], m[1][c], m[2][c]}; }
};
inline Mat3 operator*(const Mat3& a, const Mat3& b) {
  Mat3 r = Mat3::zero();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) r(i, j) += a(i, k) * b(k, j);
  return r;
}
inline Vec3 operator*(const Mat3& a, const Vec3& x) {
  return {a(0, 0) * x[0] + a(0, 1) * x[1] + a(0, 2) * x[2], a(1, 0) * x[0] + a(1, 1) * x[1] + a(1, 2) * x[2], a(2, 0) * x[0] + a(2, 1) * x[1] + a(2, 2) * x[2]};
}
inline Mat3 operator+(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(i, j) + b(i, j);
  return r;
}
inline Mat3 operator*(double s, const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = s * a(i, j);
  return r;
}
inline Mat3 transpose(const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(j, i);
  return r;
}

// Eigen 3.4.0 Matrix3d::inverse(): cofactors / determinant (Eigen/src/LU/InverseImpl.h, compute_inverse<..,3>).
inline Mat3 inverse(const Mat3& a) {
  Mat3 c;
  c(0, 0) = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
  c(0, 1) = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
  c(0, 2) = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
  c(1, 0) = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
  c(1, 1) = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
  c(1, 2) = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
  c(2, 0) = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  c(2, 1) = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
  c(2, 2) = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
  const double det = a(0, 0) * c(0, 0) + a(0, 1) * c(1, 0) + a(0, 2) * c(2, 0);
  return (1.0 / det) * c;
}

// util/lie.hpp:13-23
inline Mat3 skew(const Vec3& x) {
  Mat3 s = Mat3::zero();
  s(0, 1) = -x[2];
  s(0, 2) = x[1];
  s(1, 0) = x[2];
  s(1, 2) = -x[0];
  s(2, 0) = -x[1];
  s(2, 1) = x[0];
  return s;
}

// Rigid transform (Eigen::Isometry3d): x -> R x + t.
struct SE3 {
  Mat3 R;
  Vec3 t;
  static SE3 identity() { return {Mat3::identity(), {0, 0, 0}}; }
};
inline SE3 operator*(const SE3& a, const SE3& b) { return {a.R * b.R, a.R * b.t + a.t}; }
inline Vec3 operator*(const SE3& a, const Vec3& x) { return a.R * x + a.t; }
inline SE3 inverse(const SE3& a) {
  const Mat3 Rt = transpose(a.R);
  return {Rt, -1.0 * (Rt * a.t)};
}
// 4x4 column-major (Eigen's default storage) <-> SE3
inline SE3 se3_from_colmajor16(const double* m) {
  SE3 T;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T.R(r, c) = m[c * 4 + r];
    T.t[r] = m[12 + r];
  }
  return T;
}
inline void se3_to_colmajor16(const SE3& T, double* m) {
  for (int i = 0; i < 16; i++) m[i] = 0.0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) m[c * 4 + r] = T.R(r, c);
    m[12 + r] = T.t[r];
  }
  m[15] = 1.0;
}

using Vec6 = std::array<double, 6>;
struct Mat6 {
  double m[6][6];
  double& operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
  static Mat6 zero() {
    Mat6 z;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) z(i, j) = 0.0;
    return z;
  }
  Mat6& operator+=(const Mat6& o) {
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) m[i][j] += o.m[i][j];
    return *this;
  }
};

// util/lie.hpp:54-71 (Sophus-derived quaternion exp map), returned as (w, x, y, z).
inline std::array<double, 4> so3_exp_quat(const Vec3& omega) {
  const double theta_sq = dot(omega, omega);
  double imag_factor, real_factor;
  if (theta_sq < 1e-10) {
    const double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    const double theta = std::sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    imag_factor = std::sin(half_theta) / theta;
    real_factor = std::cos(half_theta);
  }
  return {real_factor, imag_factor * omega[0], imag_factor * omega[1], imag_factor * omega[2]};
}

// Eigen 3.4.0 QuaternionBase::toRotationMatrix() (no normalisation).
inline Mat3 quat_to_rotation(const std::array<double, 4>& q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Mat3 R;
  R(0, 0) = 1 - (tyy + tzz);
  R(0, 1) = txy - twz;
  R(0, 2) = txz + twy;
  R(1, 0) = txy + twz;
  R(1, 1) = 1 - (txx + tzz);
  R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy;
  R(2, 1) = tyz + twx;
  R(2, 2) = 1 - (txx + tyy);
  return R;
}

// util/lie.hpp:77-96  se3_exp, twist = [rx ry rz tx ty tz] (rotation first).
inline SE3 se3_exp(const Vec6& a) {
  const Vec3 omega{a[0], a[1], a[2]};
  const Vec3 v{a[3], a[4], a[5]};
  const double theta_sq = dot(omega, omega);
  const double theta = std::sqrt(theta_sq);
  SE3 se3;
  se3.R = quat_to_rotation(so3_exp_quat(omega));
  if (theta < 1e-10) {
    se3.t = se3.R * v;
  } else {
    const Mat3 Omega = skew(omega);
    const Mat3 V = Mat3::identity() + ((1.0 - std::cos(theta)) / theta_sq) * Omega + ((theta - std::sin(theta)) / (theta_sq * theta)) * (Omega * Omega);
    se3.t = V * v;
  }
  return se3;
}

// (H + lambda I)^-1 rhs via LDL^T with diagonal pivoting (the algorithm class of Eigen 3.4.0 LDLT, Eigen/src/Cholesky/LDLT.h).
inline Vec6 ldlt_solve(const Mat6& A_in, const Vec6& rhs) {
  constexpr int n = 6;
  double A[n][n];
  int perm[n];
  for (int i = 0; i < n; i++) {
    perm[i] = i;
    for (int j = 0; j < n; j++) A[i][j] = A_in(i, j);
  }
  // In-place lower LDL^T with symmetric pivoting on the largest remaining |diagonal|.
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::abs(A[k][k]);
    for (int i = k + 1; i < n; i++) {
      if (std::abs(A[i][i]) > best) {
        best = std::abs(A[i][i]);
        p = i;
      }
    }
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(A[k][j], A[p][j]);
      for (int i = 0; i < n; i++) std::swap(A[i][k], A[i][p]);
      std::swap(perm[k], perm[p]);
    }
    const double d = A[k][k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < n; i++) A[i][k] /= d;
    for (int i = k + 1; i < n; i++) {
      for (int j = k + 1; j <= i; j++) {
        A[i][j] -= A[i][k] * d * A[j][k];
        A[j][i] = A[i][j];
      }
    }
  }
  double y[n];
  for (int i = 0; i < n; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < n; i++) y[i] = (A[i][i] != 0.0) ? y[i] / A[i][i] : 0.0;
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) y[i] -= A[j][i] * y[j];
  Vec6 x;
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
  return x;
}

// ---- Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3d>::computeDirect (Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h,
//      direct_selfadjoint_eigenvalues<SolverType,3,false>) restated.  Eigenvalues ascending, eigenvectors in columns.
namespace detail {
inline void compute_roots(const Mat3& m, double roots[3]) {
  const double s_inv3 = 1.0 / 3.0;
  const double s_sqrt3 = std::sqrt(3.0);
  const double c0 = m(0, 0) * m(1, 1) * m(2, 2) + 2.0 * m(1, 0) * m(2, 0) * m(2, 1) - m(0, 0) * m(2, 1) * m(2, 1) - m(1, 1) * m(2, 0) * m(2, 0) - m(2, 2) * m(1, 0) * m(1, 0);
  const double c1 = m(0, 0) * m(1, 1) - m(1, 0) * m(1, 0) + m(0, 0) * m(2, 2) - m(2, 0) * m(2, 0) + m(1, 1) * m(2, 2) - m(2, 1) * m(2, 1);
  const double c2 = m(0, 0) + m(1, 1) + m(2, 2);
  const double c2_over_3 = c2 * s_inv3;
  double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = std::max(a_over_3, 0.0);
  const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
  double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = std::max(q, 0.0);
  const double rho = std::sqrt(a_over_3);
  const double theta = std::atan2(std::sqrt(q), half_b) * s_inv3;
  const double cos_theta = std::cos(theta);
  const double sin_theta = std::sin(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 + 2.0 * rho * cos_theta;
}
inline void extract_kernel(const Mat3& mat, Vec3& res, Vec3& representative) {
  int i0 = 0;
  double best = std::abs(mat(0, 0));
  for (int i = 1; i < 3; i++) {
    if (std::abs(mat(i, i)) > best) {
      best = std::abs(mat(i, i));
      i0 = i;
    }
  }
  representative = mat.col(i0);
  const Vec3 c0 = cross(representative, mat.col((i0 + 1) % 3));
  const Vec3 c1 = cross(representative, mat.col((i0 + 2) % 3));
  const double n0 = sqnorm(c0), n1 = sqnorm(c1);
  res = (n0 > n1) ? (1.0 / std::sqrt(n0)) * c0 : (1.0 / std::sqrt(n1)) * c1;
}
}  // namespace detail

inline void eigen_sym3_direct(const Mat3& mat_in, double eivals[3], Mat3& eivecs) {
  // Symmetrise from the lower triangle (Eigen reads the lower triangular part only).
  Mat3 mat = mat_in;
  mat(0, 1) = mat(1, 0);
  mat(0, 2) = mat(2, 0);
  mat(1, 2) = mat(2, 1);
  const double shift = (mat(0, 0) + mat(1, 1) + mat(2, 2)) / 3.0;
  Mat3 scaled = mat;
  for (int i = 0; i < 3; i++) scaled(i, i) -= shift;
  double scale = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j <= i; j++) scale = std::max(scale, std::abs(scaled(i, j)));
  if (scale > 0.0) scaled = (1.0 / scale) * scaled;
  detail::compute_roots(scaled, eivals);
  const double eps = std::numeric_limits<double>::epsilon();
  Vec3 cols[3];
  if ((eivals[2] - eivals[0]) <= eps) {
    cols[0] = {1, 0, 0};
    cols[1] = {0, 1, 0};
    cols[2] = {0, 0, 1};
  } else {
    double d0 = eivals[2] - eivals[1];
    const double d1 = eivals[1] - eivals[0];
    int k = 0, l = 2;
    if (d0 > d1) {
      std::swap(k, l);
      d0 = d1;
    }
    {
      Mat3 tmp = scaled;
      for (int i = 0; i < 3; i++) tmp(i, i) -= eivals[k];
      detail::extract_kernel(tmp, cols[k], cols[l]);
    }
    if (d0 <= 2.0 * eps * d1) {
      cols[l] = cols[l] - dot(cols[k], cols[l]) * cols[l];
      cols[l] = (1.0 / norm(cols[l])) * cols[l];
    } else {
      Mat3 tmp = scaled;
      for (int i = 0; i < 3; i++) tmp(i, i) -= eivals[l];
      Vec3 dummy;
      detail::extract_kernel(tmp, cols[l], dummy);
    }
    cols[1] = cross(cols[2], cols[0]);
    cols[1] = (1.0 / norm(cols[1])) * cols[1];
  }
  for (int i = 0; i < 3; i++) eivals[i] = eivals[i] * scale + shift;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) eivecs(r, c) = cols[c][r];
}

// Independent cross-check used by the oracle's own tests: cyclic Jacobi, eigenvalues ascending.
inline void eigen_sym3_jacobi(const Mat3& mat_in, double eivals[3], Mat3& eivecs) {
  double a[3][3];
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) a[i][j] = (i >= j) ? mat_in(i, j) : mat_in(j, i);
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++) {
      for (int q = p + 1; q < 3; q++) {
        if (a[p][q] == 0.0) continue;
        const double tau = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::abs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < 3; k++) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
    }
  }
  int order[3] = {0, 1, 2};
  std::sort(order, order + 3, [&](int i, int j) { return a[i][i] < a[j][j]; });
  for (int c = 0; c < 3; c++) {
    eivals[c] = a[order[c]][order[c]];
    for (int r = 0; r < 3; r++) eivecs(r, c) = v[r][order[c]];
  }
}

}  // namespace orc
