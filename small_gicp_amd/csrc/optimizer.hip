// Host side of Registration<>::align: Levenberg-Marquardt / Gauss-Newton over the 6x6 normal equations.
// Mirrors registration/optimizer.hpp:24-149, termination_criteria.hpp:11-20, general_factor.hpp:41-75 and
// util/lie.hpp:54-96 of the reference (/root/reference) including their control-flow quirks:
//   * LM accepts a step iff new_e <= e, then lambda /= factor, else lambda *= factor (<= 10 trials per outer iteration);
//   * result.iterations is the index of the last executed outer iteration; result.H/b are the last linearization;
//   * convergence is tested on the accepted delta only: |rot| <= rotation_eps && |trans| <= translation_eps.
// The reductions are callbacks so the same optimizer drives one GPU (sga_align) or sharded GPUs + all-reduce (sga_optimize).
#include <cmath>
#include <cstdio>

#include "common.hpp"

namespace sga {

// ---- tiny fixed-size double algebra (host) ------------------------------------------------------------------------------------
struct M4 {
  double a[16];  // column-major
};

static M4 mul(const M4& A, const M4& B) {
  M4 C;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A.a[4 * k + r] * B.a[4 * c + k];
      C.a[4 * c + r] = s;
    }
  return C;
}

// util/lie.hpp:77-96 — rotation-first twist; quaternion exp map after Sophus (lie.hpp:54-71), Eigen's toRotationMatrix.
static M4 se3_exp(const double d[6]) {
  const double wx = d[0], wy = d[1], wz = d[2];
  const double theta_sq = wx * wx + wy * wy + wz * wz;
  const double theta = std::sqrt(theta_sq);
  double imag, real;
  if (theta_sq < 1e-10) {
    const double t4 = theta_sq * theta_sq;
    imag = 0.5 - theta_sq / 48.0 + t4 / 3840.0;
    real = 1.0 - theta_sq / 8.0 + t4 / 384.0;
  } else {
    imag = std::sin(0.5 * theta) / theta;
    real = std::cos(0.5 * theta);
  }
  const double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
  double R[3][3];
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0][0] = 1 - (tyy + tzz);
    R[0][1] = txy - twz;
    R[0][2] = txz + twy;
    R[1][0] = txy + twz;
    R[1][1] = 1 - (txx + tzz);
    R[1][2] = tyz - twx;
    R[2][0] = txz - twy;
    R[2][1] = tyz + twx;
    R[2][2] = 1 - (txx + tyy);
  }
  double t[3];
  const double v[3] = {d[3], d[4], d[5]};
  if (theta < 1e-10) {
    for (int r = 0; r < 3; r++) t[r] = R[r][0] * v[0] + R[r][1] * v[1] + R[r][2] * v[2];
  } else {
    const double W[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    const double c1 = (1.0 - std::cos(theta)) / theta_sq, c2 = (theta - std::sin(theta)) / (theta_sq * theta);
    for (int r = 0; r < 3; r++) {
      t[r] = 0;
      for (int c = 0; c < 3; c++) {
        double w2 = 0;
        for (int k = 0; k < 3; k++) w2 += W[r][k] * W[k][c];
        t[r] += ((r == c ? 1.0 : 0.0) + c1 * W[r][c] + c2 * w2) * v[c];
      }
    }
  }
  M4 T;
  for (int i = 0; i < 16; i++) T.a[i] = 0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T.a[4 * c + r] = R[r][c];
    T.a[12 + r] = t[r];
  }
  T.a[15] = 1;
  return T;
}

// Solve (H + lambda I) x = -b: symmetric-indefinite-safe LDL^T with diagonal pivoting (the reference calls Eigen's ldlt()).
static void solve_damped(const double H[36], const double b[6], double lambda, double x[6]) {
  double A[6][6];
  int p[6];
  for (int i = 0; i < 6; i++) {
    p[i] = i;
    for (int j = 0; j < 6; j++) A[i][j] = H[6 * i + j] + (i == j ? lambda : 0.0);
  }
  for (int k = 0; k < 6; k++) {
    int piv = k;
    for (int i = k + 1; i < 6; i++)
      if (std::fabs(A[i][i]) > std::fabs(A[piv][piv])) piv = i;
    if (piv != k) {
      for (int j = 0; j < 6; j++) std::swap(A[k][j], A[piv][j]);
      for (int i = 0; i < 6; i++) std::swap(A[i][k], A[i][piv]);
      std::swap(p[k], p[piv]);
    }
    const double dk = A[k][k];
    if (dk == 0.0) continue;
    for (int i = k + 1; i < 6; i++) A[i][k] /= dk;
    for (int i = k + 1; i < 6; i++)
      for (int j = k + 1; j <= i; j++) {
        A[i][j] -= A[i][k] * dk * A[j][k];
        A[j][i] = A[i][j];
      }
  }
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = -b[p[i]];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < 6; i++) y[i] = A[i][i] != 0.0 ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
  for (int i = 0; i < 6; i++) x[p[i]] = y[i];
}

static bool converged(const sga_registration_setting& s, const double d[6]) {
  return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) <= s.rotation_eps && std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]) <= s.translation_eps;
}

static void apply_general_factor(const sga_registration_setting& s, double H[36]) {
  // general_factor.hpp:66: *H += lambda * (mask - 1).abs().asDiagonal()
  if (s.restrict_dof_lambda > 0)
    for (int i = 0; i < 6; i++) H[7 * i] += s.restrict_dof_lambda * std::fabs(s.restrict_dof_mask[i] - 1.0);
}

static int optimize_impl(const sga_registration_setting& s, const double init_T[16], sga_linearize_fn lin, sga_error_fn err, void* user, sga_result* out) {
  M4 T;
  for (int i = 0; i < 16; i++) T.a[i] = init_T[i];
  out->converged = 0;
  out->iterations = 0;
  out->num_inliers = 0;
  for (int i = 0; i < 36; i++) out->H[i] = 0;
  for (int i = 0; i < 6; i++) out->b[i] = 0;
  out->error = 0;
  double H[36], b[6], e = 0;
  uint64_t inliers = 0;
  if (s.optimizer == SGA_GAUSS_NEWTON) {
    if (s.verbose) std::printf("--- GN optimization ---\n");
    for (int i = 0; i < s.max_iterations && !out->converged; i++) {
      if (lin(user, T.a, H, b, &e, &inliers)) return fail(SGA_ERR_CALLBACK, "linearize callback failed");
      apply_general_factor(s, H);
      double delta[6];
      solve_damped(H, b, s.gn_lambda, delta);
      if (s.verbose)
        std::printf("iter=%d e=%g lambda=%g dt=%g dr=%g\n", i, e, s.gn_lambda, std::sqrt(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]), std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]));
      out->converged = converged(s, delta);
      T = mul(T, se3_exp(delta));
      out->iterations = i;
      memcpy(out->H, H, sizeof(H));
      memcpy(out->b, b, sizeof(b));
      out->error = e;
    }
  } else {
    if (s.verbose) std::printf("--- LM optimization ---\n");
    double lambda = s.init_lambda;
    for (int i = 0; i < s.max_iterations && !out->converged; i++) {
      if (lin(user, T.a, H, b, &e, &inliers)) return fail(SGA_ERR_CALLBACK, "linearize callback failed");
      apply_general_factor(s, H);
      bool success = false;
      for (int j = 0; j < s.max_inner_iterations; j++) {
        double delta[6];
        solve_damped(H, b, lambda, delta);
        const M4 new_T = mul(T, se3_exp(delta));
        double new_e = 0;
        if (err(user, new_T.a, &new_e)) return fail(SGA_ERR_CALLBACK, "error callback failed");
        if (s.verbose)
          std::printf(
            "iter=%d inner=%d e=%g new_e=%g lambda=%g dt=%g dr=%g\n", i, j, e, new_e, lambda, std::sqrt(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]),
            std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]));
        if (new_e <= e) {
          out->converged = converged(s, delta);
          T = new_T;
          lambda /= s.lambda_factor;
          success = true;
          e = new_e;
          break;
        }
        lambda *= s.lambda_factor;
      }
      out->iterations = i;
      memcpy(out->H, H, sizeof(H));
      memcpy(out->b, b, sizeof(b));
      out->error = e;
      if (!success) break;
    }
  }
  out->num_inliers = inliers;  // == count_if(factors, inlier()) after the last linearize (optimizer.hpp:146)
  memcpy(out->T_target_source, T.a, sizeof(T.a));
  return SGA_OK;
}

struct GpuReduction {
  sga_context* ctx;
  sga_problem* pb;
  const sga_factor_params* fp;
};

static int gpu_linearize(void* user, const double T[16], double H[36], double b[6], double* e, uint64_t* inl) {
  auto* g = static_cast<GpuReduction*>(user);
  return sga_linearize(g->ctx, g->pb, g->fp, T, H, b, e, inl);
}
static int gpu_error(void* user, const double T[16], double* e) {
  auto* g = static_cast<GpuReduction*>(user);
  return sga_error(g->ctx, g->pb, g->fp, T, e);
}

}  // namespace sga

using namespace sga;

extern "C" {

void sga_registration_setting_default(sga_registration_setting* s) {
  if (!s) return;
  sga_factor_params_default(&s->factor);
  s->optimizer = SGA_LEVENBERG_MARQUARDT;
  s->max_iterations = 20;
  s->max_inner_iterations = 10;
  s->init_lambda = 1e-3;
  s->lambda_factor = 10.0;
  s->gn_lambda = 1e-6;
  s->translation_eps = 1e-3;
  s->rotation_eps = 0.1 * M_PI / 180.0;
  s->verbose = 0;
  s->restrict_dof_lambda = 0.0;
  for (int i = 0; i < 6; i++) s->restrict_dof_mask[i] = 1.0;
}

void sga_se3_exp(const double twist[6], double T[16]) {
  const M4 m = se3_exp(twist);
  memcpy(T, m.a, sizeof(m.a));
}

int sga_optimize(const sga_registration_setting* setting, const double init_T[16], sga_linearize_fn linearize, sga_error_fn error, void* user, sga_result* out) {
  if (!setting || !init_T || !linearize || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (setting->optimizer == SGA_LEVENBERG_MARQUARDT && !error) return fail(SGA_ERR_INVALID, "LM needs an error callback");
  return optimize_impl(*setting, init_T, linearize, error, user, out);
}

int sga_align_problem(sga_context* ctx, sga_problem* problem, const double init_T[16], const sga_registration_setting* setting, sga_result* out) {
  if (!ctx || !problem || !setting || !out) return fail(SGA_ERR_INVALID, "null argument");
  static const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  // registration.hpp:34-39 warns (does not fail) on tiny clouds
  if (problem->target->n <= 10) std::fprintf(stderr, "warning: target point cloud is too small. |target|=%zu\n", problem->target->n);
  if (problem->n <= 10) std::fprintf(stderr, "warning: source point cloud is too small. |source|=%zu\n", problem->n);
  // every registration starts without search hints: its result must not depend on earlier calls on the same problem
  // (with the canonical tie rule of kd_search.hpp they could not change it anyway; what this guarantees is that no search work is
  // carried over from one registration to the next)
  if (!problem->state_fresh) {  // (a problem nothing has searched yet holds no hints: two fills less for every scan of an odometry stream)
    SGA_HIP(hipSetDevice(ctx->device));
    if (problem->n > 0) SGA_HIP(hipMemsetAsync(problem->hint.p, 0xff, problem->n * sizeof(int), ctx->stream));
    if (problem->n > 0 && problem->hint2.n >= problem->n) SGA_HIP(hipMemsetAsync(problem->hint2.p, 0xff, problem->n * sizeof(int), ctx->stream));
  }
  problem->prev_valid = false;
  GpuReduction g{ctx, problem, &setting->factor};
  return optimize_impl(*setting, init_T ? init_T : I16, gpu_linearize, gpu_error, &g, out);
}

int sga_align(sga_context* ctx, const sga_index* target, const sga_cloud* source, const double init_T[16], const sga_registration_setting* setting, sga_result* out) {
  if (!ctx || !target || !source || !setting || !out) return fail(SGA_ERR_INVALID, "null argument");
  sga_problem* pb = nullptr;
  SGA_TRY(sga_problem_create(ctx, target, source, init_T, &pb));
  const int rc = sga_align_problem(ctx, pb, init_T, setting, out);
  sga_problem_destroy(pb);
  return rc;
}

}  // extern "C"
