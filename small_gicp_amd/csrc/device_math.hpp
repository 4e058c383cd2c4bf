// Device-side math of the registration hot path (gfx950).  Real = float (default) or double (SGA_MATH_FP64).
// Formulas follow SURVEY.md Appendix A; reference citations: factors/gicp_factor.hpp:49-70, plane_icp_factor.hpp:44-54,
// icp_factor.hpp:34-52, robust_kernel.hpp:70-98 (relative to /root/reference).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace sga {

template <typename Real>
struct Sym3 {
  Real xx, xy, xz, yy, yz, zz;
};

template <typename Real>
struct Rigid {
  Real r[9];  // row-major rotation
  Real t[3];
};

// host-side: column-major double[16] -> Rigid<Real>
template <typename Real>
inline Rigid<Real> rigid_from_colmajor(const double T[16]) {
  Rigid<Real> g;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) g.r[3 * r + c] = static_cast<Real>(T[4 * c + r]);
    g.t[r] = static_cast<Real>(T[12 + r]);
  }
  return g;
}

template <typename Real>
__device__ __forceinline__ void transform_point(const Rigid<Real>& g, Real x, Real y, Real z, Real& qx, Real& qy, Real& qz) {
  qx = fma(g.r[0], x, fma(g.r[1], y, fma(g.r[2], z, g.t[0])));
  qy = fma(g.r[3], x, fma(g.r[4], y, fma(g.r[5], z, g.t[1])));
  qz = fma(g.r[6], x, fma(g.r[7], y, fma(g.r[8], z, g.t[2])));
}

// R C R^T for symmetric C.
template <typename Real>
__device__ __forceinline__ Sym3<Real> rotate_sym(const Real* R, const Sym3<Real>& C) {
  // A = R C
  const Real a00 = R[0] * C.xx + R[1] * C.xy + R[2] * C.xz, a01 = R[0] * C.xy + R[1] * C.yy + R[2] * C.yz, a02 = R[0] * C.xz + R[1] * C.yz + R[2] * C.zz;
  const Real a10 = R[3] * C.xx + R[4] * C.xy + R[5] * C.xz, a11 = R[3] * C.xy + R[4] * C.yy + R[5] * C.yz, a12 = R[3] * C.xz + R[4] * C.yz + R[5] * C.zz;
  const Real a20 = R[6] * C.xx + R[7] * C.xy + R[8] * C.xz, a21 = R[6] * C.xy + R[7] * C.yy + R[8] * C.yz, a22 = R[6] * C.xz + R[7] * C.yz + R[8] * C.zz;
  Sym3<Real> o;
  o.xx = a00 * R[0] + a01 * R[1] + a02 * R[2];
  o.xy = a00 * R[3] + a01 * R[4] + a02 * R[5];
  o.xz = a00 * R[6] + a01 * R[7] + a02 * R[8];
  o.yy = a10 * R[3] + a11 * R[4] + a12 * R[5];
  o.yz = a10 * R[6] + a11 * R[7] + a12 * R[8];
  o.zz = a20 * R[6] + a21 * R[7] + a22 * R[8];
  return o;
}

// R^T M R for symmetric M.
template <typename Real>
__device__ __forceinline__ Sym3<Real> rotate_sym_t(const Real* R, const Sym3<Real>& M) {
  // A = R^T M  (A_ij = sum_k R_ki M_kj)
  const Real a00 = R[0] * M.xx + R[3] * M.xy + R[6] * M.xz, a01 = R[0] * M.xy + R[3] * M.yy + R[6] * M.yz, a02 = R[0] * M.xz + R[3] * M.yz + R[6] * M.zz;
  const Real a10 = R[1] * M.xx + R[4] * M.xy + R[7] * M.xz, a11 = R[1] * M.xy + R[4] * M.yy + R[7] * M.yz, a12 = R[1] * M.xz + R[4] * M.yz + R[7] * M.zz;
  const Real a20 = R[2] * M.xx + R[5] * M.xy + R[8] * M.xz, a21 = R[2] * M.xy + R[5] * M.yy + R[8] * M.yz, a22 = R[2] * M.xz + R[5] * M.yz + R[8] * M.zz;
  Sym3<Real> o;  // (A R)_ij = sum_k A_ik R_kj
  o.xx = a00 * R[0] + a01 * R[3] + a02 * R[6];
  o.xy = a00 * R[1] + a01 * R[4] + a02 * R[7];
  o.xz = a00 * R[2] + a01 * R[5] + a02 * R[8];
  o.yy = a10 * R[1] + a11 * R[4] + a12 * R[7];
  o.yz = a10 * R[2] + a11 * R[5] + a12 * R[8];
  o.zz = a20 * R[2] + a21 * R[5] + a22 * R[8];
  return o;
}

// Closed-form inverse of a symmetric 3x3 (cofactors / determinant, as Eigen's Matrix3d::inverse()).
template <typename Real>
__device__ __forceinline__ Sym3<Real> inverse_sym(const Sym3<Real>& S) {
  const Real c00 = S.yy * S.zz - S.yz * S.yz;
  const Real c01 = S.xz * S.yz - S.xy * S.zz;
  const Real c02 = S.xy * S.yz - S.xz * S.yy;
  const Real c11 = S.xx * S.zz - S.xz * S.xz;
  const Real c12 = S.xy * S.xz - S.xx * S.yz;
  const Real c22 = S.xx * S.yy - S.xy * S.xy;
  const Real det = S.xx * c00 + S.xy * c01 + S.xz * c02;
  const Real inv = Real(1) / det;
  return {c00 * inv, c01 * inv, c02 * inv, c11 * inv, c12 * inv, c22 * inv};
}

// robust_kernel.hpp:24-27 (Huber), :47 (Cauchy); x = sqrt(e)
template <typename Real>
__device__ __forceinline__ Real robust_weight(int kind, Real c, Real e) {
  if (kind == SGA_ROBUST_HUBER) {
    const Real x = sqrt(e);
    return x < c ? Real(1) : c / x;
  }
  if (kind == SGA_ROBUST_CAUCHY) return c / (c + e);  // x*x = e
  return Real(1);
}

// One correspondence -> 28 values: out[0..20] = upper triangle of H (row-wise), out[21..26] = b, out[27] = e.
//   J = [R skew(ps) | -R],  H = J^T M J,  b = J^T M r,  e = 1/2 r^T M r   (M, r in the target frame)
// evaluated in the source frame: M' = R^T M R, w = R^T (M r):
//   H_tt = M', H_rt = S M', H_rr = (S M') S^T, b_r = -S w, b_t = -w   with S = skew(ps).
// Optionally hands back M' (Mp_out) and g = M' R^T r = R^T (M r) (g_out): the ingredients of the quadratic error model (linearize.hip).
template <typename Real>
__device__ __forceinline__ void pair_system(const Real* R, Real px, Real py, Real pz, Real rx, Real ry, Real rz, const Sym3<Real>& M, Real weight, Real* out, Sym3<Real>* Mp_out = nullptr, Real* g_out = nullptr) {
  const Real vx = M.xx * rx + M.xy * ry + M.xz * rz;
  const Real vy = M.xy * rx + M.yy * ry + M.yz * rz;
  const Real vz = M.xz * rx + M.yz * ry + M.zz * rz;
  const Real e = Real(0.5) * (rx * vx + ry * vy + rz * vz);
  const Real w0 = R[0] * vx + R[3] * vy + R[6] * vz;
  const Real w1 = R[1] * vx + R[4] * vy + R[7] * vz;
  const Real w2 = R[2] * vx + R[5] * vy + R[8] * vz;
  const Sym3<Real> Mp = rotate_sym_t(R, M);
  if (Mp_out != nullptr) *Mp_out = Mp;
  if (g_out != nullptr) {
    g_out[0] = w0;
    g_out[1] = w1;
    g_out[2] = w2;
  }
  // K = S M'
  const Real k00 = -pz * Mp.xy + py * Mp.xz, k01 = -pz * Mp.yy + py * Mp.yz, k02 = -pz * Mp.yz + py * Mp.zz;
  const Real k10 = pz * Mp.xx - px * Mp.xz, k11 = pz * Mp.xy - px * Mp.yz, k12 = pz * Mp.xz - px * Mp.zz;
  const Real k20 = -py * Mp.xx + px * Mp.xy, k21 = -py * Mp.xy + px * Mp.yy, k22 = -py * Mp.xz + px * Mp.yz;
  // H_rr = K S^T
  out[0] = weight * (-pz * k01 + py * k02);
  out[1] = weight * (pz * k00 - px * k02);
  out[2] = weight * (-py * k00 + px * k01);
  out[6] = weight * (pz * k10 - px * k12);
  out[7] = weight * (-py * k10 + px * k11);
  out[11] = weight * (-py * k20 + px * k21);
  // H_rt = K
  out[3] = weight * k00;
  out[4] = weight * k01;
  out[5] = weight * k02;
  out[8] = weight * k10;
  out[9] = weight * k11;
  out[10] = weight * k12;
  out[12] = weight * k20;
  out[13] = weight * k21;
  out[14] = weight * k22;
  // H_tt = M'
  out[15] = weight * Mp.xx;
  out[16] = weight * Mp.xy;
  out[17] = weight * Mp.xz;
  out[18] = weight * Mp.yy;
  out[19] = weight * Mp.yz;
  out[20] = weight * Mp.zz;
  // b_r = -S w, b_t = -w
  out[21] = weight * (pz * w1 - py * w2);
  out[22] = weight * (-pz * w0 + px * w2);
  out[23] = weight * (py * w0 - px * w1);
  out[24] = -weight * w0;
  out[25] = -weight * w1;
  out[26] = -weight * w2;
  out[27] = weight * e;
}

// ---- wave64 / block reductions ---------------------------------------------------------------------------------------------
// DPP butterfly inside each row of 16 lanes (full-rate VALU), then row_bcast15 / row_bcast31: the wave total lands in lane 63.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  // old = 0 so lanes excluded by ROW_MASK contribute 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_mov_f32<0xB1>(v);         // quad_perm [1,0,3,2]  (lane ^ 1)
  v += dpp_mov_f32<0x4E>(v);         // quad_perm [2,3,0,1]  (lane ^ 2)
  v += dpp_mov_f32<0x141>(v);        // row_half_mirror      (other quad of the 8-lane half)
  v += dpp_mov_f32<0x140>(v);        // row_mirror           (other half of the 16-lane row)
  v += dpp_mov_f32<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
  v += dpp_mov_f32<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
  return v;                          // valid in lanes 48..63
}

__device__ __forceinline__ double wave_sum_f64(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- transposing wave reduction: N sums over the 64 lanes in ~3 N instructions ----------------------------------------------------
// Summing N per-lane values one at a time costs a 6-step DPP chain each.  Here every step HALVES the number of live values instead:
// of a pair (a, b) a lane keeps the one its lane bit selects, hands the other to its partner lane and adds what the partner sends
// (one DPP add, or one v_permlane{16,32}_swap + add across rows).  After the six steps lane l holds the wave totals of the values
// number l and 64 + l.  Partners: lane ^ 1, lane ^ 2 (quad_perm), then rotations by 4 and by 8 inside the row of 16 (a perfect
// matching between the lanes with the step's bit clear and set that keeps the lower bits), then the row and half-wave swaps.
template <int CTRL, int N>
__device__ __forceinline__ void wave_halve_dpp(const float (&v)[N], float (&o)[(N + 1) / 2], bool bit) {
#pragma unroll
  for (int k = 0; k < N / 2; k++) {
    const float keep = bit ? v[2 * k + 1] : v[2 * k], send = bit ? v[2 * k] : v[2 * k + 1];
    o[k] = keep + dpp_mov_f32<CTRL>(send);
  }
  if constexpr ((N & 1) != 0) o[N / 2] = v[N - 1] + dpp_mov_f32<CTRL>(v[N - 1]);  // lanes with the bit set then hold a value that does not exist (index >= N)
}

typedef unsigned int sga_u32x2 __attribute__((ext_vector_type(2)));
template <bool HALVES>
__device__ __forceinline__ sga_u32x2 wave_swap(unsigned a, unsigned b) {
  if constexpr (HALVES)
    return __builtin_amdgcn_permlane32_swap(a, b, false, false);  // upper 32 lanes of a <-> lower 32 lanes of b
  else
    return __builtin_amdgcn_permlane16_swap(a, b, false, false);  // odd rows of a <-> even rows of b
}
// After the swap both registers hold, lane by lane, partial sums of the SAME value of the pair; which one is read off a swapped
// marker (0 = the first of the pair, 1 = the second), so nothing here depends on which rows the instruction exchanges.
template <bool HALVES, int N>
__device__ __forceinline__ void wave_halve_swap(const float (&v)[N], float (&o)[(N + 1) / 2], unsigned& which) {
  which = wave_swap<HALVES>(0u, 1u).x;
#pragma unroll
  for (int k = 0; k < N / 2; k++) {
    const sga_u32x2 r = wave_swap<HALVES>(__float_as_uint(v[2 * k]), __float_as_uint(v[2 * k + 1]));
    o[k] = __uint_as_float(r.x) + __uint_as_float(r.y);
  }
  if constexpr ((N & 1) != 0) {
    const sga_u32x2 r = wave_swap<HALVES>(__float_as_uint(v[N - 1]), 0u);
    o[N / 2] = __uint_as_float(r.x) + __uint_as_float(r.y);
  }
}

// in: N <= 128 values per lane (all 64 lanes active).  out: lane l holds the totals of value `slot` (lo) and, for N > 64, `64 + slot`
// (hi); slot is a permutation of the lane numbers.  Totals of values >= N are meaningless.
template <int N>
__device__ __forceinline__ void wave_transpose_sum(const float (&v)[N], int lane, float& lo, float& hi, int& slot) {
  static_assert(N > 32 && N <= 128, "sized for the moment sums (72, or two halves of 36)");
  float a[(N + 1) / 2];
  wave_halve_dpp<0xB1>(v, a, (lane & 1) != 0);   // quad_perm [1,0,3,2]
  float b[(N + 3) / 4];
  wave_halve_dpp<0x4E>(a, b, (lane & 2) != 0);   // quad_perm [2,3,0,1]
  float c[(N + 7) / 8];
  wave_halve_dpp<0x124>(b, c, (lane & 4) != 0);  // row_ror:4
  float d[(N + 15) / 16];
  wave_halve_dpp<0x128>(c, d, (lane & 8) != 0);  // row_ror:8
  float e[(N + 31) / 32];
  unsigned m4, m5;
  wave_halve_swap<false>(d, e, m4);
  float f[(N + 63) / 64];
  wave_halve_swap<true>(e, f, m5);
  lo = f[0];
  if constexpr (N > 64)
    hi = f[1];
  else
    hi = 0.f;
  slot = (lane & 15) | static_cast<int>(m4 << 4) | static_cast<int>(m5 << 5);
}

}  // namespace sga
