// Batched P3P hypothesis generation and reprojection scoring for gfx950.
//
// Reference behaviour (src/vo/pose_estimation/p3p.py, src/vo/algorithms/ransac.py):
//   model_fn  p3p.py:51-79    pose from 4 sampled 3D-2D correspondences: P3P on the
//                             first three, the fourth picks among <= 4 solutions
//   error_fn  p3p.py:81-108   squared reprojection error of every correspondence,
//                             formed as (sqrt(dx^2 + dy^2))^2 after
//                             x' = X' * (1/Z'), u = x' * fx + cx
//   inliers   ransac.py:104-106  error < threshold (strict), counted per hypothesis
// The reference runs these one RANSAC iteration at a time; here all pre-drawn
// samples are solved in one launch (one lane per hypothesis) and scored in a second
// (one workgroup per hypothesis, wave ballots give the inlier bit-mask rows), and
// the host replays the sequential accept/adapt rule over (valid, count).
//
// fp64 throughout, FP contraction off, only + - * / sqrt: same rounding sequence
// as the CPU oracle.
#include "vo_internal.h"

#pragma clang fp contract(off)

namespace {

struct pose_t {
  double R[9];
  double t[3];
};

__device__ __forceinline__ double cubic_eval(double A, double B, double C, double x) { return ((x + A) * x + B) * x + C; }

// value of lane k (0..3) of the caller's quad, in all four lanes
template <int K>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true);   // quad_perm [K,K,K,K]
}

// Same operations in the same order as oracle/csrc/p3p.c:cubic_positive_root (a real root in (0, U]
// of m^3 + A m^2 + B m + C for C < 0: the bracket [L, U] cut in four per round down to 2^-36, then
// one Newton step).  Called by all four lanes of a hypothesis with equal arguments: lanes 1..3
// evaluate the three cut points of a round, the signs travel by DPP.
__device__ __forceinline__ double cubic_root_bracketed(double A, double B, double C, int sub) {
  double U = 3.0 * fabs(A);
  const double sb = sqrt(3.0 * fabs(B));
  if (sb > U) U = sb;
  int e;
  (void)frexp(3.0 * fabs(C), &e);                          // 3|C| < 2^e
  const int e3 = (e >= 0) ? (e + 2) / 3 : -((-e) / 3);     // ceil(e / 3)
  const double cb = ldexp(1.0, e3);
  if (cb > U) U = cb;
  double xl = fabs(C) / ((U + fabs(A)) * U + fabs(B));
  double xh = 1.0625 * U;
  if (!(cubic_eval(A, B, C, xl) < 0.0)) xl = 0.0;          // rounding spoiled the lower bound
  for (int it = 0; it < 64; ++it) {
    if (!(xh - xl > 1.4551915228366852e-11 * xh)) break;   // 2^-36 (uniform over the quad)
    double p1, p2, p3;
    if (xl > 0.0 && xh > 16.0 * xl) {
      const double r2 = sqrt(xh / xl);
      const double r1 = sqrt(r2);
      p1 = xl * r1;
      p2 = xl * r2;
      p3 = p2 * r1;
    } else {
      const double w = 0.25 * (xh - xl);
      p1 = xl + w;
      p2 = xl + 2.0 * w;
      p3 = xl + 3.0 * w;
    }
    const double pm = sub == 1 ? p1 : (sub == 2 ? p2 : p3);   // (lane 0 duplicates lane 3)
    const int sm = cubic_eval(A, B, C, pm) >= 0.0 ? 1 : 0;
    const int s1 = quad_bcast<1>(sm), s2 = quad_bcast<2>(sm), s3 = quad_bcast<3>(sm);
    if (s1) xh = p1;
    else if (s2) { xl = p1; xh = p2; }
    else if (s3) { xl = p2; xh = p3; }
    else xl = p3;
  }
  double m = 0.5 * (xl + xh);
  {
    const double f = cubic_eval(A, B, C, m);
    const double df = (3.0 * m + 2.0 * A) * m + B;
    if (df != 0.0) m = m - f / df;
  }
  return m;
}

__device__ __forceinline__ int quad_roots(double b, double c, double* r) {
  const double disc = b * b - 4.0 * c;
  if (disc < 0.0) return 0;
  const double sq = sqrt(disc);
  const double q = (b >= 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
  r[0] = q;
  r[1] = (q != 0.0) ? c / q : 0.0;
  return 2;
}

// quad-cooperative: lane `sub` polishes (and returns in roots[sub]) only the root it will use
__device__ int quartic_roots(double c0, double c1, double c2, double c3, double c4, double* roots, int sub) {
  if (c4 == 0.0) return 0;
  const double a3 = c3 / c4, a2 = c2 / c4, a1 = c1 / c4, a0 = c0 / c4;
  const double a3sq = a3 * a3;
  const double p = a2 - 0.375 * a3sq;
  const double q = a1 - 0.5 * a2 * a3 + 0.125 * a3sq * a3;
  const double r = a0 - 0.25 * a1 * a3 + 0.0625 * a2 * a3sq - (3.0 / 256.0) * a3sq * a3sq;
  // the (up to four) roots of the depressed quartic in the order the oracle lists them; scalars, not an
  // array filled through a running index (that array lived in scratch memory: a memory round trip in the
  // middle of the dependent chain, and a private segment for every wave of the launch)
  double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
  int n = 0;
  if (q == 0.0) {
    double z[2];
    if (quad_roots(p, r, z) != 0) {
      const bool ua = z[0] >= 0.0, ub = z[1] >= 0.0;
      const double sa = ua ? sqrt(z[0]) : 0.0, sb = ub ? sqrt(z[1]) : 0.0;
      if (ua) {
        y0 = sa;
        y1 = -sa;
        n = 2;
        if (ub) {
          y2 = sb;
          y3 = -sb;
          n = 4;
        }
      } else if (ub) {
        y0 = sb;
        y1 = -sb;
        n = 2;
      }
    }
  } else {
    const double m = cubic_root_bracketed(p, 0.25 * p * p - r, -0.125 * q * q, sub);
    if (!(m > 0.0)) return 0;
    const double s = sqrt(2.0 * m);
    const double h = 0.5 * p + m;
    const double g = q / (2.0 * s);
    double za[2], zb[2];
    const int na = quad_roots(s, h - g, za), nb = quad_roots(-s, h + g, zb);
    if (na != 0) {
      y0 = za[0];
      y1 = za[1];
      if (nb != 0) {
        y2 = zb[0];
        y3 = zb[1];
      }
    } else if (nb != 0) {
      y0 = zb[0];
      y1 = zb[1];
    }
    n = na + nb;
  }
  const double shift = 0.25 * a3;
  {
    double ys = y0;
    if (sub == 1) ys = y1;
    if (sub == 2) ys = y2;
    if (sub == 3) ys = y3;
    double x = ys - shift;
    if (sub < n) {
      for (int it = 0; it < 3; ++it) {
        const double f = (((x + a3) * x + a2) * x + a1) * x + a0;
        const double df = ((4.0 * x + 3.0 * a3) * x + 2.0 * a2) * x + a1;
        if (df == 0.0) break;
        x = x - f / df;
      }
    }
    roots[0] = roots[1] = roots[2] = roots[3] = x;   // (each lane reads its own entry)
  }
  return n;
}

// orthonormal frame of a point triple; E row-major with the frame vectors as columns
__device__ __forceinline__ bool frame_of(const double* p1, const double* p2, const double* p3, double* E) {
  const double ax = p2[0] - p1[0], ay = p2[1] - p1[1], az = p2[2] - p1[2];
  const double bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
  const double na = sqrt(ax * ax + ay * ay + az * az);
  if (!(na > 0.0)) return false;
  const double e1x = ax / na, e1y = ay / na, e1z = az / na;
  double e3x = e1y * bz - e1z * by;
  double e3y = e1z * bx - e1x * bz;
  double e3z = e1x * by - e1y * bx;
  const double n3 = sqrt(e3x * e3x + e3y * e3y + e3z * e3z);
  if (!(n3 > 0.0)) return false;
  e3x /= n3;
  e3y /= n3;
  e3z /= n3;
  const double e2x = e3y * e1z - e3z * e1y;
  const double e2y = e3z * e1x - e3x * e1z;
  const double e2z = e3x * e1y - e3y * e1x;
  E[0] = e1x; E[1] = e2x; E[2] = e3x;
  E[3] = e1y; E[4] = e2y; E[5] = e3y;
  E[6] = e1z; E[7] = e2z; E[8] = e3z;
  return true;
}

__device__ __forceinline__ double reproj_sq(const double* R, const double* t, double fx, double fy, double cx,
                                            double cy, double X, double Y, double Z, double u0, double v0) {
  const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
  const double yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
  const double zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
  const double iz = (zc != 0.0) ? 1.0 / zc : 1.0;
  const double xn = xc * iz, yn = yc * iz;
  const double u = xn * fx + cx;
  const double v = yn * fy + cy;
  const double dx = u0 - u, dy = v0 - v;
  const double nrm = sqrt(dx * dx + dy * dy);
  return nrm * nrm;
}

// The same up to the sum of squares s (the error is fl(fl(sqrt(s))^2)).  fl(fl(sqrt(.))^2) is monotone in s
// (sqrt is correctly rounded on both sides, rounding is monotone), so  error < thr  <=>  s <= sum_sq_limit(thr):
// the scoring loops compare s and skip the square root and the square of every evaluation, same decisions.
__device__ __forceinline__ double reproj_sum_sq(const double* R, const double* t, double fx, double fy, double cx,
                                                double cy, double X, double Y, double Z, double u0, double v0) {
  const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
  const double yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
  const double zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
  const double iz = (zc != 0.0) ? 1.0 / zc : 1.0;
  const double xn = xc * iz, yn = yc * iz;
  const double u = xn * fx + cx;
  const double v = yn * fy + cy;
  const double dx = u0 - u, dy = v0 - v;
  return dx * dx + dy * dy;
}

// largest s >= 0 with fl(fl(sqrt(s))^2) < thr (bisection over the bit patterns of the non-negative doubles, which
// order like the numbers); -1 when there is none.  The host's sqrt and product are the device's: both IEEE.
static double sum_sq_limit(double thr) {
  auto err_of = [](double s_) {
    volatile double r = std::sqrt(s_);
    volatile double e = r * r;
    return (double)e;
  };
  if (!(thr > 0.0) || !(err_of(0.0) < thr)) return -1.0;
  unsigned long long lo = 0ull, hi = 0x7ff0000000000000ull;   // err_of(lo) < thr, err_of(hi = inf) >= thr
  while (hi - lo > 1ull) {
    const unsigned long long mid = lo + (hi - lo) / 2;
    double m;
    memcpy(&m, &mid, 8);
    if (err_of(m) < thr) lo = mid;
    else hi = mid;
  }
  double r;
  memcpy(&r, &lo, 8);
  return r;
}

// NumPy's bounded draw on one 32-bit output of the generator (Lemire, ransac_host.hip):
// `risky` is raised when the draw could have been rejected, i.e. when the sequential
// generator might have consumed one more output than this position-based view assumes.
__device__ __forceinline__ unsigned bounded_from_raw(unsigned raw, unsigned rng, bool& risky) {
  const unsigned rex = rng + 1u;
  const unsigned long long m = (unsigned long long)raw * rex;
  if ((unsigned)m < rex) risky = true;
  return (unsigned)(m >> 32);
}

// RAW == false: sample indices are given.
// RAW == true : hypothesis h derives its sample from generator outputs raws[7h .. 7h+6] and the
//   population size *d_n, both of which may still be in flight when the kernel is enqueued:
//   Generator.choice(n, 4, replace=False) is four bounded draws (Floyd) and three for the
//   shuffle, so without rejections sample h sits at a fixed offset of the stream.  Any possible
//   rejection (or n < 8, where the draw count changes) sets *flag and the host redoes the batch
//   with the sequential sampler.
template <bool RAW>
__device__ __forceinline__ void p3p_solve_quad(int gt, const double* __restrict__ Xw, const double* __restrict__ xi,
                                               const int* __restrict__ samples, const unsigned* __restrict__ raws,
                                               const unsigned long long* __restrict__ d_rawpos, unsigned raw_mask,
                                               const int* __restrict__ d_n, unsigned* __restrict__ flag, int Hyp,
                                               double fx, double fy, double cx, double cy, double* __restrict__ Rout,
                                               double* __restrict__ tout, uint8_t* __restrict__ valid) {
  // four lanes (a DPP quad) per hypothesis: the set-up and the quartic are computed by all four,
  // then lane `sub` takes root `sub` through the triad alignment and the fourth-point test
  const int h = gt >> 2, sub = gt & 3;
  // (device-resident frame state: the position in the generator's output stream is a word the previous
  //  step's replay kernel left in HBM, the outputs live in a power-of-two ring)
  const unsigned ring_pos = (RAW && d_rawpos) ? (unsigned)(*d_rawpos) : 0u;
  if (h >= Hyp) return;                 // whole quads leave together
  int sidx[4];
  bool risky = false;                   // a draw of this sample could have been rejected: valid[h] bit 1
  if (RAW) {
    const int n = *d_n;
    if (n < 8) {
      if (gt == 0) atomicOr(flag, 1u);
      if (sub == 0) {
        for (int k = 0; k < 9; ++k) Rout[9 * h + k] = 0.0;
        for (int k = 0; k < 3; ++k) tout[3 * h + k] = 0.0;
        valid[h] = 0;
      }
      return;
    }
    unsigned rw[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) rw[k] = raws[(ring_pos + 7u * (unsigned)h + (unsigned)k) & raw_mask];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned j = (unsigned)(n - 4 + k);
      const int val = (int)bounded_from_raw(rw[k], j, risky);
      bool seen = false;
#pragma unroll
      for (int q = 0; q < k; ++q) seen |= (sidx[q] == val);
      sidx[k] = seen ? (int)j : val;
    }
#pragma unroll
    for (int i = 3; i >= 1; --i) {
      const int j = (int)bounded_from_raw(rw[4 + (3 - i)], (unsigned)i, risky);
      int vj = sidx[0];
#pragma unroll
      for (int q = 1; q <= i; ++q) vj = (j == q) ? sidx[q] : vj;
      const int vi = sidx[i];
#pragma unroll
      for (int q = 0; q <= i; ++q)
        if (q == j) sidx[q] = vi;
      sidx[i] = vj;
    }
    if (risky && sub == 0 && !d_rawpos) atomicOr(flag, 1u);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) sidx[k] = samples[4 * h + k];
  }
  double P[4][3], px[4][2];
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = sidx[k];
      P[k][0] = Xw[3 * idx];
      P[k][1] = Xw[3 * idx + 1];
      P[k][2] = Xw[3 * idx + 2];
      px[k][0] = xi[2 * idx];
      px[k][1] = xi[2 * idx + 1];
    }
  }
  double f[3][3];
  for (int i = 0; i < 3; ++i) {
    const double mu = (px[i][0] - cx) / fx, mv = (px[i][1] - cy) / fy;
    const double nrm = sqrt(mu * mu + mv * mv + 1.0);
    f[i][0] = mu / nrm;
    f[i][1] = mv / nrm;
    f[i][2] = 1.0 / nrm;
  }
  double d12s = 0, d13s = 0, d23s = 0;
  for (int k = 0; k < 3; ++k) {
    const double a = P[0][k] - P[1][k], b = P[0][k] - P[2][k], c = P[1][k] - P[2][k];
    d12s += a * a;
    d13s += b * b;
    d23s += c * c;
  }
  bool found = false;
  double bestR[9], bestt[3], best = 0.0;
  double Ew[9];
  if (d12s > 0.0 && d13s > 0.0 && d23s > 0.0 && frame_of(P[0], P[1], P[2], Ew)) {
    const double c12 = f[0][0] * f[1][0] + f[0][1] * f[1][1] + f[0][2] * f[1][2];
    const double c13 = f[0][0] * f[2][0] + f[0][1] * f[2][1] + f[0][2] * f[2][2];
    const double c23 = f[1][0] * f[2][0] + f[1][1] * f[2][1] + f[1][2] * f[2][2];
    // depth ratios u = s2/s1, v = s3/s1:  u = Nn(v)/Dd(v), quartic in v (see oracle/csrc/p3p.c header)
    const double a = d12s / d13s, b = d23s / d13s, g = a - b;
    const double n2 = 1.0 + g, n1 = -2.0 * g * c13, n0 = g - 1.0;
    const double e1 = 2.0 * c23, e0 = -2.0 * c12;
    const double w2 = -a, w1 = 2.0 * a * c13, w0 = 1.0 - a;
    const double dd2 = e1 * e1, dd1 = 2.0 * e1 * e0, dd0 = e0 * e0;
    const double nd3 = n2 * e1, nd2 = n2 * e0 + n1 * e1, nd1 = n1 * e0 + n0 * e1, nd0 = n0 * e0;
    const double tc = 2.0 * c12;
    const double q4 = n2 * n2 + w2 * dd2;
    const double q3 = 2.0 * n2 * n1 - tc * nd3 + (w2 * dd1 + w1 * dd2);
    const double q2 = (2.0 * n2 * n0 + n1 * n1) - tc * nd2 + (w2 * dd0 + w1 * dd1 + w0 * dd2);
    const double q1 = 2.0 * n1 * n0 - tc * nd1 + (w1 * dd0 + w0 * dd1);
    const double q0 = n0 * n0 - tc * nd0 + w0 * dd0;
    double roots[4];
    const int nr = quartic_roots(q0, q1, q2, q3, q4, roots, sub);
    do {
      if (sub >= nr) break;
      double v = roots[0];
      if (sub == 1) v = roots[1];
      if (sub == 2) v = roots[2];
      if (sub == 3) v = roots[3];
      if (!(v > 0.0)) break;
      const double Dd = e1 * v + e0;
      if (fabs(Dd) < 1e-12) break;
      const double Nn = (n2 * v + n1) * v + n0;
      const double u = Nn / Dd;
      if (!(u > 0.0)) break;
      const double qv = (v - 2.0 * c13) * v + 1.0;
      if (!(qv > 0.0)) break;
      const double s1 = sqrt(d13s / qv);
      const double s2 = u * s1, s3 = v * s1;
      const double C1[3] = {s1 * f[0][0], s1 * f[0][1], s1 * f[0][2]};
      const double C2[3] = {s2 * f[1][0], s2 * f[1][1], s2 * f[1][2]};
      const double C3[3] = {s3 * f[2][0], s3 * f[2][1], s3 * f[2][2]};
      double Ec[9];
      if (!frame_of(C1, C2, C3, Ec)) break;
      double R[9], t[3];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          R[3 * r + c] = Ec[3 * r + 0] * Ew[3 * c + 0] + Ec[3 * r + 1] * Ew[3 * c + 1] + Ec[3 * r + 2] * Ew[3 * c + 2];
      for (int r = 0; r < 3; ++r)
        t[r] = C1[r] - (R[3 * r] * P[0][0] + R[3 * r + 1] * P[0][1] + R[3 * r + 2] * P[0][2]);
      const double e = reproj_sq(R, t, fx, fy, cx, cy, P[3][0], P[3][1], P[3][2], px[3][0], px[3][1]);
      if (!(e == e)) break;
      found = true;
      best = e;
      for (int k = 0; k < 9; ++k) bestR[k] = R[k];
      for (int k = 0; k < 3; ++k) bestt[k] = t[k];
    } while (0);
  }
  // the sequential rule keeps the first root with the smallest fourth-point error: the minimum
  // of (error, root index) over the quad's candidates; lanes without a candidate sort last
  {
    double ke = found ? best : __longlong_as_double(0x7ff0000000000000ll);
    int ki = found ? sub : sub + 4;
#pragma unroll
    for (int step = 0; step < 2; ++step) {
      const int lo = __double2loint(ke), hi = __double2hiint(ke);
      int olo, ohi, oi;
      if (step == 0) {
        olo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        ohi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
        oi = __builtin_amdgcn_update_dpp(0, ki, 0xB1, 0xF, 0xF, true);
      } else {
        olo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        ohi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
        oi = __builtin_amdgcn_update_dpp(0, ki, 0x4E, 0xF, 0xF, true);
      }
      const double oe = __hiloint2double(ohi, olo);
      if (oe < ke || (oe == ke && oi < ki)) {
        ke = oe;
        ki = oi;
      }
    }
    if (ki >= 4) {                       // no root gave a pose
      if (sub == 0) {
        for (int k = 0; k < 9; ++k) Rout[9 * h + k] = 0.0;
        for (int k = 0; k < 3; ++k) tout[3 * h + k] = 0.0;
        valid[h] = risky ? 2 : 0;
      }
    } else if (ki == sub) {
      for (int k = 0; k < 9; ++k) Rout[9 * h + k] = bestR[k];
      for (int k = 0; k < 3; ++k) tout[3 * h + k] = bestt[k];
      valid[h] = risky ? 3 : 1;
    }
  }
}

template <bool RAW>
__global__ __launch_bounds__(64) void p3p_solve_kernel(const double* __restrict__ Xw, const double* __restrict__ xi,
                                                       const int* __restrict__ samples,
                                                       const unsigned* __restrict__ raws,
                                                       const unsigned long long* __restrict__ d_rawpos, unsigned raw_mask,
                                                       const int* __restrict__ d_n, unsigned* __restrict__ flag, int Hyp,
                                                       double fx, double fy, double cx, double cy,
                                                       double* __restrict__ Rout, double* __restrict__ tout,
                                                       uint8_t* __restrict__ valid) {
  p3p_solve_quad<RAW>(blockIdx.x * blockDim.x + threadIdx.x, Xw, xi, samples, raws, d_rawpos, raw_mask, d_n, flag, Hyp, fx,
                      fy, cx, cy, Rout, tout, valid);
}

// Frame loop: hypotheses AND their inlier counts in one launch.  A workgroup of 256 owns HG hypotheses:
//   wave 0 solves them (4 lanes each, p3p_solve_quad) while waves 1-3 already fetch the population;
//   then every thread keeps up to HP correspondences in registers and scores all HG poses against them --
//   the population is read from memory once per workgroup (Hyp / HG times in all, not Hyp times: the one-workgroup-
//   per-hypothesis kernel streams ~70 MB through L2 at cfg-2 and slows down 3x when the tracker and the detector
//   run beside it), inlier bits leave as one ballot word per wave, counts as popcounts.
// HG hypotheses per workgroup: 4 for one sequence (250 workgroups at 1000 hypotheses, one per compute unit; 8 per workgroup left
// 131 of the 256 units idle: 22.4 -> 20.0 us, the rest is the solve's own dependent chain -- 2 per workgroup: 19.8), 8 for a few
// sequences, 16 when a launch holds
// several sequences (wave 0's 64 lanes all solve, half as many workgroups share the solve's latency: throughput).
// Correspondences per thread and tile: 7 for HG = 8 (256 * 7 = 1792 = 28 mask words: one tile at every population the
// frame loop sees), 4 for HG = 16, which with five waves per SIMD asked for (96 registers: the solve spills 11 words,
// the scoring loop none) puts the 72 x 16 workgroups of a 16-sequence launch on the chip in one round instead of two.
template <int HG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HG == 16 ? 5 : 4, HG == 16 ? 5 : 4))) void
p3p_hyp_kernel(const double* __restrict__ Xw, const double* __restrict__ xi,
                                                      const unsigned* __restrict__ raws,
                                                      const unsigned long long* __restrict__ d_rawpos, unsigned raw_mask,
                                                      const int* __restrict__ d_n, unsigned* __restrict__ flag, int Hyp,
                                                      double fx, double fy, double cx, double cy, double lim,
                                                      double* __restrict__ Rout, double* __restrict__ tout,
                                                      uint8_t* __restrict__ valid, int* __restrict__ counts,
                                                      unsigned long long* __restrict__ masks, int words,
                                                      unsigned long long* __restrict__ ts_out, vo_hyp_batch B) {
  constexpr int HP = HG == 16 ? 4 : 7;
  __shared__ int s_cnt[4][HG];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (blockIdx.y != 0) {               // several sequences per launch: grid.y = sequence
    const size_t q = blockIdx.y;
    Xw += q * B.X;
    xi += q * B.x;
    raws += q * B.raws;
    d_rawpos = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(d_rawpos) + q * B.ctl);
    d_n = reinterpret_cast<const int*>(reinterpret_cast<const char*>(d_n) + q * B.ctl);
    flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(flag) + q * B.ctl);
    if (ts_out) ts_out = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(ts_out) + q * B.ctl);
    Rout += q * (size_t)Hyp * 9;
    tout += q * (size_t)Hyp * 3;
    valid += q * (size_t)Hyp;
    counts += q * (size_t)Hyp;
    if (masks) masks += q * (size_t)Hyp * words;
  }
  if (ts_out && blockIdx.x == 0 && tid == 0) *ts_out = wall_clock64();
  const int N = *d_n;
  const int h0 = blockIdx.x * HG;
  double pX[HP], pY[HP], pZ[HP], pu[HP], pv[HP];
  auto load_tile = [&](int tile) {
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const int i = min(tile * (256 * HP) + k * 256 + tid, max(N - 1, 0));
      pX[k] = Xw[3 * i];
      pY[k] = Xw[3 * i + 1];
      pZ[k] = Xw[3 * i + 2];
      pu[k] = xi[2 * i];
      pv[k] = xi[2 * i + 1];
    }
  };
  // the solving wave changes with the workgroup: a workgroup's wave k runs on SIMD k, and with every solver on
  // SIMD 0 the solves of the workgroups sharing a CU would queue there while the other three SIMDs wait
  if (wv == ((blockIdx.x + blockIdx.y) & 3)) {
    if (lane < 4 * HG)
      p3p_solve_quad<true>(h0 * 4 + lane, Xw, xi, nullptr, raws, d_rawpos, raw_mask, d_n, flag, Hyp, fx, fy, cx, cy, Rout,
                           tout, valid);
    if (N > 0) load_tile(0);
  } else if (N > 0) {
    load_tile(0);
  }
  __syncthreads();            // (the poses wave 0 wrote are visible to the workgroup)
  int cnt[HG];
#pragma unroll
  for (int g = 0; g < HG; ++g) cnt[g] = 0;
  const int words_n = (N + 63) >> 6;
  const int tiles = (N + 256 * HP - 1) / (256 * HP);
  for (int tile = 0; tile < tiles; ++tile) {
    if (tile > 0) load_tile(tile);
#pragma unroll
    for (int g = 0; g < HG; ++g) {
      const int h = h0 + g;
      if (h >= Hyp) break;
      const bool ok = (valid[h] & 1) != 0;
      double R[9], t[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) R[k] = Rout[9 * h + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) t[k] = tout[3 * h + k];
#pragma unroll
      for (int k = 0; k < HP; ++k) {
        const int i = tile * (256 * HP) + k * 256 + tid;
        bool in = false;
        if (ok && i < N) in = reproj_sum_sq(R, t, fx, fy, cx, cy, pX[k], pY[k], pZ[k], pu[k], pv[k]) <= lim;
        const unsigned long long m = __ballot(in);
        const int w = __builtin_amdgcn_readfirstlane(i >> 6);   // (uniform in the wave: the counts stay in scalar registers)
        if (w < words_n) {
          if (lane == 0 && masks) masks[(size_t)h * words + w] = m;
          cnt[g] += __popcll(m);
        }
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < HG; ++g) s_cnt[wv][g] = cnt[g];
  }
  __syncthreads();
  if (tid < HG && h0 + tid < Hyp) counts[h0 + tid] = s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
}

constexpr int SC_T = 256;

// one workgroup per hypothesis; mask row h holds ceil(N/64) 64-bit words
__global__ __launch_bounds__(SC_T) void p3p_score_kernel(const double* __restrict__ Xw, const double* __restrict__ xi,
                                                         int N_arg, const int* __restrict__ d_n,
                                                         const double* __restrict__ Rall,
                                                         const double* __restrict__ tall,
                                                         const uint8_t* __restrict__ valid, double fx, double fy,
                                                         double cx, double cy, double thr, int* __restrict__ counts,
                                                         unsigned long long* __restrict__ masks, int words) {
  const int h = blockIdx.x;
  const int tid = threadIdx.x;
  const int N = d_n ? *d_n : N_arg;         // population size known only on the device (pipeline) or given
  const int words_n = (N + 63) >> 6;        // words in use; `words` is the row stride
  __shared__ int s_cnt[SC_T / 64];
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rall[9 * h + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = tall[3 * h + k];
  const bool ok = (valid[h] & 1) != 0;   // (bit 1: see p3p_solve_kernel)
  int cnt = 0;
  for (int base = 0; base < words_n * 64; base += SC_T) {
    const int i = base + tid;
    bool in = false;
    if (ok && i < N) {
      const double e = reproj_sq(R, t, fx, fy, cx, cy, Xw[3 * i], Xw[3 * i + 1], Xw[3 * i + 2], xi[2 * i], xi[2 * i + 1]);
      in = e < thr;
    }
    const unsigned long long m = __ballot(in);
    const int w = i >> 6;
    if ((tid & 63) == 0 && w < words_n) {
      if (masks) masks[(size_t)h * words + w] = m;
      cnt += __popcll(m);
    }
  }
  if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) {
    int s = 0;
    for (int k = 0; k < SC_T / 64; ++k) s += s_cnt[k];
    counts[h] = s;
  }
}

__global__ __launch_bounds__(256) void reproj_kernel(const double* __restrict__ Xw, const double* __restrict__ xi, int N,
                                                     const double* __restrict__ Rt, double fx, double fy, double cx,
                                                     double cy, double thr, uint8_t* __restrict__ mask,
                                                     double* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = Rt[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = Rt[9 + k];
  const double e = reproj_sq(R, t, fx, fy, cx, cy, Xw[3 * i], Xw[3 * i + 1], Xw[3 * i + 2], xi[2 * i], xi[2 * i + 1]);
  if (mask) mask[i] = e < thr ? 1 : 0;
  if (err) err[i] = e;
}

}  // namespace

// Frame-loop form of vo_p3p_hypotheses_dev: the population size, the position in the generator's output stream and
// the outputs themselves (a power-of-two ring, raw_mask = length - 1) are read on the device, so the launch can be
// enqueued before any of them exists; a draw that NumPy might have rejected marks its hypothesis (valid[h] bit 1),
// a population below 8 raises *d_flag.  One launch: hypotheses and inlier counts (p3p_hyp_kernel).
int vo_p3p_hypotheses_ring_dev(vo_ctx* ctx, const double* d_X, const double* d_x, const int32_t* d_n, int n_cap,
                               const double* K, const uint32_t* d_raws, const uint64_t* d_rawpos, uint32_t raw_mask,
                               int Hyp, double thr_sq, double* d_R, double* d_t, uint8_t* d_valid, int32_t* d_counts,
                               uint64_t* d_masks, uint32_t* d_flag, uint64_t* d_ts, const vo_hyp_batch* batch) {
  if (!ctx) return VO_EINVAL;
  const vo_hyp_batch B = batch ? *batch : vo_hyp_batch();
  const int S = B.S > 0 ? B.S : 1;
  VO_REQUIRE(ctx, d_X && d_x && d_n && K && d_raws && d_rawpos && d_R && d_t && d_valid && d_counts && d_flag,
             "p3p_hypotheses_ring: null pointer");
  VO_REQUIRE(ctx, n_cap >= 4 && Hyp >= 1, "p3p_hypotheses_ring: need n_cap >= 4 and Hyp >= 1");
  VO_REQUIRE(ctx, K[0] != 0.0 && K[4] != 0.0, "p3p_hypotheses_ring: singular intrinsics");
  const double lim = sum_sq_limit(thr_sq);   // (lim: see reproj_sum_sq)
  {
    vo_prof_scope ps(ctx, VO_K_P3P_SOLVE);
    static const int forced = getenv("VO_HYP_GROUP") ? atoi(getenv("VO_HYP_GROUP")) : 0;   // measurements: 2 / 4 / 8 / 16
    if (forced == 16 || (forced != 8 && S >= 8))
      hipLaunchKernelGGL(p3p_hyp_kernel<16>, dim3(vo_cdiv(Hyp, 16), S), dim3(256), 0, ctx->stream, d_X, d_x, d_raws,
                         (const unsigned long long*)d_rawpos, raw_mask, d_n, d_flag, Hyp, K[0], K[4], K[2], K[5], lim, d_R,
                         d_t, d_valid, d_counts, (unsigned long long*)d_masks, vo_cdiv(n_cap, 64), (unsigned long long*)d_ts,
                         B);
    else if (forced == 4 || (forced == 0 && S == 1))
      hipLaunchKernelGGL(p3p_hyp_kernel<4>, dim3(vo_cdiv(Hyp, 4), S), dim3(256), 0, ctx->stream, d_X, d_x, d_raws,
                         (const unsigned long long*)d_rawpos, raw_mask, d_n, d_flag, Hyp, K[0], K[4], K[2], K[5], lim, d_R,
                         d_t, d_valid, d_counts, (unsigned long long*)d_masks, vo_cdiv(n_cap, 64), (unsigned long long*)d_ts,
                         B);
    else if (forced == 2)
      hipLaunchKernelGGL(p3p_hyp_kernel<2>, dim3(vo_cdiv(Hyp, 2), S), dim3(256), 0, ctx->stream, d_X, d_x, d_raws,
                         (const unsigned long long*)d_rawpos, raw_mask, d_n, d_flag, Hyp, K[0], K[4], K[2], K[5], lim, d_R,
                         d_t, d_valid, d_counts, (unsigned long long*)d_masks, vo_cdiv(n_cap, 64), (unsigned long long*)d_ts,
                         B);
    else
      hipLaunchKernelGGL(p3p_hyp_kernel<8>, dim3(vo_cdiv(Hyp, 8), S), dim3(256), 0, ctx->stream, d_X, d_x, d_raws,
                         (const unsigned long long*)d_rawpos, raw_mask, d_n, d_flag, Hyp, K[0], K[4], K[2], K[5], lim, d_R,
                         d_t, d_valid, d_counts, (unsigned long long*)d_masks, vo_cdiv(n_cap, 64), (unsigned long long*)d_ts,
                         B);
  }
  return vo_check_launch(ctx, "p3p_hyp_kernel");
}

extern "C" {

double vo_inlier_sum_sq_limit(double thr_sq) { return sum_sq_limit(thr_sq); }

int vo_p3p_hypotheses_dev(vo_ctx* ctx, const double* d_X, const double* d_x, int N, const double* K,
                          const int32_t* d_samples, int Hyp, double thr_sq, double* d_R, double* d_t,
                          uint8_t* d_valid, int32_t* d_counts, uint64_t* d_masks) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_X && d_x && K && d_samples && d_R && d_t && d_valid && d_counts, "p3p_hypotheses: null pointer");
  VO_REQUIRE(ctx, N >= 4 && Hyp >= 1, "p3p_hypotheses: need N >= 4 and Hyp >= 1");
  VO_REQUIRE(ctx, K[0] != 0.0 && K[4] != 0.0, "p3p_hypotheses: singular intrinsics");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  {
    vo_prof_scope ps(ctx, VO_K_P3P_SOLVE);
    hipLaunchKernelGGL(p3p_solve_kernel<false>, dim3(vo_cdiv(Hyp, 16)), dim3(64), 0, ctx->stream, d_X, d_x,
                       d_samples, (const unsigned*)nullptr, (const unsigned long long*)nullptr, 0xffffffffu,
                       (const int*)nullptr, (unsigned*)nullptr, Hyp, fx, fy, cx, cy, d_R, d_t, d_valid);
  }
  VO_TRY(vo_check_launch(ctx, "p3p_solve_kernel"));
  const int words = vo_cdiv(N, 64);
  {
    vo_prof_scope ps(ctx, VO_K_P3P_SCORE);
    hipLaunchKernelGGL(p3p_score_kernel, dim3(Hyp), dim3(SC_T), 0, ctx->stream, d_X, d_x, N, (const int*)nullptr, d_R,
                       d_t, d_valid, fx, fy, cx, cy, thr_sq, d_counts, (unsigned long long*)d_masks, words);
  }
  return vo_check_launch(ctx, "p3p_score_kernel");
}

int vo_reproj_inliers_dev(vo_ctx* ctx, const double* d_X, const double* d_x, int N, const double* K,
                          const double* d_Rt /* 9 + 3 */, double thr_sq, uint8_t* d_mask, double* d_err) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, d_X && d_x && K && d_Rt && (d_mask || d_err), "reproj_inliers: null pointer");
  VO_REQUIRE(ctx, N >= 0, "reproj_inliers: bad N");
  if (N == 0) return VO_OK;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    vo_prof_scope ps(ctx, VO_K_REPROJ);
    hipLaunchKernelGGL(reproj_kernel, dim3(vo_cdiv(N, 256)), dim3(256), 0, ctx->stream, d_X, d_x, N, d_Rt, K[0], K[4],
                       K[2], K[5], thr_sq, d_mask, d_err);
  }
  return vo_check_launch(ctx, "reproj_kernel");
}

// ---- host-buffer wrappers ------------------------------------------------------------

int vo_p3p_hypotheses(vo_ctx* ctx, const double* X, const double* x, int N, const double* K, const int32_t* samples,
                      int Hyp, double thr_sq, double* R, double* t, uint8_t* valid, int32_t* counts,
                      uint64_t* masks /* nullable, Hyp*ceil(N/64) */) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, X && x && K && samples && R && t && valid && counts, "p3p_hypotheses: null pointer");
  VO_REQUIRE(ctx, N >= 4 && Hyp >= 1, "p3p_hypotheses: need N >= 4 and Hyp >= 1");
  for (int i = 0; i < 4 * Hyp; ++i)
    VO_REQUIRE(ctx, samples[i] >= 0 && samples[i] < N, "p3p_hypotheses: sample index %d out of range", samples[i]);
  const int words = vo_cdiv(N, 64);
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)N * 24));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[2], (size_t)Hyp * 16));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)Hyp * 72));
  VO_TRY(vo_ensure(ctx, s[4], (size_t)Hyp * 24));
  VO_TRY(vo_ensure(ctx, s[5], (size_t)Hyp));
  VO_TRY(vo_ensure(ctx, s[6], (size_t)Hyp * 4));
  VO_TRY(vo_ensure(ctx, s[7], (size_t)Hyp * words * 8));
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, X, (size_t)N * 24, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, x, (size_t)N * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, samples, (size_t)Hyp * 16, hipMemcpyHostToDevice, st));
  VO_TRY(vo_p3p_hypotheses_dev(ctx, (const double*)s[0].p, (const double*)s[1].p, N, K, (const int32_t*)s[2].p, Hyp,
                               thr_sq, (double*)s[3].p, (double*)s[4].p, (uint8_t*)s[5].p, (int32_t*)s[6].p,
                               (uint64_t*)s[7].p));
  VO_HIP_TRY(ctx, hipMemcpyAsync(R, s[3].p, (size_t)Hyp * 72, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(t, s[4].p, (size_t)Hyp * 24, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(valid, s[5].p, (size_t)Hyp, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(counts, s[6].p, (size_t)Hyp * 4, hipMemcpyDeviceToHost, st));
  if (masks) VO_HIP_TRY(ctx, hipMemcpyAsync(masks, s[7].p, (size_t)Hyp * words * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

int vo_reproj_inliers(vo_ctx* ctx, const double* X, const double* x, int N, const double* K, const double* R,
                      const double* t, double thr_sq, uint8_t* mask, double* err) {
  if (!ctx) return VO_EINVAL;
  VO_REQUIRE(ctx, X && x && K && R && t && (mask || err), "reproj_inliers: null pointer");
  VO_REQUIRE(ctx, N >= 0, "reproj_inliers: bad N");
  if (N == 0) return VO_OK;
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], (size_t)N * 24));
  VO_TRY(vo_ensure(ctx, s[1], (size_t)N * 16));
  VO_TRY(vo_ensure(ctx, s[2], 96));
  VO_TRY(vo_ensure(ctx, s[5], (size_t)N));
  VO_TRY(vo_ensure(ctx, s[3], (size_t)N * 8));
  double Rt[12];
  memcpy(Rt, R, 72);
  memcpy(Rt + 9, t, 24);
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, X, (size_t)N * 24, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[1].p, x, (size_t)N * 16, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[2].p, Rt, 96, hipMemcpyHostToDevice, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));   // Rt is a stack buffer
  VO_TRY(vo_reproj_inliers_dev(ctx, (const double*)s[0].p, (const double*)s[1].p, N, K, (const double*)s[2].p, thr_sq,
                               mask ? (uint8_t*)s[5].p : nullptr, err ? (double*)s[3].p : nullptr));
  if (mask) VO_HIP_TRY(ctx, hipMemcpyAsync(mask, s[5].p, (size_t)N, hipMemcpyDeviceToHost, st));
  if (err) VO_HIP_TRY(ctx, hipMemcpyAsync(err, s[3].p, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

}  // extern "C"
