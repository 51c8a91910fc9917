/*
 * CPU oracle: P3P pose from 4 correspondences + reprojection scoring.
 * TEST INFRASTRUCTURE (see oracle/__init__.py) -- never linked into the product.
 *
 * Follows the reference call sites
 *   model_fn  src/vo/pose_estimation/p3p.py:51-79   cv2.solvePnP(..., SOLVEPNP_P3P) on 4 points
 *   error_fn  src/vo/pose_estimation/p3p.py:81-108  cv2.projectPoints + ||.||^2
 *   inliers   src/vo/algorithms/ransac.py:104-106   errors < inlier_threshold
 *
 * PARITY UNPINNED against OpenCV: the P3P solve and projectPoints live in
 * opencv-python==4.8.1.78 (environment.yml:22), which is neither under
 * /root/reference nor installable here.  OpenCV's SOLVEPNP_P3P is the algebraic
 * P3P of Gao et al. (2003): a quartic whose real roots give the point depths,
 * up to four poses from the first three points, the fourth point selecting the
 * pose with the least reprojection error.  This file restates that scheme with
 * Grunert's elimination (same solution set): ratios u = s2/s1, v = s3/s1 of the
 * three depths, one quartic in v, closed-form (Ferrari) roots polished by
 * Newton, rigid alignment of the two point triads, fourth-point selection.
 * Checked against analytic ground truth (tests/test_oracle_p3p.py), and against
 * the reference's own tolerance (tests/test_p3p.py:93-98: R, t within 1e-3).
 *
 * Only +, -, *, /, sqrt are used so that a second implementation following the
 * same operation order reproduces the results bit for bit.  Compile with
 * -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static double cubic_eval(double A, double B, double C, double x) { return ((x + A) * x + B) * x + C; }

static double cubic_positive_root(double A, double B, double C) {
  /* a real root in (0, U] of m^3 + A m^2 + B m + C, given C < 0.  U bounds every positive root:
   * beyond it each of |A| m^2, |B| m, |C| is below m^3 / 3.  L bounds them from below:
   * -C = m (m^2 + A m + B) <= m (U^2 + |A| U + |B|).  The sign-change bracket [L, U] is cut in
   * four per round -- at geometric points while it spans more than four octaves, evenly after
   * that -- keeping the leftmost sign change, down to a relative width of 2^-36; one Newton
   * step finishes.  No data-dependent branches on anything but signs, and the three cut points of
   * a round are independent (the GPU evaluates them in three lanes).  Fixed operation order;
   * + - * / sqrt and exponent arithmetic only, so every implementation gets the same bits. */
  double U = 3.0 * fabs(A);
  const double sb = sqrt(3.0 * fabs(B));
  if (sb > U) U = sb;
  int e;
  (void)frexp(3.0 * fabs(C), &e);                          /* 3|C| < 2^e */
  const int e3 = (e >= 0) ? (e + 2) / 3 : -((-e) / 3);     /* ceil(e / 3) */
  const double cb = ldexp(1.0, e3);
  if (cb > U) U = cb;
  double xl = fabs(C) / ((U + fabs(A)) * U + fabs(B));
  double xh = 1.0625 * U;
  if (!(cubic_eval(A, B, C, xl) < 0.0)) xl = 0.0;          /* rounding spoiled the lower bound */
  for (int it = 0; it < 64; ++it) {
    if (!(xh - xl > 1.4551915228366852e-11 * xh)) break;   /* 2^-36 */
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
    const int s1 = cubic_eval(A, B, C, p1) >= 0.0;
    const int s2 = cubic_eval(A, B, C, p2) >= 0.0;
    const int s3 = cubic_eval(A, B, C, p3) >= 0.0;
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

static int quadratic_real(double b, double c, double* r) {
  /* y^2 + b y + c = 0 */
  double disc = b * b - 4.0 * c;
  if (disc < 0.0) return 0;
  double sq = sqrt(disc);
  double q = (b >= 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
  r[0] = q;
  r[1] = (q != 0.0) ? c / q : 0.0;
  return 2;
}

/* real roots of c[4] x^4 + ... + c[0]; returns their number (0..4) */
static int quartic_real(const double c[5], double roots[4]) {
  if (c[4] == 0.0) return 0;
  double a3 = c[3] / c[4], a2 = c[2] / c[4], a1 = c[1] / c[4], a0 = c[0] / c[4];
  double a3sq = a3 * a3;
  double p = a2 - 0.375 * a3sq;
  double q = a1 - 0.5 * a2 * a3 + 0.125 * a3sq * a3;
  double r = a0 - 0.25 * a1 * a3 + 0.0625 * a2 * a3sq - (3.0 / 256.0) * a3sq * a3sq;
  double y[4];
  int n = 0;
  if (q == 0.0) {
    double z[2];
    int nz = quadratic_real(p, r, z);
    for (int i = 0; i < nz; ++i) {
      if (z[i] >= 0.0) {
        double s = sqrt(z[i]);
        y[n++] = s;
        y[n++] = -s;
      }
    }
  } else {
    double m = cubic_positive_root(p, 0.25 * p * p - r, -0.125 * q * q);
    if (!(m > 0.0)) return 0;
    double s = sqrt(2.0 * m);
    double h = 0.5 * p + m;
    double g = q / (2.0 * s);
    n += quadratic_real(s, h - g, y + n);
    n += quadratic_real(-s, h + g, y + n);
  }
  double shift = 0.25 * a3;
  for (int i = 0; i < n; ++i) {
    double x = y[i] - shift;
    for (int it = 0; it < 3; ++it) { /* Newton polish on the monic quartic */
      double f = (((x + a3) * x + a2) * x + a1) * x + a0;
      double df = ((4.0 * x + 3.0 * a3) * x + 2.0 * a2) * x + a1;
      if (df == 0.0) break;
      x = x - f / df;
    }
    roots[i] = x;
  }
  return n;
}

static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

static int triad(const double p1[3], const double p2[3], const double p3[3], double E[9]) {
  /* columns e1, e2, e3 stored row-major: E[3*r + c] = e_c[r] */
  double a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double b[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]};
  double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (!(na > 0.0)) return 0;
  double e1[3] = {a[0] / na, a[1] / na, a[2] / na};
  double e3[3];
  cross3(e1, b, e3);
  double n3 = sqrt(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
  if (!(n3 > 0.0)) return 0;
  e3[0] /= n3;
  e3[1] /= n3;
  e3[2] /= n3;
  double e2[3];
  cross3(e3, e1, e2);
  for (int r = 0; r < 3; ++r) {
    E[3 * r + 0] = e1[r];
    E[3 * r + 1] = e2[r];
    E[3 * r + 2] = e3[r];
  }
  return 1;
}

/* squared reprojection error exactly as the reference forms it:
 * cv2.projectPoints (x' = X' * (1/Z'), u = x' fx + cx) then
 * np.linalg.norm(diff, axis=(1,2)) ** 2 = (sqrt(dx^2 + dy^2))^2  (p3p.py:101-107) */
static double reproj_err(const double R[9], const double t[3], const double K[9], const double X[3],
                         const double x[2]) {
  double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  double iz = (zc != 0.0) ? 1.0 / zc : 1.0;
  double xn = xc * iz, yn = yc * iz;
  double u = xn * K[0] + K[2];
  double v = yn * K[4] + K[5];
  double dx = x[0] - u, dy = x[1] - v;
  double nrm = sqrt(dx * dx + dy * dy);
  return nrm * nrm;
}

/* X4: 4x3 world points, x4: 4x2 pixels, K: 3x3 row-major.  Returns 1 and R (world->camera), t on success. */
int oracle_p3p_solve(const double* X4, const double* x4, const double* K, double* R_out, double* t_out) {
  const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  double f[3][3];
  for (int i = 0; i < 3; ++i) {
    double mu = (x4[2 * i] - cx) / fx, mv = (x4[2 * i + 1] - cy) / fy;
    double nrm = sqrt(mu * mu + mv * mv + 1.0);
    f[i][0] = mu / nrm;
    f[i][1] = mv / nrm;
    f[i][2] = 1.0 / nrm;
  }
  const double* P1 = X4;
  const double* P2 = X4 + 3;
  const double* P3 = X4 + 6;
  double d12s = 0, d13s = 0, d23s = 0;
  for (int k = 0; k < 3; ++k) {
    double a = P1[k] - P2[k], b = P1[k] - P3[k], c = P2[k] - P3[k];
    d12s += a * a;
    d13s += b * b;
    d23s += c * c;
  }
  if (!(d12s > 0.0) || !(d13s > 0.0) || !(d23s > 0.0)) return 0;
  double c12 = f[0][0] * f[1][0] + f[0][1] * f[1][1] + f[0][2] * f[1][2];
  double c13 = f[0][0] * f[2][0] + f[0][1] * f[2][1] + f[0][2] * f[2][2];
  double c23 = f[1][0] * f[2][0] + f[1][1] * f[2][1] + f[1][2] * f[2][2];

  /* s2 = u s1, s3 = v s1;  a = d12^2/d13^2, b = d23^2/d13^2, q(v) = v^2 - 2 c13 v + 1
   *   u^2 - 2 c12 u + 1 - a q = 0,   u^2 - 2 c23 v u + v^2 - b q = 0
   * =>  u = Nn(v) / Dd(v),  Nn = v^2 - 1 + (a - b) q,  Dd = 2 (c23 v - c12)
   * =>  Nn^2 - 2 c12 Nn Dd + (1 - a q) Dd^2 = 0   (quartic in v)                  */
  double a = d12s / d13s, b = d23s / d13s, g = a - b;
  double n2 = 1.0 + g, n1 = -2.0 * g * c13, n0 = g - 1.0;       /* Nn */
  double e1 = 2.0 * c23, e0 = -2.0 * c12;                         /* Dd */
  double w2 = -a, w1 = 2.0 * a * c13, w0 = 1.0 - a;               /* 1 - a q */
  double dd2 = e1 * e1, dd1 = 2.0 * e1 * e0, dd0 = e0 * e0;       /* Dd^2 */
  double nd3 = n2 * e1, nd2 = n2 * e0 + n1 * e1, nd1 = n1 * e0 + n0 * e1, nd0 = n0 * e0; /* Nn Dd */
  double tc = 2.0 * c12;
  double coef[5];
  coef[4] = n2 * n2 + w2 * dd2;
  coef[3] = 2.0 * n2 * n1 - tc * nd3 + (w2 * dd1 + w1 * dd2);
  coef[2] = (2.0 * n2 * n0 + n1 * n1) - tc * nd2 + (w2 * dd0 + w1 * dd1 + w0 * dd2);
  coef[1] = 2.0 * n1 * n0 - tc * nd1 + (w1 * dd0 + w0 * dd1);
  coef[0] = n0 * n0 - tc * nd0 + w0 * dd0;

  double roots[4];
  int nr = quartic_real(coef, roots);
  double Ew[9];
  if (!triad(P1, P2, P3, Ew)) return 0;

  int found = 0;
  double best = 0.0;
  for (int i = 0; i < nr; ++i) {
    double v = roots[i];
    if (!(v > 0.0)) continue;
    double Dd = e1 * v + e0;
    if (fabs(Dd) < 1e-12) continue;
    double Nn = (n2 * v + n1) * v + n0;
    double u = Nn / Dd;
    if (!(u > 0.0)) continue;
    double qv = (v - 2.0 * c13) * v + 1.0;
    if (!(qv > 0.0)) continue;
    double s1 = sqrt(d13s / qv);
    double s2 = u * s1, s3 = v * s1;
    double C1[3] = {s1 * f[0][0], s1 * f[0][1], s1 * f[0][2]};
    double C2[3] = {s2 * f[1][0], s2 * f[1][1], s2 * f[1][2]};
    double C3[3] = {s3 * f[2][0], s3 * f[2][1], s3 * f[2][2]};
    double Ec[9];
    if (!triad(C1, C2, C3, Ec)) continue;
    double R[9], t[3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        R[3 * r + c] = Ec[3 * r + 0] * Ew[3 * c + 0] + Ec[3 * r + 1] * Ew[3 * c + 1] + Ec[3 * r + 2] * Ew[3 * c + 2];
    for (int r = 0; r < 3; ++r) t[r] = C1[r] - (R[3 * r] * P1[0] + R[3 * r + 1] * P1[1] + R[3 * r + 2] * P1[2]);
    double e = reproj_err(R, t, K, X4 + 9, x4 + 6);
    if (!(e == e)) continue; /* NaN */
    if (!found || e < best) {
      found = 1;
      best = e;
      memcpy(R_out, R, sizeof(R));
      memcpy(t_out, t, sizeof(t));
    }
  }
  return found;
}

void oracle_reproj_errors(const double* X, const double* x, int N, const double* K, const double* R,
                          const double* t, double* err) {
  for (int i = 0; i < N; ++i) err[i] = reproj_err(R, t, K, X + 3 * i, x + 2 * i);
}

/* Batched: for each sample of 4 indices solve P3P, then count points with err < thr (strict). */
void oracle_p3p_hypotheses(const double* X, const double* x, int N, const double* K, const int32_t* samples,
                           int Hyp, double thr, double* R, double* t, uint8_t* valid, int32_t* counts,
                           uint8_t* masks /* Hyp*N, nullable */) {
  for (int h = 0; h < Hyp; ++h) {
    double X4[12], x4[8];
    for (int k = 0; k < 4; ++k) {
      int idx = samples[4 * h + k];
      memcpy(X4 + 3 * k, X + 3 * idx, 3 * sizeof(double));
      memcpy(x4 + 2 * k, x + 2 * idx, 2 * sizeof(double));
    }
    double* Rh = R + 9 * h;
    double* th = t + 3 * h;
    int ok = oracle_p3p_solve(X4, x4, K, Rh, th);
    valid[h] = (uint8_t)ok;
    int cnt = 0;
    if (ok) {
      for (int i = 0; i < N; ++i) {
        int in = reproj_err(Rh, th, K, X + 3 * i, x + 2 * i) < thr;
        cnt += in;
        if (masks) masks[(size_t)h * N + i] = (uint8_t)in;
      }
    } else {
      memset(Rh, 0, 9 * sizeof(double));
      memset(th, 0, 3 * sizeof(double));
      if (masks) memset(masks + (size_t)h * N, 0, N);
    }
    counts[h] = cnt;
  }
}
