// v_SR(d): the real-space pair kernel and its derivative, shared by the pair kernels.
// Reference: Potential.sr_from_dist / from_dist / f_cutoff (potentials/potential.py:59-138),
// CoulombPotential (potentials/coulomb.py:80-120), InversePowerLawPotential (potentials/inversepowerlaw.py:55-106).
#pragma once

#include "common.h"

namespace mipme {

static constexpr double kPiR = 3.14159265358979323846;

// Device-side description of v_SR(d)
struct SRPot {
  int mode;        // 0: bare v, 1: v - v_LR (range separated), 2: -v_LR * f_cut, 3: v * (1 - f_cut)
  int p;           // exponent 1..6
  double pref;
  double inv_2s2;  // 1/(2 sigma^2)
  double rx;       // exclusion radius
  int deg;         // exclusion degree
};

inline int make_srpot(const mipme_potential_t* pot, SRPot& s) {
  MIPME_REQUIRE(pot != nullptr, "potential descriptor is NULL");
  s.p = pot->kind == MIPME_COULOMB ? 1 : pot->exponent;
  MIPME_REQUIRE(s.p >= 1 && s.p <= 6, "Unsupported exponent: %d", s.p);
  const bool smeared = pot->smearing > 0;
  const bool excl = pot->exclusion_radius > 0;
  s.mode = smeared ? (excl ? 2 : 1) : (excl ? 3 : 0);
  s.pref = pot->prefactor;
  s.inv_2s2 = smeared ? 0.5 / (pot->smearing * pot->smearing) : 0.0;
  s.rx = pot->exclusion_radius;
  s.deg = pot->exclusion_degree;
  return MIPME_OK;
}

__device__ __forceinline__ float fexp(float x) { return expf(x); }
__device__ __forceinline__ double fexp(double x) { return exp(x); }
__device__ __forceinline__ float ferfc(float x) { return erfcf(x); }
__device__ __forceinline__ double ferfc(double x) { return erfc(x); }
__device__ __forceinline__ float fsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float fsin(float x) { return sinf(x); }
__device__ __forceinline__ double fsin(double x) { return sin(x); }
__device__ __forceinline__ float fcos(float x) { return cosf(x); }
__device__ __forceinline__ double fcos(double x) { return cos(x); }

// erfc(y) given e = exp(-y^2), y >= 0.  float: erfc = e * t * P8(t), t = 1/(1 + 0.4 y), P8 fitted (Chebyshev nodes) to
// erfcx(y)/t on [0, 6.5]: relative error <= 3.7e-7 in fp32 arithmetic (the libm erfcf costs ~4x the instructions
// and made the pair kernels VALU-bound); beyond y = 6.5, where erfc < 4e-20, the error grows to 2e-5 relative.
__device__ __forceinline__ float erfc_from_exp(float y, float e) {
  const float t = 1.0f / (1.0f + 0.4f * y);
  float p = 2.646481385e-02f;
  p = p * t + -6.557867191e-02f;
  p = p * t + -7.738398321e-02f;
  p = p * t + 2.820383187e-01f;
  p = p * t + -6.220284696e-02f;
  p = p * t + 2.579626189e-01f;
  p = p * t + 1.840778096e-01f;
  p = p * t + 2.291638826e-01f;
  p = p * t + 2.254580744e-01f;
  return e * t * p;
}
// double: erfc(y) = e * t * P20(x), x = (2 t - (1 + tlo)) / (1 - tlo) in [-1, 1] for t in [1/(1+0.4*27), 1]: a degree-20 fit of
// erfcx(y)/t at Chebyshev nodes (max relative error 2e-15 for 0 <= y <= 27, beyond which e = exp(-y^2) underflows anyway), given
// by its MONOMIAL coefficients in x and evaluated with Horner's rule: 20 FMAs (the Chebyshev form with Clenshaw's recurrence,
// round 2, needed 40 operations for the same accuracy; the coefficients decay faster than the basis grows, so the monomial
// form loses nothing: 3.8e-15 against 4.2e-15).
#define MIPME_ERFC_CHEB                                                                                                   \
  {4.50235299459692540e-01,  3.24125536248557999e-01,  1.67075628387939157e-01,  5.52651679935470402e-02,                \
   6.82018143828926407e-03,  -2.76262477755241610e-03, -1.03953621374069873e-03, 1.92114029446880925e-04,                \
   1.22950360871657786e-04,  -2.59317092799155009e-05, -1.43585248578882984e-05, 5.12388472713849280e-06,                \
   1.35756476325816326e-06,  -1.02658932393014442e-06, -3.47187889529154472e-09, 1.73400820276635575e-07,                \
   -4.17454076893041247e-08, -2.09125782862454419e-08, 1.11490864632721937e-08,  1.28815160249804723e-09,                \
   -1.26127363551349624e-09}
static constexpr int kErfcChebTerms = 21;
// 1/x and 1/sqrt(x) in double from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26) and two Newton steps each: within an
// ulp or two of the IEEE sequences (v_div_scale / fmas / fixup, sqrt + division) at a third of their instruction count
__device__ __forceinline__ double rcp_newton(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}
// c: the 21 coefficients.  The fused pair kernels pass the copy that travels in their kernel arguments (FastRS::cheb): it is
// loaded once into scalar registers, where 21 double literals would be re-materialised with two v_mov each per evaluation
// (65 of the ~300 instructions of the fp64 pair loop).
__device__ __forceinline__ double erfc_from_exp(double y, double e, const double* __restrict__ c) {
  constexpr double tlo = 0.0847457627118644;
  constexpr double xs = 2.0 / (1.0 - tlo), x0 = -(1.0 + tlo) / (1.0 - tlo);
  const double t = rcp_newton(1.0 + 0.4 * y);
  const double x = __builtin_fma(xs, t, x0);  // (2 t - (1 + tlo)) / (1 - tlo)
  double p = c[kErfcChebTerms - 1];
#pragma unroll
  for (int k = kErfcChebTerms - 2; k >= 0; --k) p = __builtin_fma(p, x, c[k]);
  return e * t * p;
}
// exp(-x), x >= 0, in double: n = round(-x log2 e), r = -x - n ln 2 (two-step reduction), e^r from its degree-13 Taylor
// polynomial (|r| <= ln 2 / 2: 2e-16), scaled by 2^n -- 18 instructions where libm's exp takes about twice as many (its range
// checks and table look-ups are for arguments this one never sees)
__device__ __forceinline__ double exp_neg_fast(double x) {
  x = x < 700.0 ? x : 700.0;
  const double n = __builtin_rint(-x * 1.4426950408889634);
  double r = __builtin_fma(n, -0.693147180369123816490, -x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;  // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);
  p = __builtin_fma(p, r, 2.505210838544172e-08);
  p = __builtin_fma(p, r, 2.755731922398589e-07);
  p = __builtin_fma(p, r, 2.7557319223985893e-06);
  p = __builtin_fma(p, r, 2.48015873015873e-05);
  p = __builtin_fma(p, r, 0.0001984126984126984);
  p = __builtin_fma(p, r, 0.001388888888888889);
  p = __builtin_fma(p, r, 0.008333333333333333);
  p = __builtin_fma(p, r, 0.041666666666666664);
  p = __builtin_fma(p, r, 0.16666666666666666);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, int(n));
}
__device__ __forceinline__ double erfc_from_exp(double y, double e) {
  constexpr double c[kErfcChebTerms] = MIPME_ERFC_CHEB;
  return erfc_from_exp(y, e, c);
}
// same fit with the hardware reciprocal (1 ulp) for t
__device__ __forceinline__ float erfc_from_exp_fast(float y, float e) {
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.4f * y);
  float p = 2.646481385e-02f;
  p = p * t + -6.557867191e-02f;
  p = p * t + -7.738398321e-02f;
  p = p * t + 2.820383187e-01f;
  p = p * t + -6.220284696e-02f;
  p = p * t + 2.579626189e-01f;
  p = p * t + 1.840778096e-01f;
  p = p * t + 2.291638826e-01f;
  p = p * t + 2.254580744e-01f;
  return e * t * p;
}

template <typename T>
__device__ __forceinline__ T powi(T x, int n) {
  T r = T(1);
  for (int i = 0; i < n; ++i) r *= x;
  return r;
}

// Q(p/2, x) regularised upper incomplete gamma and x^(p/2-1) e^-x / Gamma(p/2) for integer p in 1..6.
//   integer a:      Q(a,x) = e^-x sum_{k<a} x^k/k!
//   half-integer a: Q(1/2,x) = erfc(sqrt x);  Q(a+1,x) = Q(a,x) + x^a e^-x / Gamma(a+1)
template <typename T>
__device__ __forceinline__ void upper_gamma(int p, T x, T& Q, T& dens) {
  const T ex = fexp(-x);
  if ((p & 1) == 0) {
    const int a = p / 2;  // 1,2,3
    T term = T(1), sum = T(1);
    for (int k = 1; k < a; ++k) {
      term *= x / T(k);
      sum += term;
    }
    Q = ex * sum;
    dens = ex * term;  // x^(a-1)/(a-1)!
  } else {
    const T sx = fsqrt(x);
    const T isp = T(0.56418958354775628695);  // 1/sqrt(pi) = 1/Gamma(1/2)
    Q = erfc_from_exp(sx, ex);
    // term_k = x^(k-1/2) e^-x / Gamma(k+1/2)
    T term = (x > T(0)) ? ex * isp / sx : T(0);  // k = 0: x^(-1/2)/Gamma(1/2)
    dens = term;
    for (int k = 1; k <= (p - 1) / 2; ++k) {
      term *= x / (T(k) - T(0.5));
      Q += term;
      dens = term;
    }
  }
}

// P(p/2, x) = 1 - Q, the regularised LOWER incomplete gamma, for small x from its power series
//   P(a, x) = x^a e^-x sum_{k>=0} x^k / Gamma(a + k + 1)
// (what the reference evaluates directly: torch.special.gammainc, inversepowerlaw.py:98-103).  1 - Q loses all relative
// precision as x -> 0 -- P ~ x^a / Gamma(a+1) -- which matters inside an exclusion radius (mode 2: v = -v_LR f_cut): in fp32
// with p = 5, 6 it rounded to zero below d ~ 0.13 sigma.  Used for x < 1, where 12 (float) / 22 (double) terms reach the
// rounding level (term ratio x / (a + k)).
template <typename T>
__device__ __forceinline__ T lower_gamma_series(int p, T x) {
  // 1 / Gamma(p/2 + 1), p = 1..6
  constexpr double inv_gamma[6] = {1.1283791670955126, 1.0, 0.7522527780636751, 0.5, 0.30090111122547003, 1.0 / 6.0};
  const T a = T(0.5) * T(p);
  T term = T(inv_gamma[p - 1]), sum = term;
  constexpr int K = sizeof(T) == 4 ? 12 : 22;
#pragma unroll
  for (int k = 1; k <= K; ++k) {
    term *= x / (a + T(k));
    sum += term;
  }
  T xa = powi(x, p / 2);
  if (p & 1) xa *= fsqrt(x);
  return xa * fexp(-x) * sum;
}

// v_SR(d) and dv_SR/dd (see oracle/pme_numpy.py::sr_pair for the derivation).
template <typename T, bool DERIV>
__device__ __forceinline__ void sr_eval(const SRPot& s, T d, T& v, T& dv) {
  const T pref = T(s.pref);
  const T dc = d > T(1e-15) ? d : T(1e-15);
  const T inv = T(1) / dc;
  const T invp = powi(inv, s.p);
  T fc = T(0), dfc = T(0);
  if (s.mode >= 2) {
    const T rx = T(s.rx);
    if (d < rx) {
      const T arg = T(kPiR) * d / rx;
      const T base = T(0.5) * (T(1) - fcos(arg));
      fc = T(1) - powi(base, s.deg);
      if constexpr (DERIV) dfc = -T(s.deg) * powi(base, s.deg - 1) * T(0.5) * T(kPiR) / rx * fsin(arg);
    }
  }
  if (s.mode == 0 || s.mode == 3) {
    const T vb = pref * invp;
    const T dvb = -T(s.p) * vb * inv;
    if (s.mode == 0) {
      v = vb;
      if constexpr (DERIV) dv = dvb;
    } else {
      v = vb * (T(1) - fc);
      if constexpr (DERIV) dv = dvb * (T(1) - fc) - vb * dfc;
    }
    return;
  }
  const T x = dc * dc * T(s.inv_2s2);
  T Q, dens;
  upper_gamma<T>(s.p, x, Q, dens);
  const T dxdd = T(2) * dc * T(s.inv_2s2);
  if (s.mode == 1) {
    v = pref * Q * invp;
    if constexpr (DERIV) dv = pref * (-dens * dxdd * invp - T(s.p) * Q * invp * inv);
  } else {
    const T P = x < T(1) ? lower_gamma_series<T>(s.p, x) : T(1) - Q;
    const T vl = pref * P * invp;
    v = -vl * fc;
    if constexpr (DERIV) {
      const T dvl = pref * (dens * dxdd * invp - T(s.p) * P * invp * inv);
      dv = -(dvl * fc + vl * dfc);
    }
  }
}

// Fast path of the fused row kernels for range-separated potentials without exclusion (mode 1) and a compile-time
// exponent P:   v = pref Q(P/2, x) / d^P,   (dv/dd) / d = -pref (2 x dens + P Q) / d^(P+2),   x = d^2 / (2 sigma^2),
// dens = x^(P/2-1) e^-x / Gamma(P/2), evaluated from d^2.  The float version uses the hardware rsq / rcp / exp2 (1 ulp
// each) instead of IEEE division sequences and libm range reduction: the generic sr_eval costs ~190 VALU slots per pair,
// which made the fused kernels VALU-bound.  P = 1 is the Coulomb potential.
struct FastRS {
  double inv_2s2, c1, pref;  // c1 = 1/(sigma sqrt 2)
  double cheb[kErfcChebTerms];  // erfc coefficients of the fp64 path (see erfc_from_exp)
};
inline FastRS make_fast_rs(const SRPot& s) {
  FastRS f{s.inv_2s2, sqrt(s.inv_2s2), s.pref, MIPME_ERFC_CHEB};
  return f;
}
// exponents with a dedicated instantiation (others take the generic sr_eval): Coulomb and dispersion
inline int fast_rs_exponent(const SRPot& s) { return (s.mode == 1 && (s.p == 1 || s.p == 6)) ? s.p : 0; }

__device__ __forceinline__ float rs_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rs_rsqrt(double x) { return rsqrt_newton(x); }
__device__ __forceinline__ float rs_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rs_rcp(double x) { return rcp_newton(x); }
__device__ __forceinline__ float rs_exp_neg(float x) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * x); }
__device__ __forceinline__ double rs_exp_neg(double x) { return exp_neg_fast(x); }
__device__ __forceinline__ float rs_erfc(float y, float e, const double*) { return erfc_from_exp_fast(y, e); }
__device__ __forceinline__ double rs_erfc(double y, double e, const double* c) { return erfc_from_exp(y, e, c); }

template <int P, bool DERIV, typename T>
__device__ __forceinline__ void fast_rs_eval(T inv_2s2, T c1, T pref, T d2, T& v, T& dvd, const double* cheb) {
  d2 = d2 > T(1e-30) ? d2 : T(1e-30);
  const T inv = rs_rsqrt(d2);
  const T inv2 = inv * inv;
  const T x = d2 * inv_2s2;
  const T e = rs_exp_neg(x);
  T Q, two_x_dens;  // Q(P/2, x) and 2 x dens(x)
  if constexpr (P % 2 == 0) {
    T term = T(1), sum = T(1);
#pragma unroll
    for (int k = 1; k < P / 2; ++k) {
      term *= x * T(1.0 / k);
      sum += term;
    }
    Q = e * sum;
    two_x_dens = T(2) * x * e * term;
  } else {
    // half-integer a = m + 1/2: term_k = x^(k-1/2) e^-x / Gamma(k+1/2); term_1 = 2 y e / sqrt(pi) with y = sqrt(x), so no
    // division by y is needed: Q = erfc(y) + sum_{k=1..m} term_k and 2 x dens = 2 x term_m = (2m+1) term_{m+1}
    constexpr int m = (P - 1) / 2;
    const T y = c1 * (d2 * inv);
    Q = rs_erfc(y, e, cheb);
    T term = T(2.0 * 0.56418958354775628695) * y * e;  // term_1
#pragma unroll
    for (int k = 1; k <= m; ++k) {
      Q += term;
      term *= x * T(1.0 / (k + 0.5));  // -> term_{k+1}
    }
    two_x_dens = T(2 * m + 1) * term;
  }
  T invp = inv;
#pragma unroll
  for (int k = 1; k < P; ++k) invp *= inv;
  const T pi = pref * invp;
  v = pi * Q;
  if constexpr (DERIV) dvd = -(pi * inv2) * (two_x_dens + T(P) * Q);
}

}  // namespace mipme
