// v_LR^(k^2): the reciprocal-space kernel of the range-separated potentials and its derivative, shared by the mesh
// filter (kfilter.hip) and the explicit Ewald sum (ewald.hip).
// Reference: Potential.lr_from_k_sq (potentials/coulomb.py:122-142, potentials/inversepowerlaw.py:109-141, lib/math.py:85-104).
#pragma once

#include <cmath>

#include "common.h"

namespace mipme {

static constexpr double kPi = 3.14159265358979323846;

struct KPot {
  int p;          // exponent
  double c0;      // prefactor * pi^1.5 / Gamma(p/2) * (2 sigma^2)^((3-p)/2)
  double hs2;     // sigma^2 / 2
  double a;       // (3-p)/2
  double k0;      // value at k = 0
};

static inline int make_kpot(const mipme_potential_t* pot, KPot& k) {
  MIPME_REQUIRE(pot != nullptr, "potential descriptor is NULL");
  MIPME_REQUIRE(pot->smearing > 0, "`smearing` is %g but must be positive", pot->smearing);
  k.p = pot->kind == MIPME_COULOMB ? 1 : pot->exponent;
  MIPME_REQUIRE(k.p >= 1 && k.p <= 6, "Unsupported exponent: %d", k.p);
  k.a = 0.5 * (3 - k.p);
  const double two_s2 = 2.0 * pot->smearing * pot->smearing;
  k.c0 = pot->prefactor * std::pow(kPi, 1.5) / std::tgamma(0.5 * k.p) * std::pow(two_s2, k.a);
  k.hs2 = 0.5 * pot->smearing * pot->smearing;
  k.k0 = k.p > 3 ? -k.c0 / k.a : 0.0;
  return MIPME_OK;
}

// E1(z), z > 0: power series for z <= 1, continued fraction (backward recurrence) above.
__device__ inline double exp1_dev(double z) {
  if (z <= 1.0) {
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 40; ++k) {
      term *= -z * double(k) / (double(k + 1) * double(k + 1));
      sum += term;
      if (fabs(term) <= fabs(sum) * 1e-17) break;
    }
    return -0.57721566490153286061 - log(z) + z * sum;
  }
  const int m = 20 + int(80.0 / z);
  double t = 0.0;
  for (int k = m; k >= 1; --k) t = double(k) / (1.0 + double(k) / (z + t));
  return exp(-z) / (z + t);
}

// f_p(z) = Gamma(a, z) / z^a with a = (3-p)/2 (closed forms for integer p) and its derivative
// f_p' = -(exp(-z) + a f_p)/z.
__device__ inline void lr_kernel_dev(const KPot& kp, double k2, double& v, double& dv_dk2) {
  if (k2 == 0.0) {
    v = kp.k0;
    dv_dk2 = 0.0;
    return;
  }
  const double z = kp.hs2 * k2;
  const double ez = exp(-z);
  double f;
  switch (kp.p) {
    case 1: f = ez / z; break;
    case 2: f = sqrt(kPi / z) * erfc(sqrt(z)); break;
    case 3: f = exp1_dev(z); break;
    case 4: f = 2.0 * (ez - sqrt(kPi * z) * erfc(sqrt(z))); break;
    case 5: f = ez - z * exp1_dev(z); break;
    default: f = ((2.0 - 4.0 * z) * ez + 4.0 * sqrt(kPi * z * z * z) * erfc(sqrt(z))) / 3.0; break;
  }
  v = kp.c0 * f;
  dv_dk2 = kp.c0 * (-(ez + kp.a * f) / z) * kp.hs2;
}

}  // namespace mipme
