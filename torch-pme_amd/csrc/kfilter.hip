// Reciprocal-space part: G(k) table, hipFFT R2C -> * G -> C2R, and the k-grid reductions of the
// cell gradient.
//
// Replaces generate_kvectors_for_mesh (reference lib/kvectors.py:24-74), KSpaceFilter.update/forward
// (lib/kspace_filter.py:97-197), P3MKSpaceFilter.update/_compute_influence/_charge_assignment
// (lib/kspace_filter.py:293-329,349-361) and Potential.lr_from_k_sq (potentials/coulomb.py:122-142,
// potentials/inversepowerlaw.py:109-141, lib/math.py:85-104).  The reference materialises the
// (nx,ny,nz/2+1,3) k-vector grid and k^2; here every thread derives its k-vector from its index.
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdlib>
#include <new>

#include "common.h"
#include "kpot.h"
#include "fft_lds.h"

namespace mipme {

struct KGeom {
  double inv[9];  // inverse cell
  double h[3];    // |a_c| / n_c (P3M charge-assignment spacing, kspace_filter.py:308-311)
  int nx, ny, nz, nzh;
  int scheme, order;
};


static inline KGeom make_kgeom(const mipme_mesh_t* m) {
  KGeom g;
  for (int i = 0; i < 9; ++i) g.inv[i] = m->inv_cell[i];
  const int ns[3] = {m->nx, m->ny, m->nz};
  for (int c = 0; c < 3; ++c) {
    const double* a = m->cell + 3 * c;
    g.h[c] = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) / double(ns[c]);
  }
  g.nx = m->nx;
  g.ny = m->ny;
  g.nz = m->nz;
  g.nzh = m->nz / 2 + 1;
  g.scheme = m->scheme;
  g.order = m->order;
  return g;
}

__device__ inline int fft_freq(int i, int n) { return i < (n + 1) / 2 ? i : i - n; }

struct KPoint {
  double k[3];
  int f[3];
  double G, dGdk[3], dGdh[3];
  // dG/dk_c = alpha k_c - beta_c h_c,  dG/dh_c = -beta_c k_c: the four values per k-point the derivative table keeps
  double alpha, beta[3];
};

template <bool DERIV>
__device__ inline void eval_point(const KGeom& g, const KPot& kp, int ix, int iy, int iz, KPoint& o) {
  o.f[0] = fft_freq(ix, g.nx);
  o.f[1] = fft_freq(iy, g.ny);
  o.f[2] = iz;
  double k2 = 0.0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o.k[c] = 2.0 * kPi * (o.f[0] * g.inv[3 * c + 0] + o.f[1] * g.inv[3 * c + 1] + o.f[2] * g.inv[3 * c + 2]);
    k2 += o.k[c] * o.k[c];
  }
  double v, dv;
  lr_kernel_dev(kp, k2, v, dv);
  if (g.scheme == MIPME_LAGRANGE) {
    o.G = v;
    if constexpr (DERIV) {
      o.alpha = 2.0 * dv;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o.dGdk[c] = 2.0 * o.k[c] * dv;
        o.dGdh[c] = 0.0;
        o.beta[c] = 0.0;
      }
    }
    return;
  }
  // (one sincos per axis when the derivatives are wanted: the double-precision sin / cos of libm are ~250 instructions each, and
  // the cell-gradient sums of the x stage evaluate this for every k-point of the half grid -- most of that kernel's extra 19 us)
  double s = 1.0, t[3], sn[3], cs[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t[c] = 0.5 * o.k[c] * g.h[c];
    if constexpr (DERIV) {
      sincos(t[c], &sn[c], &cs[c]);
    } else {
      sn[c] = sin(t[c]);
      cs[c] = 0.0;
    }
    s *= (t[c] == 0.0) ? 1.0 : sn[c] / t[c];
  }
  double U2 = 1.0;
  const double s2 = s * s;
  for (int i = 0; i < g.order; ++i) U2 *= s2;
  if (U2 == 0.0) {
    o.G = 0.0;
    if constexpr (DERIV) {
      o.alpha = 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) o.dGdk[c] = o.dGdh[c] = o.beta[c] = 0.0;
    }
    return;
  }
  const double inv = 1.0 / U2;
  o.G = v * inv;
  if constexpr (DERIV) {
    o.alpha = 2.0 * dv * inv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double tc = t[c];
      // d/dt ln(sin t / t) = cot t - 1/t
      const double L = fabs(tc) < 1e-4 ? (-tc / 3.0 - tc * tc * tc / 45.0) : (cs[c] / sn[c] - 1.0 / tc);
      const double w = o.G * double(2 * g.order) * L;
      o.dGdk[c] = 2.0 * o.k[c] * dv * inv - w * 0.5 * g.h[c];
      o.dGdh[c] = -w * 0.5 * o.k[c];
      o.beta[c] = 0.5 * w;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kfilter_kernel(KGeom g, KPot kp, T* __restrict__ G) {
  const int64_t Mh = int64_t(g.nx) * g.ny * g.nzh;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= Mh) return;
  const int iz = int(t % g.nzh);
  const int64_t r = t / g.nzh;
  const int iy = int(r % g.ny);
  const int ix = int(r / g.ny);
  KPoint o;
  eval_point<false>(g, kp, ix, iy, iz, o);
  G[t] = T(o.G);
}

// Derivative table of the filter, 4 reals per half-grid point {alpha, beta_x, beta_y, beta_z}:
//   dG/dk_c = alpha k_c - beta_c h_c,   dG/dh_c = -beta_c k_c     (k Cartesian, h_c = |a_c| / n_c; beta = 0 for PME)
// with alpha = 2 v_LR^'(k^2) / U^2 and beta_c = G n (cot t_c - 1/t_c), t_c = k_c h_c / 2.  Built once per (cell, potential) like
// G itself; the x stage of the fused convolution forms the k-grid sums of the cell gradient from it with a dozen FMAs per
// k-point instead of evaluating the influence function's derivatives in double precision for every k-point of every step
// (~700 instructions each: 7.8 -> 15 us at 64^3).
template <typename T>
__global__ __launch_bounds__(256) void kfilter_deriv_kernel(KGeom g, KPot kp, T* __restrict__ D) {
  const int64_t Mh = int64_t(g.nx) * g.ny * g.nzh;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= Mh) return;
  const int iz = int(t % g.nzh);
  const int64_t r = t / g.nzh;
  const int iy = int(r % g.ny);
  const int ix = int(r / g.ny);
  KPoint o;
  eval_point<true>(g, kp, ix, iy, iz, o);
  D[4 * t + 0] = T(o.alpha);
  D[4 * t + 1] = T(o.beta[0]);
  D[4 * t + 2] = T(o.beta[1]);
  D[4 * t + 3] = T(o.beta[2]);
}

// hat_work = hat * G ; dc[c] = Re hat[c, 0]
template <typename T>
__global__ __launch_bounds__(256) void apply_filter_kernel(int64_t Mh, int C, const T* __restrict__ hat,
                                                          const T* __restrict__ G, T* __restrict__ out,
                                                          T* __restrict__ dc) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= Mh) return;
  const T gk = G[t];
  for (int c = 0; c < C; ++c) {
    const int64_t o = 2 * (c * Mh + t);
    const T re = hat[o], im = hat[o + 1];
    out[o] = re * gk;
    out[o + 1] = im * gk;
    if (t == 0 && dc) dc[c] = re;
  }
}

static constexpr int kFinalizeBlocksDecl = 64, kFinalizeNVDecl = 22;  // layout of the finalisation scratch, see below

// Backward variant: also accumulates, per block, the 12 k-grid sums of the cell gradient
//   K[c][d] = 2 pi sum_k dL/dG(k) dG/dk_c f_d ,  H[c] = sum_k dL/dG(k) dG/dh_c ,
//   dL/dG(k) = mu(k) sum_ch Re[rho^_ch(k) conj psi^_ch(k)]          
template <typename T>
__global__ __launch_bounds__(256) void apply_filter_cellgrad_kernel(KGeom g, KPot kp, int C,
                                                                   const T* __restrict__ psi_hat,
                                                                   const T* __restrict__ rho_hat,
                                                                   const T* __restrict__ G, T* __restrict__ out,
                                                                   T* __restrict__ dc, double* __restrict__ partials) {
  const int64_t Mh = int64_t(g.nx) * g.ny * g.nzh;
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  double acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0;
  if (t < Mh) {
    const int iz = int(t % g.nzh);
    const int64_t r = t / g.nzh;
    const int iy = int(r % g.ny);
    const int ix = int(r / g.ny);
    const T gk = G[t];
    double dLdG = 0.0;
    for (int c = 0; c < C; ++c) {
      const int64_t o = 2 * (c * Mh + t);
      const T re = psi_hat[o], im = psi_hat[o + 1];
      if (out) {  // energy mode passes psi_hat = rho_hat and needs only the sums
        out[o] = re * gk;
        out[o + 1] = im * gk;
      }
      if (t == 0 && dc) dc[c] = re;
      dLdG += double(rho_hat[o]) * double(re) + double(rho_hat[o + 1]) * double(im);
    }
    const bool edge = (iz == 0) || ((g.nz % 2 == 0) && (iz == g.nz / 2));
    dLdG *= edge ? 1.0 : 2.0;
    KPoint p;
    eval_point<true>(g, kp, ix, iy, iz, p);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[3 * c + d] = 2.0 * kPi * dLdG * p.dGdk[c] * double(p.f[d]);
      acc[9 + c] = dLdG * p.dGdh[c];
    }
  }
  __shared__ double red[4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double v = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    partials[int64_t(blockIdx.x) * 12 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
  // clear the ticket counter of cellgrad_finalize_kernel (stored behind its block sums, after the k-grid partials)
  if (blockIdx.x == 0 && threadIdx.x == 0)
    *reinterpret_cast<int*>(partials + int64_t(gridDim.x) * 12 + int64_t(kFinalizeBlocksDecl) * kFinalizeNVDecl) = 0;
}

// Single-block finalisation of dL/dcell:
//   D = r^T T + K,  T = grad_pos A^T ;  gA = -Ainv^T D Ainv^T + rows[(H_c/(|a_c| n_c)) a_c] + dL/dV * V * Ainv^T
//   dL/dV = -(1/V) sum_ic g_ic/2 Phi_ic + (2 bg / V) sum_c Q_c dc(psi_c)
static constexpr int kFinalizeBlocks = kFinalizeBlocksDecl;  // partial-sum blocks of the finalisation (+ a ticket counter after them)
static constexpr int kFinalizeNV = kFinalizeNVDecl;          // 12 k-grid sums, 9 r^T grad_pos, 1 sum g*Phi/2

// Multi-block: every block sums its share of the k-grid partials and of the atoms into scratch[b][22]; the block that draws
// the last ticket adds the block sums in index order (deterministic) and does the 3x3 algebra.  `scratch` = the tail of the
// partials buffer: kFinalizeBlocks * 22 doubles + one int ticket counter that is zero on entry (cleared by
// apply_filter_cellgrad_kernel) and is left at zero.
template <typename T>
__global__ __launch_bounds__(256) void cellgrad_finalize_kernel(mipme_mesh_t m, double bg, int64_t n_atoms, int nblocks,
                                                               const double* __restrict__ partials, double* scratch,
                                                               const T* __restrict__ pos,
                                                               const T* __restrict__ grad_pos,
                                                               const T* __restrict__ gout, const T* __restrict__ phi_atoms,
                                                               const T* __restrict__ rho_dc, const T* __restrict__ psi_dc,
                                                               const T* __restrict__ energy_scale,
                                                               T* __restrict__ grad_cell, const T* __restrict__ field,
                                                               const T* __restrict__ q) {
  // grad_pos == NULL (energy mode only): the mesh part of dL/dr_a is energy_scale * q_a * field_a, with the per-atom mesh
  // field the forward gather wrote
  // energy_scale != NULL (energy mode, g = gE * charges): the k-grid partials were formed with psi^ = rho^ and psi_dc is
  // rho_dc; both are multiplied by gE / 2V here
  constexpr int NV = kFinalizeNV;
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  const int64_t t0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  for (int64_t b = t0; b < nblocks; b += stride) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] += partials[b * 12 + i];
  }
  const int C = m.n_channels;
  for (int64_t a = t0; a < n_atoms; a += stride) {
    const double r[3] = {double(pos[3 * a]), double(pos[3 * a + 1]), double(pos[3 * a + 2])};
    double gp[3];
    if (grad_pos) {
      gp[0] = double(grad_pos[3 * a]);
      gp[1] = double(grad_pos[3 * a + 1]);
      gp[2] = double(grad_pos[3 * a + 2]);
    } else {
      const double w = double(energy_scale[0]) * double(q[a]);
      gp[0] = w * double(field[3 * a]);
      gp[1] = w * double(field[3 * a + 1]);
      gp[2] = w * double(field[3 * a + 2]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < 3; ++e) acc[12 + 3 * c + e] += r[c] * gp[e];
    for (int c = 0; c < C; ++c) acc[21] += 0.5 * double(gout[a * C + c]) * double(phi_atoms[a * C + c]);
  }
  __shared__ double red[4][NV];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double v = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  int* counter = reinterpret_cast<int*>(scratch + size_t(kFinalizeBlocks) * NV);
  if (threadIdx.x < NV)
    __hip_atomic_store(&scratch[size_t(blockIdx.x) * NV + threadIdx.x],
                       red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = ticket == int(gridDim.x) - 1;
  }
  __syncthreads();
  if (!last) return;
  __shared__ double s[NV];
  __shared__ double all[kFinalizeBlocks * NV];  // independent loads first (a dependent chain of 64 L2 round trips took 20 us)
  for (int k = threadIdx.x; k < int(gridDim.x) * NV; k += blockDim.x)
    all[k] = __hip_atomic_load(&scratch[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x < NV) {
    double v = 0.0;
    for (int b = 0; b < int(gridDim.x); ++b) v += all[b * NV + threadIdx.x];
    s[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double es = energy_scale ? double(energy_scale[0]) * 0.5 / m.volume : 1.0;
    for (int i = 0; i < 12; ++i) s[i] *= es;
    const double* A = m.cell;
    const double* Ai = m.inv_cell;
    double D[9];
    // R = R' A^T : R[c][d] = sum_e R'[c][e] A[d][e]
    for (int c = 0; c < 3; ++c)
      for (int d = 0; d < 3; ++d) {
        double v = 0.0;
        for (int e = 0; e < 3; ++e) v += s[12 + 3 * c + e] * A[3 * d + e];
        D[3 * c + d] = v + s[3 * c + d];
      }
    double sumQ = 0.0;
    for (int c = 0; c < C; ++c) sumQ += double(rho_dc[c]) * double(psi_dc[c]) * es;
    const double dLdV = -s[21] / m.volume + 2.0 * bg / m.volume * sumQ;
    const int ns[3] = {m.nx, m.ny, m.nz};
    for (int a = 0; a < 3; ++a) {
      const double norm = sqrt(A[3 * a] * A[3 * a] + A[3 * a + 1] * A[3 * a + 1] + A[3 * a + 2] * A[3 * a + 2]);
      for (int b = 0; b < 3; ++b) {
        double v = 0.0;
        for (int c = 0; c < 3; ++c)
          for (int d = 0; d < 3; ++d) v += Ai[3 * c + a] * D[3 * c + d] * Ai[3 * b + d];
        double out = -v + dLdV * m.volume * Ai[3 * b + a];
        out += s[9 + a] / (norm * double(ns[a])) * A[3 * a + b];
        grad_cell[3 * a + b] = T(out);
      }
    }
  }
}

// ---- hipFFT plans ------------------------------------------------------------------------------
}  // namespace mipme

struct mipme_fft_plan {
  hipfftHandle fwd = 0, inv = 0;
  // (y, z) plane transforms for the fused convolution (convolve_xfused): the x direction is done by xconv_kernel
  hipfftHandle fwd2d = 0, inv2d = 0;
  // own (y, z) plane kernels (yz_planes_kernel) instead of the two hipFFT plans above: power-of-two ny, nz whose half-complex
  // plane fits 64 KB of LDS
  bool own_yz = false;
  // planes too large for one workgroup's LDS (256-wide meshes): the same transforms as TWO launches per direction -- z rows
  // (yz_planes_kernel without its y stage, `split_rows` rows per workgroup) and y columns (ycols_kernel) -- still no hipFFT
  bool split_yz = false;
  int split_rows = 0;
  int dtype = 0, nx = 0, ny = 0, nz = 0, batch = 0;
  // per-brick atom counters of the binning pass (csrc/bricks.hip): zero between calls -- the spread kernel that consumes
  // the bins clears them again, which saves a memset launch per evaluation
  int* brick_count = nullptr;
  // scratch of the gather tail (csrc/bricks.hip GatherTail: per-brick energy partials + tickets), zero between calls
  void* tail_scratch = nullptr;
  int64_t tail_bytes = 0;
  // set by the plane spread (bricks.hip plane_spread_yz_body) when it has already written the forward (y,z) transform of the
  // charge mesh into the convolution's half-complex buffer: the next convolve_xfused skips its forward plane launch
  bool forward_done = false;
  // ... as `forward_parts` partial transforms that add up: part 0 in the convolution's buffer, the others in hat_parts
  int forward_parts = 1;
  // ... with the y columns still to do (planes spread in bands of rows: only the z rows are transformed)
  bool forward_ycols = false;
  void* hat_parts = nullptr;
  int64_t hat_parts_bytes = 0;
};

namespace mipme {

static const char* fft_err(hipfftResult r) {
  switch (r) {
    case HIPFFT_SUCCESS: return "HIPFFT_SUCCESS";
    case HIPFFT_INVALID_PLAN: return "HIPFFT_INVALID_PLAN";
    case HIPFFT_ALLOC_FAILED: return "HIPFFT_ALLOC_FAILED";
    case HIPFFT_INVALID_TYPE: return "HIPFFT_INVALID_TYPE";
    case HIPFFT_INVALID_VALUE: return "HIPFFT_INVALID_VALUE";
    case HIPFFT_INTERNAL_ERROR: return "HIPFFT_INTERNAL_ERROR";
    case HIPFFT_EXEC_FAILED: return "HIPFFT_EXEC_FAILED";
    case HIPFFT_SETUP_FAILED: return "HIPFFT_SETUP_FAILED";
    case HIPFFT_INVALID_SIZE: return "HIPFFT_INVALID_SIZE";
    default: return "HIPFFT_<other>";
  }
}

#define MIPME_CHECK_FFT(expr)                                                           \
  do {                                                                                  \
    hipfftResult _r = (expr);                                                           \
    if (_r != HIPFFT_SUCCESS) {                                                         \
      mipme::set_error("%s failed: %s (%s:%d)", #expr, fft_err(_r), __FILE__, __LINE__); \
      return MIPME_EFFT;                                                                \
    }                                                                                   \
  } while (0)

// ---- fused reciprocal-space convolution ------------------------------------------------------------------------
// irfftn(rfftn(rho) * G) with hipFFT doing the (y, z) plane transforms and ONE kernel doing everything along x:
// forward FFT over x, multiplication by G, inverse FFT over x -- per (ky, kz) column these three steps are independent
// of every other column, so they need no pass through memory in between (the 3-D hipFFT plans spend a kernel on each
// and the filter a third).  Decimation-in-frequency forward (natural -> bit-reversed order), pointwise product in
// bit-reversed order, decimation-in-time inverse (bit-reversed -> natural): no reordering pass.  nx = 2^k (the mesh
// sizes of get_ns_mesh are always powers of two, lib/kvectors.py:17-21); other sizes use the 3-D plans.
static bool xfused_dims_ok(int nx) { return nx >= 2 && nx <= 2048 && (nx & (nx - 1)) == 0; }

// (Cplx, the LDS FFT passes and yz_plane_body live in fft_lds.h: bricks.hip uses them too)

// ---- (y, z) plane transforms in LDS ----------------------------------------------------------------------------------
// One workgroup per (channel, x) plane.  The real z rows are transformed as complex rows of half the length (even / odd
// samples as real / imaginary part) plus the usual split step, so the plane lives in LDS as ny x (nz/2 + 1) complex values,
// in place.  Bit reversal is folded into the loads / stores: every 1-D transform is either decimation in time (bit-reversed
// in, natural out) or decimation in frequency (natural in, bit-reversed out), radix 2, one barrier per stage.  Same
// conventions as the hipFFT plans they replace: un-normalised in both directions (kspace_filter.py:169-187), layouts
// (C, nx, ny, nz) real and (C, nx, ny, nz/2 + 1) complex.  Besides being one launch of our own per direction, they keep
// hipFFT out of the hot path (see the plan self-test above for why that matters).
// a single workgroup may use (almost) all 160 KB of a CU's LDS; above 64 KB the kernel needs its dynamic-LDS limit raised
static constexpr size_t kYzMaxLds = 152 * 1024;

static bool own_yz_dims_ok(int dtype, int ny, int nz) {
  const bool pow2 = ny >= 2 && nz >= 4 && (ny & (ny - 1)) == 0 && (nz & (nz - 1)) == 0;
  const size_t cs = dtype == MIPME_F32 ? 8 : 16;
  return pow2 && cs * (size_t(ny) * (nz / 2 + 1) + size_t(ny > nz / 2 ? ny : nz / 2) / 2 + size_t(nz / 2 + 1)) <= kYzMaxLds;
}



// ---- cell-gradient riders of an energy step ------------------------------------------------------------------------------
// Workgroups appended to the launch that FOLLOWS the x stage (the inverse (y,z) planes: 64 workgroups on 256 CUs at 64^3, so the
// riders run on idle CUs and end before the planes do).  Each takes a slice of the half grid and forms the 15 moments of
// w = mu |rho^|^2 (written by the x stage) against the filter's derivative table {alpha, beta_x, beta_y, beta_z},
//     M1[e][d] = sum w alpha f_e f_d  (00 01 02 11 12 22),      M2[c][d] = sum w beta_c f_d,          f = integer frequencies,
// a slice of the pair kernel's per-wave cell sums (9 doubles per wavefront), and a slice of the x stage's energy sums; one row
// of kCellRow doubles per rider: {15 moments, 9 pair sums, E_k}.  cell_tail_finalize_kernel turns the column sums into dE/dcell.
static constexpr int kCellRow = 25;
struct CellRider {
  int n_riders;  // 0: none
  int nx, ny, nzh;
  const void* wbuf;   // (nx, ny, nzh) reals
  const void* dG4;    // (nx, ny, nzh, 4) reals
  const double* cwave;    // nullable
  int n_waves;
  const double* epart_k;  // nullable
  int n_tiles;
  double* rows;  // [n_riders][kCellRow]
  // general adjoint (nullable): also the 12 sums K[c][d], H[c] of this rider's moments, as cellgrad_finalize_kernel reads them
  // (one 12-value row per rider), and its ticket counter cleared
  double* kh_rows;
  int* ticket;
  double inv[9], h[3];
};

template <typename T>
__device__ __forceinline__ void cell_rider_body(const CellRider& r, unsigned rider, int nthr, double* red /* [16][kCellRow] LDS */) {
  typedef T T4v __attribute__((ext_vector_type(4)));
  const T* __restrict__ wbuf = static_cast<const T*>(r.wbuf);
  const T4v* __restrict__ dG4 = static_cast<const T4v*>(r.dG4);
  const int tid = threadIdx.x;
  // (32-bit index arithmetic: the host side refuses half grids of 2^31 points)
  const unsigned Mh = unsigned(r.nx) * unsigned(r.ny) * unsigned(r.nzh);
  const unsigned gid = rider * unsigned(nthr) + unsigned(tid), TT = unsigned(r.n_riders) * unsigned(nthr);
  T m[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) m[i] = T(0);
  constexpr int U = 4;  // points in flight per thread (all loads of a batch issued before the first use)
  for (unsigned i0 = gid; i0 < Mh; i0 += U * TT) {
    T w[U];
    T4v d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = i0 + unsigned(u) * TT;
      const bool ok = i < Mh;
      w[u] = ok ? wbuf[i] : T(0);
      d[u] = dG4[ok ? i : 0u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = i0 + unsigned(u) * TT;
      const unsigned q = i / unsigned(r.nzh), iz = i - q * unsigned(r.nzh);
      const unsigned ix = q / unsigned(r.ny), iy = q - ix * unsigned(r.ny);
      const T fx = T(fft_freq(int(ix), r.nx)), fy = T(fft_freq(int(iy), r.ny)), fz = T(int(iz));
      const T wa = w[u] * d[u].x;
      const T wax = wa * fx, way = wa * fy, waz = wa * fz;
      m[0] += wax * fx; m[1] += wax * fy; m[2] += wax * fz;
      m[3] += way * fy; m[4] += way * fz; m[5] += waz * fz;
      const T wb0 = w[u] * d[u].y, wb1 = w[u] * d[u].z, wb2 = w[u] * d[u].w;
      m[6] += wb0 * fx; m[7] += wb0 * fy; m[8] += wb0 * fz;
      m[9] += wb1 * fx; m[10] += wb1 * fy; m[11] += wb1 * fz;
      m[12] += wb2 * fx; m[13] += wb2 * fy; m[14] += wb2 * fz;
    }
  }
  // the pair kernel's per-wave sums: a wavefront of the rider takes 64 / 9 ... simply one pair-kernel wave per thread and round
  double cp[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) cp[k] = 0.0;
  if (r.cwave)
    for (unsigned i = gid; i < unsigned(r.n_waves); i += TT) {
#pragma unroll
      for (int k = 0; k < 9; ++k) cp[k] += r.cwave[9 * size_t(i) + k];
    }
  double ek = 0.0;
  if (r.epart_k)
    for (unsigned i = gid; i < unsigned(r.n_tiles); i += TT) ek += r.epart_k[i];
  // wave sums in the working precision (DPP + readlane: no LDS traffic), waves added in double
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < kCellRow; ++i) {
    const T mine = i < 15 ? m[i < 15 ? i : 0] : (i < 24 ? T(cp[(i >= 15 && i < 24) ? i - 15 : 0]) : T(ek));
    const T v = wave_sum_dpp(mine);
    if (lane == 0) red[wave * kCellRow + i] = double(v);
  }
  __syncthreads();
  double total = 0.0;
  if (tid < kCellRow) {
    for (int w = 0; w < (nthr + 63) / 64; ++w) total += red[w * kCellRow + tid];
    r.rows[size_t(rider) * kCellRow + tid] = total;
  }
  if (r.kh_rows) {  // uniform
    __syncthreads();
    if (tid < 15) red[tid] = total;
    __syncthreads();
    if (tid < 12) {
      double out;
      if (tid < 9) {
        const int cc = tid / 3, d = tid % 3;
        double a = 0.0;
        for (int e = 0; e < 3; ++e) {
          const int lo = e < d ? e : d, hi = e < d ? d : e;
          const double inv_ce = cc == 0 ? (e == 0 ? r.inv[0] : e == 1 ? r.inv[1] : r.inv[2])
                                        : cc == 1 ? (e == 0 ? r.inv[3] : e == 1 ? r.inv[4] : r.inv[5])
                                                  : (e == 0 ? r.inv[6] : e == 1 ? r.inv[7] : r.inv[8]);
          a += inv_ce * red[lo * 3 - lo * (lo - 1) / 2 + (hi - lo)];
        }
        const double hc = cc == 0 ? r.h[0] : (cc == 1 ? r.h[1] : r.h[2]);
        out = 2.0 * kPi * (2.0 * kPi * a - hc * red[6 + 3 * cc + d]);
      } else {
        const int cc = tid - 9;
        double a = 0.0;
        for (int e = 0; e < 3; ++e) {
          const double inv_ce = cc == 0 ? (e == 0 ? r.inv[0] : e == 1 ? r.inv[1] : r.inv[2])
                                        : cc == 1 ? (e == 0 ? r.inv[3] : e == 1 ? r.inv[4] : r.inv[5])
                                                  : (e == 0 ? r.inv[6] : e == 1 ? r.inv[7] : r.inv[8]);
          a += inv_ce * red[6 + 3 * cc + e];
        }
        out = -2.0 * kPi * a;
      }
      r.kh_rows[size_t(rider) * 12 + tid] = out;
    }
    if (r.ticket && rider == 0 && tid == 0) *r.ticket = 0;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void cell_rider_kernel(CellRider r) {
  __shared__ double red[16 * kCellRow];
  cell_rider_body<T>(r, blockIdx.x, int(blockDim.x), red);
}

template <typename T, bool INVERSE, bool YSTAGE = true>
__global__ __launch_bounds__(1024) void yz_planes_kernel(int ny, int nz, int logny, int loglz, const T* __restrict__ real_in,
                                                       Cplx<T>* __restrict__ hat, T* __restrict__ real_out,
                                                       const int* __restrict__ skip, unsigned n_planes, CellRider rider) {
  MIPME_SKIP_IF_SET(skip);
  extern __shared__ __attribute__((aligned(16))) char smem_yz[];
  if constexpr (INVERSE && YSTAGE) {
    if (blockIdx.x >= n_planes) {  // (uniform; rider.n_riders workgroups behind the planes)
      cell_rider_body<T>(rider, blockIdx.x - n_planes, int(blockDim.x), reinterpret_cast<double*>(smem_yz));
      return;
    }
  }
  yz_plane_body<T, INVERSE, YSTAGE>(ny, nz, logny, loglz, real_in, hat, real_out, blockIdx.x, smem_yz);  // plane = (channel, x)
}

// y columns of the half-complex mesh for planes that do not fit LDS (split_yz): one workgroup transforms the columns
// (x, kz0 .. kz0 + KZ) along y in LDS as tile[y][z] -- forward: decimation in frequency, stored back through the bit reversal;
// inverse: loaded through the bit reversal, decimation in time, conjugate twiddles -- in place, natural order in memory both
// ways, un-normalised.  Segments of KZ complex values (>= 64 B) keep the strided accesses coalesced.
// Workgroups go to the 8 XCDs round robin by their linear index, and each XCD has its own L2.  The tiles of the strided stages
// are segments of KZ complex values of rows whose pitch (nzh elements) is not a multiple of the cache line, so neighbouring
// chunks share lines: give every XCD a CONTIGUOUS range of tiles (b -> the (b / 8)-th tile of XCD b % 8's range) and the
// shared lines are fetched once per L2 instead of once per neighbour.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned n) {
  const unsigned x = b & 7u, per = n >> 3, rem = n & 7u;
  return x * per + (x < rem ? x : rem) + (b >> 3);
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void ycols_kernel(int ny, int nzh, int logny, int kzs, int nchunk, Cplx<T>* __restrict__ hat,
                                                   const int* __restrict__ skip) {
  MIPME_SKIP_IF_SET(skip);
  extern __shared__ __attribute__((aligned(16))) char smem_yc[];
  const int KZ = 1 << kzs, KP = KZ + 1;                 // rows padded by one element: the butterflies stride over y, and an
  Cplx<T>* tile = reinterpret_cast<Cplx<T>*>(smem_yc);  // even row length would put a wave's 64 accesses on the same banks
  Cplx<T>* tw = tile + size_t(ny) * KP;                 // exp(-2 pi i j / ny), j < ny / 2
  const int tid = threadIdx.x, nthr = blockDim.x;
  const unsigned tile_id = xcd_tile(blockIdx.x, gridDim.x);
  const int chunk = tile_id % nchunk;
  const int64_t plane = tile_id / nchunk;  // (channel, x)
  const int kz0 = chunk << kzs, kzn = min(KZ, nzh - kz0);
  for (int j = tid; j < (ny >> 1); j += nthr) unit_root(j, ny, tw[j].re, tw[j].im);
  Cplx<T>* col = hat + plane * int64_t(ny) * nzh + kz0;  // element (y, z): col[y * nzh + z]
  const int n_el = ny << kzs;
  for (int idx = tid; idx < n_el; idx += nthr) {
    const int y = idx >> kzs, z = idx & (KZ - 1);
    const int yt = INVERSE ? int(__brev(unsigned(y)) >> (32 - logny)) : y;
    tile[yt * KP + z] = z < kzn ? col[int64_t(y) * nzh + z] : Cplx<T>{T(0), T(0)};
  }
  __syncthreads();
  lds_fft_radix2<T, INVERSE, INVERSE, MIPME_FFT_BFAST ? 1 : 0>(tile, logny, KZ, 1, KP, tw, ny);  // forward: DIF; inverse: DIT
  for (int idx = tid; idx < n_el; idx += nthr) {
    const int y = idx >> kzs, z = idx & (KZ - 1);
    const int ys = INVERSE ? y : int(__brev(unsigned(y)) >> (32 - logny));
    if (z < kzn) col[int64_t(ys) * nzh + z] = tile[y * KP + z];
  }
}

template <typename T>
static int ycols(mipme_fft_plan* p, hipStream_t st, bool inverse, void* hat) {
  const int nzh = p->nz / 2 + 1;
  int logny = 0;
  while ((1 << logny) < p->ny) ++logny;
  int kzs = sizeof(T) == 4 ? 4 : 3;  // 128-byte segments; tile <= 32 KiB
  while (kzs > 0 && sizeof(Cplx<T>) * (size_t(p->ny) << kzs) > 32768) --kzs;
  const int nchunk = (nzh + (1 << kzs) - 1) >> kzs;
  const size_t lds = sizeof(Cplx<T>) * (size_t(p->ny) * ((size_t(1) << kzs) + 1) + size_t(p->ny / 2));
  const unsigned grid = unsigned(nchunk) * unsigned(p->nx) * unsigned(p->batch);
  if (inverse)
    ycols_kernel<T, true><<<grid, 256, lds, st>>>(p->ny, nzh, logny, kzs, nchunk, (Cplx<T>*)hat, skip_flag_slot());
  else
    ycols_kernel<T, false><<<grid, 256, lds, st>>>(p->ny, nzh, logny, kzs, nchunk, (Cplx<T>*)hat, skip_flag_slot());
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// rows per workgroup of the z-rows launch of split planes: the largest power of two (<= ny) whose tile fits 64 KB of LDS
static int split_rows_for(int dtype, int ny, int nz) {
  const size_t cs = dtype == MIPME_F32 ? 8 : 16;
  const size_t Lz = size_t(nz / 2), RZ = Lz + 1;
  int r = ny;
  while (r > 1 && cs * (size_t(r) * RZ + Lz / 2 + RZ) > 64 * 1024) r >>= 1;
  return cs * (size_t(r) * RZ + Lz / 2 + RZ) <= 64 * 1024 ? r : 0;
}

template <typename T>
static int zrows(mipme_fft_plan* p, hipStream_t st, bool inverse, const void* real_in, void* hat, void* real_out) {
  int loglz = 0;
  while ((1 << loglz) < p->nz / 2) ++loglz;
  const int Lz = p->nz / 2, R = p->split_rows;
  const size_t lds = sizeof(Cplx<T>) * (size_t(R) * (Lz + 1) + size_t(Lz) / 2 + size_t(Lz + 1));
  const unsigned grid = unsigned(int64_t(p->nx) * p->ny * p->batch / R);
  const int work = R * (Lz + 1);
  const int threads = work >= 2048 ? 1024 : (work >= 512 ? 256 : 64);
  if (inverse)
    yz_planes_kernel<T, true, false><<<grid, threads, lds, st>>>(R, p->nz, 0, loglz, nullptr, (Cplx<T>*)hat, (T*)real_out, skip_flag_slot(), grid, CellRider{});
  else
    yz_planes_kernel<T, false, false><<<grid, threads, lds, st>>>(R, p->nz, 0, loglz, (const T*)real_in, (Cplx<T>*)hat, nullptr, skip_flag_slot(), grid, CellRider{});
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int cell_riders_alone(hipStream_t st, const CellRider& r) {
  cell_rider_kernel<T><<<unsigned(r.n_riders), 1024, 0, st>>>(r);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// shape of the single-launch plane kernels (yz_planes_kernel with its y stage)
static void yz_launch_shape(const mipme_fft_plan* p, size_t real_bytes, int& logny, int& loglz, size_t& lds, int& threads) {
  logny = loglz = 0;
  while ((1 << logny) < p->ny) ++logny;
  while ((1 << loglz) < p->nz / 2) ++loglz;
  const int Lz = p->nz / 2, Ltab = p->ny > Lz ? p->ny : Lz;
  lds = 2 * real_bytes * (size_t(p->ny) * (Lz + 1) + size_t(Ltab) / 2 + size_t(Lz + 1));
  const int work = p->ny * (Lz + 1);
  // latency-bound, one workgroup per plane: the widest workgroup wins (1024 threads 24.0 us per convolution at 64^3 fp32,
  // 512 threads 25.6, 256 threads 30.9; the two hipFFT plans it replaces 25.3)
  threads = work >= 2048 ? 1024 : (work >= 512 ? 256 : 64);
}

// rider (inverse only, nullable): the cell-gradient riders of an energy step, appended to the launch where there is ONE launch
// per direction, a launch of their own otherwise
template <typename T>
static int yz_planes(mipme_fft_plan* p, hipStream_t st, bool inverse, const void* real_in, void* hat, void* real_out,
                     const CellRider* rider = nullptr) {
  if (p->split_yz) {  // planes beyond one workgroup's LDS: z rows and y columns as two launches
    int rc;
    if (!inverse) {
      if ((rc = zrows<T>(p, st, false, real_in, hat, nullptr))) return rc;
      return ycols<T>(p, st, false, hat);
    }
    if (rider && rider->n_riders > 0 && (rc = cell_riders_alone<T>(st, *rider))) return rc;
    if ((rc = ycols<T>(p, st, true, hat))) return rc;
    return zrows<T>(p, st, true, nullptr, hat, real_out);
  }
  int logny = 0, loglz = 0, threads = 0;
  size_t lds = 0;
  yz_launch_shape(p, sizeof(T), logny, loglz, lds, threads);
  const unsigned grid = unsigned(p->nx) * unsigned(p->batch);
  const unsigned n_riders = (inverse && rider) ? unsigned(rider->n_riders) : 0u;
  if (n_riders && lds < sizeof(double) * 16 * kCellRow) lds = sizeof(double) * 16 * kCellRow;  // the riders' reduction scratch
  if (lds > 64 * 1024) {  // once per instantiation: allow the large dynamic allocation
    static bool raised[2] = {false, false};
    if (!raised[inverse ? 1 : 0]) {
      const void* fn = inverse ? (const void*)yz_planes_kernel<T, true> : (const void*)yz_planes_kernel<T, false>;
      MIPME_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(kYzMaxLds)));
      raised[inverse ? 1 : 0] = true;
    }
  }
  if (inverse)
    yz_planes_kernel<T, true><<<grid + n_riders, threads, lds, st>>>(p->ny, p->nz, logny, loglz, nullptr, (Cplx<T>*)hat, (T*)real_out,
                                                                     skip_flag_slot(), grid, n_riders ? *rider : CellRider{});
  else
    yz_planes_kernel<T, false><<<grid, threads, lds, st>>>(p->ny, p->nz, logny, loglz, (const T*)real_in, (Cplx<T>*)hat, nullptr,
                                                           skip_flag_slot(), grid, CellRider{});
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// One block: the columns (ky, kz0 .. kz0 + KZ) of channel c, all nx points of each, in LDS as tile[x][z] (KZ = 2^kzs
// columns, padded ones compute on zeros).  Two radix-2 stages are done per pass by the thread that owns the four points
// they couple, which halves the LDS round trips and barriers of the textbook radix-2 schedule without changing its
// (bit-reversed) data order.
// CELLSUMS: also form, per block, the 12 k-grid sums of the cell gradient for the energy mode (psi^ = rho^, i.e.
// dL/dG(k) = mu(k) |rho^(k)|^2 up to the factor gE / 2V applied by cellgrad_finalize_kernel) from the transformed columns
// while they are in LDS -- what apply_filter_cellgrad_kernel computes from a stored rfftn(rho) in the backward pass.
// The body serves one TILE with a GROUP of `nthr` threads (`tid` = index inside the group, `grp` = index of the group in the
// workgroup, `smem_x` = the group's own LDS region): the stand-alone kernel runs one group per workgroup, the persistent
// convolution kernel several side by side.  Every __syncthreads() is workgroup-wide and unconditional, so all groups of a
// workgroup run the same number of stages (same nx); a group without a tile (`active` = false) computes on zeros and touches
// no global memory.  tile_id / n_tiles stand for blockIdx.x / gridDim.x of the stand-alone launch.
#ifndef MIPME_X_PAD
#define MIPME_X_PAD 1
#endif
static constexpr int kXPad = MIPME_X_PAD;  // elements of padding per LDS row of the x stage
// CELLSUMS: 0 = none; 1 = the 12 k-grid sums of the cell gradient with the filter's derivatives evaluated in place (double
// precision, eval_point<true>), one 12-value row per tile (general autograd nodes).
// The ENERGY STEP's cell gradient takes nothing but one store per k-point from this kernel: wbuf (nullable) receives
// w = mu |rho^|^2 on the half grid, and the sums against the filter's derivative table are formed by rider workgroups of the
// NEXT launch (cell_rider_body below) -- the x stage is one tile per CU at 64^3, a chain of latencies, and whatever is put on
// that chain is paid in full (sums formed here, in the middle or behind the stores, cost 7 us of a 7.8 us kernel).
// rho_hat_out (nullable): the tile before the product, i.e. rfftn(mesh_in) in its natural (kx, ky, kz) layout -- kept by a forward
// pass whose backward may need it; rho_hat_in (nullable, backward pass): w = mu Re[rho^ conj psi^] instead (the general adjoint's
// dL/dG), rho^ prefetched with the tile.
struct XCellExtra {
  void* wbuf;
  void* rho_hat_out;
  const void* rho_hat_in;
  // the forward (y,z) transforms of `n_more` MORE partial charge meshes (plane spread with several workgroups per plane: each
  // transforms its own partial plane, and the transform is linear), `more_stride` complex values apart: added on load
  const void* hat_more = nullptr;
  int n_more = 0;
  int64_t more_stride = 0;
};
template <typename T, int CELLSUMS>
__device__ __forceinline__ void xconv_tile_body(int nx, int ny, int nzh, int log2nx, int kzs, int nchunk,
                                                Cplx<T>* __restrict__ hat, const T* __restrict__ G, int64_t G_stride,
                                                T* __restrict__ dc, const KGeom& kg, const KPot& kp,
                                                double* __restrict__ partials, double* __restrict__ epart,
                                                const double* __restrict__ sr_part, int n_sr_part, unsigned tile_id,
                                                unsigned n_tiles, bool active, int tid, int nthr, int grp, char* smem_x,
                                                const XCellExtra& xc = XCellExtra{nullptr, nullptr, nullptr}) {
  // rows padded by one element (as in the y-column stage): the passes give consecutive lanes consecutive groups of x, i.e. a
  // stride of whole rows -- with KZ = 8 complex floats (64 bytes) per row that is 4 distinct bank groups for 32 lanes
  const int KZ = 1 << kzs, KP = KZ + kXPad;
  Cplx<T>* tile = reinterpret_cast<Cplx<T>*>(smem_x);  // [nx][KP]
  Cplx<T>* tw = tile + size_t(nx) * KP;                 // [nx/2]: exp(-2 pi i j / nx)
  const int chunk = tile_id % nchunk;
  const int ky = (tile_id / nchunk) % ny;
  const int c = tile_id / (nchunk * ny);
  const int kz0 = chunk << kzs;
  const int kzn = min(KZ, nzh - kz0);
  const int half_n = nx >> 1;
  // sr_part (nullable, single batch entry): pre-reduce this block's slice of the pair kernel's per-wave energy partial sums
  // {sum q V_sr, sum q^2} (rows_body.h FusedRowsArgs::epart) into epart[gridDim.x + 2 b ...] -- they were written two
  // launches ago, and the gather's tail then adds up gridDim.x pairs instead of N / 4
  double sr0 = 0.0, sr1 = 0.0;
  if (sr_part && active && tid < 64) {
    const int per = (n_sr_part + int(n_tiles) - 1) / int(n_tiles);
    const int lo = int(tile_id) * per, hi = min(lo + per, n_sr_part);
    for (int i = lo + tid; i < hi; i += 64) {
      sr0 += sr_part[2 * i];
      sr1 += sr_part[2 * i + 1];
    }
  }
  Cplx<T>* col = hat + (int64_t(c) * nx * ny + ky) * nzh + kz0;  // element (x, z): col[x * ny * nzh + z]
  const int64_t xs = int64_t(ny) * nzh;
  const int n_el = nx << kzs;
  // The tile's loads are issued FIRST, into registers, and the twiddle table is computed while they are in flight: at 64^3 the
  // kernel is one tile per CU, a chain of latencies in which the sincos sequence used to sit in front of loads that the plane
  // kernel has just written on other XCDs (9.3 -> 8.6 us; the same reordering in the plane kernels changed nothing).
  constexpr int kGPrefetch = 8;
  const bool g_prefetched = n_el <= kGPrefetch * nthr;
  Cplx<T> tpre[kGPrefetch];
  if (g_prefetched) {
#pragma unroll
    for (int u = 0; u < kGPrefetch; ++u) {
      const int idx = tid + u * nthr;
      const int x = idx >> kzs, z = idx & (KZ - 1);
      tpre[u] = (active && idx < n_el && z < kzn) ? col[x * xs + z] : Cplx<T>{T(0), T(0)};
    }
  }
  // partial transforms of a plane spread with several workgroups per plane: the first extra part travels in registers of its own
  // (all loads of the tile in flight together: one round trip, not one per part), further parts are added one after the other
  Cplx<T> tpre2[kGPrefetch];
  if (g_prefetched && xc.n_more > 0) {  // uniform
    const Cplx<T>* more = static_cast<const Cplx<T>*>(xc.hat_more) + (col - hat);
#pragma unroll
    for (int u = 0; u < kGPrefetch; ++u) {
      const int idx = tid + u * nthr;
      const int x = idx >> kzs, z = idx & (KZ - 1);
      tpre2[u] = (active && idx < n_el && z < kzn) ? more[x * xs + z] : Cplx<T>{T(0), T(0)};
    }
    more += xc.more_stride;
    for (int p = 1; p < xc.n_more; ++p, more += xc.more_stride) {
#pragma unroll
      for (int u = 0; u < kGPrefetch; ++u) {
        const int idx = tid + u * nthr;
        const int x = idx >> kzs, z = idx & (KZ - 1);
        if (active && idx < n_el && z < kzn) tpre2[u] = cadd(tpre2[u], more[x * xs + z]);
      }
    }
  }
  for (int j = tid; j < half_n; j += nthr) unit_root(j, nx, tw[j].re, tw[j].im);
  if (g_prefetched) {
#pragma unroll
    for (int u = 0; u < kGPrefetch; ++u) {
      const int idx = tid + u * nthr;
      if (idx < n_el) tile[(idx >> kzs) * KP + (idx & (KZ - 1))] = xc.n_more > 0 ? cadd(tpre[u], tpre2[u]) : tpre[u];
    }
  } else {
    for (int idx = tid; idx < n_el; idx += nthr) {
      const int x = idx >> kzs, z = idx & (KZ - 1);
      Cplx<T> v = (active && z < kzn) ? col[x * xs + z] : Cplx<T>{T(0), T(0)};
      if (xc.n_more > 0 && active && z < kzn) {
        const Cplx<T>* more = static_cast<const Cplx<T>*>(xc.hat_more) + (col - hat);
        for (int p = 0; p < xc.n_more; ++p, more += xc.more_stride) v = cadd(v, more[x * xs + z]);
      }
      tile[x * KP + z] = v;
    }
  }
  // the filter values this thread will need after the forward transform: fetched now, behind the tile's own loads, instead of
  // after it (one more exposed round trip in a kernel that is one tile per CU at 64^3: a chain of latencies)
  T gpre[kGPrefetch];
  if (g_prefetched) {
#pragma unroll
    for (int u = 0; u < kGPrefetch; ++u) {
      const int idx = tid + u * nthr;
      const int x = idx >> kzs, z = idx & (KZ - 1);
      gpre[u] = T(0);
      if (idx < n_el && z < kzn) {
        const int kx = int(__brev(unsigned(x)) >> (32 - log2nx));
        gpre[u] = G[c * G_stride + (int64_t(kx) * ny + ky) * nzh + kz0 + z];
      }
    }
  }
  Cplx<T> rpre[kGPrefetch];
  const Cplx<T>* __restrict__ rho_in = static_cast<const Cplx<T>*>(xc.rho_hat_in);
  if (rho_in && g_prefetched) {  // uniform
#pragma unroll
    for (int u = 0; u < kGPrefetch; ++u) {
      const int idx = tid + u * nthr;
      const int x = idx >> kzs, z = idx & (KZ - 1);
      rpre[u] = Cplx<T>{T(0), T(0)};
      if (idx < n_el && z < kzn) {
        const int kx = int(__brev(unsigned(x)) >> (32 - log2nx));
        rpre[u] = rho_in[(int64_t(c) * nx + kx) * ny * nzh + int64_t(ky) * nzh + kz0 + z];
      }
    }
  }
  __syncthreads();
  // ---- forward, decimation in frequency (natural in, bit-reversed out): KZ sequences of length nx, element x of column z at
  //      tile[x * KP + z] ----
  lds_fft_radix2<T, false, false, MIPME_FFT_BFAST ? 1 : 0>(tile, log2nx, KZ, 1, KP, tw, nx, tid, nthr);
  if (dc && active && ky == 0 && kz0 == 0 && tid == 0) dc[c] = tile[0].re;  // k = 0 (bit reversal maps 0 to 0)
  if constexpr (CELLSUMS == 1) {
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0;
    for (int idx = tid; idx < n_el; idx += nthr) {
      const int x = idx >> kzs, z = idx & (KZ - 1);
      if (z < kzn) {
        const int kx = int(__brev(unsigned(x)) >> (32 - log2nx));
        const int iz = kz0 + z;
        const Cplx<T> v = tile[x * KP + z];
        const bool edge = (iz == 0) || ((kg.nz % 2 == 0) && (iz == kg.nz / 2));
        const double dLdG = (double(v.re) * double(v.re) + double(v.im) * double(v.im)) * (edge ? 1.0 : 2.0);
        KPoint p;
        eval_point<true>(kg, kp, kx, ky, iz, p);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
          for (int d = 0; d < 3; ++d) acc[3 * a + d] += 2.0 * kPi * dLdG * p.dGdk[a] * double(p.f[d]);
          acc[9 + a] += dLdG * p.dGdh[a];
        }
      }
    }
    __shared__ double red[8][12];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      double v = acc[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < 12) {
      double v = 0.0;
      for (int w = 0; w < (nthr + 63) / 64; ++w) v += red[w][tid];
      partials[int64_t(tile_id) * 12 + tid] = v;
    }
    // the ticket counter of cellgrad_finalize_kernel lives behind its block sums, after the k-grid partials
    if (tile_id == 0 && tid == 0)
      *reinterpret_cast<int*>(partials + int64_t(n_tiles) * 12 + int64_t(kFinalizeBlocksDecl) * kFinalizeNVDecl) = 0;
  }
  // ---- product with G: position x holds kx = bitrev(x) ----
  // epart (nullable): this block's share of sum_k mu_k G_k |rho^_k|^2 (mu = 2 except on the kz = 0 and kz = nz/2 planes of
  // the half grid) -- the mesh part of the energy, assembled by the gather's tail (bricks.hip GatherTail)
  double esum = 0.0;
  const int nz_full = 2 * (nzh - 1);
  int u_pre = 0;
  for (int idx = tid; idx < n_el; idx += nthr, ++u_pre) {
    const int x = idx >> kzs, z = idx & (KZ - 1);
    if (z < kzn) {
      const int kx = int(__brev(unsigned(x)) >> (32 - log2nx));
      T gk;  // G_stride: one filter table per batch entry, or 0
      if (g_prefetched) {
        gk = T(0);
#pragma unroll
        for (int u = 0; u < kGPrefetch; ++u) gk = u == u_pre ? gpre[u] : gk;
      } else {
        gk = G[c * G_stride + (int64_t(kx) * ny + ky) * nzh + kz0 + z];
      }
      Cplx<T> v = tile[x * KP + z];
      if (epart) {
        const int iz = kz0 + z;
        const bool edge = iz == 0 || iz == nz_full / 2;
        esum += (edge ? 1.0 : 2.0) * double(gk) * (double(v.re) * double(v.re) + double(v.im) * double(v.im));
      }
      if (xc.wbuf || xc.rho_hat_out) {  // uniform
        const int iz = kz0 + z;
        const int64_t at = (int64_t(c) * nx + kx) * ny * nzh + int64_t(ky) * nzh + iz;
        if (xc.rho_hat_out) static_cast<Cplx<T>*>(xc.rho_hat_out)[at] = v;
        if (xc.wbuf) {
          const bool edge = iz == 0 || iz == nz_full / 2;
          Cplx<T> r = v;
          if (rho_in) {
            if (g_prefetched) {
              r = Cplx<T>{T(0), T(0)};
#pragma unroll
              for (int u = 0; u < kGPrefetch; ++u) r = u == u_pre ? rpre[u] : r;
            } else {
              r = rho_in[at];
            }
          }
          static_cast<T*>(xc.wbuf)[at] = (edge ? T(1) : T(2)) * (r.re * v.re + r.im * v.im);
        }
      }
      v.re *= gk;
      v.im *= gk;
      tile[x * KP + z] = v;
    }
  }
  if (epart) {  // uniform
    __shared__ double ered[16][8];  // [group][wave of the group]
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) esum += __shfl_xor(esum, off, 64);
    if (lane == 0) ered[grp][wave] = esum;
    __syncthreads();
    if (tid == 0 && active) {
      double v = 0.0;
      for (int w = 0; w < (nthr + 63) / 64; ++w) v += ered[grp][w];
      epart[tile_id] = v;
    }
    if (sr_part && tid < 64) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        sr0 += __shfl_xor(sr0, off, 64);
        sr1 += __shfl_xor(sr1, off, 64);
      }
      if (tid == 0 && active) {
        epart[int64_t(n_tiles) + 2 * int64_t(tile_id)] = sr0;
        epart[int64_t(n_tiles) + 2 * int64_t(tile_id) + 1] = sr1;
      }
    }
  }
  __syncthreads();
  // ---- inverse, decimation in time (bit-reversed in, natural out); conjugate twiddles, no normalisation
  //      (kspace_filter.py:169-187: norm="backward" forward, norm="forward" inverse) ----
  lds_fft_radix2<T, true, true, MIPME_FFT_BFAST ? 1 : 0>(tile, log2nx, KZ, 1, KP, tw, nx, tid, nthr);
  for (int idx = tid; idx < n_el; idx += nthr) {
    const int x = idx >> kzs, z = idx & (KZ - 1);
    if (active && z < kzn) col[x * xs + z] = tile[x * KP + z];
  }
}

template <typename T, int CELLSUMS>
__global__ __launch_bounds__(256) void xconv_kernel(int nx, int ny, int nzh, int log2nx, int kzs, int nchunk,
                                                   Cplx<T>* __restrict__ hat, const T* __restrict__ G, int64_t G_stride,
                                                   T* __restrict__ dc, KGeom kg, KPot kp, double* __restrict__ partials,
                                                   double* __restrict__ epart, const double* __restrict__ sr_part,
                                                   int n_sr_part, const int* __restrict__ skip, XCellExtra xc) {
  MIPME_SKIP_IF_SET(skip);
  extern __shared__ __attribute__((aligned(16))) char smem_x[];
  xconv_tile_body<T, CELLSUMS>(nx, ny, nzh, log2nx, kzs, nchunk, hat, G, G_stride, dc, kg, kp, partials, epart, sr_part,
                               n_sr_part, xcd_tile(blockIdx.x, gridDim.x), gridDim.x, true, int(threadIdx.x), int(blockDim.x), 0, smem_x,
                               xc);
}

bool fft_plan_xfused(const mipme_fft_plan* p) { return p->own_yz || (p->fwd2d != 0 && p->inv2d != 0); }
int fft_plan_batch(const mipme_fft_plan* p) { return p->batch; }

// mesh_in (C,nx,ny,nz) -> mesh_out, hat: one half-complex work buffer; dc[c] = Re rfftn(mesh_in)[c,0,0,0] (nullable)
// blocks of the x stage (= number of 12-value partial sums it writes when asked for the cell sums)
int64_t xconv_blocks(const mipme_fft_plan* p) {
  const int nzh = p->nz / 2 + 1;
  const size_t cs = p->dtype == MIPME_F32 ? 8 : 16;
  int kzs = p->dtype == MIPME_F32 ? 3 : 2;
  while (kzs > 0 && cs * (size_t(p->nx) << kzs) > 32768) --kzs;
  const int KZ = 1 << kzs;
  return int64_t((nzh + KZ - 1) / KZ) * p->ny * p->batch;
}

// wavefronts per tile of the x stage in its default shape (rows per tile of the energy step's cell sums, XCellExtra)
int64_t xconv_waves(const mipme_fft_plan* p) {
  const size_t cs = p->dtype == MIPME_F32 ? 8 : 16;
  int kzs = p->dtype == MIPME_F32 ? 3 : 2;
  while (kzs > 0 && cs * (size_t(p->nx) << kzs) > 32768) --kzs;
  int threads = (p->nx >> 2) << kzs;
  threads = threads < 64 ? 64 : (threads > 256 ? 256 : threads);
  return (threads + 63) / 64;
}

// cell_mesh + cell_pot + cell_partials (all nullable together): also write the energy-mode k-grid sums of the cell gradient
template <typename T>
static int convolve_xfused_t(mipme_fft_plan* p, hipStream_t st, const void* mesh_in, const void* G, void* hat, void* mesh_out,
                             void* dc, int64_t G_stride, const mipme_mesh_t* cell_mesh, const mipme_potential_t* cell_pot,
                             void* cell_partials, void* epart, const void* sr_part, int64_t n_sr_part, const ConvCell* cc) {
  const int nzh = p->nz / 2 + 1;
  int log2nx = 0;
  while ((1 << log2nx) < p->nx) ++log2nx;
  const size_t cs = sizeof(Cplx<T>);
  // columns per block: a power of two giving >= 64-byte segments (8 complex floats / 4 complex doubles), tile <= 32 KiB
  int kzs = sizeof(T) == 4 ? 3 : 2;
  while (kzs > 0 && cs * (size_t(p->nx) << kzs) > 32768) --kzs;
  const int KZ = 1 << kzs;
  const int nchunk = (nzh + KZ - 1) / KZ;
  const unsigned grid = unsigned(nchunk) * unsigned(p->ny) * unsigned(p->batch);
  const size_t lds = cs * (size_t(p->nx) * (size_t(KZ) + kXPad) + size_t(p->nx / 2));
  int threads = (p->nx >> 2) << kzs;  // one 4-point group per thread and pass
  threads = threads < 64 ? 64 : (threads > 256 ? 256 : threads);
  // with the cell sums evaluated in place every element of the tile costs ~700 double-precision instructions
  // (eval_point<true>): at 64^3 that was four elements per thread in ONE wave per SIMD, a serial stream with nothing to overlap
  // its latencies -- 256 threads (0.1217 -> 0.1170 ms for energy + forces + dE/dcell as a graph; 512 threads with launch bounds
  // to match: 0.1252).  The energy step (cc) leaves them to the riders of the next launch.
  if (cell_partials) threads = 256;
  KGeom kg{};
  KPot kp{};
  if (cell_partials) {
    MIPME_REQUIRE(cell_mesh && cell_pot, "the cell sums of the x stage need the mesh and potential descriptors");
    int rc = make_kpot(cell_pot, kp);
    if (rc) return rc;
    kg = make_kgeom(cell_mesh);
  }
  XCellExtra xc{cc ? cc->wbuf : nullptr, cc ? cc->rho_hat_out : nullptr, cc ? cc->rho_hat_in : nullptr};
  if (p->forward_done && p->forward_parts > 1) {
    xc.hat_more = p->hat_parts;
    xc.n_more = p->forward_parts - 1;
    xc.more_stride = int64_t(p->nx) * p->ny * nzh * p->batch;
  }
  CellRider rider{};
  const bool riders = cc && cc->n_riders > 0;
  if (riders) {
    MIPME_REQUIRE(p->batch == 1 && cc->G_deriv && cc->wbuf && cc->rows,
                  "the cell riders serve a single mesh and need the derivative table and their buffers");
    MIPME_REQUIRE(int64_t(p->nx) * p->ny * nzh < (int64_t(1) << 31), "mesh too large for the cell riders' 32-bit indices");
    if (cc->kh_rows) {
      MIPME_REQUIRE(cell_mesh, "the K / H sums of the riders need the mesh descriptor");
      const KGeom kg2 = make_kgeom(cell_mesh);
      for (int i = 0; i < 9; ++i) rider.inv[i] = kg2.inv[i];
      for (int i = 0; i < 3; ++i) rider.h[i] = kg2.h[i];
      rider.kh_rows = cc->kh_rows;
      rider.ticket = cc->ticket;
    }
    rider.n_riders = cc->n_riders;
    rider.nx = p->nx;
    rider.ny = p->ny;
    rider.nzh = nzh;
    rider.wbuf = cc->wbuf;
    rider.dG4 = cc->G_deriv;
    rider.cwave = cc->cwave;
    rider.n_waves = int(cc->n_waves);
    rider.epart_k = (const double*)epart;
    rider.n_tiles = int(grid);
    rider.rows = cc->rows;
  }
  const bool forward_done = p->forward_done, forward_ycols = p->forward_ycols;
  p->forward_done = p->forward_ycols = false;
  if (forward_done) {
    // the plane spread left the transformed planes in `hat` -- or, spread in bands, their z rows: the y columns in place
    if (forward_ycols) {
      int rc = ycols<T>(p, st, false, hat);
      if (rc) return rc;
    }
  } else if (p->own_yz) {
    int rc = yz_planes<T>(p, st, false, mesh_in, hat, nullptr);
    if (rc) return rc;
  } else if (sizeof(T) == 4) {
    MIPME_CHECK_FFT(hipfftExecR2C(p->fwd2d, (hipfftReal*)mesh_in, (hipfftComplex*)hat));
  } else {
    MIPME_CHECK_FFT(hipfftExecD2Z(p->fwd2d, (hipfftDoubleReal*)mesh_in, (hipfftDoubleComplex*)hat));
  }
#define MIPME_XCONV_LAUNCH(MODE)                                                                                              \
  xconv_kernel<T, MODE><<<grid, threads, lds, st>>>(p->nx, p->ny, nzh, log2nx, kzs, nchunk, (Cplx<T>*)hat, (const T*)G,      \
                                                    G_stride, (T*)dc, kg, kp, (double*)cell_partials, (double*)epart,        \
                                                    (const double*)sr_part, int(n_sr_part), skip_flag_slot(), xc)
  if (cell_partials)
    MIPME_XCONV_LAUNCH(1);
  else
    MIPME_XCONV_LAUNCH(0);
#undef MIPME_XCONV_LAUNCH
  MIPME_LAUNCH_CHECK();
  if (p->own_yz) {
    int rc = yz_planes<T>(p, st, true, nullptr, hat, mesh_out, riders ? &rider : nullptr);
    if (rc) return rc;
  } else {
    if (riders) {
      int rc = cell_riders_alone<T>(st, rider);
      if (rc) return rc;
    }
    if (sizeof(T) == 4)
      MIPME_CHECK_FFT(hipfftExecC2R(p->inv2d, (hipfftComplex*)hat, (hipfftReal*)mesh_out));
    else
      MIPME_CHECK_FFT(hipfftExecZ2D(p->inv2d, (hipfftDoubleComplex*)hat, (hipfftDoubleReal*)mesh_out));
  }
  return MIPME_OK;
}

int convolve_xfused(mipme_fft_plan* p, hipStream_t st, const void* mesh_in, const void* G, void* hat, void* mesh_out,
                    void* dc, int64_t G_stride, const mipme_mesh_t* cell_mesh, const mipme_potential_t* cell_pot,
                    void* cell_partials, void* epart, const void* sr_part, int64_t n_sr_part, const RowRideHost* rh,
                    void* err_flag, const ConvCell* cc) {
  MIPME_REQUIRE(!rh || rh->n_rows <= 0, "row riders on the convolution launches were an experiment of round 2 (tools/r06/pruned_experiments.patch)");
  if (!p->own_yz) {
    MIPME_CHECK_FFT(hipfftSetStream(p->fwd2d, st));
    MIPME_CHECK_FFT(hipfftSetStream(p->inv2d, st));
  }
  if (p->dtype == MIPME_F32)
    return convolve_xfused_t<float>(p, st, mesh_in, G, hat, mesh_out, dc, G_stride, cell_mesh, cell_pot, cell_partials, epart,
                                    sr_part, n_sr_part, cc);
  return convolve_xfused_t<double>(p, st, mesh_in, G, hat, mesh_out, dc, G_stride, cell_mesh, cell_pot, cell_partials, epart,
                                   sr_part, n_sr_part, cc);
}

int fft_forward(mipme_fft_plan* p, hipStream_t st, const void* in, void* out);
int fft_inverse(mipme_fft_plan* p, hipStream_t st, void* in, void* out);

// the plane spread can stand in for the forward (y,z) launch of convolve_xfused_t: own single-launch plane kernels, one mesh
bool fft_plan_plane_forward_ok(const mipme_fft_plan* p) {
  return p && p->own_yz && !p->split_yz && p->batch == 1;
}
// ... the same for a batched plan (frame batches: bricks.hip frames_plane_rows_kernel)
bool fft_plan_plane_forward_ok_batched(const mipme_fft_plan* p) { return p && p->own_yz && !p->split_yz; }
void fft_plan_set_forward_done(mipme_fft_plan* p, bool done, int parts) {
  if (!p) return;
  p->forward_done = done;
  p->forward_parts = done ? parts : 1;
  p->forward_ycols = false;
}
void fft_plan_set_forward_ycols(mipme_fft_plan* p, bool pending) {
  if (p) p->forward_ycols = pending && p->forward_done;
}
// room for `n_more` more half-complex meshes (the partial transforms of a plane spread with several workgroups per plane);
// allocated on first use -- not possible during stream capture: NULL then (the caller runs with one part)
void* fft_plan_hat_parts(mipme_fft_plan* p, hipStream_t st, int n_more) {
  const int64_t bytes = int64_t(n_more) * p->nx * p->ny * (p->nz / 2 + 1) * p->batch * (p->dtype == MIPME_F32 ? 8 : 16);
  if (p->hat_parts) return p->hat_parts_bytes >= bytes ? p->hat_parts : nullptr;  // (never re-allocated: graphs hold the pointer)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  void* buf = nullptr;
  if (hipMalloc(&buf, size_t(bytes)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  p->hat_parts = buf;
  p->hat_parts_bytes = bytes;
  return buf;
}

// ---- plan self-test ------------------------------------------------------------------------------------------------
// hipFFT plans created AFTER another user of hipFFT in the same process has run certain transforms can return wrong
// results without any error (observed with PyTorch 2.10+rocm7.0: after torch.fft.rfftn of a 64^3 fp64 tensor, the C2R
// transforms of a (32,32,128) fp64 plan created later are wrong at every odd z index; tools/fft_interference_probe.py).
// A plan is therefore checked once, when it is created: irfftn(rfftn(x)) must return M x for a pseudo-random x.
template <typename T>
__global__ void selftest_fill_kernel(int64_t n, T* __restrict__ x) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = unsigned(i) * 2654435761u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  x[i] = T(double(h & 0xffffff) / double(0x1000000) - 0.5);
}

template <typename T>
__global__ void selftest_check_kernel(int64_t n, const T* __restrict__ got, double scale, double* __restrict__ max_err) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  double e = 0.0;
  if (i < n) {
    unsigned h = unsigned(i) * 2654435761u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    const double want = double(T(double(h & 0xffffff) / double(0x1000000) - 0.5));
    e = fabs(double(got[i]) / scale - want);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) e = fmax(e, __shfl_xor(e, off, 64));
  if ((threadIdx.x & 63) == 0 && e > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(max_err), __double_as_longlong(e));
}

template <typename T>
static int plan_selftest_t(mipme_fft_plan* p) {
  const int64_t M = int64_t(p->nx) * p->ny * p->nz, Mh = int64_t(p->nx) * p->ny * (p->nz / 2 + 1);
  const int64_t nr = M * p->batch, nc = 2 * Mh * p->batch;
  T *x = nullptr, *y = nullptr, *hat = nullptr;
  double* err = nullptr;
  int rc = MIPME_OK;
  double host_err[2] = {0.0, 0.0};
  if (hipMalloc((void**)&x, sizeof(T) * nr) != hipSuccess || hipMalloc((void**)&y, sizeof(T) * nr) != hipSuccess ||
      hipMalloc((void**)&hat, sizeof(T) * nc) != hipSuccess || hipMalloc((void**)&err, 2 * sizeof(double)) != hipSuccess ||
      hipMemset(err, 0, 2 * sizeof(double)) != hipSuccess) {
    set_error("could not allocate the scratch of the FFT plan self-test");
    (void)hipGetLastError();
    rc = MIPME_EHIP;
  }
  const unsigned blocks = unsigned((nr + 255) / 256);
  hipStream_t st = nullptr;
  if (!rc) selftest_fill_kernel<T><<<blocks, 256, 0, st>>>(nr, x);
  if (!rc && p->fwd && p->inv) {
    rc = fft_forward(p, st, x, hat);
    if (!rc) rc = fft_inverse(p, st, hat, y);
    if (!rc) selftest_check_kernel<T><<<blocks, 256, 0, st>>>(nr, y, double(M), err);
  }
  if (!rc && p->own_yz) {  // our own (y, z) plane kernels: same identity
    rc = yz_planes<T>(p, st, false, x, hat, nullptr);
    if (!rc) rc = yz_planes<T>(p, st, true, nullptr, hat, y);
    if (!rc) selftest_check_kernel<T><<<blocks, 256, 0, st>>>(nr, y, double(p->ny) * p->nz, err + 1);
  }
  if (!rc && p->fwd2d && p->inv2d) {
    hipfftResult r1 = hipfftSetStream(p->fwd2d, st), r2 = hipfftSetStream(p->inv2d, st);
    if (r1 == HIPFFT_SUCCESS && r2 == HIPFFT_SUCCESS) {
      if constexpr (sizeof(T) == 4) {
        r1 = hipfftExecR2C(p->fwd2d, (hipfftReal*)x, (hipfftComplex*)hat);
        r2 = hipfftExecC2R(p->inv2d, (hipfftComplex*)hat, (hipfftReal*)y);
      } else {
        r1 = hipfftExecD2Z(p->fwd2d, (hipfftDoubleReal*)x, (hipfftDoubleComplex*)hat);
        r2 = hipfftExecZ2D(p->inv2d, (hipfftDoubleComplex*)hat, (hipfftDoubleReal*)y);
      }
    }
    if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) {
      set_error("hipFFT failed in the plan self-test");
      rc = MIPME_EFFT;
    } else {
      selftest_check_kernel<T><<<blocks, 256, 0, st>>>(nr, y, double(p->ny) * p->nz, err + 1);
    }
  }
  if (!rc && hipMemcpy(host_err, err, 2 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
    set_error("could not read back the FFT plan self-test");
    (void)hipGetLastError();
    rc = MIPME_EHIP;
  }
  (void)hipFree(x);
  (void)hipFree(y);
  (void)hipFree(hat);
  (void)hipFree(err);
  if (rc) return rc;
  const double tol = sizeof(T) == 4 ? 1e-4 : 1e-10;
  if (!(host_err[0] < tol) || !(host_err[1] < tol)) {
    set_error("hipFFT returns wrong results for the %d x %d x %d (x%d) %s plan: irfftn(rfftn(x)) differs from x by %.3g (3-D) / "
              "%.3g (2-D planes).  Known cause: another hipFFT user in the process (e.g. torch.fft) ran transforms whose "
              "kernels this plan now shares; create the calculators / run one evaluation before such calls, or set "
              "MIPME_FFT_SELFTEST=0 to skip this check",
              p->nx, p->ny, p->nz, p->batch, sizeof(T) == 4 ? "fp32" : "fp64", host_err[0], host_err[1]);
    return MIPME_EFFT;
  }
  return MIPME_OK;
}

static int plan_selftest(mipme_fft_plan* p) {
  const char* e = getenv("MIPME_FFT_SELFTEST");
  if (e && e[0] == '0') return MIPME_OK;
  return p->dtype == MIPME_F32 ? plan_selftest_t<float>(p) : plan_selftest_t<double>(p);
}

static hipfftResult create_3d_plans(mipme_fft_plan* p) {
  int n[3] = {p->nx, p->ny, p->nz};
  int rembed[3] = {p->nx, p->ny, p->nz};
  int cembed[3] = {p->nx, p->ny, p->nz / 2 + 1};
  const int rdist = p->nx * p->ny * p->nz, cdist = p->nx * p->ny * (p->nz / 2 + 1);
  hipfftResult r = hipfftPlanMany(&p->fwd, 3, n, rembed, 1, rdist, cembed, 1, cdist,
                                  p->dtype == MIPME_F32 ? HIPFFT_R2C : HIPFFT_D2Z, p->batch);
  if (r == HIPFFT_SUCCESS)
    r = hipfftPlanMany(&p->inv, 3, n, cembed, 1, cdist, rembed, 1, rdist, p->dtype == MIPME_F32 ? HIPFFT_C2R : HIPFFT_Z2D,
                       p->batch);
  return r;
}

// 3-D plans on first use (plans with own_yz): created and self-tested here -- not possible during stream capture
static int ensure_3d_plans(mipme_fft_plan* p) {
  if (p->fwd && p->inv) return MIPME_OK;
  const hipfftResult r = create_3d_plans(p);
  if (r != HIPFFT_SUCCESS) {
    set_error("hipfftPlanMany(%d,%d,%d x%d) failed: %s", p->nx, p->ny, p->nz, p->batch, fft_err(r));
    if (p->fwd) hipfftDestroy(p->fwd);
    if (p->inv) hipfftDestroy(p->inv);
    p->fwd = p->inv = 0;
    return MIPME_EFFT;
  }
  const int rc = plan_selftest(p);
  if (rc) {
    hipfftDestroy(p->fwd);
    hipfftDestroy(p->inv);
    p->fwd = p->inv = 0;
  }
  return rc;
}

int fft_plan_create(int dtype, int nx, int ny, int nz, int batch, mipme_fft_plan** out) {
  MIPME_REQUIRE(out != nullptr, "plan output pointer is NULL");
  MIPME_REQUIRE(nx > 0 && ny > 0 && nz > 0 && batch > 0, "invalid FFT dimensions %d %d %d x%d", nx, ny, nz, batch);
  MIPME_REQUIRE(dtype == MIPME_F32 || dtype == MIPME_F64, "invalid dtype %d", dtype);
  mipme_fft_plan* p = new (std::nothrow) mipme_fft_plan();
  MIPME_REQUIRE(p != nullptr, "out of host memory");
  p->dtype = dtype;
  p->nx = nx;
  p->ny = ny;
  p->nz = nz;
  p->batch = batch;
  {
    const char* e = getenv("MIPME_OWN_YZ");
    p->own_yz = xfused_dims_ok(nx) && own_yz_dims_ok(dtype, ny, nz) && !(e && e[0] == '0');
    // power-of-two planes that do not fit one workgroup's LDS: split transforms (z rows + y columns), still own kernels
    const bool pow2 = ny >= 2 && nz >= 4 && (ny & (ny - 1)) == 0 && (nz & (nz - 1)) == 0;
    if (!p->own_yz && xfused_dims_ok(nx) && pow2 && !(e && e[0] == '0')) {
      const int rows = split_rows_for(dtype, ny, nz);
      const size_t cs = dtype == MIPME_F32 ? 8 : 16;
      if (rows > 0 && cs * (size_t(ny) + size_t(ny / 2)) <= 32768) {  // a y column of one kz also has to fit its tile
        p->own_yz = p->split_yz = true;
        p->split_rows = rows;
      }
    }
  }
  // With our own plane kernels the fused convolution -- the hot path -- needs no hipFFT at all: the 3-D hipFFT plans (general
  // backward with a cell gradient, mipme_convolve, mipme_fft_r2c) are then created on first use (ensure_3d_plans).
  hipfftResult r = HIPFFT_SUCCESS;
  if (!p->own_yz) r = create_3d_plans(p);
  if (r == HIPFFT_SUCCESS && xfused_dims_ok(nx) && !p->own_yz) {
    int n2[2] = {ny, nz};
    int rembed2[2] = {ny, nz};
    int cembed2[2] = {ny, nz / 2 + 1};
    r = hipfftPlanMany(&p->fwd2d, 2, n2, rembed2, 1, ny * nz, cembed2, 1, ny * (nz / 2 + 1),
                       dtype == MIPME_F32 ? HIPFFT_R2C : HIPFFT_D2Z, nx * batch);
    if (r == HIPFFT_SUCCESS)
      r = hipfftPlanMany(&p->inv2d, 2, n2, cembed2, 1, ny * (nz / 2 + 1), rembed2, 1, ny * nz,
                         dtype == MIPME_F32 ? HIPFFT_C2R : HIPFFT_Z2D, nx * batch);
  }
  if (r != HIPFFT_SUCCESS) {
    set_error("hipfftPlanMany(%d,%d,%d x%d) failed: %s", nx, ny, nz, batch, fft_err(r));
    if (p->fwd) hipfftDestroy(p->fwd);
    if (p->inv) hipfftDestroy(p->inv);
    if (p->fwd2d) hipfftDestroy(p->fwd2d);
    if (p->inv2d) hipfftDestroy(p->inv2d);
    delete p;
    return MIPME_EFFT;
  }
  // brick counters + overflow counter, then the plane-list counters of the plane spread + their overflow counter (bricks.hip)
  const size_t n_count = plan_counter_words(nx, ny, nz);
  if (hipMalloc((void**)&p->brick_count, n_count * sizeof(int)) != hipSuccess ||
      hipMemset(p->brick_count, 0, n_count * sizeof(int)) != hipSuccess) {
    set_error("could not allocate the brick counters of the plan (plans cannot be created during stream capture)");
    (void)hipGetLastError();
    if (p->brick_count) (void)hipFree(p->brick_count);
    if (p->fwd) hipfftDestroy(p->fwd);
    if (p->inv) hipfftDestroy(p->inv);
    if (p->fwd2d) hipfftDestroy(p->fwd2d);
    if (p->inv2d) hipfftDestroy(p->inv2d);
    delete p;
    return MIPME_EHIP;
  }
  const int st_rc = plan_selftest(p);
  if (st_rc) {
    (void)hipFree(p->brick_count);
    if (p->fwd) hipfftDestroy(p->fwd);
    if (p->inv) hipfftDestroy(p->inv);
    if (p->fwd2d) hipfftDestroy(p->fwd2d);
    if (p->inv2d) hipfftDestroy(p->inv2d);
    delete p;
    return st_rc;
  }
  *out = p;
  return MIPME_OK;
}

struct FftDims { int dtype, nx, ny, nz, batch; };
FftDims fft_plan_dims(const mipme_fft_plan* p) { return FftDims{p->dtype, p->nx, p->ny, p->nz, p->batch}; }
int* fft_plan_brick_count(const mipme_fft_plan* p) { return p->brick_count; }

// Allocated on first use (a synchronous hipMalloc + hipMemset: not during stream capture -- the callers warm up first).
void* fft_plan_tail_scratch(mipme_fft_plan* p, int64_t bytes) {
  if (p->tail_scratch && p->tail_bytes >= bytes) return p->tail_scratch;
  if (p->tail_scratch) return nullptr;  // the size is a function of the plan's mesh: cannot change
  void* buf = nullptr;
  if (hipMalloc(&buf, size_t(bytes)) != hipSuccess || hipMemset(buf, 0, size_t(bytes)) != hipSuccess) {
    (void)hipGetLastError();
    if (buf) (void)hipFree(buf);
    return nullptr;
  }
  p->tail_scratch = buf;
  p->tail_bytes = bytes;
  return buf;
}

int fft_plan_destroy(mipme_fft_plan* p) {
  if (!p) return MIPME_OK;
  if (p->fwd) hipfftDestroy(p->fwd);
  if (p->inv) hipfftDestroy(p->inv);
  if (p->fwd2d) hipfftDestroy(p->fwd2d);
  if (p->inv2d) hipfftDestroy(p->inv2d);
  if (p->brick_count) (void)hipFree(p->brick_count);
  if (p->tail_scratch) (void)hipFree(p->tail_scratch);
  if (p->hat_parts) (void)hipFree(p->hat_parts);
  delete p;
  return MIPME_OK;
}

int fft_forward(mipme_fft_plan* p, hipStream_t st, const void* in, void* out) {
  int rc = ensure_3d_plans(p);
  if (rc) return rc;
  MIPME_CHECK_FFT(hipfftSetStream(p->fwd, st));
  if (p->dtype == MIPME_F32)
    MIPME_CHECK_FFT(hipfftExecR2C(p->fwd, (hipfftReal*)in, (hipfftComplex*)out));
  else
    MIPME_CHECK_FFT(hipfftExecD2Z(p->fwd, (hipfftDoubleReal*)in, (hipfftDoubleComplex*)out));
  return MIPME_OK;
}

int fft_inverse(mipme_fft_plan* p, hipStream_t st, void* in, void* out) {
  int rc = ensure_3d_plans(p);
  if (rc) return rc;
  MIPME_CHECK_FFT(hipfftSetStream(p->inv, st));
  if (p->dtype == MIPME_F32)
    MIPME_CHECK_FFT(hipfftExecC2R(p->inv, (hipfftComplex*)in, (hipfftReal*)out));
  else
    MIPME_CHECK_FFT(hipfftExecZ2D(p->inv, (hipfftDoubleComplex*)in, (hipfftDoubleReal*)out));
  return MIPME_OK;
}

template <typename T>
int kfilter_build_impl(hipStream_t st, const mipme_mesh_t* m, const mipme_potential_t* pot, void* G) {
  KPot kp;
  int rc = make_kpot(pot, kp);
  if (rc) return rc;
  const KGeom g = make_kgeom(m);
  const int64_t Mh = int64_t(g.nx) * g.ny * g.nzh;
  kfilter_kernel<T><<<unsigned((Mh + 255) / 256), 256, 0, st>>>(g, kp, (T*)G);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int apply_filter_impl(hipStream_t st, int64_t Mh, int C, const void* hat, const void* G, void* out, void* dc) {
  apply_filter_kernel<T><<<unsigned((Mh + 255) / 256), 256, 0, st>>>(Mh, C, (const T*)hat, (const T*)G, (T*)out, (T*)dc);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int64_t cellgrad_blocks(const mipme_mesh_t* m) {
  const int64_t Mh = int64_t(m->nx) * m->ny * (m->nz / 2 + 1);
  return (Mh + 255) / 256;
}

template <typename T>
int apply_filter_cellgrad_impl(hipStream_t st, const mipme_mesh_t* m, const mipme_potential_t* pot, const void* psi_hat,
                               const void* rho_hat, const void* G, void* out, void* dc, void* partials) {
  KPot kp;
  int rc = make_kpot(pot, kp);
  if (rc) return rc;
  const KGeom g = make_kgeom(m);
  apply_filter_cellgrad_kernel<T><<<unsigned(cellgrad_blocks(m)), 256, 0, st>>>(
      g, kp, m->n_channels, (const T*)psi_hat, (const T*)rho_hat, (const T*)G, (T*)out, (T*)dc, (double*)partials);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int64_t cellgrad_scratch_doubles() { return int64_t(kFinalizeBlocks) * kFinalizeNV + 1; }

template <typename T>
int cellgrad_finalize_impl(hipStream_t st, const mipme_mesh_t* m, double bg, int64_t n_atoms, void* partials,
                           const void* pos, const void* grad_pos, const void* gout, const void* phi_atoms,
                           const void* rho_dc, const void* psi_dc, const void* energy_scale, void* grad_cell,
                           int64_t kgrid_blocks, const void* field, const void* q) {
  // kgrid_blocks: number of 12-value k-grid partial sums in `partials` (0: as written by apply_filter_cellgrad_impl)
  const int64_t nb = kgrid_blocks > 0 ? kgrid_blocks : cellgrad_blocks(m);
  MIPME_REQUIRE(grad_pos || (field && q && energy_scale), "cell gradient needs grad_positions or the mesh field + charges");
  MIPME_REQUIRE(m->n_channels == 1 || grad_pos, "the mesh field is single-channel");
  double* scratch = (double*)partials + 12 * nb;
  cellgrad_finalize_kernel<T><<<kFinalizeBlocks, 256, 0, st>>>(*m, bg, n_atoms, int(nb),
                                                 (const double*)partials, scratch,
                                                 (const T*)pos, (const T*)grad_pos, (const T*)gout,
                                                 (const T*)phi_atoms, (const T*)rho_dc, (const T*)psi_dc,
                                                 (const T*)energy_scale, (T*)grad_cell, (const T*)field, (const T*)q);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int kfilter_deriv_impl(hipStream_t st, const mipme_mesh_t* m, const mipme_potential_t* pot, void* D) {
  KPot kp;
  int rc = make_kpot(pot, kp);
  if (rc) return rc;
  const KGeom g = make_kgeom(m);
  const int64_t Mh = int64_t(g.nx) * g.ny * g.nzh;
  kfilter_deriv_kernel<T><<<unsigned((Mh + 255) / 256), 256, 0, st>>>(g, kp, (T*)D);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// ---- dE/dcell of an energy step (E = sum q V, seed s), assembled by ONE workgroup from partial sums the step's kernels left
// behind (SURVEY.md Appendix A.5 with g = s q, i.e. psi = (s / 2V) rho):
//   rows[r][25]    cell riders (cell_rider_body): 15 moments of mu |rho^|^2 against the filter's derivative table
//                    M1[e][d] = sum w alpha f_e f_d (00 01 02 11 12 22),  M2[c][d] = sum w beta_c f_d
//                  => K[c][d] = 2 pi sum_k w dG/dk_c f_d = 2 pi (2 pi sum_e Ai[c][e] M1[e][d] - h_c M2[c][d]),
//                     H[c]    = sum_k w dG/dh_c = -2 pi sum_e Ai[c][e] M2[c][e],           h_c = |a_c| / n_c
//                  then 9 sums of the pair kernel  C[i][c] = sum_rows q_a sum_e w_e v'/d sh_i u_c,
//                  then E_k = sum_k mu G |rho^|^2  (= V sum_a q_a Phi_a: the gather is the adjoint of the spread)
//   rpart[b][9]    gather:   R[c][e] = sum_a r_{a,c} (s q_a field_{a,e})                                           (per brick)
// mesh part  gA[a][b] = -sum_c Ai[c][a] R[c][b] - (s/2V) sum_cd Ai[c][a] K[c][d] Ai[b][d] + dLdV V Ai[b][a]
//                       + (s/2V) H[a] / (|a_a| n_a) A[a][b],      dLdV = -s E_k / (2 V^2) + s bg Q^2 / V^2
// pair part  gP[m][c] = s f/2 sum_i Ai[i][m] C[i][c]   (every pair sits in two rows; f = 1/2 for a full list)
// out: 27 reals: mesh part, pair part, their sum.  Column sums: 25 groups of 16 lanes (rider rows) and 9 groups of 32 lanes
// (brick rows), sixteen loads in flight per lane -- a plain accumulation loop is a chain of dependent cold loads (the inputs
// were written by other XCDs a moment ago): 6.7 us for this kernel --, then 18 threads for the 3x3 algebra.
template <typename T>
__global__ __launch_bounds__(1024) void cell_tail_finalize_kernel(mipme_mesh_t m, double bg, double pair_scale, int n_rows,
                                                                 int n_bricks, const double* __restrict__ rows,
                                                                 const double* __restrict__ rpart, const T* __restrict__ dc,
                                                                 const T* __restrict__ seed, T* __restrict__ out) {
  __shared__ double gs[34], geo[18], s[31], res[18];
  {
    const int t = threadIdx.x;
    const double* base = nullptr;
    int n = 0, stride = 1, lanes = 16, l = 0, g = -1;
    if (t < 16 * kCellRow) {
      g = t >> 4, l = t & 15, lanes = 16;
      base = rows + g, n = n_rows, stride = kCellRow;
    } else if (t < 16 * kCellRow + 32 * 9) {
      const int u = t - 16 * kCellRow;
      g = kCellRow + (u >> 5), l = u & 31, lanes = 32;
      base = rpart + (u >> 5), n = n_bricks, stride = 9;
    }
    double v = 0.0;
    constexpr int B = 16;
    for (int i0 = l; i0 < n; i0 += lanes * B) {
      double tmp[B];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + lanes * u;
        tmp[u] = i < n ? base[int64_t(i) * stride] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < B; ++u) v += tmp[u];
    }
    // (16-lane groups are aligned to 16, 32-lane groups to 32 -- 16 * 25 = 400 is a multiple of 16 but not of 32: reduce both
    // with xor steps inside aligned 16-lane halves and combine the halves of a 32-lane group through LDS atomics-free adds)
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
    if (g >= 0 && g < kCellRow && l == 0) gs[g] = v;
    __shared__ double half[9][2];
    if (g >= kCellRow && (l & 15) == 0) half[g - kCellRow][l >> 4] = v;
    if (threadIdx.x == 1023) {  // cell and inverse cell to LDS with static indices (scalar loads of the by-value argument; the
                                // threads below index them by thread)
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        geo[i] = m.cell[i];
        geo[9 + i] = m.inv_cell[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < 9) gs[kCellRow + threadIdx.x] = half[threadIdx.x][0] + half[threadIdx.x][1];
  }
  __syncthreads();
  const double* A = geo;
  const double* Ai = geo + 9;
  // s[0..8] = K, s[9..11] = H, s[12..20] = C, s[21..29] = R, s[30] = E_k
  if (threadIdx.x < 31) {
    const int t = threadIdx.x;
    double v;
    if (t < 9) {
      const int cc = t / 3, d = t % 3;
      double a = 0.0;
      for (int e = 0; e < 3; ++e) {
        const int lo = e < d ? e : d, hi = e < d ? d : e;
        a += Ai[3 * cc + e] * gs[lo * 3 - lo * (lo - 1) / 2 + (hi - lo)];
      }
      const double hc = sqrt(A[3 * cc] * A[3 * cc] + A[3 * cc + 1] * A[3 * cc + 1] + A[3 * cc + 2] * A[3 * cc + 2]) /
                        double(cc == 0 ? m.nx : (cc == 1 ? m.ny : m.nz));
      v = 2.0 * kPi * (2.0 * kPi * a - hc * gs[6 + 3 * cc + d]);
    } else if (t < 12) {
      const int cc = t - 9;
      double a = 0.0;
      for (int e = 0; e < 3; ++e) a += Ai[3 * cc + e] * gs[6 + 3 * cc + e];
      v = -2.0 * kPi * a;
    } else if (t < 21) {
      v = gs[15 + (t - 12)];
    } else if (t < 30) {
      v = gs[kCellRow + (t - 21)];
    } else {
      v = gs[24];
    }
    s[t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    const double sd = seed ? double(seed[0]) : 1.0;
    const double V = m.volume, es = 0.5 * sd / V;
    if (threadIdx.x < 9) {
      const int a = threadIdx.x / 3, b = threadIdx.x % 3;
      const int nsa = a == 0 ? m.nx : (a == 1 ? m.ny : m.nz);
      const double Q = double(dc[0]);
      const double dLdV = -sd * s[30] / (2.0 * V * V) + sd * bg * Q * Q / (V * V);
      double v = 0.0;
      for (int c = 0; c < 3; ++c) {
        v -= Ai[3 * c + a] * s[21 + 3 * c + b];
        for (int d = 0; d < 3; ++d) v -= es * Ai[3 * c + a] * s[3 * c + d] * Ai[3 * b + d];
      }
      const double norm = sqrt(A[3 * a] * A[3 * a] + A[3 * a + 1] * A[3 * a + 1] + A[3 * a + 2] * A[3 * a + 2]);
      v += dLdV * V * Ai[3 * b + a] + es * s[9 + a] / (norm * double(nsa)) * A[3 * a + b];
      out[threadIdx.x] = T(v);
      res[threadIdx.x] = v;
    } else {
      const int mm = (threadIdx.x - 9) / 3, c = (threadIdx.x - 9) % 3;
      double v = 0.0;
      for (int i = 0; i < 3; ++i) v += Ai[3 * i + mm] * s[12 + 3 * i + c];
      v *= sd * pair_scale;
      out[threadIdx.x] = T(v);
      res[threadIdx.x] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < 9) out[18 + threadIdx.x] = T(res[threadIdx.x] + res[9 + threadIdx.x]);  // the sum, for callers with ONE cell tensor
}

template <typename T>
int cell_tail_finalize_impl(hipStream_t st, const mipme_mesh_t* m, double bg, double pair_scale, int64_t n_rows,
                            int64_t n_bricks, const void* rows, const void* rpart, const void* dc, const void* seed, void* out) {
  cell_tail_finalize_kernel<T><<<1, 1024, 0, st>>>(*m, bg, pair_scale, int(n_rows), int(n_bricks), (const double*)rows,
                                                  (const double*)rpart, (const T*)dc, (const T*)seed, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template int kfilter_deriv_impl<float>(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, void*);
template int kfilter_deriv_impl<double>(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, void*);
template int cell_tail_finalize_impl<float>(hipStream_t, const mipme_mesh_t*, double, double, int64_t, int64_t, const void*,
                                            const void*, const void*, const void*, void*);
template int cell_tail_finalize_impl<double>(hipStream_t, const mipme_mesh_t*, double, double, int64_t, int64_t, const void*,
                                             const void*, const void*, const void*, void*);
template int kfilter_build_impl<float>(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, void*);
template int kfilter_build_impl<double>(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, void*);
template int apply_filter_impl<float>(hipStream_t, int64_t, int, const void*, const void*, void*, void*);
template int apply_filter_impl<double>(hipStream_t, int64_t, int, const void*, const void*, void*, void*);
template int apply_filter_cellgrad_impl<float>(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, const void*,
                                               const void*, const void*, void*, void*, void*);
template int apply_filter_cellgrad_impl<double>(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*,
                                                const void*, const void*, const void*, void*, void*, void*);
template int cellgrad_finalize_impl<float>(hipStream_t, const mipme_mesh_t*, double, int64_t, void*, const void*,
                                           const void*, const void*, const void*, const void*, const void*, const void*,
                                           void*, int64_t, const void*, const void*);
template int cellgrad_finalize_impl<double>(hipStream_t, const mipme_mesh_t*, double, int64_t, void*, const void*,
                                            const void*, const void*, const void*, const void*, const void*, const void*,
                                            void*, int64_t, const void*, const void*);

}  // namespace mipme
