// Differentiable building blocks of the path: the primitives whose backward passes are made of the same primitives.
//
// The reference is a chain of ATen ops (calculators/calculator.py:43-87,103-189; calculators/pme.py:88-143;
// lib/mesh_interpolator.py:303-457), so PyTorch differentiates it to any order: a loss on forces with learned charges, a
// Hessian-vector product.  The fused kernels of this library are first order.  This file holds the second route
// (`calculator.double_backward = "analytic"`, torch-pme_amd/analytic.py): four linear maps, each closed under differentiation,
//
//   spread_jet (u, x; k)  mesh[c,m] = sum_i x[i,c] D^k W_i(m)        D^k = d^kx/du_x^kx d^ky/du_y^ky d^kz/du_z^kz of the
//   gather_jet (u, phi; k) out[i,c] = sum_m phi[c,m] D^k W_i(m)       interpolation weights W_i(m) = w(x_x) w(x_y) w(x_z)
//   pair_sum   (w, x)      out[i,c] = sum_p w_p x[j_p,c] (+ roles swapped for a half list)
//   pair_dot   (a, b)      out[p]   = sum_c a[i_p,c] b[j_p,c] (+ roles swapped)
//
// taking the FRACTIONAL mesh coordinates u = n r A^-1 (N,3) as a tensor, so that the chain to positions and cell is autograd's:
//   d spread_jet / du_d = x * gather_jet(., k + e_d),   d spread_jet / dx = gather_jet(., k)   and the same with the roles swapped.
// Lane layout as in mesh.hip (a group of pow2(n^2) lanes per atom, lane = (t_y, t_z), walking t_x; atomics for the spread).
// These kernels serve second-order and inspection paths: O(N n^3) each, not tuned further.
#include "common.h"

namespace mipme {

// k-th derivative (k = 0..3) of the 1-D weights with respect to x (common.h: generating rules).
//   P3M      : w_t = N_n(f + n-1-t), N_n' (y) = N_{n-1}(y) - N_{n-1}(y-1)  =>  the k-th derivative is the k-th backward difference
//              of the order n-k spline values; 0 for k >= n (the pieces have degree n-1; the reference differentiates the same
//              piecewise polynomials).
//   Lagrange : Leibniz recurrence over the linear factors of prod_{s != t} (x - xi_s).
template <int K, int N, typename T>
__device__ __forceinline__ void bspline_raise_to(T f, int m, T (&a)[N]) {
  if constexpr (K <= N) {
    if (K <= m) bspline_step<K, N, T>(f, a);
    bspline_raise_to<K + 1, N, T>(f, m, a);
  }
}

template <int SCHEME, int N, typename T>
__device__ __forceinline__ void weights_1d_jet(T x, int k, T (&w)[N]) {
  if constexpr (SCHEME == MIPME_P3M) {
    const int m = N - k;  // order of the spline whose values are differenced
    T a[N];
#pragma unroll
    for (int j = 0; j < N; ++j) a[j] = T(0);
    if (m >= 1) {
      a[0] = T(1);  // N_1(f) on [0,1)
      bspline_raise_to<2, N, T>(x + T(0.5), m, a);
#pragma unroll
      for (int rep = 0; rep < 3; ++rep) {
        if (rep < k) {
#pragma unroll
          for (int j = N - 1; j >= 1; --j) a[j] -= a[j - 1];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < N; ++t) w[t] = a[N - 1 - t];
  } else {
    T e[N];
#pragma unroll
    for (int s = 0; s < N; ++s) e[s] = x - (T(s) - T(0.5) * T(N - 1));
#pragma unroll
    for (int t = 0; t < N; ++t) {
      double den = 1.0;
#pragma unroll
      for (int s = 0; s < N; ++s)
        if (s != t) den *= double(t - s);
      T p0 = T(1), p1 = T(0), p2 = T(0), p3 = T(0);
#pragma unroll
      for (int s = 0; s < N; ++s) {
        if (s != t) {
          p3 = p3 * e[s] + T(3) * p2;
          p2 = p2 * e[s] + T(2) * p1;
          p1 = p1 * e[s] + p0;
          p0 = p0 * e[s];
        }
      }
      const T sel = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
      w[t] = sel * T(1.0 / den);
    }
  }
}

template <int N, typename T>
struct JetStencil {
  T wx[N];
  T wyz;
  int bx, rowoff;
  bool active;
};

template <int SCHEME, int N, typename T>
__device__ __forceinline__ void make_jet_stencil(int nx, int ny, int nz, const T* __restrict__ u, int64_t atom, int l, int kx,
                                                 int ky, int kz, JetStencil<N, T>& s) {
  int mx, my, mz;
  double xx, xy, xz;
  split_coordinate<N>(double(u[3 * atom + 0]), mx, xx);
  split_coordinate<N>(double(u[3 * atom + 1]), my, xy);
  split_coordinate<N>(double(u[3 * atom + 2]), mz, xz);
  T wy[N], wz[N];
  weights_1d_jet<SCHEME, N, T>(T(xx), kx, s.wx);
  weights_1d_jet<SCHEME, N, T>(T(xy), ky, wy);
  weights_1d_jet<SCHEME, N, T>(T(xz), kz, wz);
  const int ty = l / N, tz = l - ty * N;
  s.active = l < N * N;
  s.wyz = pick<N, T>(wy, ty) * pick<N, T>(wz, tz);
  s.rowoff = posmod(my + stencil_start<N>() + ty, ny) * nz + posmod(mz + stencil_start<N>() + tz, nz);
  s.bx = posmod(mx + stencil_start<N>(), nx);
}

template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(256) void spread_jet_kernel(int nx, int ny, int nz, int64_t n_atoms, int C, const T* __restrict__ u,
                                                        const T* __restrict__ val, int kx, int ky, int kz, T* __restrict__ mesh) {
  constexpr int LANES = StencilGroup<N>::LANES;
  constexpr int APB = 256 / LANES;
  const int l = threadIdx.x % LANES;
  const int64_t atom = int64_t(blockIdx.x) * APB + threadIdx.x / LANES;
  if (atom >= n_atoms) return;
  JetStencil<N, T> s;
  make_jet_stencil<SCHEME, N, T>(nx, ny, nz, u, atom, l, kx, ky, kz, s);
  if (!s.active) return;
  const int64_t plane = int64_t(ny) * nz, M = plane * nx;
  for (int c = 0; c < C; ++c) {
    const T q = val[atom * C + c] * s.wyz;
    T* mc = mesh + c * M + s.rowoff;
    int ix = s.bx;
#pragma unroll
    for (int tx = 0; tx < N; ++tx) {
      atomic_add(mc + ix * plane, q * s.wx[tx]);
      ix = (ix + 1 == nx) ? 0 : ix + 1;
    }
  }
}

template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(256) void gather_jet_kernel(int nx, int ny, int nz, int64_t n_atoms, int C, const T* __restrict__ u,
                                                        const T* __restrict__ mesh, int kx, int ky, int kz, T* __restrict__ out) {
  constexpr int LANES = StencilGroup<N>::LANES;
  constexpr int APB = 256 / LANES;
  const int l = threadIdx.x % LANES;
  int64_t atom = int64_t(blockIdx.x) * APB + threadIdx.x / LANES;
  const bool valid = atom < n_atoms;
  if (!valid) atom = n_atoms - 1;  // the whole group stays alive for the shuffles
  JetStencil<N, T> s;
  make_jet_stencil<SCHEME, N, T>(nx, ny, nz, u, atom, l, kx, ky, kz, s);
  const int64_t plane = int64_t(ny) * nz, M = plane * nx;
  const T wyz = s.active ? s.wyz : T(0);
  const int rowoff = s.active ? s.rowoff : 0;
  for (int c = 0; c < C; ++c) {
    const T* mc = mesh + c * M + rowoff;
    T acc = T(0);
    int ix = s.bx;
#pragma unroll
    for (int tx = 0; tx < N; ++tx) {
      acc += mc[ix * plane] * s.wx[tx];
      ix = (ix + 1 == nx) ? 0 : ix + 1;
    }
    acc *= wyz;
#pragma unroll
    for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, LANES);
    if (l == 0 && valid) out[atom * C + c] = acc;
  }
}

// The three gathers that the derivative with respect to u needs -- orders k + e_x, k + e_y, k + e_z -- in one walk over the
// stencil: out[i, c, d] = sum_m mesh[c, m] D^(k + e_d) W_i(m).  (Same mesh reads as one gather; a third of the launches of an
// evaluation that is bound by the host.)
template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(256) void gather_jet3_kernel(int nx, int ny, int nz, int64_t n_atoms, int C, const T* __restrict__ u,
                                                         const T* __restrict__ mesh, int kx, int ky, int kz, T* __restrict__ out) {
  constexpr int LANES = StencilGroup<N>::LANES;
  constexpr int APB = 256 / LANES;
  const int l = threadIdx.x % LANES;
  int64_t atom = int64_t(blockIdx.x) * APB + threadIdx.x / LANES;
  const bool valid = atom < n_atoms;
  if (!valid) atom = n_atoms - 1;
  int mx, my, mz;
  double xx, xy, xz;
  split_coordinate<N>(double(u[3 * atom + 0]), mx, xx);
  split_coordinate<N>(double(u[3 * atom + 1]), my, xy);
  split_coordinate<N>(double(u[3 * atom + 2]), mz, xz);
  T wx0[N], wx1[N], wy0[N], wy1[N], wz0[N], wz1[N];
  weights_1d_jet<SCHEME, N, T>(T(xx), kx, wx0);
  weights_1d_jet<SCHEME, N, T>(T(xx), kx + 1, wx1);
  weights_1d_jet<SCHEME, N, T>(T(xy), ky, wy0);
  weights_1d_jet<SCHEME, N, T>(T(xy), ky + 1, wy1);
  weights_1d_jet<SCHEME, N, T>(T(xz), kz, wz0);
  weights_1d_jet<SCHEME, N, T>(T(xz), kz + 1, wz1);
  const int ty = l / N, tz = l - ty * N;
  const bool active = l < N * N;
  const T y0 = pick<N, T>(wy0, ty), y1 = pick<N, T>(wy1, ty), z0 = pick<N, T>(wz0, tz), z1 = pick<N, T>(wz1, tz);
  const int rowoff = active ? posmod(my + stencil_start<N>() + ty, ny) * nz + posmod(mz + stencil_start<N>() + tz, nz) : 0;
  const int bx = posmod(mx + stencil_start<N>(), nx);
  const int64_t plane = int64_t(ny) * nz, M = plane * nx;
  for (int c = 0; c < C; ++c) {
    const T* mc = mesh + c * M + rowoff;
    T a0 = T(0), a1 = T(0);
    int ix = bx;
#pragma unroll
    for (int tx = 0; tx < N; ++tx) {
      const T v = mc[ix * plane];
      a0 += v * wx0[tx];
      a1 += v * wx1[tx];
      ix = (ix + 1 == nx) ? 0 : ix + 1;
    }
    T o[3] = {active ? a1 * y0 * z0 : T(0), active ? a0 * y1 * z0 : T(0), active ? a0 * y0 * z1 : T(0)};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int off = LANES / 2; off > 0; off >>= 1) o[d] += __shfl_xor(o[d], off, LANES);
      if (l == 0 && valid) out[(atom * C + c) * 3 + d] = o[d];
    }
  }
}

// mode: 0 = half list (both directions), 1 = full list (i <- j), 2 = full list transposed (j <- i: the adjoint of mode 1)
template <typename T, typename I>
__global__ __launch_bounds__(256) void pair_sum_kernel(int64_t P, int C, const I* __restrict__ pairs, const T* __restrict__ w,
                                                      const T* __restrict__ x, int mode, T* __restrict__ out) {
  const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t i = int64_t(pairs[2 * p]), j = int64_t(pairs[2 * p + 1]);
  const T wp = w[p];
  for (int c = 0; c < C; ++c) {
    if (mode != 2) atomic_add(out + i * C + c, wp * x[j * C + c]);
    if (mode != 1) atomic_add(out + j * C + c, wp * x[i * C + c]);
  }
}

// The same sum from the transposed list (topology.hip: row_ptr[2a], [2a+1], [2a+2] bound atom a's role-i and role-j entries
// {other atom, pair index}): 16 lanes per atom, no atomics, a fixed summation order -- the atomic version above spends 0.6 ms
// on the 4.76 M pairs of the 32k-atom water box (300 updates per address), this one streams the entries once.
template <typename T>
__global__ __launch_bounds__(256) void pair_sum_rows_kernel(int64_t N, int C, const int* __restrict__ row_ptr,
                                                           const int2* __restrict__ entries, const T* __restrict__ w,
                                                           const T* __restrict__ x, int mode, T* __restrict__ out) {
  constexpr int LANES = 16;
  const int sub = threadIdx.x % LANES;
  int64_t a = int64_t(blockIdx.x) * (256 / LANES) + threadIdx.x / LANES;
  const bool valid = a < N;
  if (!valid) a = N - 1;
  const int begin = row_ptr[2 * a], mid = row_ptr[2 * a + 1], end = row_ptr[2 * a + 2];
  const int lo = mode == 2 ? mid : begin, hi = mode == 1 ? mid : end;
  for (int c = 0; c < C; ++c) {
    T acc = T(0);
    for (int e = lo + sub; e < hi; e += LANES) {
      const int2 en = entries[e];
      acc += w[en.y] * x[int64_t(en.x) * C + c];
    }
#pragma unroll
    for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, LANES);
    if (sub == 0 && valid) out[a * C + c] = acc;
  }
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void pair_dot_kernel(int64_t P, int C, const I* __restrict__ pairs, const T* __restrict__ a,
                                                      const T* __restrict__ b, int half, T* __restrict__ out) {
  const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t i = int64_t(pairs[2 * p]), j = int64_t(pairs[2 * p + 1]);
  T acc = T(0);
  for (int c = 0; c < C; ++c) {
    acc += a[i * C + c] * b[j * C + c];
    if (half) acc += a[j * C + c] * b[i * C + c];
  }
  out[p] = acc;
}

// The caller-side distance helper (tests/helpers.py:278-304) differentiated twice needs the pair DIFFERENCE and its adjoint:
//   pair_diff    out[p,c] = x[j_p,c] - x[i_p,c]                    adjoint: pair_scatter
//   pair_scatter out[a,c] = sum_{p: j_p = a} v[p,c] - sum_{p: i_p = a} v[p,c]     adjoint: pair_diff
// (ATen's index_add_ does the second one with compare-and-swap loops in double precision: 4 ms for 1.2 M pairs.)
template <typename T, typename I>
__global__ __launch_bounds__(256) void pair_diff_kernel(int64_t P, int C, const I* __restrict__ pairs, const T* __restrict__ x,
                                                       T* __restrict__ out) {
  const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t i = int64_t(pairs[2 * p]), j = int64_t(pairs[2 * p + 1]);
  for (int c = 0; c < C; ++c) out[p * C + c] = x[j * C + c] - x[i * C + c];
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void pair_scatter_kernel(int64_t P, int C, const I* __restrict__ pairs, const T* __restrict__ v,
                                                          T* __restrict__ out) {
  const int64_t p = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t i = int64_t(pairs[2 * p]), j = int64_t(pairs[2 * p + 1]);
  for (int c = 0; c < C; ++c) {
    const T val = v[p * C + c];
    atomic_add(out + j * C + c, val);
    atomic_add(out + i * C + c, -val);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pair_scatter_rows_kernel(int64_t N, int C, const int* __restrict__ row_ptr,
                                                               const int2* __restrict__ entries, const T* __restrict__ v,
                                                               T* __restrict__ out) {
  constexpr int LANES = 16;
  const int sub = threadIdx.x % LANES;
  int64_t a = int64_t(blockIdx.x) * (256 / LANES) + threadIdx.x / LANES;
  const bool valid = a < N;
  if (!valid) a = N - 1;
  const int begin = row_ptr[2 * a], mid = row_ptr[2 * a + 1], end = row_ptr[2 * a + 2];
  // (channels in chunks of four per walk over the row: the common case, three Cartesian components, reads the entries once)
  for (int c0 = 0; c0 < C; c0 += 4) {
    const int nc = min(4, C - c0);
    T acc[4] = {T(0), T(0), T(0), T(0)};
    for (int e = begin + sub; e < end; e += LANES) {
      const T* vp = v + int64_t(entries[e].y) * C + c0;
      const T sgn = e < mid ? T(-1) : T(1);  // role-i entries first (the atom is i of the pair), then role-j
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nc) acc[c] += sgn * vp[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int off = LANES / 2; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, LANES);
      if (c < nc && sub == 0 && valid) out[a * C + c0 + c] = acc[c];
    }
  }
}

template <int N>
static inline unsigned jet_blocks(int64_t n_atoms) {
  constexpr int APB = 256 / StencilGroup<N>::LANES;
  return unsigned((n_atoms + APB - 1) / APB);
}

template <typename T>
static int spread_jet_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* u, const void* val, int kx, int ky,
                           int kz, void* mesh) {
  MIPME_CHECK_HIP(zero_async(mesh, sizeof(T) * size_t(m->n_channels) * m->nx * m->ny * m->nz, st));
  if (n_atoms == 0) return MIPME_OK;
  MIPME_DISPATCH_STENCIL(m->scheme, m->order,
                         (spread_jet_kernel<S, N, T><<<jet_blocks<N>(n_atoms), 256, 0, st>>>(
                             m->nx, m->ny, m->nz, n_atoms, m->n_channels, (const T*)u, (const T*)val, kx, ky, kz, (T*)mesh)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int gather_jet_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* u, const void* mesh, int kx, int ky,
                           int kz, void* out) {
  if (n_atoms == 0) return MIPME_OK;
  MIPME_DISPATCH_STENCIL(m->scheme, m->order,
                         (gather_jet_kernel<S, N, T><<<jet_blocks<N>(n_atoms), 256, 0, st>>>(
                             m->nx, m->ny, m->nz, n_atoms, m->n_channels, (const T*)u, (const T*)mesh, kx, ky, kz, (T*)out)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int gather_jet3_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* u, const void* mesh, int kx, int ky,
                            int kz, void* out) {
  if (n_atoms == 0) return MIPME_OK;
  MIPME_DISPATCH_STENCIL(m->scheme, m->order,
                         (gather_jet3_kernel<S, N, T><<<jet_blocks<N>(n_atoms), 256, 0, st>>>(
                             m->nx, m->ny, m->nz, n_atoms, m->n_channels, (const T*)u, (const T*)mesh, kx, ky, kz, (T*)out)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
static int pair_sum_impl(hipStream_t st, int64_t P, int64_t N, int C, const void* pairs, const void* w, const void* x, int mode,
                         void* out) {
  MIPME_CHECK_HIP(zero_async(out, sizeof(T) * size_t(N) * C, st));
  if (P == 0) return MIPME_OK;
  pair_sum_kernel<T, I><<<unsigned((P + 255) / 256), 256, 0, st>>>(P, C, (const I*)pairs, (const T*)w, (const T*)x, mode, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int pair_sum_rows_impl(hipStream_t st, int64_t N, int C, const void* row_ptr, const void* entries, const void* w,
                              const void* x, int mode, void* out) {
  if (N == 0) return MIPME_OK;
  pair_sum_rows_kernel<T><<<unsigned((N + 15) / 16), 256, 0, st>>>(N, C, (const int*)row_ptr, (const int2*)entries, (const T*)w,
                                                                  (const T*)x, mode, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
static int pair_dot_impl(hipStream_t st, int64_t P, int C, const void* pairs, const void* a, const void* b, int half, void* out) {
  if (P == 0) return MIPME_OK;
  pair_dot_kernel<T, I><<<unsigned((P + 255) / 256), 256, 0, st>>>(P, C, (const I*)pairs, (const T*)a, (const T*)b, half, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
static int pair_diff_impl(hipStream_t st, int64_t P, int C, const void* pairs, const void* x, void* out) {
  if (P == 0) return MIPME_OK;
  pair_diff_kernel<T, I><<<unsigned((P + 255) / 256), 256, 0, st>>>(P, C, (const I*)pairs, (const T*)x, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
static int pair_scatter_impl(hipStream_t st, int64_t P, int64_t N, int C, const void* pairs, const void* row_ptr,
                             const void* entries, const void* v, void* out) {
  if (N == 0) return MIPME_OK;
  if (row_ptr && entries) {
    pair_scatter_rows_kernel<T><<<unsigned((N + 15) / 16), 256, 0, st>>>(N, C, (const int*)row_ptr, (const int2*)entries,
                                                                        (const T*)v, (T*)out);
  } else {
    MIPME_CHECK_HIP(zero_async(out, sizeof(T) * size_t(N) * C, st));
    if (P > 0) pair_scatter_kernel<T, I><<<unsigned((P + 255) / 256), 256, 0, st>>>(P, C, (const I*)pairs, (const T*)v, (T*)out);
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // namespace mipme

using namespace mipme;

static inline bool jet_orders_ok(int kx, int ky, int kz) {
  return kx >= 0 && ky >= 0 && kz >= 0 && kx <= 3 && ky <= 3 && kz <= 3;
}

extern "C" {

int mipme_spread_jet(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* u, const void* values,
                     int kx, int ky, int kz, void* mesh_out) {
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && mesh_out && (n_atoms == 0 || (u && values)), "NULL buffer passed to mipme_spread_jet");
  MIPME_REQUIRE(jet_orders_ok(kx, ky, kz), "derivative orders (%d,%d,%d) outside 0..3", kx, ky, kz);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return spread_jet_impl<float>(st, mesh, n_atoms, u, values, kx, ky, kz, mesh_out);
  if (dtype == MIPME_F64) return spread_jet_impl<double>(st, mesh, n_atoms, u, values, kx, ky, kz, mesh_out);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_gather_jet(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* u, const void* mesh_in,
                     int kx, int ky, int kz, void* out) {
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && mesh_in && (n_atoms == 0 || (u && out)), "NULL buffer passed to mipme_gather_jet");
  MIPME_REQUIRE(jet_orders_ok(kx, ky, kz), "derivative orders (%d,%d,%d) outside 0..3", kx, ky, kz);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return gather_jet_impl<float>(st, mesh, n_atoms, u, mesh_in, kx, ky, kz, out);
  if (dtype == MIPME_F64) return gather_jet_impl<double>(st, mesh, n_atoms, u, mesh_in, kx, ky, kz, out);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_gather_jet3(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* u, const void* mesh_in,
                      int kx, int ky, int kz, void* out) {
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && mesh_in && (n_atoms == 0 || (u && out)), "NULL buffer passed to mipme_gather_jet3");
  MIPME_REQUIRE(jet_orders_ok(kx + 1, ky + 1, kz + 1), "derivative orders (%d,%d,%d) + 1 outside 0..3", kx, ky, kz);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return gather_jet3_impl<float>(st, mesh, n_atoms, u, mesh_in, kx, ky, kz, out);
  if (dtype == MIPME_F64) return gather_jet3_impl<double>(st, mesh, n_atoms, u, mesh_in, kx, ky, kz, out);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_pair_sum(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels, const void* pairs,
                   const void* weights, const void* x, int mode, void* out) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0 && n_channels > 0 && mode >= 0 && mode <= 2, "invalid argument of mipme_pair_sum");
  MIPME_REQUIRE(n_atoms == 0 || out, "NULL output passed to mipme_pair_sum");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && weights && x), "NULL buffer passed to mipme_pair_sum");
  MIPME_REQUIRE((dtype == MIPME_F32 || dtype == MIPME_F64) && (idx_dtype == MIPME_I64 || idx_dtype == MIPME_I32),
                "invalid dtype %d / index dtype %d", dtype, idx_dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return idx_dtype == MIPME_I64 ? pair_sum_impl<float, int64_t>(st, n_pairs, n_atoms, n_channels, pairs, weights, x, mode, out)
                                  : pair_sum_impl<float, int32_t>(st, n_pairs, n_atoms, n_channels, pairs, weights, x, mode, out);
  return idx_dtype == MIPME_I64 ? pair_sum_impl<double, int64_t>(st, n_pairs, n_atoms, n_channels, pairs, weights, x, mode, out)
                                : pair_sum_impl<double, int32_t>(st, n_pairs, n_atoms, n_channels, pairs, weights, x, mode, out);
}

int mipme_pair_sum_rows(void* stream, int dtype, int64_t n_atoms, int n_channels, const void* row_ptr, const void* entries,
                        const void* weights, const void* x, int mode, void* out) {
  MIPME_REQUIRE(n_atoms >= 0 && n_channels > 0 && mode >= 0 && mode <= 2, "invalid argument of mipme_pair_sum_rows");
  MIPME_REQUIRE(n_atoms == 0 || (row_ptr && entries && weights && x && out), "NULL buffer passed to mipme_pair_sum_rows");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return pair_sum_rows_impl<float>(st, n_atoms, n_channels, row_ptr, entries, weights, x, mode, out);
  if (dtype == MIPME_F64) return pair_sum_rows_impl<double>(st, n_atoms, n_channels, row_ptr, entries, weights, x, mode, out);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_pair_dot(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int n_channels, const void* pairs, const void* a,
                   const void* b, int half, void* out) {
  MIPME_REQUIRE(n_pairs >= 0 && n_channels > 0, "invalid argument of mipme_pair_dot");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && a && b && out), "NULL buffer passed to mipme_pair_dot");
  MIPME_REQUIRE((dtype == MIPME_F32 || dtype == MIPME_F64) && (idx_dtype == MIPME_I64 || idx_dtype == MIPME_I32),
                "invalid dtype %d / index dtype %d", dtype, idx_dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return idx_dtype == MIPME_I64 ? pair_dot_impl<float, int64_t>(st, n_pairs, n_channels, pairs, a, b, half, out)
                                  : pair_dot_impl<float, int32_t>(st, n_pairs, n_channels, pairs, a, b, half, out);
  return idx_dtype == MIPME_I64 ? pair_dot_impl<double, int64_t>(st, n_pairs, n_channels, pairs, a, b, half, out)
                                : pair_dot_impl<double, int32_t>(st, n_pairs, n_channels, pairs, a, b, half, out);
}

int mipme_pair_diff(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int n_channels, const void* pairs, const void* x,
                    void* out) {
  MIPME_REQUIRE(n_pairs >= 0 && n_channels > 0, "invalid argument of mipme_pair_diff");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && x && out), "NULL buffer passed to mipme_pair_diff");
  MIPME_REQUIRE((dtype == MIPME_F32 || dtype == MIPME_F64) && (idx_dtype == MIPME_I64 || idx_dtype == MIPME_I32),
                "invalid dtype %d / index dtype %d", dtype, idx_dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return idx_dtype == MIPME_I64 ? pair_diff_impl<float, int64_t>(st, n_pairs, n_channels, pairs, x, out)
                                  : pair_diff_impl<float, int32_t>(st, n_pairs, n_channels, pairs, x, out);
  return idx_dtype == MIPME_I64 ? pair_diff_impl<double, int64_t>(st, n_pairs, n_channels, pairs, x, out)
                                : pair_diff_impl<double, int32_t>(st, n_pairs, n_channels, pairs, x, out);
}

int mipme_pair_scatter(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels, const void* pairs,
                       const void* row_ptr, const void* entries, const void* values, void* out) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0 && n_channels > 0, "invalid argument of mipme_pair_scatter");
  MIPME_REQUIRE(n_atoms == 0 || out, "NULL output passed to mipme_pair_scatter");
  MIPME_REQUIRE(n_pairs == 0 || (values && (pairs || (row_ptr && entries))), "NULL buffer passed to mipme_pair_scatter");
  MIPME_REQUIRE((row_ptr == nullptr) == (entries == nullptr), "row_ptr and entries go together");
  MIPME_REQUIRE((dtype == MIPME_F32 || dtype == MIPME_F64) && (idx_dtype == MIPME_I64 || idx_dtype == MIPME_I32),
                "invalid dtype %d / index dtype %d", dtype, idx_dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return idx_dtype == MIPME_I64
               ? pair_scatter_impl<float, int64_t>(st, n_pairs, n_atoms, n_channels, pairs, row_ptr, entries, values, out)
               : pair_scatter_impl<float, int32_t>(st, n_pairs, n_atoms, n_channels, pairs, row_ptr, entries, values, out);
  return idx_dtype == MIPME_I64
             ? pair_scatter_impl<double, int64_t>(st, n_pairs, n_atoms, n_channels, pairs, row_ptr, entries, values, out)
             : pair_scatter_impl<double, int32_t>(st, n_pairs, n_atoms, n_channels, pairs, row_ptr, entries, values, out);
}

}  // extern "C"
