// Particle <-> mesh kernels: charge spreading, potential gathering, and the gradient gather.
//
// Replaces MeshInterpolator.compute_weights / points_to_mesh / mesh_to_points
// (reference lib/mesh_interpolator.py:303-457), which materialise (n,N,3) weights plus three
// (n^3,N) int64 index tables in HBM (98 MB at 32k atoms, n=5) and then scatter/gather through them.
// Here nothing is materialised: a group of LANES = pow2(n^2) lanes owns one atom, each lane one
// (ty,tz) column of the n^3 stencil and walks tx.  Consecutive lanes hit consecutive z addresses, so
// the n contiguous mesh points of a stencil row share one cache line and one atomic request.
// The meshes of the benchmark configurations (<= 8 MiB) are L2 / Infinity-Cache resident.
#include "common.h"

namespace mipme {

template <int SCHEME, int N, bool DERIV, typename T>
struct AtomStencil {
  T wx[N], dwx[N];
  T wy, wz, dwy, dwz;
  int bx;      // wrapped base index along x (offset 0 of the stencil)
  int rowoff;  // (iy*nz + iz) for this lane
  bool active;
};

// Compute the stencil of one atom for lane `l` of its group.
template <int SCHEME, int N, bool DERIV, typename T>
__device__ __forceinline__ void make_stencil(const Geom& g, const T* __restrict__ pos, int64_t atom, int l,
                                             AtomStencil<SCHEME, N, DERIV, T>& s) {
  const double rx = double(pos[3 * atom + 0]);
  const double ry = double(pos[3 * atom + 1]);
  const double rz = double(pos[3 * atom + 2]);
  const double ux = double(g.nx) * (rx * g.inv[0] + ry * g.inv[3] + rz * g.inv[6]);
  const double uy = double(g.ny) * (rx * g.inv[1] + ry * g.inv[4] + rz * g.inv[7]);
  const double uz = double(g.nz) * (rx * g.inv[2] + ry * g.inv[5] + rz * g.inv[8]);
  int mx, my, mz;
  double xx, xy, xz;
  split_coordinate<N>(ux, mx, xx);
  split_coordinate<N>(uy, my, xy);
  split_coordinate<N>(uz, mz, xz);
  T wy[N], wz[N], dwy[N], dwz[N];
  weights_1d<SCHEME, N, DERIV, T>(T(xx), s.wx, s.dwx);
  weights_1d<SCHEME, N, DERIV, T>(T(xy), wy, dwy);
  weights_1d<SCHEME, N, DERIV, T>(T(xz), wz, dwz);
  const int ty = l / N, tz = l - ty * N;
  s.active = l < N * N;
  s.wy = pick<N, T>(wy, ty);
  s.wz = pick<N, T>(wz, tz);
  if constexpr (DERIV) {
    s.dwy = pick<N, T>(dwy, ty);
    s.dwz = pick<N, T>(dwz, tz);
  }
  const int iy = posmod(my + stencil_start<N>() + ty, g.ny);
  const int iz = posmod(mz + stencil_start<N>() + tz, g.nz);
  s.rowoff = iy * g.nz + iz;
  s.bx = posmod(mx + stencil_start<N>(), g.nx);
}

template <int LANES, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LANES);
  return v;
}

// ---- spread ------------------------------------------------------------------------------------
template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(256) void spread_kernel(Geom g, int64_t n_atoms, int C, const T* __restrict__ pos,
                                                    const T* __restrict__ val, T scale, T* __restrict__ mesh) {
  constexpr int LANES = StencilGroup<N>::LANES;
  constexpr int APB = 256 / LANES;
  const int l = threadIdx.x % LANES;
  const int64_t atom = int64_t(blockIdx.x) * APB + threadIdx.x / LANES;
  if (atom >= n_atoms) return;
  AtomStencil<SCHEME, N, false, T> s;
  make_stencil<SCHEME, N, false, T>(g, pos, atom, l, s);
  if (!s.active) return;
  const int64_t plane = int64_t(g.ny) * g.nz;
  const int64_t M = plane * g.nx;
  const T wyz = s.wy * s.wz;
  for (int c = 0; c < C; ++c) {
    const T q = val[atom * C + c] * scale * wyz;
    T* mc = mesh + c * M + s.rowoff;
    int ix = s.bx;
#pragma unroll
    for (int tx = 0; tx < N; ++tx) {
      atomic_add(mc + ix * plane, q * s.wx[tx]);
      ix = (ix + 1 == g.nx) ? 0 : ix + 1;
    }
  }
}

// ---- gather (forward) --------------------------------------------------------------------------
// EPILOGUE: out = 1/2 (acc*inv_vol - self*q - 2*bg*inv_vol*Q_c)   [pme.py:113-143], and raw = acc*inv_vol
template <int SCHEME, int N, bool EPILOGUE, typename T>
__global__ __launch_bounds__(256) void gather_kernel(Geom g, int64_t n_atoms, int C, const T* __restrict__ pos,
                                                    const T* __restrict__ mesh, const T* __restrict__ q,
                                                    const T* __restrict__ qsum, T inv_vol, T self_c, T bg_c,
                                                    bool accumulate, T* __restrict__ out, T* __restrict__ raw,
                                                    int* __restrict__ nan_flag = nullptr) {
  constexpr int LANES = StencilGroup<N>::LANES;
  constexpr int APB = 256 / LANES;
  const int l = threadIdx.x % LANES;
  int64_t atom = int64_t(blockIdx.x) * APB + threadIdx.x / LANES;
  const bool valid = atom < n_atoms;
  if (!valid) atom = n_atoms - 1;  // keep the whole group alive for the shuffles
  AtomStencil<SCHEME, N, false, T> s;
  make_stencil<SCHEME, N, false, T>(g, pos, atom, l, s);
  const int64_t plane = int64_t(g.ny) * g.nz;
  const int64_t M = plane * g.nx;
  const T wyz = s.active ? s.wy * s.wz : T(0);
  const int rowoff = s.active ? s.rowoff : 0;
  for (int c = 0; c < C; ++c) {
    const T* mc = mesh + c * M + rowoff;
    T acc = T(0);
    int ix = s.bx;
#pragma unroll
    for (int tx = 0; tx < N; ++tx) {
      acc += mc[ix * plane] * s.wx[tx];
      ix = (ix + 1 == g.nx) ? 0 : ix + 1;
    }
    acc = group_sum<LANES, T>(acc * wyz);
    if (l == 0 && valid) {
      if constexpr (EPILOGUE) {
        const T phi = acc * inv_vol;
        const T qi = q[atom * C + c];
        const T lr = T(0.5) * (phi - self_c * qi - T(2) * bg_c * inv_vol * qsum[c]);
        out[atom * C + c] = accumulate ? out[atom * C + c] + lr : lr;
        if (nan_flag && lr != lr) *nan_flag = 1;  // NaN guard of kspace_filter.py:189-195 (see mipme.h, nan_flag)
        if (raw) raw[atom * C + c] = phi;
      } else {
        out[atom * C + c] = acc;
      }
    }
  }
}

// ---- gather of the gradient (backward) ---------------------------------------------------------
// With h = g/(2V):  dL/dW_i(g) = sum_c [h_ic phi_c(g) + q_ic chi_c(g)];  dL/du_d = gather with one derivative
// weight;  grad_pos_c = sum_d n_d Ainv[c][d] dL/du_d;  grad_q_ic = gather(chi_c) - self/2 g_ic - bg/V sum_j g_jc.
template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(256) void gather_grad_kernel(Geom g, int64_t n_atoms, int C, const T* __restrict__ pos,
                                                         const T* __restrict__ q, const T* __restrict__ gout,
                                                         const T* __restrict__ phi, const T* __restrict__ chi,
                                                         const T* __restrict__ psi_dc, const T* __restrict__ gscale,
                                                         T half_inv_vol, T self_c, T bg_c, T* __restrict__ grad_pos,
                                                         T* __restrict__ grad_q) {
  constexpr int LANES = StencilGroup<N>::LANES;
  constexpr int APB = 256 / LANES;
  const T cs = gscale ? gscale[0] * half_inv_vol : T(1);  // energy mode: chi = cs * phi (see bricks.hip)
  const int l = threadIdx.x % LANES;
  int64_t atom = int64_t(blockIdx.x) * APB + threadIdx.x / LANES;
  const bool valid = atom < n_atoms;
  if (!valid) atom = n_atoms - 1;
  AtomStencil<SCHEME, N, true, T> s;
  make_stencil<SCHEME, N, true, T>(g, pos, atom, l, s);
  const int64_t plane = int64_t(g.ny) * g.nz;
  const int64_t M = plane * g.nx;
  const int rowoff = s.active ? s.rowoff : 0;
  const T act = s.active ? T(1) : T(0);
  T sx = T(0), sdx = T(0);
  for (int c = 0; c < C; ++c) {
    const T hc = gout[atom * C + c] * half_inv_vol;
    const T qc = q[atom * C + c];
    const T* pc = phi + c * M + rowoff;
    const T* cc = chi + c * M + rowoff;
    T schi = T(0);
    int ix = s.bx;
#pragma unroll
    for (int tx = 0; tx < N; ++tx) {
      const T vchi = cc[ix * plane] * cs;
      const T v = hc * pc[ix * plane] + qc * vchi;
      sx += v * s.wx[tx];
      sdx += v * s.dwx[tx];
      schi += vchi * s.wx[tx];
      ix = (ix + 1 == g.nx) ? 0 : ix + 1;
    }
    if (grad_q) {
      schi = group_sum<LANES, T>(schi * s.wy * s.wz * act);
      if (l == 0 && valid) {
        // (bg/V) sum_j g_jc = 2 bg * dc(psi_c), psi = spread(g/2V)
        grad_q[atom * C + c] = schi - T(0.5) * self_c * gout[atom * C + c] - T(2) * bg_c * psi_dc[c] * cs;
      }
    }
  }
  if (grad_pos) {
    const T ax = group_sum<LANES, T>(sdx * s.wy * s.wz * act) * T(g.nx);
    const T ay = group_sum<LANES, T>(sx * s.dwy * s.wz * act) * T(g.ny);
    const T az = group_sum<LANES, T>(sx * s.wy * s.dwz * act) * T(g.nz);
    if (l == 0 && valid) {
      grad_pos[3 * atom + 0] = T(g.inv[0]) * ax + T(g.inv[1]) * ay + T(g.inv[2]) * az;
      grad_pos[3 * atom + 1] = T(g.inv[3]) * ax + T(g.inv[4]) * ay + T(g.inv[5]) * az;
      grad_pos[3 * atom + 2] = T(g.inv[6]) * ax + T(g.inv[7]) * ay + T(g.inv[8]) * az;
    }
  }
}

// ---- dispatch ----------------------------------------------------------------------------------
template <int N>
static inline unsigned blocks_for(int64_t n_atoms) {
  constexpr int APB = 256 / StencilGroup<N>::LANES;
  return unsigned((n_atoms + APB - 1) / APB);
}

template <typename T>
int spread_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* pos, const void* val, double scale,
                void* mesh) {
  const Geom g = make_geom(m);
  const size_t bytes = sizeof(T) * size_t(m->n_channels) * m->nx * m->ny * m->nz;
  MIPME_CHECK_HIP(zero_async(mesh, bytes, st));
  if (n_atoms == 0) return MIPME_OK;
  MIPME_DISPATCH_STENCIL(m->scheme, m->order, (spread_kernel<S, N, T><<<blocks_for<N>(n_atoms), 256, 0, st>>>(
                                                  g, n_atoms, m->n_channels, (const T*)pos, (const T*)val, T(scale), (T*)mesh)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int gather_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* pos, const void* mesh, void* out) {
  if (n_atoms == 0) return MIPME_OK;
  const Geom g = make_geom(m);
  MIPME_DISPATCH_STENCIL(m->scheme, m->order,
                         (gather_kernel<S, N, false, T><<<blocks_for<N>(n_atoms), 256, 0, st>>>(
                             g, n_atoms, m->n_channels, (const T*)pos, (const T*)mesh, nullptr, nullptr, T(0), T(0),
                             T(0), false, (T*)out, nullptr)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int gather_epilogue_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* pos, const void* mesh,
                         const void* q, const void* qsum, double self_c, double bg_c, void* out, void* raw,
                         int accumulate, void* nan_flag) {
  if (n_atoms == 0) return MIPME_OK;
  const Geom g = make_geom(m);
  MIPME_DISPATCH_STENCIL(m->scheme, m->order,
                         (gather_kernel<S, N, true, T><<<blocks_for<N>(n_atoms), 256, 0, st>>>(
                             g, n_atoms, m->n_channels, (const T*)pos, (const T*)mesh, (const T*)q, (const T*)qsum,
                             T(1.0 / m->volume), T(self_c), T(bg_c), accumulate != 0, (T*)out, (T*)raw, (int*)nan_flag)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int gather_grad_impl(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* pos, const void* q,
                     const void* gout, const void* phi, const void* chi, const void* gsum_dc, const void* gscale,
                     double self_c, double bg_c, void* grad_pos, void* grad_q) {
  if (n_atoms == 0) return MIPME_OK;
  const Geom g = make_geom(m);
  MIPME_DISPATCH_STENCIL(
      m->scheme, m->order,
      (gather_grad_kernel<S, N, T><<<blocks_for<N>(n_atoms), 256, 0, st>>>(
          g, n_atoms, m->n_channels, (const T*)pos, (const T*)q, (const T*)gout, (const T*)phi, (const T*)chi,
          (const T*)gsum_dc, (const T*)gscale, T(0.5 / m->volume), T(self_c), T(bg_c), (T*)grad_pos, (T*)grad_q)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// explicit instantiations used by api.hip
template int spread_impl<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, double, void*);
template int spread_impl<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, double, void*);
template int gather_impl<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, void*);
template int gather_impl<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, void*);
template int gather_epilogue_impl<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*,
                                         const void*, const void*, double, double, void*, void*, int, void*);
template int gather_epilogue_impl<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*,
                                          const void*, const void*, double, double, void*, void*, int, void*);
template int gather_grad_impl<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, const void*,
                                     const void*, const void*, const void*, const void*, double, double, void*, void*);
template int gather_grad_impl<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, const void*,
                                      const void*, const void*, const void*, const void*, double, double, void*, void*);

}  // namespace mipme
