// Real-space pair kernels: the short-range pair sum, its adjoint, and the caller-side pair-distance op.
//
// Replaces Calculator._compute_rspace (reference calculators/calculator.py:43-87: sr_from_dist ->
// charges[atom_js] gather -> two index_add_ -> /2, each an ATen op with a P-sized temporary) and the
// elementwise potential functions of potentials/potential.py:59-138, potentials/coulomb.py:80-120,
// potentials/inversepowerlaw.py:55-106.  One pass over the pair stream: 2 indices + 1 distance per pair
// are read once (coalesced, 16-byte index loads), charges are gathered from the L2-resident (N,C) table,
// and contributions are added with hardware float atomics.
#include "common.h"
#include "srpot.h"

namespace mipme {

template <typename I>
__device__ __forceinline__ void load_pair(const I* __restrict__ pairs, int64_t p, int64_t& i, int64_t& j) {
  if constexpr (sizeof(I) == 8) {
    const longlong2 ij = reinterpret_cast<const longlong2*>(pairs)[p];
    i = ij.x;
    j = ij.y;
  } else {
    const int2 ij = reinterpret_cast<const int2*>(pairs)[p];
    i = ij.x;
    j = ij.y;
  }
}

// ---- forward -----------------------------------------------------------------------------------
template <typename T, typename I, int CT>
__global__ __launch_bounds__(256) void rspace_forward_kernel(SRPot s, int64_t P, int Cdyn, const I* __restrict__ pairs,
                                                            const T* __restrict__ dist, const T* __restrict__ q,
                                                            const uint8_t* __restrict__ mask, bool full,
                                                            T* __restrict__ out) {
  const int C = CT > 0 ? CT : Cdyn;
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
    if (mask && !mask[p]) continue;
    int64_t i, j;
    load_pair<I>(pairs, p, i, j);
    T v, dv;
    sr_eval<T, false>(s, dist[p], v, dv);
    v *= T(0.5);
    for (int c = 0; c < C; ++c) {
      atomic_add(out + i * C + c, q[j * C + c] * v);
      if (!full) atomic_add(out + j * C + c, q[i * C + c] * v);
    }
  }
}

// ---- backward ----------------------------------------------------------------------------------
// gscale != NULL ("energy mode"): the upstream gradient is g = gscale * charges, so
// g_i q_j + g_j q_i = 2 gscale q_i q_j and the two gathers of g are not needed.
template <typename T, typename I>
__global__ __launch_bounds__(256) void rspace_backward_kernel(SRPot s, int64_t P, int C, const I* __restrict__ pairs,
                                                             const T* __restrict__ dist, const T* __restrict__ q,
                                                             const uint8_t* __restrict__ mask, bool full,
                                                             const T* __restrict__ g, const T* __restrict__ gscale,
                                                             T* __restrict__ grad_d, T* __restrict__ grad_q) {
  const T gs = gscale ? gscale[0] : T(0);
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
    if (mask && !mask[p]) {
      if (grad_d) grad_d[p] = T(0);
      continue;
    }
    int64_t i, j;
    load_pair<I>(pairs, p, i, j);
    T v, dv;
    sr_eval<T, true>(s, dist[p], v, dv);
    T acc = T(0);
    if (gscale && !grad_q) {
      for (int c = 0; c < C; ++c) acc += q[i * C + c] * q[j * C + c];
      acc *= gs * (full ? T(1) : T(2));
    } else {
      for (int c = 0; c < C; ++c) {
        const T gi = g[i * C + c], gj = g[j * C + c];
        acc += gi * q[j * C + c];
        if (!full) acc += gj * q[i * C + c];
        if (grad_q) {
          atomic_add(grad_q + j * C + c, T(0.5) * v * gi);
          if (!full) atomic_add(grad_q + i * C + c, T(0.5) * v * gj);
        }
      }
    }
    if (grad_d) grad_d[p] = T(0.5) * dv * acc;
  }
}

// ---- pair distances (caller side, tests/helpers.py:278-304) ------------------------------------
template <typename T, typename I>
__global__ __launch_bounds__(256) void distance_forward_kernel(int64_t P, const I* __restrict__ pairs,
                                                              const T* __restrict__ pos, const T* __restrict__ cell,
                                                              const T* __restrict__ shifts, T* __restrict__ out) {
  T A[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = cell ? cell[k] : T(0);
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
    int64_t i, j;
    load_pair<I>(pairs, p, i, j);
    T vx = pos[3 * j] - pos[3 * i], vy = pos[3 * j + 1] - pos[3 * i + 1], vz = pos[3 * j + 2] - pos[3 * i + 2];
    if (shifts) {
      const T sx = shifts[3 * p], sy = shifts[3 * p + 1], sz = shifts[3 * p + 2];
      vx += sx * A[0] + sy * A[3] + sz * A[6];
      vy += sx * A[1] + sy * A[4] + sz * A[7];
      vz += sx * A[2] + sy * A[5] + sz * A[8];
    }
    out[p] = fsqrt(vx * vx + vy * vy + vz * vz);
  }
}


// Same op with the pair stream compressed to 16 bytes per pair: (i, j) as int32 and the three integer cell shifts as
// int8 in one word (the float (P,3) shift tensor alone is 12 of the 28 bytes the generic kernel moves per pair).
template <typename T>
__global__ void pack_pair_shifts_kernel(int64_t P, const T* __restrict__ shifts, int* __restrict__ packed,
                                        int* __restrict__ flag) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int word = 0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const T sh = shifts[3 * p + k];
    const T r = rint(sh);
    bad |= (r != sh) || (r > T(127)) || (r < T(-127));
    word |= (int(r) & 0xff) << (8 * k);
  }
  packed[p] = word;
  if (bad) atomicOr(flag, 1);
}

template <typename T>
__global__ __launch_bounds__(256) void distance_forward_packed_kernel(int64_t P, const int2* __restrict__ pairs,
                                                                     const int* __restrict__ packed,
                                                                     const T* __restrict__ pos, const T* __restrict__ cell,
                                                                     T* __restrict__ out) {
  T A[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = cell[k];
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
    const int2 ij = pairs[p];
    const int w = packed[p];
    const int64_t i = ij.x, j = ij.y;
    const T sx = T((w << 24) >> 24), sy = T((w << 16) >> 24), sz = T((w << 8) >> 24);
    const T vx = pos[3 * j] - pos[3 * i] + (sx * A[0] + sy * A[3] + sz * A[6]);
    const T vy = pos[3 * j + 1] - pos[3 * i + 1] + (sx * A[1] + sy * A[4] + sz * A[7]);
    const T vz = pos[3 * j + 2] - pos[3 * i + 2] + (sx * A[2] + sy * A[5] + sz * A[8]);
    out[p] = fsqrt(vx * vx + vy * vy + vz * vz);
  }
}

template <typename T, typename I, bool CELLGRAD>
__global__ __launch_bounds__(256) void distance_backward_kernel(int64_t P, const I* __restrict__ pairs,
                                                               const T* __restrict__ pos, const T* __restrict__ cell,
                                                               const T* __restrict__ shifts,
                                                               const T* __restrict__ grad_d, T* __restrict__ grad_pos,
                                                               double* __restrict__ partials) {
  T A[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = cell ? cell[k] : T(0);
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; p < P; p += int64_t(gridDim.x) * blockDim.x) {
    int64_t i, j;
    load_pair<I>(pairs, p, i, j);
    T vx = pos[3 * j] - pos[3 * i], vy = pos[3 * j + 1] - pos[3 * i + 1], vz = pos[3 * j + 2] - pos[3 * i + 2];
    T sx = T(0), sy = T(0), sz = T(0);
    if (shifts) {
      sx = shifts[3 * p], sy = shifts[3 * p + 1], sz = shifts[3 * p + 2];
      vx += sx * A[0] + sy * A[3] + sz * A[6];
      vy += sx * A[1] + sy * A[4] + sz * A[7];
      vz += sx * A[2] + sy * A[5] + sz * A[8];
    }
    const T d = fsqrt(vx * vx + vy * vy + vz * vz);
    const T sc = grad_d[p] / d;
    const T gx = sc * vx, gy = sc * vy, gz = sc * vz;
    atomic_add(grad_pos + 3 * j + 0, gx);
    atomic_add(grad_pos + 3 * j + 1, gy);
    atomic_add(grad_pos + 3 * j + 2, gz);
    atomic_add(grad_pos + 3 * i + 0, -gx);
    atomic_add(grad_pos + 3 * i + 1, -gy);
    atomic_add(grad_pos + 3 * i + 2, -gz);
    if constexpr (CELLGRAD) {
      acc[0] += double(sx * gx); acc[1] += double(sx * gy); acc[2] += double(sx * gz);
      acc[3] += double(sy * gx); acc[4] += double(sy * gy); acc[5] += double(sy * gz);
      acc[6] += double(sz * gx); acc[7] += double(sz * gy); acc[8] += double(sz * gz);
    }
  }
  if constexpr (CELLGRAD) {
    __shared__ double red[4][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      double v = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
      partials[int64_t(blockIdx.x) * 9 + threadIdx.x] =
          red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

template <typename T>
__global__ void reduce9_kernel(int nblocks, const double* __restrict__ partials, T* __restrict__ out) {
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] += partials[int64_t(b) * 9 + k];
  __shared__ double red[4][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) out[threadIdx.x] = T(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- launch helpers ----------------------------------------------------------------------------
static constexpr int kPairGridMax = 256 * 16;  // 256 CUs x 16 blocks of 256 threads; grid-stride beyond

static inline unsigned pair_grid(int64_t P) {
  const int64_t b = (P + 255) / 256;
  return unsigned(b < kPairGridMax ? (b > 0 ? b : 1) : kPairGridMax);
}

int64_t pair_partials_blocks(int64_t P) { return pair_grid(P); }

template <typename T, typename I>
int rspace_forward_impl(hipStream_t st, int64_t P, int64_t N, int C, const void* pairs, const void* dist, const void* q,
                        const void* mask, int full, const mipme_potential_t* pot, int accumulate, void* out) {
  SRPot s;
  int rc = make_srpot(pot, s);
  if (rc) return rc;
  if (!accumulate) MIPME_CHECK_HIP(zero_async(out, sizeof(T) * size_t(N) * C, st));
  if (P == 0) return MIPME_OK;
  if (C == 1)
    rspace_forward_kernel<T, I, 1><<<pair_grid(P), 256, 0, st>>>(s, P, C, (const I*)pairs, (const T*)dist, (const T*)q,
                                                                 (const uint8_t*)mask, full != 0, (T*)out);
  else
    rspace_forward_kernel<T, I, 0><<<pair_grid(P), 256, 0, st>>>(s, P, C, (const I*)pairs, (const T*)dist, (const T*)q,
                                                                 (const uint8_t*)mask, full != 0, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
int rspace_backward_impl(hipStream_t st, int64_t P, int64_t N, int C, const void* pairs, const void* dist,
                         const void* q, const void* mask, int full, const mipme_potential_t* pot, const void* g,
                         const void* gscale, void* grad_d, void* grad_q) {
  SRPot s;
  int rc = make_srpot(pot, s);
  if (rc) return rc;
  if (P == 0) return MIPME_OK;
  rspace_backward_kernel<T, I><<<pair_grid(P), 256, 0, st>>>(s, P, C, (const I*)pairs, (const T*)dist, (const T*)q,
                                                             (const uint8_t*)mask, full != 0, (const T*)g,
                                                             (const T*)gscale, (T*)grad_d, (T*)grad_q);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
int distance_forward_impl(hipStream_t st, int64_t P, const void* pairs, const void* pos, const void* cell,
                          const void* shifts, void* out) {
  if (P == 0) return MIPME_OK;
  distance_forward_kernel<T, I><<<pair_grid(P), 256, 0, st>>>(P, (const I*)pairs, (const T*)pos, (const T*)cell,
                                                              (const T*)shifts, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T, typename I>
int distance_backward_impl(hipStream_t st, int64_t P, int64_t N, const void* pairs, const void* pos, const void* cell,
                           const void* shifts, const void* grad_d, void* partials, void* grad_pos, void* grad_cell) {
  MIPME_CHECK_HIP(zero_async(grad_pos, sizeof(T) * size_t(N) * 3, st));
  if (P == 0) {
    if (grad_cell) MIPME_CHECK_HIP(zero_async(grad_cell, sizeof(T) * 9, st));
    return MIPME_OK;
  }
  const unsigned grid = pair_grid(P);
  if (grad_cell) {
    MIPME_REQUIRE(partials != nullptr, "partials scratch required for the cell gradient");
    distance_backward_kernel<T, I, true><<<grid, 256, 0, st>>>(P, (const I*)pairs, (const T*)pos, (const T*)cell,
                                                               (const T*)shifts, (const T*)grad_d, (T*)grad_pos,
                                                               (double*)partials);
    MIPME_LAUNCH_CHECK();
    reduce9_kernel<T><<<1, 256, 0, st>>>(int(grid), (const double*)partials, (T*)grad_cell);
  } else {
    distance_backward_kernel<T, I, false><<<grid, 256, 0, st>>>(P, (const I*)pairs, (const T*)pos, (const T*)cell,
                                                                (const T*)shifts, (const T*)grad_d, (T*)grad_pos,
                                                                nullptr);
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}


template <typename T>
int pack_pair_shifts_impl(hipStream_t st, int64_t P, const void* shifts, void* packed, void* flag) {
  MIPME_CHECK_HIP(zero_async(flag, sizeof(int), st));
  if (P == 0) return MIPME_OK;
  pack_pair_shifts_kernel<T><<<unsigned((P + 255) / 256), 256, 0, st>>>(P, (const T*)shifts, (int*)packed, (int*)flag);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int distance_forward_packed_impl(hipStream_t st, int64_t P, const void* pairs, const void* packed, const void* pos,
                                 const void* cell, void* out) {
  if (P == 0) return MIPME_OK;
  distance_forward_packed_kernel<T><<<pair_grid(P), 256, 0, st>>>(P, (const int2*)pairs, (const int*)packed,
                                                                  (const T*)pos, (const T*)cell, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}
template int pack_pair_shifts_impl<float>(hipStream_t, int64_t, const void*, void*, void*);
template int pack_pair_shifts_impl<double>(hipStream_t, int64_t, const void*, void*, void*);
template int distance_forward_packed_impl<float>(hipStream_t, int64_t, const void*, const void*, const void*, const void*, void*);
template int distance_forward_packed_impl<double>(hipStream_t, int64_t, const void*, const void*, const void*, const void*, void*);

#define MIPME_INST(T, I)                                                                                               \
  template int rspace_forward_impl<T, I>(hipStream_t, int64_t, int64_t, int, const void*, const void*, const void*,    \
                                         const void*, int, const mipme_potential_t*, int, void*);                     \
  template int rspace_backward_impl<T, I>(hipStream_t, int64_t, int64_t, int, const void*, const void*, const void*,   \
                                          const void*, int, const mipme_potential_t*, const void*, const void*, void*, \
                                          void*);                                                                      \
  template int distance_forward_impl<T, I>(hipStream_t, int64_t, const void*, const void*, const void*, const void*,   \
                                           void*);                                                                     \
  template int distance_backward_impl<T, I>(hipStream_t, int64_t, int64_t, const void*, const void*, const void*,      \
                                            const void*, const void*, void*, void*, void*);
MIPME_INST(float, int64_t)
MIPME_INST(float, int32_t)
MIPME_INST(double, int64_t)
MIPME_INST(double, int32_t)

}  // namespace mipme
