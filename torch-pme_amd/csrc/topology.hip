// Pair-list topology ("incidence CSR") and the owner-computes pair kernels built on it.
//
// Why: a half neighbour list makes every pair contribute to two atoms.  The reference does this with
// index_add_ (calculators/calculator.py:78-84) and, for the distance gradient, with autograd's scatter-add
// (tests/helpers.py:286-304).  On MI355X scattered float atomics top out at ~21 G atomics/s regardless of
// scope or table size (measured, tools/atomic_bench.hip) -- 30 M atomics per 32k-atom step = 1.4 ms --
// whereas random 4-byte *reads* of L2-resident per-atom tables run at > 250 G/s.  So the pair list is
// transposed once per list into rows of incident entries per atom:
//     row_ptr[2a]   .. row_ptr[2a+1]  : entries where atom a is the FIRST index  (role i), p ascending
//     row_ptr[2a+1] .. row_ptr[2a+2]  : entries where atom a is the SECOND index (role j), p ascending
//     entry = { other atom (int32), pair index p (int32) }
// and every per-atom result (potentials, charge gradients, position gradients) is produced by the one
// wavefront that owns the atom: coalesced entry stream, gathers of q/positions from L2, no atomics, and a
// deterministic summation order (stable radix sort).
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "rows_body.h"
#include "srpot.h"

namespace mipme {

// ---- build -------------------------------------------------------------------------------------
template <typename I>
__global__ void topo_keys_kernel(int64_t P, const I* __restrict__ pairs, unsigned* __restrict__ keys,
                                 unsigned* __restrict__ vals) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= P) return;
  keys[p] = unsigned(pairs[2 * p]) * 2u;
  vals[p] = unsigned(p);
  keys[P + p] = unsigned(pairs[2 * p + 1]) * 2u + 1u;
  vals[P + p] = unsigned(p);
}

template <typename I>
__global__ void topo_finish_kernel(int64_t E, int64_t N, const I* __restrict__ pairs,
                                   const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                   int* __restrict__ row_ptr, int2* __restrict__ entries) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e > E) return;
  const int64_t kprev = e > 0 ? int64_t(keys[e - 1]) : -1;
  const int64_t kcur = e < E ? int64_t(keys[e]) : 2 * N;
  for (int64_t k = kprev + 1; k <= kcur; ++k) row_ptr[k] = int(e);  // also fills empty rows
  if (e < E) {
    const unsigned p = vals[e];
    const int role = int(keys[e] & 1u);
    entries[e] = make_int2(int(pairs[2 * int64_t(p) + (role ? 0 : 1)]), int(p));
  }
}

struct TopoWorkspace {
  size_t sort_bytes, keys_off, vals_off, keys2_off, vals2_off, sort_off, total;
};

static TopoWorkspace topo_layout(int64_t P) {
  const int64_t E = 2 * P;
  TopoWorkspace w;
  size_t sb = 0;
  unsigned* nul = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, sb, nul, nul, nul, nul, size_t(E > 0 ? E : 1), 0u, 32u, hipStream_t(0), false);
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  const size_t arr = align(size_t(E > 0 ? E : 1) * sizeof(unsigned));
  w.sort_bytes = sb;
  w.keys_off = 0;
  w.vals_off = arr;
  w.keys2_off = 2 * arr;
  w.vals2_off = 3 * arr;
  w.sort_off = 4 * arr;
  w.total = 4 * arr + align(sb);
  return w;
}

template <typename I>
static int topology_build_impl(hipStream_t st, int64_t P, int64_t N, const void* pairs, void* workspace,
                               int64_t ws_bytes, void* row_ptr, void* entries) {
  const int64_t E = 2 * P;
  MIPME_REQUIRE(N < (int64_t(1) << 30) && P < (int64_t(1) << 30), "pair list too large for 32-bit topology");
  if (P == 0) {
    MIPME_CHECK_HIP(zero_async(row_ptr, sizeof(int) * size_t(2 * N + 1), st));
    return MIPME_OK;
  }
  const TopoWorkspace w = topo_layout(P);
  MIPME_REQUIRE(workspace && size_t(ws_bytes) >= w.total, "topology workspace too small: %lld < %zu", (long long)ws_bytes, w.total);
  char* base = (char*)workspace;
  unsigned* keys = (unsigned*)(base + w.keys_off);
  unsigned* vals = (unsigned*)(base + w.vals_off);
  unsigned* keys2 = (unsigned*)(base + w.keys2_off);
  unsigned* vals2 = (unsigned*)(base + w.vals2_off);
  topo_keys_kernel<I><<<unsigned((P + 255) / 256), 256, 0, st>>>(P, (const I*)pairs, keys, vals);
  MIPME_LAUNCH_CHECK();
  unsigned bits = 1;
  while ((int64_t(1) << bits) < 2 * N) ++bits;
  size_t sb = w.sort_bytes;
  MIPME_CHECK_HIP(rocprim::radix_sort_pairs(base + w.sort_off, sb, keys, keys2, vals, vals2, size_t(E), 0u, bits, st, false));
  topo_finish_kernel<I><<<unsigned((E + 1 + 255) / 256), 256, 0, st>>>(E, N, (const I*)pairs, keys2, vals2, (int*)row_ptr,
                                                                      (int2*)entries);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// Pack integer cell shifts into the entry stream: 3 x int8 (little end first) per entry.
// flag[0] is set non-zero if any shift is non-integral or outside [-127, 127].
template <typename T>
__global__ void topo_pack_shifts_kernel(int64_t E, const int2* __restrict__ entries, const T* __restrict__ shifts,
                                        int* __restrict__ packed, int* __restrict__ flag) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t p = entries[e].y;
  int word = 0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const T s = shifts[3 * p + k];
    const T r = rint(s);
    bad |= (r != s) || (r > T(127)) || (r < T(-127));
    word |= (int(r) & 0xff) << (8 * k);
  }
  packed[e] = word;
  if (bad) atomicOr(flag, 1);
}


// (the shift-code formats of the fused kernels, ShiftFormat, are defined in rows_body.h)

template <typename T>
__global__ void topo_pack_entries_kernel(int64_t E, int64_t n_rows, const int* __restrict__ row_ptr,
                                         const int2* __restrict__ entries, const T* __restrict__ shifts, int format,
                                         int2* __restrict__ ent_sh, int* __restrict__ flag) {
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int2 en = entries[e];
  // role of the entry: the half-row r (largest r with row_ptr[r] <= e) is odd for role j
  int64_t lo = 0, hi = n_rows;
  while (lo < hi) {
    const int64_t m = (lo + hi + 1) >> 1;
    if (row_ptr[m] <= e)
      lo = m;
    else
      hi = m - 1;
  }
  const int sgn = (lo & 1) ? -1 : 1;
  int sh[3] = {0, 0, 0};
  int bits = 0;
  if (shifts) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const T v = shifts[3 * int64_t(en.y) + k];
      const T r = rint(v);
      if ((r != v) || (r > T(127)) || (r < T(-127))) bits |= 1;
      sh[k] = sgn * int(r);
      if (sh[k] > kShiftTableRange || sh[k] < -kShiftTableRange) bits |= 2;
    }
  }
  int word;
  if (format == kShiftTable || format == kShiftTable32)
    word = (sh[0] + kShiftTableRange) + kShiftTableBase * ((sh[1] + kShiftTableRange) + kShiftTableBase * (sh[2] + kShiftTableRange));
  else
    word = (sh[0] & 0xff) | ((sh[1] & 0xff) << 8) | ((sh[2] & 0xff) << 16);
  if (format == kShiftTable32) {
    const bool fits = !(bits & 2);  // codes of out-of-range shifts do not fit 9 bits: the caller sees flag bit 1 and discards
    reinterpret_cast<int*>(ent_sh)[e] = en.x | ((fits ? word : 0) << kCompactAtomBits);
  } else {
    ent_sh[e] = make_int2(en.x, word);
  }
  if (bits) atomicOr(flag, bits);
}


template <typename T, int CMAX>
__global__ __launch_bounds__(256) void rspace_rows_kernel(SRPot s, int64_t N, int C, const int* __restrict__ row_ptr,
                                                         const int2* __restrict__ entries, const T* __restrict__ dist,
                                                         const T* __restrict__ src, const uint8_t* __restrict__ mask,
                                                         int role_lo, int role_hi, bool accumulate,
                                                         T* __restrict__ out) {
  constexpr int U = kRowUnroll;
  const int sub = threadIdx.x % kRowLanes;
  int64_t a = int64_t(blockIdx.x) * kRowsPerBlock + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = N - 1;  // keep the lanes alive for the shuffles; nothing is written
  const int beg = row_ptr[2 * a + role_lo], end = valid ? row_ptr[2 * a + role_hi + 1] : beg;
  for (int c0 = 0; c0 < C; c0 += CMAX) {
    T acc[CMAX];
#pragma unroll
    for (int k = 0; k < CMAX; ++k) acc[k] = T(0);
    // software pipeline: the entries of macro-iteration k+1 are requested before the gathers of iteration k are used
    int2 en_next[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = beg + u * kRowLanes + sub;
      en_next[u] = entries[e < end ? e : beg];
    }
    for (int base = beg; base < end; base += kRowLanes * U) {
      int2 en[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = base + u * kRowLanes + sub < end;
        en[u] = en_next[u];
      }
      T d[U], sv[U][CMAX];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (mask) ok[u] = ok[u] && mask[en[u].y];
        d[u] = dist[en[u].y];
#pragma unroll
        for (int k = 0; k < CMAX; ++k) sv[u][k] = (c0 + k < C) ? src[int64_t(en[u].x) * C + c0 + k] : T(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + kRowLanes * U + u * kRowLanes + sub;
        en_next[u] = entries[e < end ? e : beg];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        T v, dv;
        sr_eval<T, false>(s, d[u], v, dv);
        v = ok[u] ? v : T(0);
#pragma unroll
        for (int k = 0; k < CMAX; ++k) acc[k] += sv[u][k] * v;
      }
    }
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const T tot = row_sum(acc[k]);
      if (sub == 0 && valid && c0 + k < C) {
        T* o = out + a * C + c0 + k;
        *o = (accumulate ? *o : T(0)) + T(0.5) * tot;
      }
    }
  }
}


// ---- constant distances: v_SR(d) per row entry, computed once (mipme.h, mipme_rspace_rows_tabulate) ------------------------------
// One record per row entry, in row order: the partner atom and the pair's potential value (zero for a masked pair) -- the
// pair sum over them is a sparse matrix-vector product that streams 8 (fp32) / 16 (fp64) bytes per entry and gathers only the
// partner's source value; no distance gather (64-byte sectors for the scattered role-j entries), no erfc.
template <typename T>
struct EntVal {
  int other;
  T v;
};
static_assert(sizeof(EntVal<float>) == 8 && sizeof(EntVal<double>) == 16, "EntVal layout");

template <typename T>
__global__ __launch_bounds__(256) void rows_tabulate_kernel(SRPot s, int64_t N, const int* __restrict__ row_ptr,
                                                           const int2* __restrict__ entries, const T* __restrict__ dist,
                                                           const uint8_t* __restrict__ mask, int t_lo, int t_hi,
                                                           EntVal<T>* __restrict__ ev, T* __restrict__ row_sum_t) {
  constexpr int U = kRowUnroll;
  const int sub = threadIdx.x % kRowLanes;
  int64_t a = int64_t(blockIdx.x) * kRowsPerBlock + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = N - 1;
  const int beg = row_ptr[2 * a], end = valid ? row_ptr[2 * a + 2] : beg;
  const int s_beg = row_ptr[2 * a + t_lo], s_end = row_ptr[2 * a + t_hi + 1];  // the roles of the transposed sum
  T acc = T(0);
  for (int base = beg; base < end; base += kRowLanes * U) {
    int2 en[U];
    T d[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * kRowLanes + sub;
      ok[u] = e < end;
      en[u] = entries[ok[u] ? e : beg];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) d[u] = dist[en[u].y];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * kRowLanes + sub;
      T v, dv;
      sr_eval<T, false>(s, d[u], v, dv);
      if (mask && !mask[en[u].y]) v = T(0);
      if (ok[u]) {
        ev[e] = EntVal<T>{en[u].x, v};
        if (e >= s_beg && e < s_end) acc += v;
      }
    }
  }
  if (row_sum_t) {
    const T tot = row_sum(acc);
    if (sub == 0 && valid) row_sum_t[a] = T(0.5) * tot;
  }
}

// out[a] (+)= 1/2 sum_{entries of a in the roles} src[other] v_e
template <typename T>
__global__ __launch_bounds__(256) void rows_tabulated_kernel(int64_t N, const int* __restrict__ row_ptr,
                                                            const EntVal<T>* __restrict__ ev, const T* __restrict__ src,
                                                            int role_lo, int role_hi, bool accumulate, T* __restrict__ out) {
  constexpr int U = 4;
  const int sub = threadIdx.x % kRowLanes;
  int64_t a = int64_t(blockIdx.x) * kRowsPerBlock + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = N - 1;
  const int beg = row_ptr[2 * a + role_lo], end = valid ? row_ptr[2 * a + role_hi + 1] : beg;
  T acc = T(0);
  EntVal<T> nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = beg + u * kRowLanes + sub;
    nxt[u] = ev[e < end ? e : beg];
  }
  for (int base = beg; base < end; base += kRowLanes * U) {
    EntVal<T> cur[U];
    T sv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < U; ++u) sv[u] = src[cur[u].other];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + kRowLanes * U + u * kRowLanes + sub;
      nxt[u] = ev[e < end ? e : beg];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + u * kRowLanes + sub < end) acc += sv[u] * cur[u].v;
  }
  const T tot = row_sum(acc);
  if (sub == 0 && valid) out[a] = (accumulate ? out[a] : T(0)) + T(0.5) * tot;
}

// grad_pos[a] = sum_{role i} -(g_p/d_p) vec_p + sum_{role j} +(g_p/d_p) vec_p ,  vec_p = r_j - r_i + S_p A
// grad_cell = sum_p S_p^T (g_p/d_p) vec_p, accumulated from the role-i entries (each pair once).
template <typename T, bool CELLGRAD>
__global__ __launch_bounds__(256) void distance_backward_rows_kernel(int64_t N, const int* __restrict__ row_ptr,
                                                                    const int2* __restrict__ entries,
                                                                    const int* __restrict__ packed,
                                                                    const T* __restrict__ pos, const T* __restrict__ cell,
                                                                    const T* __restrict__ shifts,
                                                                    const T* __restrict__ grad_d,
                                                                    T* __restrict__ grad_pos,
                                                                    double* __restrict__ partials) {
  constexpr int U = kRowUnroll;
  T A[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = cell ? cell[k] : T(0);
  const int sub = threadIdx.x % kRowLanes;
  int64_t a = int64_t(blockIdx.x) * kRowsPerBlock + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = N - 1;
  T gx = T(0), gy = T(0), gz = T(0);
  double cg[9];
  if constexpr (CELLGRAD) {
#pragma unroll
    for (int k = 0; k < 9; ++k) cg[k] = 0.0;
  }
  const T ax = pos[3 * a], ay = pos[3 * a + 1], az = pos[3 * a + 2];
  const int beg = row_ptr[2 * a], mid = row_ptr[2 * a + 1], end = valid ? row_ptr[2 * a + 2] : beg;
  int2 en_next[U];
  int pk_next[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = beg + u * kRowLanes + sub;
    const int ec = e < end ? e : beg;
    en_next[u] = entries[ec];
    pk_next[u] = packed ? packed[ec] : 0;
  }
  for (int base = beg; base < end; base += kRowLanes * U) {
    int2 en[U];
    int pk[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = base + u * kRowLanes + sub < end;
      en[u] = en_next[u];
      pk[u] = pk_next[u];
    }
    T gd[U], ox[U], oy[U], oz[U], shx[U], shy[U], shz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      gd[u] = grad_d[en[u].y];
      ox[u] = pos[3 * int64_t(en[u].x)];
      oy[u] = pos[3 * int64_t(en[u].x) + 1];
      oz[u] = pos[3 * int64_t(en[u].x) + 2];
      if (packed) {
        shx[u] = T(unpack8(pk[u], 0));
        shy[u] = T(unpack8(pk[u], 1));
        shz[u] = T(unpack8(pk[u], 2));
      } else if (shifts) {
        shx[u] = shifts[3 * int64_t(en[u].y)];
        shy[u] = shifts[3 * int64_t(en[u].y) + 1];
        shz[u] = shifts[3 * int64_t(en[u].y) + 2];
      } else {
        shx[u] = shy[u] = shz[u] = T(0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + kRowLanes * U + u * kRowLanes + sub;
      const int ec = e < end ? e : beg;
      en_next[u] = entries[ec];
      pk_next[u] = packed ? packed[ec] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * kRowLanes + sub;
      const T sign = e < mid ? T(-1) : T(1);  // role i: a is the tail of vec (gradient -gvec); role j: +gvec
      const T sx = shx[u], sy = shy[u], sz = shz[u];
      // vec = r_j - r_i + S A ; with o = other atom: role i -> r_o - r_a + S A ; role j -> r_a - r_o + S A
      const T vx = -sign * (ox[u] - ax) + (sx * A[0] + sy * A[3] + sz * A[6]);
      const T vy = -sign * (oy[u] - ay) + (sx * A[1] + sy * A[4] + sz * A[7]);
      const T vz = -sign * (oz[u] - az) + (sx * A[2] + sy * A[5] + sz * A[8]);
      const T d2 = vx * vx + vy * vy + vz * vz;
      const T sc = ok[u] ? gd[u] / fsqrt(d2) : T(0);
      gx += sign * sc * vx;
      gy += sign * sc * vy;
      gz += sign * sc * vz;
      if constexpr (CELLGRAD) {
        if (ok[u] && e < mid) {
          const double px = double(sc * vx), py = double(sc * vy), pz = double(sc * vz);
          cg[0] += double(sx) * px; cg[1] += double(sx) * py; cg[2] += double(sx) * pz;
          cg[3] += double(sy) * px; cg[4] += double(sy) * py; cg[5] += double(sy) * pz;
          cg[6] += double(sz) * px; cg[7] += double(sz) * py; cg[8] += double(sz) * pz;
        }
      }
    }
  }
  gx = row_sum(gx);
  gy = row_sum(gy);
  gz = row_sum(gz);
  if (sub == 0 && valid) {
    grad_pos[3 * a] = gx;
    grad_pos[3 * a + 1] = gy;
    grad_pos[3 * a + 2] = gz;
  }
  if constexpr (CELLGRAD) {
    __shared__ double red[4][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double v = wave_sum(cg[k]);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += red[w][threadIdx.x];
      partials[int64_t(blockIdx.x) * 9 + threadIdx.x] = v;
    }
  }
}


// ---- fused distance + pair-sum row kernels ------------------------------------------------------------------------
// When the distances handed to the calculator were produced by mipme_pair_distance_forward from (positions, cell,
// shifts), the row kernels can recompute d_e = |r_o - r_a + S A| from the L2-resident positions (one gather per entry)
// instead of gathering d[p] (and, backward, grad_d[p] and the shifts) at random from P-sized arrays, and the chain rule
// through d is applied in the same kernel:
//   POT    out[a] (+)= 1/2 sum_{e in potential roles} src[o] v_SR(d_e)
//   FORCE  F[a]    = sum_{all e} sign_e w_e v_SR'(d_e) vec_e / d_e        (sign: -1 role i, +1 role j)
//          w_e = q[o]                                   if g == NULL  (caller multiplies by gE q[a] f: energy mode)
//              = 1/2 (g[a] q[o] + g[o] q[a])            half list
//              = 1/2 g[a] q[o] (role i), 1/2 g[o] q[a] (role j)   full list
//   CELLGRAD partial sums of sum_{role-i e} (q[a] if g == NULL) w_e v' / d  S_e^T vec_e  per block (fp64)

template <typename T>
__global__ void pack_atom_records_kernel(int64_t N, const T* __restrict__ pos, const T* __restrict__ w,
                                         AtomRecord<T>* __restrict__ rec) {
  const int64_t a = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= N) return;
  AtomRecord<T> r;
  r.x = pos[3 * a];
  r.y = pos[3 * a + 1];
  r.z = pos[3 * a + 2];
  r.w = w[a];
  rec[a] = r;
}


template <typename T, int MODE, bool CELLGRAD, int PFAST, bool MASK, bool TABLE>
__global__ __launch_bounds__(256) void sr_fused_rows_kernel(FusedRowsArgs<T> a) {
  MIPME_SKIP_IF_SET(a.skip);
  __shared__ AtomRecord<T> shift_tab[TABLE ? kShiftTableSize : 1];
  sr_fused_rows_body<T, MODE, CELLGRAD, PFAST, MASK, TABLE, 256>(a, blockIdx.x, shift_tab);
}

// 4-byte entry stream (kShiftTable32), potential + force sums: the generic body (distance by-product) and the packed one
template <typename T, int PFAST>
__global__ __launch_bounds__(256) void sr_fused_rows_compact_kernel(FusedRowsArgs<T> a) {
  __shared__ AtomRecord<T> shift_tab[kShiftTableSize];
  sr_fused_rows_body<T, kPotForce, false, PFAST, false, true, 256, 0, true>(a, blockIdx.x, shift_tab);
}
template <int PFAST>
__global__ __launch_bounds__(256) void sr_rows_pk_kernel(FusedRowsArgs<float> a) {
  __shared__ AtomRecord<float> shift_tab[kShiftTableSize];
  sr_rows_pk_body<PFAST, 256>(a, blockIdx.x, shift_tab);
}

// energy mode: grad_pos[a] = gE q[a] (f F[a] + field[a]), grad_cell = f gE sum_b partials[b]   (f = 1/2 for a full list;
// field = the mesh part from the forward gather, nullable, as is force)
template <typename T>
__global__ void sr_fused_finalize_kernel(int64_t N, const T* __restrict__ force, const T* __restrict__ field,
                                         const T* __restrict__ q, const T* __restrict__ gscale, T f, int64_t nblocks,
                                         const double* __restrict__ partials, T* __restrict__ grad_pos,
                                         T* __restrict__ grad_cell) {
  const T ge = gscale[0];
  const T sc = f * ge;
  if (grad_pos) {
    for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < 3 * N; t += int64_t(gridDim.x) * blockDim.x) {
      const T sr = force ? f * force[t] : T(0), lr = field ? field[t] : T(0);
      grad_pos[t] = ge * q[t / 3] * (sr + lr);
    }
  }
  if (grad_cell && blockIdx.x == 0) {
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[k] += partials[b * 9 + k];
    __shared__ double red[4][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double v = wave_sum(acc[k]);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
      double v = 0.0;
      for (int w = 0; w < int(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
      grad_cell[threadIdx.x] = T(double(sc) * v);
    }
  }
}

template <typename T>
__global__ void reduce9_rows_kernel(int64_t nblocks, const double* __restrict__ partials, T* __restrict__ out) {
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  for (int64_t b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] += partials[b * 9 + k];
  __shared__ double red[16][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    double v = 0.0;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
    out[threadIdx.x] = T(v);
  }
}

static inline unsigned row_blocks(int64_t N) { return unsigned((N + kRowsPerBlock - 1) / kRowsPerBlock); }

template <typename T>
static int rspace_rows_impl(hipStream_t st, int64_t N, int C, const void* row_ptr, const void* entries,
                            const void* dist, const void* src, const void* mask, int role_lo, int role_hi,
                            const mipme_potential_t* pot, int accumulate, void* out) {
  SRPot s;
  int rc = make_srpot(pot, s);
  if (rc) return rc;
  if (N == 0) return MIPME_OK;
  if (C == 1)
    rspace_rows_kernel<T, 1><<<row_blocks(N), 256, 0, st>>>(s, N, C, (const int*)row_ptr, (const int2*)entries,
                                                            (const T*)dist, (const T*)src, (const uint8_t*)mask, role_lo,
                                                            role_hi, accumulate != 0, (T*)out);
  else
    rspace_rows_kernel<T, 4><<<row_blocks(N), 256, 0, st>>>(s, N, C, (const int*)row_ptr, (const int2*)entries,
                                                            (const T*)dist, (const T*)src, (const uint8_t*)mask, role_lo,
                                                            role_hi, accumulate != 0, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int distance_backward_rows_impl(hipStream_t st, int64_t N, const void* row_ptr, const void* entries,
                                       const void* packed, const void* pos, const void* cell, const void* shifts,
                                       const void* grad_d, void* partials, void* grad_pos, void* grad_cell) {
  if (N == 0) {
    if (grad_cell) MIPME_CHECK_HIP(zero_async(grad_cell, sizeof(T) * 9, st));
    return MIPME_OK;
  }
  if (grad_cell) {
    MIPME_REQUIRE(partials != nullptr, "partials scratch required for the cell gradient");
    distance_backward_rows_kernel<T, true><<<row_blocks(N), 256, 0, st>>>(
        N, (const int*)row_ptr, (const int2*)entries, (const int*)packed, (const T*)pos, (const T*)cell,
        (const T*)shifts, (const T*)grad_d, (T*)grad_pos, (double*)partials);
    MIPME_LAUNCH_CHECK();
    reduce9_rows_kernel<T><<<1, 1024, 0, st>>>(int64_t(row_blocks(N)), (const double*)partials, (T*)grad_cell);
  } else {
    distance_backward_rows_kernel<T, false><<<row_blocks(N), 256, 0, st>>>(
        N, (const int*)row_ptr, (const int2*)entries, (const int*)packed, (const T*)pos, (const T*)cell,
        (const T*)shifts, (const T*)grad_d, (T*)grad_pos, nullptr);
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}


template <typename T>
static int sr_fused_rows_impl(hipStream_t st, int64_t N, const void* row_ptr, const void* ent_sh, const void* entries,
                              const void* mask, const void* pos, const void* cell, const void* q, const void* src,
                              const void* g, int transpose, int full_list, const mipme_potential_t* pot, int accumulate,
                              int shift_format, void* records, int records_ready, void* out, void* force,
                              void* partials, void* grad_cell, void* dist_out) {
  SRPot s;
  int rc = make_srpot(pot, s);
  if (rc) return rc;
  if (N == 0) {
    if (grad_cell) MIPME_CHECK_HIP(zero_async(grad_cell, sizeof(T) * 9, st));
    return MIPME_OK;
  }
  int lo = 0, hi = 1;
  if (full_list) lo = hi = transpose ? 1 : 0;
  const unsigned grid = row_blocks(N);
  const FastRS cf = make_fast_rs(s);
  const int pfast = fast_rs_exponent(s);
  const int row_flags = shift_format & ~kShiftFormatMask;  // kRowsPadded: rows written by mipme_nl_stream
  shift_format &= kShiftFormatMask;
  MIPME_REQUIRE((shift_format == kShiftPacked || shift_format == kShiftTable || shift_format == kShiftTable32) &&
                    (row_flags & ~kRowsPadded) == 0,
                "invalid shift format %d", shift_format | row_flags);
  MIPME_REQUIRE(!(mask && shift_format != kShiftPacked), "a pair mask needs entries packed in the int8 shift format");
  MIPME_REQUIRE(!(row_flags & kRowsPadded) || (!mask && !dist_out),
                "padded rows (device neighbour stream) carry no pair indices: no pair mask, no distance by-product");
  if (!records_ready) {
    pack_atom_records_kernel<T><<<unsigned((N + 255) / 256), 256, 0, st>>>(
        N, (const T*)pos, (const T*)(out && !force ? src : q), (AtomRecord<T>*)records);
    MIPME_LAUNCH_CHECK();
  }
  const bool want_pot = out != nullptr, want_force = force != nullptr, want_cg = partials != nullptr;
  int mode;
  if (want_pot && !want_force)
    mode = kPot;
  else if (want_pot && want_force) {
    MIPME_REQUIRE(src == q && !g, "potential + force sums are computed together only for src == charges, grad_out == NULL");
    mode = kPotForce;
  } else if (want_force)
    mode = g ? kForceG : kForceQ;
  else {
    set_error("mipme_sr_rows_fused: nothing to compute");
    return MIPME_EINVAL;
  }
  FusedRowsArgs<T> args = make_fused_rows_args<T>(s, cf, N, row_ptr, ent_sh, entries, mask, pos, records, cell, q, g, lo, hi,
                                                  full_list, accumulate, out, force, partials, dist_out,
                                                  shift_format | row_flags);
  args.skip = skip_flag_slot();
  if (shift_format == kShiftTable32) {
    MIPME_REQUIRE(mode == kPotForce && !want_cg && pfast > 0 && N <= kCompactMaxAtoms,
                  "4-byte entries serve the potential + force pass of the Coulomb / dispersion fast paths only");
    static const bool packed_ok = env_flag("MIPME_ROWS_PK", true);
    if constexpr (std::is_same<T, float>::value) {
      if (!dist_out && packed_ok) {
        if (pfast == 1)
          sr_rows_pk_kernel<1><<<grid, 256, 0, st>>>(args);
        else
          sr_rows_pk_kernel<6><<<grid, 256, 0, st>>>(args);
        MIPME_LAUNCH_CHECK();
        return MIPME_OK;
      }
    }
    if (pfast == 1)
      sr_fused_rows_compact_kernel<T, 1><<<grid, 256, 0, st>>>(args);
    else
      sr_fused_rows_compact_kernel<T, 6><<<grid, 256, 0, st>>>(args);
    MIPME_LAUNCH_CHECK();
    return MIPME_OK;
  }
#define MIPME_FUSED_LAUNCH_(MODE, CG, CF, MK, TB) sr_fused_rows_kernel<T, MODE, CG, CF, MK, TB><<<grid, 256, 0, st>>>(args)
#define MIPME_FUSED_MK(MODE, CG, CF)                                                                                  \
  do {                                                                                                                \
    if (mask)                                                                                                         \
      MIPME_FUSED_LAUNCH_(MODE, CG, CF, true, false);                                                                 \
    else if (shift_format == kShiftTable)                                                                             \
      MIPME_FUSED_LAUNCH_(MODE, CG, CF, false, true);                                                                 \
    else                                                                                                              \
      MIPME_FUSED_LAUNCH_(MODE, CG, CF, false, false);                                                                \
  } while (0)
#define MIPME_FUSED_CF(MODE, CG)                                                                                      \
  do {                                                                                                                \
    if (pfast == 1)                                                                                                   \
      MIPME_FUSED_MK(MODE, CG, 1);                                                                                    \
    else if (pfast == 6)                                                                                              \
      MIPME_FUSED_MK(MODE, CG, 6);                                                                                    \
    else                                                                                                              \
      MIPME_FUSED_MK(MODE, CG, 0);                                                                                    \
  } while (0)
#define MIPME_FUSED_CG(MODE)                                                                                          \
  do {                                                                                                                \
    if (want_cg)                                                                                                      \
      MIPME_FUSED_CF(MODE, true);                                                                                     \
    else                                                                                                              \
      MIPME_FUSED_CF(MODE, false);                                                                                    \
  } while (0)
  switch (mode) {
    case kPot: MIPME_FUSED_CF(kPot, false); break;
    case kPotForce: MIPME_FUSED_CG(kPotForce); break;
    case kForceQ: MIPME_FUSED_CG(kForceQ); break;
    default: MIPME_FUSED_CG(kForceG); break;
  }
#undef MIPME_FUSED_CG
#undef MIPME_FUSED_CF
#undef MIPME_FUSED_MK
#undef MIPME_FUSED_LAUNCH_
  MIPME_LAUNCH_CHECK();
  if (grad_cell) {
    reduce9_rows_kernel<T><<<1, 1024, 0, st>>>(int64_t(grid), (const double*)partials, (T*)grad_cell);
    MIPME_LAUNCH_CHECK();
  }
  return MIPME_OK;
}

template <typename T>
static int sr_fused_finalize_impl(hipStream_t st, int64_t N, const void* force, const void* field, const void* q,
                                  const void* gscale, int full_list, const void* partials, void* grad_pos,
                                  void* grad_cell) {
  if (N == 0) {
    if (grad_cell) MIPME_CHECK_HIP(zero_async(grad_cell, sizeof(T) * 9, st));
    return MIPME_OK;
  }
  const int64_t want = (3 * N + 255) / 256;
  const unsigned grid = unsigned(want < 2048 ? want : 2048);
  sr_fused_finalize_kernel<T><<<grid, 256, 0, st>>>(N, (const T*)force, (const T*)field, (const T*)q, (const T*)gscale,
                                                    full_list ? T(0.5) : T(1), int64_t(row_blocks(N)),
                                                    (const double*)partials, (T*)grad_pos, (T*)grad_cell);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // namespace mipme

using namespace mipme;

extern "C" {

int64_t mipme_topology_workspace_bytes(int64_t n_pairs) { return int64_t(topo_layout(n_pairs).total); }

int mipme_topology_build(void* stream, int idx_dtype, int64_t n_pairs, int64_t n_atoms, const void* pairs,
                         void* workspace, int64_t workspace_bytes, void* row_ptr, void* entries) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0 && row_ptr, "invalid arguments to mipme_topology_build");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && entries), "NULL buffer passed to mipme_topology_build");
  hipStream_t st = (hipStream_t)stream;
  if (idx_dtype == MIPME_I64)
    return topology_build_impl<int64_t>(st, n_pairs, n_atoms, pairs, workspace, workspace_bytes, row_ptr, entries);
  if (idx_dtype == MIPME_I32)
    return topology_build_impl<int32_t>(st, n_pairs, n_atoms, pairs, workspace, workspace_bytes, row_ptr, entries);
  set_error("invalid index dtype %d", idx_dtype);
  return MIPME_EINVAL;
}

int mipme_topology_pack_shifts(void* stream, int dtype, int64_t n_pairs, const void* entries, const void* shifts,
                               void* packed, void* flag) {
  MIPME_REQUIRE(n_pairs >= 0 && flag, "invalid arguments to mipme_topology_pack_shifts");
  hipStream_t st = (hipStream_t)stream;
  MIPME_CHECK_HIP(zero_async(flag, sizeof(int), st));
  const int64_t E = 2 * n_pairs;
  if (E == 0) return MIPME_OK;
  MIPME_REQUIRE(entries && shifts && packed, "NULL buffer passed to mipme_topology_pack_shifts");
  if (dtype == MIPME_F32)
    topo_pack_shifts_kernel<float><<<unsigned((E + 255) / 256), 256, 0, st>>>(E, (const int2*)entries, (const float*)shifts,
                                                                              (int*)packed, (int*)flag);
  else if (dtype == MIPME_F64)
    topo_pack_shifts_kernel<double><<<unsigned((E + 255) / 256), 256, 0, st>>>(E, (const int2*)entries, (const double*)shifts,
                                                                               (int*)packed, (int*)flag);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_rspace_rows(void* stream, int dtype, int64_t n_atoms, int n_channels, const void* row_ptr, const void* entries,
                      const void* dist, const void* src, const void* pair_mask, int transpose, int full_list,
                      const mipme_potential_t* pot, int accumulate, void* out) {
  MIPME_REQUIRE(n_atoms >= 0 && n_channels > 0 && row_ptr, "invalid arguments to mipme_rspace_rows");
  MIPME_REQUIRE(n_atoms == 0 || (out && src), "NULL buffer passed to mipme_rspace_rows");
  // forward: role i (+ j for a half list); transposed (charge gradient): role j (+ i for a half list)
  int lo, hi;
  if (!full_list) {
    lo = 0;
    hi = 1;
  } else {
    lo = hi = transpose ? 1 : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return rspace_rows_impl<float>(st, n_atoms, n_channels, row_ptr, entries, dist, src, pair_mask, lo, hi, pot, accumulate, out);
  if (dtype == MIPME_F64)
    return rspace_rows_impl<double>(st, n_atoms, n_channels, row_ptr, entries, dist, src, pair_mask, lo, hi, pot, accumulate, out);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int64_t mipme_rspace_rows_value_bytes(int dtype, int64_t n_pairs) {
  if (n_pairs < 0 || (dtype != MIPME_F32 && dtype != MIPME_F64)) return 0;
  return (2 * n_pairs + 1) * int64_t(dtype == MIPME_F32 ? sizeof(EntVal<float>) : sizeof(EntVal<double>));
}

int mipme_rspace_rows_tabulate(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* entries, const void* dist,
                               const void* pair_mask, int full_list, const mipme_potential_t* pot, void* values,
                               void* row_sum_transposed) {
  MIPME_REQUIRE(n_atoms >= 0 && row_ptr && entries && values && (n_atoms == 0 || dist), "invalid arguments to mipme_rspace_rows_tabulate");
  MIPME_REQUIRE(dtype == MIPME_F32 || dtype == MIPME_F64, "invalid dtype %d", dtype);
  SRPot s;
  int rc = make_srpot(pot, s);
  if (rc) return rc;
  if (n_atoms == 0) return MIPME_OK;
  hipStream_t st = (hipStream_t)stream;
  const int t_lo = full_list ? 1 : 0, t_hi = 1;  // roles of the transposed sum (mipme_rspace_rows)
  if (dtype == MIPME_F32)
    rows_tabulate_kernel<float><<<row_blocks(n_atoms), 256, 0, st>>>(s, n_atoms, (const int*)row_ptr, (const int2*)entries,
                                                                    (const float*)dist, (const uint8_t*)pair_mask, t_lo, t_hi,
                                                                    (EntVal<float>*)values, (float*)row_sum_transposed);
  else
    rows_tabulate_kernel<double><<<row_blocks(n_atoms), 256, 0, st>>>(s, n_atoms, (const int*)row_ptr, (const int2*)entries,
                                                                     (const double*)dist, (const uint8_t*)pair_mask, t_lo, t_hi,
                                                                     (EntVal<double>*)values, (double*)row_sum_transposed);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_rspace_rows_tabulated(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* values, const void* src,
                                int transpose, int full_list, int accumulate, void* out) {
  MIPME_REQUIRE(n_atoms >= 0 && row_ptr && values, "invalid arguments to mipme_rspace_rows_tabulated");
  MIPME_REQUIRE(n_atoms == 0 || (out && src), "NULL buffer passed to mipme_rspace_rows_tabulated");
  MIPME_REQUIRE(dtype == MIPME_F32 || dtype == MIPME_F64, "invalid dtype %d", dtype);
  if (n_atoms == 0) return MIPME_OK;
  int lo = 0, hi = 1;
  if (full_list) lo = hi = transpose ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    rows_tabulated_kernel<float><<<row_blocks(n_atoms), 256, 0, st>>>(n_atoms, (const int*)row_ptr, (const EntVal<float>*)values,
                                                                     (const float*)src, lo, hi, accumulate != 0, (float*)out);
  else
    rows_tabulated_kernel<double><<<row_blocks(n_atoms), 256, 0, st>>>(n_atoms, (const int*)row_ptr, (const EntVal<double>*)values,
                                                                      (const double*)src, lo, hi, accumulate != 0, (double*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_pair_distance_backward_rows(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* entries,
                                      const void* packed_shifts, const void* positions, const void* cell,
                                      const void* shifts, const void* grad_dist, void* partials, void* grad_positions,
                                      void* grad_cell) {
  MIPME_REQUIRE(n_atoms >= 0 && row_ptr, "invalid arguments to mipme_pair_distance_backward_rows");
  MIPME_REQUIRE(n_atoms == 0 || (positions && grad_dist && grad_positions), "NULL buffer passed to mipme_pair_distance_backward_rows");
  MIPME_REQUIRE((cell == nullptr) == (shifts == nullptr && packed_shifts == nullptr), "`cell` and shifts must be given together");
  MIPME_REQUIRE(!grad_cell || cell, "cell gradient requested without a cell");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return distance_backward_rows_impl<float>(st, n_atoms, row_ptr, entries, packed_shifts, positions, cell, shifts,
                                              grad_dist, partials, grad_positions, grad_cell);
  if (dtype == MIPME_F64)
    return distance_backward_rows_impl<double>(st, n_atoms, row_ptr, entries, packed_shifts, positions, cell, shifts,
                                               grad_dist, partials, grad_positions, grad_cell);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_topology_pack_entries(void* stream, int dtype, int64_t n_pairs, int64_t n_atoms, const void* row_ptr,
                                const void* entries, const void* shifts, int shift_format, void* entries_shift,
                                void* flag) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0 && flag && row_ptr, "invalid arguments to mipme_topology_pack_entries");
  MIPME_REQUIRE(shift_format == kShiftPacked || shift_format == kShiftTable || shift_format == kShiftTable32,
                "invalid shift format %d", shift_format);
  MIPME_REQUIRE(shift_format != kShiftTable32 || n_atoms <= kCompactMaxAtoms,
                "the 32-bit entry format holds atom indices below 2^22, got %lld atoms", (long long)n_atoms);
  hipStream_t st = (hipStream_t)stream;
  MIPME_CHECK_HIP(zero_async(flag, sizeof(int), st));
  const int64_t E = 2 * n_pairs;
  if (E == 0) return MIPME_OK;
  MIPME_REQUIRE(entries && entries_shift, "NULL buffer passed to mipme_topology_pack_entries");
  const unsigned grid = unsigned((E + 255) / 256);
  if (dtype == MIPME_F32)
    topo_pack_entries_kernel<float><<<grid, 256, 0, st>>>(E, 2 * n_atoms, (const int*)row_ptr, (const int2*)entries,
                                                          (const float*)shifts, shift_format, (int2*)entries_shift,
                                                          (int*)flag);
  else if (dtype == MIPME_F64)
    topo_pack_entries_kernel<double><<<grid, 256, 0, st>>>(E, 2 * n_atoms, (const int*)row_ptr, (const int2*)entries,
                                                           (const double*)shifts, shift_format, (int2*)entries_shift,
                                                           (int*)flag);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_sr_rows_fused(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* entries_shift,
                        const void* entries, const void* pair_mask, const void* positions, const void* cell,
                        const void* charges, const void* src, const void* grad_out, int transpose, int full_list,
                        const mipme_potential_t* pot, int accumulate, int shift_format, void* records,
                        int records_ready, void* out, void* force, void* partials, void* grad_cell, void* dist_out) {
  MIPME_REQUIRE(n_atoms >= 0 && row_ptr, "invalid arguments to mipme_sr_rows_fused");
  MIPME_REQUIRE(!dist_out || (out && !pair_mask && !transpose && entries),
                "`dist_out` is written by the potential pass only (no pair mask, not transposed) and needs the entry table");
  MIPME_REQUIRE(n_atoms == 0 || (entries_shift && positions && charges && records), "NULL buffer passed to mipme_sr_rows_fused");
  MIPME_REQUIRE(!pair_mask || entries, "`pair_mask` needs the (other, pair) entry table");
  MIPME_REQUIRE(!out || src, "`src` is required for the potential sum");
  MIPME_REQUIRE(!partials || force, "cell partial sums are produced together with the force sums");
  MIPME_REQUIRE(!grad_cell || (partials && grad_out), "grad_cell needs partials and an explicit upstream gradient");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return sr_fused_rows_impl<float>(st, n_atoms, row_ptr, entries_shift, entries, pair_mask, positions, cell, charges,
                                     src, grad_out, transpose, full_list, pot, accumulate, shift_format, records, records_ready, out, force, partials,
                                     grad_cell, dist_out);
  if (dtype == MIPME_F64)
    return sr_fused_rows_impl<double>(st, n_atoms, row_ptr, entries_shift, entries, pair_mask, positions, cell, charges,
                                      src, grad_out, transpose, full_list, pot, accumulate, shift_format, records, records_ready, out, force, partials,
                                     grad_cell, dist_out);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_sr_rows_finalize(void* stream, int dtype, int64_t n_atoms, const void* force, const void* field,
                           const void* charges, const void* grad_scale, int full_list, const void* partials,
                           void* grad_positions, void* grad_cell) {
  MIPME_REQUIRE(n_atoms >= 0 && grad_scale, "invalid arguments to mipme_sr_rows_finalize");
  MIPME_REQUIRE(n_atoms == 0 || !grad_positions || ((force || field) && charges), "NULL buffer passed to mipme_sr_rows_finalize");
  MIPME_REQUIRE(!grad_cell || partials, "grad_cell needs the partial sums of mipme_sr_rows_fused");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return sr_fused_finalize_impl<float>(st, n_atoms, force, field, charges, grad_scale, full_list, partials, grad_positions, grad_cell);
  if (dtype == MIPME_F64)
    return sr_fused_finalize_impl<double>(st, n_atoms, force, field, charges, grad_scale, full_list, partials, grad_positions, grad_cell);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int64_t mipme_rows_partials_size(int64_t n_atoms) { return 9 * int64_t(row_blocks(n_atoms)); }

}  // extern "C"
