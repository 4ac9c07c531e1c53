// Live bins: the particle <-> mesh kernels of an MD-like loop (mipme_md_rebin / mipme_md_step, csrc/api.hip).
// Shared device bodies: bricks_device.h.
#include "bricks_device.h"

namespace mipme {
bool sr_job_fusable(const mipme_sr_job_t* job);  // (bricks.hip)
// ==========================================================================================================================
// Live bins: the particle <-> mesh kernels of an MD-like loop (mipme_md_rebin / mipme_md_step, csrc/api.hip)
// --------------------------------------------------------------------------------------------------------------------------
// Between two refreshes of its neighbour list an MD step changes the positions by a fraction of a mesh spacing, yet the step
// above bins every atom again (7 us of pure latency at cfg3, on the critical path), and every brick of the spread scans its 27
// neighbours for the atoms whose stencils reach it (two more memory round trips and a round of LDS atomics).  Here both
// belong to the REFRESH, like the pair list: mipme_md_rebin bins the atoms once and writes, per brick, the list of atoms whose
// stencil can reach the brick while the atom stays within kLiveMargin mesh points of where it was binned.  A step then
//   spread   reads its brick's list, fetches the atoms' CURRENT (x, y, z, q) records, evaluates their 1-D weights on the fly and
//            accumulates as before (an atom whose stencil no longer overlaps contributes zeros);
//   gather   walks the brick's home atoms (bin slots), evaluates weights and derivatives on the fly from the current record and
//            reads a halo tile that is kLiveMargin points wider on every side; an atom that has moved further than the margin
//            sets a flag in pinned host memory (the results of that step are then invalid: refresh sooner).
// Every position-dependent quantity is recomputed every step; only the atom -> brick bookkeeping is reused.  One channel.
static constexpr int kLiveMargin = 1;
enum LiveFlags { kLiveListOverflow = 1, kLiveMoved = 2 };

struct LiveLists {
  int* counters;  // [nb + 1] binning counters of the rebin (zero outside it)
  int* count;     // [nb] atoms in the brick's list
  int* atoms;     // [nb][lcap]
  int lcap;
  int4* rec_now;  // [slots] {current mesh coordinates, atom} of the atom in the slot, written by every step's spread
  int4* home_rec;  // [N] {mesh coordinates at the rebin, bin slot} of every atom
  int* host_flags;  // pinned int32, nullable: LiveFlags
};

static inline int live_list_capacity(const mipme_mesh_t* m, int64_t N) {
  double frac = 1.0;
  const int ns[3] = {m->nx, m->ny, m->nz};
  for (int d = 0; d < 3; ++d) frac *= std::min(1.0, double(BRICK + m->order - 1 + 2 * kLiveMargin) / ns[d]);
  const int64_t want = (int64_t(2.0 * frac * double(N)) + 64 + 63) / 64 * 64;
  const int64_t all = (N + 63) / 64 * 64;
  return int(std::min<int64_t>(want, std::max<int64_t>(all, 64)));
}
static inline int64_t live_slots(const mipme_mesh_t* m, int64_t N) {
  const BrickGeom bg = make_brick_geom(m);
  return int64_t(bg.nb) * bin_capacity(bg.nb, N) + N;
}
int64_t live_lists_ints(const mipme_mesh_t* m, int64_t N) {
  const BrickGeom bg = make_brick_geom(m);
  const int64_t head = ((8 + (bg.nb + 1) + bg.nb + int64_t(bg.nb) * live_list_capacity(m, N)) + 3) / 4 * 4;
  return head + 4 * live_slots(m, N) + 4 * N;
}
static inline LiveLists live_view(const mipme_mesh_t* m, int64_t N, void* lists, void* host_flags) {
  const BrickGeom bg = make_brick_geom(m);
  int* b = (int*)lists;
  LiveLists l;
  l.counters = b + 8;
  l.count = l.counters + (bg.nb + 1);
  l.atoms = l.count + bg.nb;
  l.lcap = live_list_capacity(m, N);
  const int64_t head = ((8 + (bg.nb + 1) + bg.nb + int64_t(bg.nb) * l.lcap) + 3) / 4 * 4;  // 16-byte aligned
  l.rec_now = (int4*)(b + head);
  l.home_rec = l.rec_now + live_slots(m, N);
  l.host_flags = (int*)host_flags;
  return l;
}

// current mesh coordinates and 1-D weights (DERIV: and their derivatives) of an atom record; the scheme is a run-time switch
// where an order exists in both (3..5).  One call per axis with its own arrays: a [3][N] array indexed by the axis went to
// scratch memory.
template <int N, bool DERIV, typename T>
__device__ __forceinline__ void live_axis(int scheme, T x, T (&w)[N], T (&dw)[N]) {
  if constexpr (N <= 2) {
    weights_1d<MIPME_P3M, N, DERIV, T>(x, w, dw);
  } else if constexpr (N >= 6) {
    weights_1d<MIPME_LAGRANGE, N, DERIV, T>(x, w, dw);
  } else {
    if (scheme == MIPME_P3M)
      weights_1d<MIPME_P3M, N, DERIV, T>(x, w, dw);
    else
      weights_1d<MIPME_LAGRANGE, N, DERIV, T>(x, w, dw);
  }
}
template <int N, typename T>
__device__ __forceinline__ void live_coords(const Geom& g, const AtomRecord<T>& r, int& mx, int& my, int& mz, T& x0, T& x1, T& x2) {
  const double rx = double(r.x), ry = double(r.y), rz = double(r.z);
  const double ux = double(g.nx) * (rx * g.inv[0] + ry * g.inv[3] + rz * g.inv[6]);
  const double uy = double(g.ny) * (rx * g.inv[1] + ry * g.inv[4] + rz * g.inv[7]);
  const double uz = double(g.nz) * (rx * g.inv[2] + ry * g.inv[5] + rz * g.inv[8]);
  int m;
  double x;
  split_runtime(ux, (N % 2) == 0, m, x);
  mx = posmod(m, g.nx);
  x0 = T(x);
  split_runtime(uy, (N % 2) == 0, m, x);
  my = posmod(m, g.ny);
  x1 = T(x);
  split_runtime(uz, (N % 2) == 0, m, x);
  mz = posmod(m, g.nz);
  x2 = T(x);
}

// ---- rebin: slots (no weights), snapshot, per-brick lists -------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void live_bin_kernel(Geom g, BrickGeom bg, bool even, BinIndex bi, int* __restrict__ counters,
                                                      int64_t Natoms, const AtomRecord<T>* __restrict__ rec4,
                                                      int* __restrict__ over_brick, int4* __restrict__ rec,
                                                      int4* __restrict__ home_rec) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= Natoms) return;  // (whole-wave exits aside, the ballots below see the exec mask of the remaining lanes)
  const AtomRecord<T> r = rec4[i];
  const T p3[3] = {r.x, r.y, r.z};
  int m[3];
  double x[3];
  atom_mesh_coords<T>(g, even, p3, 0, m, x);
  const int b = ((m[0] / BRICK) * bg.nby + m[1] / BRICK) * bg.nbz + m[2] / BRICK;
  // lanes of the wave that fall into the same brick share one returning atomic (as bin_atoms_body)
  const int lane = threadIdx.x & 63;
  unsigned long long remaining = __ballot(true);
  int my_leader = lane, my_rank = 0, my_count = 1;
  while (remaining) {
    const int leader = __ffsll((long long)remaining) - 1;
    const int b0 = __shfl(b, leader, 64);
    const unsigned long long peers = __ballot(b == b0) & remaining;
    if (b == b0) {
      my_leader = leader;
      my_rank = __popcll(peers & ((1ull << lane) - 1ull));
      my_count = __popcll(peers);
    }
    remaining &= ~peers;
  }
  int base = 0;
  if (my_leader == lane) base = atomicAdd(&counters[b], my_count);
  base = __shfl(base, my_leader, 64);
  const int slot = base + my_rank;
  int64_t dst;
  if (slot < bi.cap) {
    dst = int64_t(b) * bi.cap + slot;
  } else {
    const int k = atomicAdd(&counters[bi.nb], 1);
    over_brick[k] = b;
    dst = bi.over_base + k;
  }
  rec[dst] = make_int4(m[0], m[1], m[2], int(i));
  home_rec[i] = make_int4(m[0], m[1], m[2], int(dst));
}

__global__ void live_snapshot_kernel(BinIndex bi, int* __restrict__ counters) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > bi.nb) return;
  const int c = counters[b];
  bi.snap[b] = b == bi.nb ? c : min(c, bi.cap);
  counters[b] = 0;
}

// atoms whose stencil -- started anywhere within kLiveMargin points of where it starts now -- reaches this brick, from the bins
// of the 27 surrounding bricks (+ the overflow region); the list is then ordered by atom index, so that the spread's sums run in
// one fixed order whatever the order of the atomics was
__device__ __forceinline__ bool live_reach(int m, int s0, int origin, int nmesh, int order) {
  const int r = rel_start(m, s0, origin, nmesh, order);
  return r <= BRICK - 1 + kLiveMargin || r >= nmesh - order + 1 - kLiveMargin;
}

template <int N>
__global__ __launch_bounds__(SPREAD_THREADS) void live_lists_kernel(Geom g, BrickGeom bg, BinIndex bins,
                                                                   const int4* __restrict__ rec, LiveLists ll) {
  __shared__ int n_list;
  __shared__ int keys[4096];
  const unsigned block = brick_of(bg, blockIdx.x);
  if (block >= unsigned(bg.nb)) return;
  int bx, by, bz;
  brick_coords(bg, block, bx, by, bz);
  const int ox = bx * BRICK, oy = by * BRICK, oz = bz * BRICK;
  const int tid = threadIdx.x, sub = tid % SPREAD_GROUP, grp = tid / SPREAD_GROUP;
  constexpr int s0 = stencil_start<N>();
  int* __restrict__ out = ll.atoms + int64_t(block) * ll.lcap;
  if (tid == 0) n_list = 0;
  __syncthreads();
  if (grp < 28) {
    int start, len;
    if (grp < 27) {
      const int dx = grp / 9 - 1, dy = (grp / 3) % 3 - 1, dz = grp % 3 - 1;
      const int nbr = (wrap1(bx + dx, bg.nbx) * bg.nby + wrap1(by + dy, bg.nby)) * bg.nbz + wrap1(bz + dz, bg.nbz);
      start = nbr * bins.cap;
      len = bins.snap[nbr];
    } else {
      start = int(bins.over_base);
      len = bins.snap[bins.nb];
    }
    for (int k = sub; k < len; k += SPREAD_GROUP) {
      const int4 a = rec[start + k];
      if (live_reach(a.x, s0, ox, g.nx, N) && live_reach(a.y, s0, oy, g.ny, N) && live_reach(a.z, s0, oz, g.nz, N)) {
        const int dst = atomicAdd(&n_list, 1);
        if (dst < ll.lcap) out[dst] = a.w;
      }
    }
  }
  __syncthreads();
  const int n = n_list;
  if (n > ll.lcap) {
    if (tid == 0) {
      ll.count[block] = ll.lcap;
      if (ll.host_flags) atomicOr(ll.host_flags, kLiveListOverflow);
    }
    return;
  }
  if (tid == 0) ll.count[block] = n;
  if (n > 1 && n <= 4096) {  // rank by counting on the (unique) atom index
    for (int k = tid; k < n; k += SPREAD_THREADS) keys[k] = out[k];
    __syncthreads();
    for (int k = tid; k < n; k += SPREAD_THREADS) {
      const int me = keys[k];
      int r = 0;
      for (int v = 0; v < n; ++v) r += keys[v] < me;
      out[r] = me;
    }
  }
}

// ---- step: spread from the lists -----------------------------------------------------------------------------------------
template <typename T>
struct LiveSpreadArgs {
  Geom g;
  BrickGeom bg;
  int scheme;
  const int* count;
  const int* atoms;
  int lcap;
  const AtomRecord<T>* rec4;
  T* mesh;
  int stage_rows;
  // home atoms of the brick: the spread also leaves their current mesh coordinates and 6 N weights / derivatives in the bins
  // (what the binning pass of the ordinary step writes), for this step's gather
  int64_t n_atoms;
  const int4* home_rec;  // [N] {mesh coordinates at the rebin, bin slot} per atom
  int4* rec_now;
  T* wts;
  int* host_flags;
};

// One thread per atom: current mesh coordinates and the 6 N weights / derivatives into the atom's bin slot -- what the binning pass
// of the ordinary step leaves there for the gather -- and the check that it has not moved further than the margin the lists were
// built with.  These workgroups sit at the FRONT of the spread's grid.  (Evaluating the weights in the gather instead, in each of
// the 8 lanes of an atom, cost that kernel 4 us; here the 4 MB of stores cost the spread 2.7 us, wherever in the kernel they are
// issued -- by the brick workgroups for their home atoms, on an otherwise idle wave of those, or here.)
template <int N, typename T>
__device__ __forceinline__ void live_home_body(const LiveSpreadArgs<T>& args, unsigned wg) {
  const int64_t i = int64_t(wg) * SPREAD_THREADS + threadIdx.x;
  if (i >= args.n_atoms) return;
  const Geom& g = args.g;
  const AtomRecord<T> r = args.rec4[i];
  const int4 was = args.home_rec[i];  // {mesh coordinates at the rebin, slot}
  int mx, my, mz;
  T x0, x1, x2;
  live_coords<N, T>(g, r, mx, my, mz, x0, x1, x2);
  T wx[N], wy[N], wz[N], dwx[N], dwy[N], dwz[N];
  live_axis<N, true, T>(args.scheme, x0, wx, dwx);
  live_axis<N, true, T>(args.scheme, x1, wy, dwy);
  live_axis<N, true, T>(args.scheme, x2, wz, dwz);
  auto far = [](int now, int then, int n) {
    int d = now - then;
    d = d > n / 2 ? d - n : (d < -(n / 2) ? d + n : d);
    return d > kLiveMargin || d < -kLiveMargin;
  };
  if ((far(mx, was.x, g.nx) || far(my, was.y, g.ny) || far(mz, was.z, g.nz)) && args.host_flags) atomicOr(args.host_flags, kLiveMoved);
  const int64_t slot = was.w;
  args.rec_now[slot] = make_int4(mx, my, mz, int(i));
  store_slot_weights<N, T>(args.wts + slot * wts_stride<N, T>(), wx, wy, wz, dwx, dwy, dwz);
}

template <int N, typename T>
__device__ __forceinline__ void live_spread_body(const LiveSpreadArgs<T>& args, unsigned block) {
  constexpr int THREADS = SPREAD_THREADS, WAVES = SPREAD_WAVES;
  const Geom& g = args.g;
  const BrickGeom& bg = args.bg;
  const int stage_rows = args.stage_rows;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int SW = 3 * BRICK;
  constexpr int PAD_ROWS = MIPME_LIVE_PADROWS ? (MIPME_SPREAD_UC - 1) * WAVES : 0;
  T* stage = reinterpret_cast<T*>(smem_raw);  // [stage_rows + PAD_ROWS][SW]
  T* part = stage;                            // [waves][512] partial bricks (aliases the stage, phase R)
  int bx, by, bz;
  brick_coords(bg, block, bx, by, bz);
  const int ox = bx * BRICK, oy = by * BRICK, oz = bz * BRICK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ns = min(args.count[block], args.lcap);
  const int* __restrict__ latoms = args.atoms + int64_t(block) * args.lcap;
  constexpr int s0 = stencil_start<N>();
  const int px = lane >> 3, py = lane & 7;
  const int64_t plane = int64_t(g.ny) * g.nz;
  T acc[BRICK];
#pragma unroll
  for (int k = 0; k < BRICK; ++k) acc[k] = T(0);
  for (int chunk = 0; chunk < ns; chunk += stage_rows) {
    const int nst = min(stage_rows, ns - chunk);
    if (tid < nst) {
      const int atom = latoms[chunk + tid];
      const AtomRecord<T> r = args.rec4[atom];
      int mx, my, mz;
      T x0, x1, x2;
      live_coords<N, T>(g, r, mx, my, mz, x0, x1, x2);
      T wx[N], wy[N], wz[N], unused[N];
      live_axis<N, false, T>(args.scheme, x0, wx, unused);
      live_axis<N, false, T>(args.scheme, x1, wy, unused);
      live_axis<N, false, T>(args.scheme, x2, wz, unused);
      // row = [wz | wx * q | wy], each placed on the brick's 8 points of its axis (zero where the stencil has no point)
      const int rz = rel_start(mz, s0, oz, g.nz, N), rx = rel_start(mx, s0, ox, g.nx, N), ry = rel_start(my, s0, oy, g.ny, N);
      T* dst = stage + tid * SW;
      // zeros, then the stencil's weights at their places (as in spread_brick_body: 3 x N conditional stores, not 3 x 8 x N selects)
#pragma unroll
      for (int k = 0; k < SW; ++k) dst[k] = T(0);
#pragma unroll
      for (int t = 0; t < N; ++t) {
        if (unsigned(rz + t) < unsigned(BRICK)) dst[rz + t] = wz[t];
        if (unsigned(rx + t) < unsigned(BRICK)) dst[BRICK + rx + t] = wx[t] * r.w;
        if (unsigned(ry + t) < unsigned(BRICK)) dst[2 * BRICK + ry + t] = wy[t];
      }
    } else if (PAD_ROWS && tid < nst + PAD_ROWS) {  // zero rows behind the staged ones (see spread_brick_body)
      T* dst = stage + tid * SW;
#pragma unroll
      for (int k = 0; k < SW; ++k) dst[k] = T(0);
    }
    __syncthreads();
    constexpr int UC = MIPME_SPREAD_UC;
    const int nstc = __builtin_amdgcn_readfirstlane(nst);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int sv0 = wave_u; sv0 < nstc; sv0 += WAVES * UC) {
#if MIPME_LIVE_PADROWS
      T wz[UC][BRICK], fx[UC], fy[UC];
      const T* sw = stage + sv0 * SW;
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        const T* su = sw + u * WAVES * SW;
        fx[u] = su[BRICK + px];
        fy[u] = su[2 * BRICK + py];
        load_row8<T>(su, wz[u]);
      }
#pragma unroll
      for (int u = 0; u < UC; ++u) fma_row8<T>(acc, fx[u] * fy[u], wz[u]);
#else
      T wxy[UC], wz[UC][BRICK], fx[UC], fy[UC];
      bool live[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        const int sv = sv0 + u * WAVES;
        live[u] = sv < nstc;
        const T* sw = stage + (live[u] ? sv : sv0) * SW;
        fx[u] = sw[BRICK + px];
        fy[u] = sw[2 * BRICK + py];
        load_row8<T>(sw, wz[u]);
      }
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        wxy[u] = live[u] ? fx[u] * fy[u] : T(0);
        fma_row8<T>(acc, wxy[u], wz[u]);
      }
#endif
    }
    __syncthreads();
  }
#pragma unroll
  for (int pz = 0; pz < BRICK; ++pz) part[wave * BRICK_PTS + (px * BRICK + py) * BRICK + pz] = acc[pz];
  __syncthreads();
  for (int k = tid; k < BRICK_PTS; k += THREADS) {
    T v = T(0);
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += part[w * BRICK_PTS + k];
    const int qx = k / (BRICK * BRICK), qy = (k / BRICK) % BRICK, qz = k % BRICK;
    const int gx = ox + qx, gy = oy + qy, gz = oz + qz;
    if (gx < g.nx && gy < g.ny && gz < g.nz) args.mesh[gx * plane + int64_t(gy) * g.nz + gz] = v;
  }
}

// the home-atom workgroups, then the bricks, then the row workgroups of the pair sum (4-byte entries) as in spread_rows_kernel
__host__ __device__ inline unsigned live_home_blocks(int64_t n_atoms, bool xcd) {
  const unsigned n = unsigned((n_atoms + SPREAD_THREADS - 1) / SPREAD_THREADS);
  return xcd ? (n + 7u) / 8u * 8u : n;  // a multiple of 8 keeps blockIdx % 8 (the XCD) of everything behind them
}
template <int N, typename T, int PFAST, bool CELL = false>
__global__ __launch_bounds__(SPREAD_THREADS, sizeof(T) == 4 ? (CELL ? MIPME_CELL_WAVES : 8) : 1) void live_spread_rows_kernel(LiveSpreadArgs<T> sa, FusedRowsArgs<T> ra,
                                                                                        unsigned n_spread, unsigned pattern) {
  const unsigned n_home = live_home_blocks(sa.n_atoms, sa.bg.xcd), n_pad = sa.bg.xcd ? pad8(n_spread) : n_spread;
  const unsigned n_row_blocks = unsigned((ra.N + kRowsPerSpreadBlock - 1) / kRowsPerSpreadBlock);
  const unsigned n_rows_pad = sa.bg.xcd ? pad8(n_row_blocks) : n_row_blocks;
  if (blockIdx.x < n_home) {
    live_home_body<N, T>(sa, blockIdx.x);
    return;
  }
  const CoSlot cs = cosched_slot(blockIdx.x - n_home, n_pad, n_rows_pad, pattern);  // (block order: see spread_rows_kernel)
  if (cs.brick) {
    const unsigned b = brick_of(sa.bg, cs.slot);
    if (cs.slot < n_pad && b < n_spread) live_spread_body<N, T>(sa, b);
  } else if (cs.slot < n_rows_pad) {
    const unsigned r = sa.bg.xcd ? xcd_contiguous(cs.slot, n_row_blocks) : cs.slot;
    if (r < n_row_blocks) {
      extern __shared__ __attribute__((aligned(16))) char smem_rows[];
      AtomRecord<T>* tab = reinterpret_cast<AtomRecord<T>*>(smem_rows);
      if constexpr (std::is_same<T, float>::value)
        sr_rows_pk_body<PFAST, SPREAD_THREADS, CELL>(ra, r, tab);
#if MIPME_ROW_LANES == 16
      else if constexpr (PFAST == 1 || PFAST == 6)
        sr_rows_f64_body<SPREAD_THREADS, CELL, PFAST>(ra, r, smem_rows);
#endif
      else if constexpr (!CELL)
        sr_fused_rows_body<T, kPotForce, false, PFAST, false, true, SPREAD_THREADS, 0, true>(ra, r, tab);
    }
  }
}

// ---- step: gather + energy + forces from the slots the spread of THIS step filled --------------------------------------------
// As gather_brick_body<TAIL>, with two differences: the records are rec_now (current mesh coordinates), which may lie up to
// kLiveMargin points outside the brick, so the halo tile is that much wider on every side; and the charge comes from the (x, y, z,
// q) record.  (The first version evaluated the weights here, in each of the 8 lanes of an atom: 12.2 us against 7.4; now the spread,
// which evaluates them anyway for its staging, leaves them in the bins for its home atoms.)
template <int N, typename T>
__global__ __launch_bounds__(GATHER_THREADS, (sizeof(T) == 4 && N <= 5) ? MIPME_GATHER_TAIL_WAVES : 1) void live_gather_tail_kernel(Geom g, BrickGeom bg, BinIndex bins,
                                                                         const int4* __restrict__ rec_now,
                                                                         const T* __restrict__ wts,
                                                                         const AtomRecord<T>* __restrict__ rec4,
                                                                         const T* __restrict__ mesh, const T* __restrict__ qsum,
                                                                         T inv_vol, T self_c, T bg_c, T* __restrict__ out,
                                                                         T* __restrict__ field, GatherTail<T> tail,
                                                                         int* __restrict__ nan_flag) {
  static_assert(N <= kGatherLanes, "one lane per z point of the stencil");
  MIPME_WG_STAMP_GATHER(0);
  constexpr int THREADS = GATHER_THREADS, LANES = kGatherLanes, GROUPS = THREADS / LANES, MG = kLiveMargin;
  constexpr int TL = BRICK + N - 1 + 2 * MG;
  __shared__ T tile[TL * TL * TL];
  const unsigned block = brick_of(bg, blockIdx.x);
  if (block >= unsigned(bg.nb)) return;
  int bx, by, bz;
  brick_coords(bg, block, bx, by, bz);
  const int ox = bx * BRICK, oy = by * BRICK, oz = bz * BRICK;
  const int beg = int(block) * bins.cap, end = beg + bins.snap[block];
  const int n_over = bins.snap[bins.nb];
  T seed = T(1);
  if (tail.seed) seed = tail.seed[0];
  const T seed_aux = tail.aux_seed ? tail.aux_seed[0] : seed;
  if (block == 0) tail_energy<T, THREADS>(tail, qsum, inv_vol, self_c, bg_c);  // uniform per workgroup
  double r3[3] = {0.0, 0.0, 0.0};
  if (beg == end && n_over == 0) {
    if (tail.rpart && threadIdx.x < 9) tail.rpart[9 * int64_t(block) + threadIdx.x] = 0.0;
    return;
  }
  const int main_iters = (end - beg + GROUPS - 1) / GROUPS, over_iters = (n_over + GROUPS - 1) / GROUPS;
  const int l = threadIdx.x % LANES, grp = threadIdx.x / LANES;
  const bool lane_active = l < N;
  const int tz = lane_active ? l : 0;
  constexpr int s0 = stencil_start<N>();
  const int64_t plane = int64_t(g.ny) * g.nz;
  bool staged = false;
  for (int it = 0; it < main_iters + over_iters; ++it) {
    bool valid;
    int id;
    if (it < main_iters) {
      const int idx = beg + it * GROUPS + grp;
      valid = idx < end;
      id = valid ? idx : beg;
    } else {
      const int k = (it - main_iters) * GROUPS + grp;
      valid = k < n_over && bins.over_brick[k < n_over ? k : 0] == int(block);
      id = int(bins.over_base) + (k < n_over ? k : 0);
    }
    int4 a = rec_now[id];
    if (!valid) a = make_int4(ox, oy, oz, 0);  // a slot that may never have been written: keep every index derived from it in range
    const T* wr = wts + int64_t(id) * wts_stride<N, T>();
    T wx[N], wy[N], dwx[N], dwy[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
      wx[t] = wr[t];
      wy[t] = wr[N + t];
      dwx[t] = wr[3 * N + t];
      dwy[t] = wr[4 * N + t];
    }
    const T wzv = lane_active ? wr[2 * N + tz] : T(0);
    const T dwzv = lane_active ? wr[5 * N + tz] : T(0);
    if (!staged) {  // halo tile, kLiveMargin points wider than the stencils of the brick's own mesh points need
      for (int k = threadIdx.x; k < TL * TL * TL; k += THREADS) {
        const int tx = k / (TL * TL), ty = (k / TL) % TL, tzz = k % TL;
        const int gx = wrap1(ox + s0 - MG + tx, g.nx), gy = wrap1(oy + s0 - MG + ty, g.ny), gz = wrap1(oz + s0 - MG + tzz, g.nz);
        tile[k] = mesh[gx * plane + int64_t(gy) * g.nz + gz];
      }
      staged = true;
    }
    const AtomRecord<T> r_early = rec4[a.w];
    const T q_early = r_early.w;
    const T out_early = out[a.w];
    const T f_early = tail.force[3 * int64_t(a.w) + (l < 3 ? l : 0)];
    if (it == 0) __syncthreads();  // tile staged (uniform: every thread runs the first pass)
    // the atom's mesh coordinates relative to the tile's origin (brick origin - margin), wrapped to the nearest image and
    // clamped into the tile (beyond the margin the spread has flagged the step invalid)
    auto tile_start = [&](int m_now, int n, int o) {
      int d = m_now - o;
      d = d > n / 2 ? d - n : (d < -(n / 2) ? d + n : d);
      d += MG;
      return d < 0 ? 0 : (d > BRICK - 1 + 2 * MG ? BRICK - 1 + 2 * MG : d);
    };
    const int rtx = tile_start(a.x, g.nx, ox), rty = tile_start(a.y, g.ny, oy), rtz = tile_start(a.z, g.nz, oz);
    const T* tp = tile + rty * TL + (rtz + tz);
    T sA = T(0), sB = T(0), sC = T(0);
#pragma unroll
    for (int ty = 0; ty < N; ++ty) {
      T sx = T(0), sdx = T(0);
#pragma unroll
      for (int tx = 0; tx < N; ++tx) {
        const T v = tp[(rtx + tx) * TL * TL + ty * TL];
        sx += v * wx[tx];
        sdx += v * dwx[tx];
      }
      sA += sx * wy[ty];
      sB += sdx * wy[ty];
      sC += sx * dwy[ty];
    }
    const T fx = group_sum_b<LANES, T>(sB * wzv) * T(g.nx) * inv_vol;
    const T fy = group_sum_b<LANES, T>(sC * wzv) * T(g.ny) * inv_vol;
    const T fz = group_sum_b<LANES, T>(sA * dwzv) * T(g.nz) * inv_vol;
    // row l of the inverse cell by selects: indexing the by-value kernel argument with a lane-dependent index makes the compiler
    // fetch it with VECTOR loads from the kernarg segment -- 8 192 waves queueing on the same few bytes of host-visible memory
    const T i0 = T(l == 1 ? g.inv[3] : (l == 2 ? g.inv[6] : g.inv[0])), i1 = T(l == 1 ? g.inv[4] : (l == 2 ? g.inv[7] : g.inv[1])),
            i2 = T(l == 1 ? g.inv[5] : (l == 2 ? g.inv[8] : g.inv[2]));
    const T fc = i0 * fx + i1 * fy + i2 * fz;
    if (l < 3 && valid) {
      const int64_t o = int64_t(a.w);
      if (field) field[3 * o + l] = fc;
      tail.grad_pos[3 * o + l] = seed * q_early * (tail.force_scale * f_early + fc);
      if (tail.rpart) {
        const double gp = double(seed_aux * q_early * fc);
        r3[0] += double(r_early.x) * gp;
        r3[1] += double(r_early.y) * gp;
        r3[2] += double(r_early.z) * gp;
      }
    }
    const T acc = group_sum_b<LANES, T>(sA * wzv);
    if (l == 0 && valid) {
      const T phi = acc * inv_vol;
      const T lr = T(0.5) * (phi - self_c * q_early - T(2) * bg_c * inv_vol * qsum[0]);
      out[a.w] = out_early + lr;
      if (tail.grad_q) tail.grad_q[a.w] = T(2) * seed_aux * (out_early + lr);
      if (nan_flag && lr != lr) *nan_flag = 1;
    }
  }
  if (tail.rpart) tail_rpart<THREADS>(r3, tail.rpart, block);  // uniform
#if MIPME_WG_TIMELINE_GATHER
  __syncthreads();
#endif
  MIPME_WG_STAMP_GATHER(1);
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
#define MIPME_DISPATCH_ORDER(ORDER_V, BODY)                       \
  do {                                                            \
    switch (ORDER_V) {                                            \
      case 1: { constexpr int N = 1; BODY; } break;               \
      case 2: { constexpr int N = 2; BODY; } break;               \
      case 3: { constexpr int N = 3; BODY; } break;               \
      case 4: { constexpr int N = 4; BODY; } break;               \
      case 5: { constexpr int N = 5; BODY; } break;               \
      case 6: { constexpr int N = 6; BODY; } break;               \
      case 7: { constexpr int N = 7; BODY; } break;               \
      default: set_error("unsupported interpolation order %d", int(ORDER_V)); return MIPME_EINVAL; \
    }                                                             \
  } while (0)

bool live_supported(const mipme_mesh_t* m, int64_t N, int dtype) {
  if (!bricks_supported(m, dtype) || m->n_channels != 1 || N <= 0) return false;
  const BrickGeom bg = make_brick_geom(m);
  if (sparse_bricks(N, bg.nb)) return false;  // (the sparse-brick variants have no live form yet)
  const size_t s = dtype == MIPME_F32 ? 4 : 8;
  const size_t tl = BRICK + m->order - 1 + 2 * kLiveMargin;
  return tl * tl * tl * s <= 60 * 1024 && live_list_capacity(m, N) >= 1;
}

template <typename T>
int live_rebin(hipStream_t st, const mipme_mesh_t* m, int64_t N, const void* rec4, void* bins, void* lists, void* host_flags) {
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  MIPME_REQUIRE(live_supported(m, N, dtype), "mesh / atom count outside the live-bin kernels' range");
  const Geom g = make_geom(m);
  const BrickGeom bg = make_brick_geom(m);
  BinsView v = bins_view(m, N, dtype, bins);
  const LiveLists ll = live_view(m, N, lists, host_flags);
  // (the counters are zero here: the lists buffer starts zeroed and live_snapshot_kernel leaves them so)
  live_bin_kernel<T><<<unsigned((N + 255) / 256), 256, 0, st>>>(g, bg, (m->order % 2) == 0, v.idx, ll.counters, N,
                                                              (const AtomRecord<T>*)rec4, v.over_brick, v.rec, ll.home_rec);
  MIPME_LAUNCH_CHECK();
  live_snapshot_kernel<<<unsigned((bg.nb + 1 + 255) / 256), 256, 0, st>>>(v.idx, ll.counters);
  MIPME_LAUNCH_CHECK();
  MIPME_DISPATCH_ORDER(m->order, (live_lists_kernel<N><<<brick_grid(bg), SPREAD_THREADS, 0, st>>>(g, bg, v.idx, v.rec, ll)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int live_spread(hipStream_t st, const mipme_mesh_t* m, int64_t N, const void* rec4, void* bins, void* lists, void* mesh,
                const mipme_sr_job_t* job, void* host_flags, double* cpart) {
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  const BrickGeom bg = make_brick_geom(m);
  const BinsView v = bins_view(m, N, dtype, bins);
  const LiveLists ll = live_view(m, N, lists, nullptr);
  const int stage_rows = spread_stage_rows(m->order, sizeof(T));
  const size_t lds = spread_lds_bytes(m->order, sizeof(T), stage_rows, false, true);
  LiveSpreadArgs<T> sa;
  sa.g = make_geom(m);
  sa.bg = bg;
  sa.scheme = m->scheme;
  sa.count = ll.count;
  sa.atoms = ll.atoms;
  sa.lcap = ll.lcap;
  sa.rec4 = (const AtomRecord<T>*)rec4;
  sa.mesh = (T*)mesh;
  sa.stage_rows = stage_rows;
  sa.n_atoms = N;
  sa.home_rec = ll.home_rec;
  sa.rec_now = ll.rec_now;
  sa.wts = (T*)v.wts;
  sa.host_flags = (int*)host_flags;
  MIPME_REQUIRE(job && sr_job_fusable(job) && (job->shift_format & kShiftFormatMask) == kShiftTable32 && !job->dist_out,
                "the live step needs a co-schedulable pair job with 4-byte entries");
  SRPot s;
  int rc = make_srpot(job->pot, s);
  if (rc) return rc;
  const FastRS cf = make_fast_rs(s);
  const int pfast = fast_rs_exponent(s);
  FusedRowsArgs<T> ra = make_fused_rows_args<T>(s, cf, job->n_atoms, job->row_ptr, job->entries_shift, job->entries, nullptr,
                                                job->positions, job->records, job->cell, job->charges, nullptr, 0,
                                                job->full_list ? 0 : 1, job->full_list, 0, job->out, job->force, nullptr, nullptr,
                                                job->shift_format);
  ra.epart = v.epart;
  ra.cpart = cpart;
  MIPME_REQUIRE(!cpart || rows_cell_supported<T>(pfast, job->shift_format, job->dist_out),
                "the cell sums of the pair kernel need 4-byte entries and 1/r or 1/r^6");
  const unsigned n_row_blocks = unsigned((job->n_atoms + kRowsPerSpreadBlock - 1) / kRowsPerSpreadBlock);
  const unsigned n_spread = unsigned(bg.nb);
  const unsigned pattern = brick_pattern(bg, n_spread, n_row_blocks, sizeof(T) == 4);
  const unsigned grid = live_home_blocks(N, bg.xcd) +
                        (bg.xcd ? cosched_grid(pad8(n_spread), pad8(n_row_blocks), pattern) : n_spread + n_row_blocks);
  note_cosched_kernel("live_spread_rows_kernel");
  if (cpart && pfast == 1)
    MIPME_DISPATCH_ORDER(m->order, (live_spread_rows_kernel<N, T, 1, true><<<grid, SPREAD_THREADS, lds, st>>>(sa, ra, n_spread, pattern)));
  else if (cpart) {
    MIPME_DISPATCH_ORDER(m->order, (live_spread_rows_kernel<N, T, 6, true><<<grid, SPREAD_THREADS, lds, st>>>(sa, ra, n_spread, pattern)));
  } else if (pfast == 1)
    MIPME_DISPATCH_ORDER(m->order, (live_spread_rows_kernel<N, T, 1><<<grid, SPREAD_THREADS, lds, st>>>(sa, ra, n_spread, pattern)));
  else
    MIPME_DISPATCH_ORDER(m->order, (live_spread_rows_kernel<N, T, 6><<<grid, SPREAD_THREADS, lds, st>>>(sa, ra, n_spread, pattern)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int live_gather(hipStream_t st, const mipme_mesh_t* m, int64_t N, const void* rec4, void* bins, void* lists, const void* mesh,
                const void* qsum, double self_c, double bg_c, void* out, void* field, const GatherTailHost* th, void* nan_flag) {
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  const Geom g = make_geom(m);
  const BrickGeom bg = make_brick_geom(m);
  const BinsView v = bins_view(m, N, dtype, bins);
  const LiveLists ll = live_view(m, N, lists, nullptr);
  MIPME_REQUIRE(th && th->force && th->grad_pos && th->energy && th->epart_k && out && qsum, "NULL buffer passed to the live gather");
  GatherTail<T> tail;
  tail.force = (const T*)th->force;
  tail.force_scale = T(th->force_scale);
  tail.seed = (const T*)th->seed;
  tail.grad_pos = (T*)th->grad_pos;
  tail.energy = (T*)th->energy;
  tail.epart_k = (const double*)th->epart_k;
  tail.n_k = int(th->n_k);
  tail.epart_sr = tail.epart_k + tail.n_k;  // pre-reduced by the x stage of the convolution
  tail.n_sr = tail.n_k;
  tail.grad_q = (T*)th->grad_q;
  tail.rpart = th->rpart;
  tail.rec4 = (const AtomRecord<T>*)rec4;
  tail.aux_seed = (const T*)th->aux_seed;
  tail.live_flags = (const int*)th->live_flags;
  tail.elog = th->elog;
  tail.elog_cursor = th->elog_cursor;
  tail.elog_cap = th->elog_cap;
  MIPME_DISPATCH_ORDER(m->order, (live_gather_tail_kernel<N, T><<<brick_grid(bg), GATHER_THREADS, 0, st>>>(
                                     g, bg, v.idx, ll.rec_now, (const T*)v.wts, (const AtomRecord<T>*)rec4, (const T*)mesh,
                                     (const T*)qsum, T(1.0 / m->volume), T(self_c), T(bg_c), (T*)out, (T*)field, tail,
                                     (int*)nan_flag)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template int live_rebin<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, void*);
template int live_rebin<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, void*);
template int live_spread<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, void*, const mipme_sr_job_t*,
                                void*, double*);
template int live_spread<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, void*, const mipme_sr_job_t*,
                                 void*, double*);
template int live_gather<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, const void*, const void*,
                                double, double, void*, void*, const GatherTailHost*, void*);
template int live_gather<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, const void*, const void*,
                                 double, double, void*, void*, const GatherTailHost*, void*);

}  // namespace mipme
