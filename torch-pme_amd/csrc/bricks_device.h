// Device code of the brick-binned particle <-> mesh path, shared by the translation units that launch it:
//   bricks.hip  bins (one-pass binning), owner-computes brick spread, plane spread, co-scheduled launches, gathers
//   frames.hip  independent frames in one launch (mipme_frames_*)
//   live.hip    live bins of an MD-like loop (mipme_md_rebin / mipme_md_step)
// Sections, in order: atom bins + binning pass; deterministic slots; spread (bricks); plane spread; gather (+ tail of the energy
// step); host-side views of the bins buffer and the stencil dispatch macro.  bricks.hip defines MIPME_BRICKS_MAIN_TU and thereby
// the few non-template functions other files link against.
#ifndef MIPME_BRICKS_DEVICE_H
#define MIPME_BRICKS_DEVICE_H
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include <map>
#include <utility>

#include "rows_body.h"
#include "fft_lds.h"

namespace mipme {

static constexpr int BRICK = 8;
static constexpr int BRICK_PTS = BRICK * BRICK * BRICK;

// Residency on gfx950 is also a matter of SCALAR registers (MI355X_MICROARCH.md, "Residency"): waves per SIMD <=
// floor(800 / (ceil(sgpr / 16) * 16 + 16)) -- 80 admit 8 waves (four 512-thread workgroups per CU), 82-96 only 7 (three).  The
// compiler takes what it likes up to 102 unless told otherwise (the attribute takes a literal: capped kernels are kernels of
// their own, chosen at dispatch for the instantiations that take the cap without spilling).
#define MIPME_SGPR_CAP __attribute__((amdgpu_num_sgpr(80)))
// spread: survivors staged together (rows of spread_row_reals reals); fewer than 256 when they would not fit next to the lists
static inline int spread_stage_rows(int order, size_t real_bytes);
// reals staged per survivor by the spread: its three 1-D weight vectors PLACED on the brick's 8 points of each axis (zero
// outside the stencil), the x vector already times the value: [wz | wx * value | wy]
static inline size_t spread_row_reals(int, size_t) { return 3 * BRICK; }

struct BrickGeom {
  int nbx, nby, nbz, nb;
  int xcd;  // 1: workgroup -> brick through xcd_contiguous() (common.h), launches padded to a multiple of 8 workgroups
};

static inline BrickGeom make_brick_geom(const mipme_mesh_t* m) {
  BrickGeom b;
  b.nbx = (m->nx + BRICK - 1) / BRICK;
  b.nby = (m->ny + BRICK - 1) / BRICK;
  b.nbz = (m->nz + BRICK - 1) / BRICK;
  b.nb = b.nbx * b.nby * b.nbz;
  b.xcd = 1;  // (XCD-contiguous workgroup -> brick / row-block mapping: +1-3 %, profiles/r02_experiments.txt; a switch until round 6)
  return b;
}
// workgroups of a one-workgroup-per-brick launch, and the brick of a workgroup (>= nb: nothing to do)
static inline unsigned brick_grid(const BrickGeom& b) { return b.xcd ? pad8(unsigned(b.nb)) : unsigned(b.nb); }
__device__ __forceinline__ unsigned brick_of(const BrickGeom& b, unsigned wg) {
  return b.xcd ? xcd_contiguous(wg, unsigned(b.nb)) : wg;
}

// a mod n for a in (-n, 2n)
__device__ __forceinline__ int wrap1(int a, int n) {
  a += a < 0 ? n : 0;
  return a >= n ? a - n : a;
}

// Brick path preconditions: >= 3 bricks per axis (the 27 neighbours are distinct bricks) and enough LDS.
#ifdef MIPME_BRICKS_MAIN_TU
bool bricks_supported(const mipme_mesh_t* m, int dtype) {
  const size_t s = dtype == MIPME_F32 ? 4 : 8;
  const int ns[3] = {m->nx, m->ny, m->nz};
  for (int d = 0; d < 3; ++d) {
    if (ns[d] <= 2 * BRICK) return false;               // need >= 3 distinct bricks per axis
    const int rem = ns[d] % BRICK;
    if (rem != 0 && rem < 4) return false;              // a narrow last brick would be skipped over by a stencil
  }
  const size_t tile = BRICK + m->order - 1;
  if (2 * size_t(m->n_channels) * tile * tile * tile * s > 60 * 1024) return false;  // gather_grad: phi+chi per channel
  if (spread_stage_rows(m->order, s) == 0) return false;  // spread staging
  return true;
}
#else
bool bricks_supported(const mipme_mesh_t* m, int dtype);
#endif

// ---- atom bins: fixed-capacity brick slots, ONE binning pass -----------------------------------------------------------
// Brick b owns the slots [b * cap, (b + 1) * cap) of the record / weight arrays; an atom takes the next free slot of its
// brick with one (wave-aggregated) returning atomic on the brick's counter and writes its record and weights there at
// once -- no counting pass, no scan, no second pass (round 1: bin_count 5.0 us + bin_fill 6.8 us at cfg3, both pure
// latency).  Atoms that find their brick full go to the overflow region behind the brick slots (counter live[nb], home
// brick in over_brick[k]); every consumer also scans it, which is a single load when it is empty.  cap = 4 x the mean
// occupancy + 8 (multiple of 8), so overflow means a density contrast above 4.
//   live  (plan / frame owned, zero between calls): int[nb + 1] counters; the binning pass fills them, the forward spread
//         reads them and copies them to `snap`, the forward gather -- the last consumer -- zeroes them again.
//   snap  (inside the bins buffer): int[nb + 1] per-call copy {min(count, cap) per brick, overflow count}, read by the
//         gathers of the forward and by everything in the backward pass.
// Plane lists come in kPlaneSub sub-lists per plane, keyed by the wavefront of the binning pass that appends (its index mod
// kPlaneSub): the appends are returning atomics on the lists' counters, and same-address atomics serialise -- 1 000 of them on
// nx = 64 counters were ~16 per address and +1.5 us on the pass; on 512 counters they are 2 per address, like the bricks'.
// (kPlaneSub = 8: common.h, next to plan_counter_words)
struct BinsLayout {
  size_t snap, over_brick, rec, wts, codes, qs, plist, pover, wmax, epart, det, det_sort_bytes, total;
  int cap;
  int pcap;  // entries per plane SUB-list (0: no plane lists for this mesh, see plane_list_capacity)
  int64_t slots;  // nb * cap + N
};

// MIPME_DETERMINISTIC=1: bit-reproducible results run to run ("race detection").  The one-pass binning hands out
// brick slots with returning atomics and the spread's candidate scan appends its survivors with LDS atomics, so in fp32 the
// mesh values -- sums over a brick's atoms -- depend on arrival order in the last bits.  In this mode the slots come from a
// stable radix sort of the atoms by brick (slot = rank by atom index inside the brick; the overflow region is ordered the
// same way) and every round of survivors is sorted by slot before it is staged: every sum then runs in one fixed order.
// Costs a sort per evaluation (three more launches); read once per process.
static bool deterministic_mode() {
  static const bool on = env_flag("MIPME_DETERMINISTIC", false);
  return on;
}
static constexpr int kRowsPerSpreadBlock = 512 / kRowLanes;  // rows per workgroup of the co-scheduled pair sum (SPREAD_THREADS)
static constexpr int kSpreadWaves = 512 / 64;

static inline int bin_capacity(int nb, int64_t N) {
  const int64_t mean = (N + nb - 1) / nb;
  int64_t cap = (4 * mean + 8 + 7) / 8 * 8;
  const int64_t all = (N + 7) / 8 * 8;  // never more than all atoms
  if (cap > all) cap = all;
  return int(cap < 8 ? 8 : cap);
}

// A slot's weights: wx, wy, wz, dwx, dwy, dwz (N each), padded to whole 16-byte chunks so that the atom's own lane writes them
// with 16-byte stores (8 instructions at N = 5 instead of 30).  Measured at 32 000 atoms this is neutral (+-0.1 us, A/B on one
// box, profiles/r03_experiments.txt); what these stores cost a kernel -- ~3 us whether four-byte or sixteen-byte, scattered by slot
// or dense by atom, cached, non-temporal or system-scope -- comes with their 4 MB however they are issued.
template <int N, typename T>
constexpr int wts_stride() {
  constexpr int V = 16 / int(sizeof(T));
  return (6 * N + V - 1) / V * V;
}
static inline size_t wts_stride_rt(int order, size_t elem) { return (6 * size_t(order) * elem + 15) / 16 * 16 / elem; }

template <int N, typename T>
__device__ __forceinline__ void store_slot_weights(T* __restrict__ wr, const T (&wx)[N], const T (&wy)[N], const T (&wz)[N],
                                                   const T (&dwx)[N], const T (&dwy)[N], const T (&dwz)[N]) {
  constexpr int W = wts_stride<N, T>(), V = 16 / int(sizeof(T));
  struct alignas(16) Chunk {
    T e[V];
  };
  T v[W];
#pragma unroll
  for (int t = 0; t < N; ++t) {
    v[t] = wx[t];
    v[N + t] = wy[t];
    v[2 * N + t] = wz[t];
    v[3 * N + t] = dwx[t];
    v[4 * N + t] = dwy[t];
    v[5 * N + t] = dwz[t];
  }
#pragma unroll
  for (int t = 6 * N; t < W; ++t) v[t] = T(0);
  Chunk* d = reinterpret_cast<Chunk*>(wr);
#pragma unroll
  for (int k = 0; k < W / V; ++k) {
    Chunk c;
#pragma unroll
    for (int e = 0; e < V; ++e) c.e[e] = v[k * V + e];
    d[k] = c;
  }
}

// Plane-list ENTRIES (round 6): what a plane workgroup needs of an atom, written by the binning pass while the atom's weights are
// in registers -- word 0 the packed stencil reference point (mx << 20 | my << 10 | mz), words 1, 2 the offsets x_y, x_z the y / z
// weights are functions of, words 3 .. 3 + N the products charge * w_x[t] of the atom's N planes --, padded to whole 16-byte
// chunks (fp32, N <= 5: 32 bytes).  Until round 5 an entry was the atom's bin SLOT and a plane workgroup gathered record (16 B),
// weights (44 B of a 120-byte row) and charge behind it: three dependent levels, ~7 gather instructions of 64 cache lines each
// per wavefront and batch.  The entries of a list are contiguous, so consecutive lanes now stream consecutive 32-byte entries
// (four lanes to a 128-byte line), there is no index to wait for, and the 2 N weights are 2 x ~25 FMAs (weights_1d).
template <int N, typename T>
constexpr int plane_entry_words() {
  constexpr int V = 16 / int(sizeof(T));
  return (3 + N + V - 1) / V * V;
}
static inline size_t plane_entry_bytes_rt(int order, size_t elem) { return ((3 + size_t(order)) * elem + 15) / 16 * 16; }
static constexpr int kPlanePackBits = 10;  // my, mz < 1024 (the plane tiles are far smaller), mx < 2048

static int plane_list_capacity(const mipme_mesh_t* m, int64_t N, int dtype);
static inline BinsLayout bins_layout(const mipme_mesh_t* m, int64_t N, int dtype) {
  const BrickGeom b = make_brick_geom(m);
  const size_t s = dtype == MIPME_F32 ? 4 : 8;
  auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
  BinsLayout l;
  l.cap = bin_capacity(b.nb, N);
  l.slots = int64_t(b.nb) * l.cap + N;
  size_t off = 0;
  l.snap = off;       off += al(sizeof(int) * size_t(b.nb + 1));
  l.over_brick = off; off += al(sizeof(int) * size_t(N));
  l.rec = off;        off += al(sizeof(int4) * size_t(l.slots));
  l.wts = off;        off += al(wts_stride_rt(m->order, s) * s * size_t(l.slots));  // per slot: see wts_stride
  l.codes = off;      off += al(size_t(l.slots));  // per slot: which neighbouring bricks the atom's stencil reaches (reach_code)
  l.qs = off;         off += al(s * size_t(l.slots));  // per slot: the atom's charge (single channel), for the plane spread
  // plane lists (plane spread): the slots of the atoms whose stencil reference point m_x is plane p, pcap per plane, and an
  // overflow list (slots of atoms whose plane list was full: normally empty) that every plane also walks
  l.pcap = plane_list_capacity(m, N, dtype);
  l.plist = off;      off += al(plane_entry_bytes_rt(m->order, s) * size_t(l.pcap) * size_t(l.pcap ? m->nx * kPlaneSub : 0));
  l.pover = off;      off += al(plane_entry_bytes_rt(m->order, s) * size_t(l.pcap ? N : 0));
  // max |charge| of every wavefront of the binning pass (one plain store each): max over them x atoms of a plane = the bound that
  // fixes the scale of the fp32 plane spread's fixed-point sums (plane_spread_yz_body)
  l.wmax = off;       off += al(sizeof(float) * size_t(l.pcap ? (N + 63) / 64 : 0));
  // energy partial sums of the co-scheduled pair sum: 2 doubles per wave of its row workgroups (FusedRowsArgs::epart)
  l.epart = off; off += al(2 * sizeof(double) * kSpreadWaves * (((size_t(N) + kRowsPerSpreadBlock - 1) / kRowsPerSpreadBlock + 1) & ~size_t(1)));
  l.det = off;
  l.det_sort_bytes = 0;
  if (deterministic_mode()) {  // keys, vals, keys2, vals2, slot_of, over_flag (N words each) + the sort's scratch
    size_t sb = 0;
    unsigned* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sb, nul, nul, nul, nul, size_t(N > 0 ? N : 1), 0u, 32u, hipStream_t(0), false);
    l.det_sort_bytes = sb;
    off += 6 * al(sizeof(int) * size_t(N > 0 ? N : 1)) + al(sb);
  }
  l.total = off;
  return l;
}

// what the consumers need to walk the bins
struct BinIndex {
  int* live;              // counters of the binning pass (see above); NULL in the backward pass
  int* snap;              // per-call snapshot
  const int* over_brick;  // home brick of every overflow atom
  int nb, cap;
  int64_t over_base;      // = nb * cap
  unsigned char* codes = nullptr;  // per brick slot: reach_code of the atom (written by the binning pass, read by the spread's scan)
  // plane lists (plane spread; pcap == 0: none): live counters int[nx * kPlaneSub + 1] behind the brick counters (zeroed by the forward
  // gather like those), slots per plane, overflow slots
  int* plive = nullptr;
  // sub-list of an atom: >= 0: its y slice, m_y >> pband_shift (bands: a workgroup reads the slices of its rows); -1: the index of
  // the binning pass's wavefront mod kPlaneSub (whole planes: spreads same-address atomics, every plane reads all sub-lists)
  int pband_shift = -1;
  void* plist = nullptr;  // entries: plane_entry_words<N, T>() reals each, [nx * kPlaneSub lists][pcap]
  void* pover = nullptr;  // entries of the atoms whose list was full
  int pcap = 0;
  // max |value| per wavefront of the binning pass (float[ceil(N / 64)], rewritten by every pass): see BinsLayout::wmax
  float* wmax = nullptr;
  int n_wmax = 0;
};

// Which of its brick's neighbours an atom's stencil reaches, from its position inside the brick: bit 2 d = the lower neighbour
// along axis d, bit 2 d + 1 = the upper one (a stencil of <= 8 points touches at most two bricks per axis; the last brick of an
// axis may be narrower than 8: bricks_supported keeps it >= 4 wide).  The spread's candidate scan of a brick then keeps an atom
// of the neighbour at offset (dx, dy, dz) iff the atom reaches back along every axis with an offset -- one byte load and one
// compare per candidate instead of its 16-byte record and three wrapped range tests.
template <int N>
__device__ __forceinline__ unsigned reach_code(const int (&m)[3], int nx, int ny, int nz) {
  constexpr int s0 = stencil_start<N>();
  const int n[3] = {nx, ny, nz};
  unsigned code = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int i = m[d] & (BRICK - 1), size = min(BRICK, n[d] - (m[d] & ~(BRICK - 1)));
    code |= (i + s0 < 0 ? 1u : 0u) << (2 * d);
    code |= (i + s0 + N - 1 >= size ? 2u : 0u) << (2 * d);
  }
  return code;
}

#ifdef MIPME_BRICKS_MAIN_TU
int plane_bins_capacity(const mipme_mesh_t* m, int64_t N, int dtype) { return plane_list_capacity(m, N, dtype); }
int plane_bands(const mipme_mesh_t* m, int dtype);
#else
int plane_bins_capacity(const mipme_mesh_t* m, int64_t N, int dtype);
int plane_bands(const mipme_mesh_t* m, int dtype);
#endif

#ifdef MIPME_BRICKS_MAIN_TU
int64_t bins_bytes(const mipme_mesh_t* m, int64_t N, int dtype) {
  if (!bricks_supported(m, dtype)) return 0;
  return int64_t(bins_layout(m, N, dtype).total);
}
#else
int64_t bins_bytes(const mipme_mesh_t* m, int64_t N, int dtype);
#endif

__device__ __forceinline__ void split_runtime(double u, bool even, int& m, double& x) {
  if (even) {
    const double fl = floor(u);
    m = int(fl);
    x = u - (fl + 0.5);
  } else {
    const double r = rint(u);
    m = int(r);
    x = u - r;
  }
}

template <typename T>
__device__ __forceinline__ void atom_mesh_coords(const Geom& g, bool even, const T* __restrict__ pos, int64_t i,
                                                 int (&m)[3], double (&x)[3]) {
  const double rx = double(pos[3 * i + 0]), ry = double(pos[3 * i + 1]), rz = double(pos[3 * i + 2]);
  const double u[3] = {double(g.nx) * (rx * g.inv[0] + ry * g.inv[3] + rz * g.inv[6]),
                       double(g.ny) * (rx * g.inv[1] + ry * g.inv[4] + rz * g.inv[7]),
                       double(g.nz) * (rx * g.inv[2] + ry * g.inv[5] + rz * g.inv[8])};
  const int n[3] = {g.nx, g.ny, g.nz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int mm;
    split_runtime(u[d], even, mm, x[d]);
    m[d] = posmod(mm, n[d]);
  }
}

// ---- binning -----------------------------------------------------------------------------------
// Lanes of a wavefront that fall into the same brick share ONE returning atomic (atoms are usually stored in a
// spatially coherent order, so a wave touches only a handful of bricks): leader election over the ballot mask.  The atom
// then writes its record {mesh coordinates, atom index} and its 1-D weights (and derivatives) -- evaluated ONCE, the four
// particle<->mesh kernels of a step only load them -- straight into its slot, and (atom_rec) the (position, charge) record
// of the fused pair kernels while the position is in registers anyway.
static constexpr int64_t kCoalescedBinAtoms = 100000;  // atoms from which the binning pass stages its weight rows (see below)
template <int SCHEME, int N, typename T, bool COALESCE = false>
__device__ __forceinline__ void bin_atoms_body(const Geom& g, const BrickGeom& bg, const BinIndex& bi, int64_t Natoms,
                                               const T* __restrict__ pos, int* __restrict__ over_brick,
                                               int4* __restrict__ rec, T* __restrict__ wts, const T* __restrict__ q,
                                               AtomRecord<T>* __restrict__ atom_rec, unsigned block,
                                               const int* __restrict__ slot_of = nullptr, T* __restrict__ qs = nullptr) {
  const int64_t i = int64_t(block) * blockDim.x + threadIdx.x;
  const bool valid = i < Natoms;
  const int lane = threadIdx.x & 63;
  int b = -1;
  int m[3] = {0, 0, 0};
  double x[3] = {0.0, 0.0, 0.0};
  if (valid) {
    atom_mesh_coords<T>(g, (N % 2) == 0, pos, i, m, x);
    b = ((m[0] / BRICK) * bg.nby + m[1] / BRICK) * bg.nbz + m[2] / BRICK;
  }
  int myslot = 0, over_k = -1;
  int pl_leader = lane, pl_rank = 0, pl_count = 0, pl_slot = -1;
  // this atom's sub-list of its plane: the wavefront's index (uniform) or, for bands, the atom's y slice
  const int pl_key = bi.pband_shift >= 0 ? ((m[1] >> bi.pband_shift) & (kPlaneSub - 1))
                                         : int(((int64_t(block) * blockDim.x + threadIdx.x) >> 6) & (kPlaneSub - 1));
  const int pl_list = m[0] * kPlaneSub + pl_key;
  if (slot_of) {  // deterministic mode: slots (and the live counters) come from the sorted atom list, see det_slots_kernel
    if (valid) {
      myslot = slot_of[i];
      if (myslot >= bi.cap) over_k = myslot - bi.cap;
    }
  } else {
    // pass 1 (no memory traffic): group the lanes by brick; every lane learns its leader lane and its rank
    unsigned long long remaining = __ballot(valid);
    int my_leader = lane, my_rank = 0, my_count = 0;
    while (remaining) {
      const int leader = __ffsll((long long)remaining) - 1;
      const int b0 = __shfl(b, leader, 64);
      const unsigned long long peers = __ballot(valid && b == b0);
      if (valid && b == b0) {
        my_leader = leader;
        my_rank = __popcll(peers & ((1ull << lane) - 1ull));
        my_count = __popcll(peers);
      }
      remaining &= ~peers;
    }
    // the same grouping by x plane for the plane lists (plane spread): their atomics travel with the bricks' ones
    if (bi.plive && bi.wmax && qs) {  // (uniform) this wavefront's largest |charge|: a plain store, no atomics (a float atomic per
      // plane and wavefront on nx addresses cost the pass 6.6 us: same-address atomics serialise at ~250 ns each)
      float a = valid ? fabsf(float(q[i])) : 0.f;
      a = fmaxf(a, dpp_mov<0xB1>(a));
      a = fmaxf(a, dpp_mov<0x4E>(a));
      a = fmaxf(a, dpp_mov<0x141>(a));
      a = fmaxf(a, dpp_mov<0x140>(a));
      a = fmaxf(fmaxf(read_lane(a, 0), read_lane(a, 16)), fmaxf(read_lane(a, 32), read_lane(a, 48)));
      if (lane == 0) bi.wmax[(int64_t(block) * blockDim.x + threadIdx.x) >> 6] = a;
    }
    if (bi.plive) {
      unsigned long long rem = __ballot(valid);
      while (rem) {
        const int leader = __ffsll((long long)rem) - 1;
        const int p0 = __shfl(pl_list, leader, 64);
        const unsigned long long peers = __ballot(valid && pl_list == p0);
        if (valid && pl_list == p0) {
          pl_leader = leader;
          pl_rank = __popcll(peers & ((1ull << lane) - 1ull));
          pl_count = __popcll(peers);
        }
        rem &= ~peers;
      }
    }
    // pass 2: all leaders issue their returning atomic together (one memory round trip per wave, not one per brick)
    int base = 0, pbase = 0;
    if (valid && my_leader == lane) base = atomicAdd(&bi.live[b], my_count);
    if (bi.plive && valid && pl_leader == lane) pbase = atomicAdd(&bi.plive[pl_list], pl_count);
    base = __shfl(base, my_leader, 64);
    myslot = base + my_rank;
    if (bi.plive) pl_slot = __shfl(pbase, pl_leader, 64) + pl_rank;
  }
  int64_t dst = 0;
  if (valid) {
    if (myslot < bi.cap) {
      dst = int64_t(b) * bi.cap + myslot;
    } else {  // brick full: overflow region (rare; one atomic per atom)
      const int k = over_k >= 0 ? over_k : atomicAdd(&bi.live[bi.nb], 1);
      over_brick[k] = b;
      dst = bi.over_base + k;
    }
    if (atom_rec) {
      AtomRecord<T> r;
      r.x = pos[3 * i];
      r.y = pos[3 * i + 1];
      r.z = pos[3 * i + 2];
      r.w = q[i];
      atom_rec[i] = r;
    }
    rec[dst] = make_int4(m[0], m[1], m[2], int(i));
    if (qs) qs[dst] = q[i];
    if (bi.codes && dst < bi.over_base) bi.codes[dst] = (unsigned char)reach_code<N>(m, g.nx, g.ny, g.nz);
  }
  // The 6N weights of an atom go to its slot, anywhere in the bins: written by the atom's own lane that is 6N four-byte stores
  // to 64 different cache lines per instruction.  Transposed through LDS instead: the wave stages its rows, then lane k of a
  // group of 6N lanes writes value k of one atom -- contiguous 24N-byte segments, two atoms per instruction at N = 5
  // (1 029 000 atoms: 95 -> ... us for this kernel).
  constexpr int W = wts_stride<N, T>();
  // the atom's plane-list entry (plane_entry_words): written once the x weights are known
  auto store_plane_entry = [&](const T (&wx)[N]) __attribute__((always_inline)) {
    if (!valid || pl_slot < 0) return;
    constexpr int EW = plane_entry_words<N, T>(), V = 16 / int(sizeof(T));
    struct alignas(16) Chunk {
      T e[V];
    };
    T* e;
    if (pl_slot < bi.pcap)  // plane list of m_x (else the plane overflow list: one atomic per atom, normally none)
      e = (T*)bi.plist + (int64_t(pl_list) * bi.pcap + pl_slot) * EW;
    else
      e = (T*)bi.pover + int64_t(atomicAdd(&bi.plive[g.nx * kPlaneSub], 1)) * EW;
    T v[EW];
    const int packed = (m[0] << (2 * kPlanePackBits)) | (m[1] << kPlanePackBits) | m[2];
    if constexpr (sizeof(T) == 4)
      v[0] = __int_as_float(packed);
    else
      v[0] = __longlong_as_double((long long)packed);
    v[1] = T(x[1]);
    v[2] = T(x[2]);
    const T qa = q[i];
#pragma unroll
    for (int t = 0; t < N; ++t) v[3 + t] = qa * wx[t];
#pragma unroll
    for (int t = 3 + N; t < EW; ++t) v[t] = T(0);
    Chunk* d = reinterpret_cast<Chunk*>(e);
#pragma unroll
    for (int k = 0; k < EW / V; ++k) {
      Chunk c;
#pragma unroll
      for (int u = 0; u < V; ++u) c.e[u] = v[k * V + u];
      d[k] = c;
    }
  };
  // COALESCE is chosen by the launcher for large systems, where the kernel is bound by its store transactions (1 029 000 atoms:
  // 95 -> 45 us); at 32k atoms it is a chain of latencies and the extra LDS round trip costs 1.3 us.  fp64 rows of n >= 5 nodes
  // do not fit 48 KB of LDS and keep the direct stores.
  constexpr bool STAGED = COALESCE && sizeof(T) * 64 * W * (256 / 64) <= 48 * 1024;
  if constexpr (STAGED) {
    __shared__ T sw[256 / 64][64 * W];
    T* mine = sw[threadIdx.x >> 6] + lane * W;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      T w[N], dw[N];
      weights_1d<SCHEME, N, true, T>(T(x[d]), w, dw);
#pragma unroll
      for (int t = 0; t < N; ++t) {
        mine[d * N + t] = w[t];
        mine[(3 + d) * N + t] = dw[t];
      }
      if (d == 0 && bi.plive) store_plane_entry(w);
    }
    // (one wave reads what the same wave wrote: no workgroup barrier needed, only the LDS writes to have landed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = 64 / W;  // atoms per store instruction
    const unsigned long long vmask = __ballot(valid);
    const int sub = lane / W, k = lane % W;
    const T* rows = sw[threadIdx.x >> 6];
    for (int a0 = 0; a0 < 64; a0 += PER) {
      const int j = a0 + (sub < PER ? sub : 0);
      const int64_t dj = __shfl(dst, j < 64 ? j : 0, 64);
      if (sub < PER && k < 6 * N && j < 64 && ((vmask >> j) & 1ull)) wts[dj * W + k] = rows[j * W + k];
    }
  } else {
    if (!valid) return;
    T wx[N], wy[N], wz[N], dwx[N], dwy[N], dwz[N];
    weights_1d<SCHEME, N, true, T>(T(x[0]), wx, dwx);
    weights_1d<SCHEME, N, true, T>(T(x[1]), wy, dwy);
    weights_1d<SCHEME, N, true, T>(T(x[2]), wz, dwz);
    store_slot_weights<N, T>(wts + dst * W, wx, wy, wz, dwx, dwy, dwz);
    if (bi.plive) store_plane_entry(wx);
  }
}

template <int SCHEME, int N, typename T, bool COALESCE>
__global__ __launch_bounds__(256) void bin_atoms_kernel(Geom g, BrickGeom bg, BinIndex bi, int64_t Natoms,
                                                       const T* __restrict__ pos, int* __restrict__ over_brick,
                                                       int4* __restrict__ rec, T* __restrict__ wts,
                                                       const T* __restrict__ q, AtomRecord<T>* __restrict__ atom_rec,
                                                       const int* __restrict__ slot_of, T* __restrict__ qs) {
  bin_atoms_body<SCHEME, N, T, COALESCE>(g, bg, bi, Natoms, pos, over_brick, rec, wts, q, atom_rec, blockIdx.x, slot_of, qs);
}

// ---- deterministic slots (MIPME_DETERMINISTIC) -----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void det_keys_kernel(Geom g, BrickGeom bg, bool even, int64_t Natoms, const T* __restrict__ pos,
                                                      unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= Natoms) return;
  int m[3];
  double x[3];
  atom_mesh_coords<T>(g, even, pos, i, m, x);
  keys[i] = unsigned(((m[0] / BRICK) * bg.nby + m[1] / BRICK) * bg.nbz + m[2] / BRICK);
  vals[i] = unsigned(i);
}

// atoms sorted by (brick, index): slot = position - first position of the brick; the first atom of a brick also writes the
// brick's live counter (= its atom count)
static __global__ __launch_bounds__(256) void det_slots_kernel(int64_t Natoms, int cap, const unsigned* __restrict__ keys2,
                                                       const unsigned* __restrict__ vals2, int* __restrict__ slot_of,
                                                       int* __restrict__ over_flag, int* __restrict__ live) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= Natoms) return;
  const unsigned b = keys2[p];
  int64_t lo = 0, hi = p;  // first position with key == b
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys2[mid] < b)
      lo = mid + 1;
    else
      hi = mid;
  }
  const int slot = int(p - lo);
  slot_of[vals2[p]] = slot;
  over_flag[p] = slot >= cap ? 1 : 0;
  if (slot == 0) {
    int64_t l2 = p, h2 = Natoms;  // first position with key > b
    while (l2 < h2) {
      const int64_t mid = (l2 + h2) >> 1;
      if (keys2[mid] <= b)
        l2 = mid + 1;
      else
        h2 = mid;
    }
    live[b] = int(l2 - p);
  }
}

// overflow atoms (slot >= cap) in sorted order: slot_of = cap + rank among them; live[nb] = their number.  One workgroup.
static __global__ __launch_bounds__(1024) void det_overflow_kernel(int64_t Natoms, int cap, int nb, const unsigned* __restrict__ vals2,
                                                           const int* __restrict__ over_flag, int* __restrict__ slot_of,
                                                           int* __restrict__ live) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int64_t per = (Natoms + 1023) / 1024;
  const int64_t lo = min(int64_t(t) * per, Natoms), hi = min(lo + per, Natoms);
  int sum = 0;
  for (int64_t k = lo; k < hi; ++k) sum += over_flag[k];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  if (sum > 0)
    for (int64_t k = lo; k < hi; ++k)
      if (over_flag[k]) slot_of[vals2[k]] = cap + run++;
  if (t == 1023) live[nb] = part[1023];
}

// number of atoms of brick `b` (clamped to the slots it has) and of the overflow region, from the live counters of a
// forward pass or from the snapshot
__device__ __forceinline__ int bin_count_of(const BinIndex& bi, int b, bool from_live) {
  const int c = from_live ? bi.live[b] : bi.snap[b];
  return b == bi.nb ? c : min(c, bi.cap);
}

// ---- shared device helpers ---------------------------------------------------------------------
template <int LANES, typename T>
__device__ __forceinline__ T group_sum_b(T v) {
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, LANES);
  return v;
}

__device__ __forceinline__ void brick_coords(const BrickGeom& bg, int b, int& bx, int& by, int& bz) {
  bz = b % bg.nbz;
  const int r = b / bg.nbz;
  by = r % bg.nby;
  bx = r / bg.nby;
}

// offset of stencil start relative to a brick origin, mapped into [-(n-1), n_mesh - n]; overlap iff <= BRICK-1
__device__ __forceinline__ int rel_start(int m, int s0, int origin, int nmesh, int order) {
  // m in [0, nmesh), -order < s0 <= 0, 0 <= origin <= nmesh - 4 (bricks_supported): m + s0 - origin lies in (-nmesh, nmesh),
  // so the non-negative residue needs one conditional add (no integer division)
  int r = m + s0 - origin;
  r = r < 0 ? r + nmesh : r;
  if (r > nmesh - order) r -= nmesh;
  return r;
}

// ---- spread: owner-computes per brick, no atomics of any kind ------------------------------------------
// (LDS float atomics turned out to be the limiter of the first brick version: ~30 us for 4 M ds_add_f32.)
//   A1  16 threads per neighbouring brick walk its atom records (independent, coalesced 16-byte loads) and append the
//       atoms whose stencil overlaps this brick to an LDS list ("survivors");
//   A2  survivors are staged up to 256 at a time: thread t writes the three 1-D weight vectors of survivor t to LDS, each
//       PLACED on the brick's 8 points of its axis (zero outside the stencil), the x vector times the value;
//   C   lane = (px,py) column of the brick, 8 z-accumulators in registers; wave w walks survivors w, w+8, ..., three per
//       iteration: the lane's x and y entries and one broadcast row (8 z entries) per survivor, one product, then 8 FMAs
//       (four packed) -- no atomics, no stencil offsets, no inside test, no conditional reads;
//   R   the eight waves' partial bricks are summed through LDS and written with coalesced stores.
// Per-phase clock stamps: tools/spread_phases.py (profiles/r01_j_spread_phases.txt).

// eight consecutive reals of a 16-byte aligned (fp32) LDS row
template <typename T>
__device__ __forceinline__ void load_row8(const T* __restrict__ row, T (&out)[BRICK]) {
  if constexpr (sizeof(T) == 4) {
    const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 4);
    out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w;
    out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
  } else {
#pragma unroll
    for (int k = 0; k < BRICK; ++k) out[k] = row[k];
  }
}

// acc[k] += w * row[k], k < 8: four packed FMAs in fp32 (v_pk_fma_f32)
template <typename T>
__device__ __forceinline__ void fma_row8(T (&acc)[BRICK], T w, const T (&row)[BRICK]) {
  if constexpr (sizeof(T) == 4) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f w2 = {w, w};
#pragma unroll
    for (int k = 0; k < BRICK; k += 2) {
      const v2f r = {row[k], row[k + 1]}, a = {acc[k], acc[k + 1]};
      const v2f o = __builtin_elementwise_fma(w2, r, a);
      acc[k] = o.x;
      acc[k + 1] = o.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < BRICK; ++k) acc[k] += w * row[k];
  }
}

#ifdef MIPME_WG_TIMELINE  // measurement builds only (tools/wg_timeline.py): when and where every workgroup of the launch ran
static __device__ long long g_wg_timeline[4 * 16384];
#define MIPME_WG_STAMP(k)                                                                                   \
  do {                                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                           \
      g_wg_timeline[blockIdx.x * 4 + (k)] = (long long)__builtin_amdgcn_s_memrealtime();                    \
      if ((k) == 0) { /* HW_ID (all 32 bits) and XCC_ID (4 bits) */                                         \
        g_wg_timeline[blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);           \
        g_wg_timeline[blockIdx.x * 4 + 3] = (long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);           \
      }                                                                                                     \
    }                                                                                                       \
  } while (0)
static __device__ long long g_wg_phase[8 * 1024];
#define MIPME_WG_PHASE(k)                                                                              \
  do {                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_phase[blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define MIPME_WG_STAMP(k)
#define MIPME_WG_PHASE(k)
#endif
// The gather launch shares the stamp buffer with the spread launch and comes after it in a step: its stamps are a build of their
// own (-DMIPME_WG_TIMELINE=2, tools/gather_timeline.py), or a step's spread timeline is overwritten from workgroup 0 up.
#if defined(MIPME_WG_TIMELINE) && MIPME_WG_TIMELINE == 2
#define MIPME_WG_TIMELINE_GATHER 1
#define MIPME_WG_STAMP_GATHER(k) MIPME_WG_STAMP(k)
#else
#define MIPME_WG_TIMELINE_GATHER 0
#define MIPME_WG_STAMP_GATHER(k)
#endif

#ifndef MIPME_STAGE_SELECT
#define MIPME_STAGE_SELECT 0  // 1: the staging rows of the brick spread through select chains (A/B builds)
#endif
#ifndef MIPME_SPREAD_UC
#define MIPME_SPREAD_UC 3  // survivors per iteration of the spread's accumulation loop (4 measured 1 % slower, r02_experiments.txt)
#endif
#ifndef MIPME_SPREAD_PADROWS
// 1: zero rows behind the staged survivors, so that the accumulation loop reads UC rows at constant offsets without testing for
// the list's end (-8 vector instructions per iteration).  Measured SLOWER in the binned launches (cfg3 20.65 -> 21.0 us, cfg5
// 174 -> 178 us, cfg2 17.3 -> 17.65 us: all six row reads of an iteration are then in flight at once on an LDS pipe that is the
// phase's limit anyway -- ~1.5 KB per survivor and wave); kept as a build option (profiles/r04_experiments.txt).
#define MIPME_SPREAD_PADROWS 0
#endif
#ifndef MIPME_CELL_WAVES
#define MIPME_CELL_WAVES 6  // waves per SIMD asked of the co-scheduled kernels that also form the cell sums (68-72 registers)
#endif
#ifndef MIPME_LIVE_PADROWS
// the same for the live-list spread (live_spread_body), where it measured FASTER: live step 0.0582 -> 0.0566 ms at cfg3
#define MIPME_LIVE_PADROWS 1
#endif
static constexpr int SPREAD_THREADS = 512;
static constexpr int SPREAD_WAVES = SPREAD_THREADS / 64;
static constexpr int SPREAD_GROUP = 16;                           // threads per neighbouring brick in the candidate scan
static constexpr size_t SPREAD_LDS_MAX = 64 * 1024;
// Sparse bricks (see GATHER_THREADS_SPARSE): a brick workgroup is a chain of memory round trips whatever it holds, so with 16
// atoms per brick (256^3 at water density: 32 768 bricks) the spread is bound by how many bricks a CU keeps in flight -- three
// of the 512-thread workgroups (384 us for 526 848 atoms).  The sparse variant runs the same phases with 128 threads, a
// 64-row staging area and shorter rounds: 11 KB of LDS, a dozen bricks per CU.
static constexpr int SPREAD_THREADS_SPARSE = 128;
// candidates per thread and round; groups are served 16 threads each, THREADS / 16 at a time
template <int THREADS> struct SpreadShape {
  static constexpr int kWaves = THREADS / 64;
  static constexpr int kGroupsPerPass = THREADS / SPREAD_GROUP;
  static constexpr int kPasses = (28 + kGroupsPerPass - 1) / kGroupsPerPass;  // 27 neighbouring bricks + the overflow region
  static constexpr int kCpt = THREADS >= 512 ? 6 : 2;                         // 96 / 32 candidates per brick and round
  static constexpr int kRound = 28 * SPREAD_GROUP * kCpt;                     // capacity of the survivor lists
};
static constexpr int SPREAD_ROUND = SpreadShape<SPREAD_THREADS>::kRound;

static_assert(SPREAD_WAVES * BRICK_PTS >= 4 * kShiftTableSize + kErfcxLdsDoubles + kExp2Tab,
              "the shift table of a row workgroup (4 reals per code) and the fp64 body's erfcx table live in the staging region");
static inline size_t spread_lds_bytes(int order, size_t real_bytes, int stage_rows, bool sparse = false, bool live = false) {
  const int waves = sparse ? SpreadShape<SPREAD_THREADS_SPARSE>::kWaves : SPREAD_WAVES;
  const int round = sparse ? SpreadShape<SPREAD_THREADS_SPARSE>::kRound : SPREAD_ROUND;
  // (+ the zero rows behind the staged survivors that the accumulation loop reads instead of testing for the list's end)
  const size_t region = std::max<size_t>(size_t(waves) * BRICK_PTS,
                                         size_t(stage_rows + ((live ? MIPME_LIVE_PADROWS : MIPME_SPREAD_PADROWS) ? (MIPME_SPREAD_UC - 1) * waves : 0)) *
                                             spread_row_reals(order, real_bytes));
  return real_bytes * region + sizeof(int) * (round + 2);
}
static inline int spread_stage_rows(int order, size_t real_bytes) {
  for (int rows : {256, 192, 128})
    if (spread_lds_bytes(order, real_bytes, rows) <= SPREAD_LDS_MAX) return rows;
  return 0;
}
static constexpr int kSpreadStageRowsSparse = 64;

template <typename T>
struct SpreadArgs {
  Geom g;
  BrickGeom bg;
  int C;
  BinIndex bins;
  bool from_live;  // forward: brick counts from the live counters (and snapshot them); backward: from the snapshot
  const int4* rec;
  const T* wts;
  const T* val;
  const T* qs = nullptr;  // per-slot copy of val (forward spread of single-channel charges after the binning pass), or NULL
  T scale;
  T* mesh;
  int stage_rows;
  bool det;  // deterministic mode: order every round of survivors by slot before staging
  const int* skip;  // nullable: return at once if *skip == 1 (mipme_set_skip_flag)
};

// block = index of the brick (workgroup index among the spread workgroups of the launch)
template <int N, typename T, int THREADS = SPREAD_THREADS>
__device__ __forceinline__ void spread_brick_body(const SpreadArgs<T>& args, unsigned block) {
  using Shape = SpreadShape<THREADS>;
  constexpr int WAVES = Shape::kWaves, PASSES = Shape::kPasses, CPT = Shape::kCpt, ROUND = Shape::kRound;
  const Geom& g = args.g;
  const BrickGeom& bg = args.bg;
  const int C = args.C;
  const BinIndex& bins = args.bins;
  const int4* __restrict__ rec = args.rec;
  const T* __restrict__ wts = args.wts;
  const T* __restrict__ val = args.val;
  const T scale = args.scale;
  T* __restrict__ mesh = args.mesh;
  const int stage_rows = args.stage_rows;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (args.from_live && threadIdx.x == 0) {  // per-call snapshot of this brick's count for the gathers and the backward pass
    bins.snap[block] = bin_count_of(bins, int(block), true);
    if (block == 0) bins.snap[bins.nb] = bin_count_of(bins, bins.nb, true);
  }
  constexpr int SW = 3 * BRICK;  // staged reals per survivor (spread_row_reals)
  constexpr int PAD_ROWS = MIPME_SPREAD_PADROWS ? (MIPME_SPREAD_UC - 1) * WAVES : 0;  // zero rows behind the staged survivors
  const int region = max(WAVES * BRICK_PTS, (stage_rows + PAD_ROWS) * SW);
  T* stage = reinterpret_cast<T*>(smem_raw);                // [stage_rows + PAD_ROWS][SW] staged weights + value
  T* part = stage;                                          // [waves][512] partial bricks (aliases the stage, phase R)
  // survivors of a round: their slot indices (the launch's LDS caps the workgroups per CU of the co-scheduled launch, rows
  // included)
  int* sidx = reinterpret_cast<int*>(stage + region);       // [ROUND]
  int& nsurv = sidx[ROUND];
  int& maxlen = sidx[ROUND + 1];
  int bx, by, bz;
  brick_coords(bg, block, bx, by, bz);
  const int ox = bx * BRICK, oy = by * BRICK, oz = bz * BRICK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // candidate scan: 16 threads per neighbouring brick (27 x 16 = 432 of the 512 threads; the 128-thread variant serves the
  // groups eight at a time) walk that brick's atom records in rounds -- the thread -> (brick, atom) mapping needs no search
  // and the 16-byte record loads stay coalesced; a 28th group walks the overflow region (atoms whose brick was full:
  // normally none)
  const int sub = tid % SPREAD_GROUP;
  int gstart[PASSES], glen[PASSES];
  unsigned need[PASSES];  // reach_code bits a candidate of this group must have; 0x100: the overflow group (tested by position)
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int grp = p * Shape::kGroupsPerPass + tid / SPREAD_GROUP;
    gstart[p] = 0;
    glen[p] = 0;
    need[p] = 0x100u;
    if (grp < 27) {
      const int dx = grp / 9 - 1, dy = (grp / 3) % 3 - 1, dz = grp % 3 - 1;
      // an atom of the neighbour at offset d reaches this brick iff it reaches back: up (bit 1) from below, down (bit 0) from above
      need[p] = (dx < 0 ? 2u : dx > 0 ? 1u : 0u) | (dy < 0 ? 8u : dy > 0 ? 4u : 0u) | (dz < 0 ? 32u : dz > 0 ? 16u : 0u);
      const int nbr = (wrap1(bx + dx, bg.nbx) * bg.nby + wrap1(by + dy, bg.nby)) * bg.nbz + wrap1(bz + dz, bg.nbz);
      gstart[p] = nbr * bins.cap;
      glen[p] = bin_count_of(bins, nbr, args.from_live);
    } else if (grp == 27) {
      gstart[p] = int(bins.over_base);
      glen[p] = bin_count_of(bins, bins.nb, args.from_live);
    }
  }
  MIPME_WG_PHASE(0);
  if (tid == 0) maxlen = 0;
  __syncthreads();
  if (sub == 0) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) atomicMax(&maxlen, glen[p]);
  }
  __syncthreads();
  const int total = maxlen;  // longest of the 27 candidate lists
  MIPME_WG_PHASE(1);
  constexpr int s0 = stencil_start<N>();
  const int px = lane >> 3, py = lane & 7;  // this lane's (x,y) column of the brick
  const int64_t plane = int64_t(g.ny) * g.nz, M = plane * g.nx;
  for (int c = 0; c < C; ++c) {
    T acc[BRICK];
#pragma unroll
    for (int k = 0; k < BRICK; ++k) acc[k] = T(0);
    for (int round = 0; round < total; round += SPREAD_GROUP * CPT) {
      if (tid == 0) nsurv = 0;
      __syncthreads();
      // A1: which candidate stencils overlap this brick?  (ROUND candidates per round = list slots)
      constexpr int NC = CPT * PASSES;
      int cidx[NC];
      unsigned ccode[NC];
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
#pragma unroll
        for (int v = 0; v < CPT; ++v) {
          const int k = round + v * SPREAD_GROUP + sub;
          cidx[p * CPT + v] = k < glen[p] ? gstart[p] + k : -1;
        }
      }
      // candidates of the 27 neighbouring bricks: one byte each (reach_code, written by the binning pass); the overflow group
      // (atoms whose brick was full: normally none) is tested by position as before
#pragma unroll
      for (int u = 0; u < NC; ++u) ccode[u] = (bins.codes && need[u / CPT] != 0x100u) ? bins.codes[cidx[u] >= 0 ? cidx[u] : 0] : 0u;
#pragma unroll
      for (int u = 0; u < NC; ++u) {
        if (cidx[u] >= 0) {
          const unsigned nd = need[u / CPT];
          bool keep;
          if (nd != 0x100u && bins.codes) {
            keep = (ccode[u] & nd) == nd;
          } else {
            const int4 cr = rec[cidx[u]];
            keep = rel_start(cr.x, s0, ox, g.nx, N) < BRICK && rel_start(cr.y, s0, oy, g.ny, N) < BRICK &&
                   rel_start(cr.z, s0, oz, g.nz, N) < BRICK;
          }
          if (keep) sidx[atomicAdd(&nsurv, 1)] = cidx[u];
        }
      }
      __syncthreads();
      MIPME_WG_PHASE(2);
      const int ns = nsurv;
      if (args.det && ns > 1) {  // rank by counting on the (unique) slot index: the order of the LDS atomics above drops out
        constexpr int PER = (ROUND + THREADS - 1) / THREADS;
        int key[PER], rnk[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int t = tid + u * THREADS;
          key[u] = 0;
          rnk[u] = 0;
          if (t < ns) {
            key[u] = sidx[t];
            int r = 0;
            for (int v = 0; v < ns; ++v) r += sidx[v] < key[u];
            rnk[u] = r;
          }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          if (tid + u * THREADS < ns) sidx[rnk[u]] = key[u];
        }
        __syncthreads();
      }
      for (int chunk = 0; chunk < ns; chunk += stage_rows) {
        const int nst = min(stage_rows, ns - chunk);
        // A2: stage the survivor's weights, placed on the brick: row = [wz | wx * value | wy], 8 entries each, entry k of an
        // axis = weight of the stencil point that falls on the brick's point k of that axis, zero if none does.  With all
        // three vectors placed the accumulation below needs neither the stencil offsets nor an "inside the stencil" test:
        // a lane reads its x and y entries at its own coordinates and everything outside the stencil multiplies by zero.
        if (tid < nst) {
          const int si = sidx[chunk + tid];
          const T* wr = wts + int64_t(si) * wts_stride<N, T>();
          T* dst = stage + tid * SW;
          const int4 sr = rec[si];  // (the scan kept the slot only: where the stencil starts relative to the brick is formed here)
          const int orig = sr.w;
          const int rx = rel_start(sr.x, s0, ox, g.nx, N), ry = rel_start(sr.y, s0, oy, g.ny, N), rz = rel_start(sr.z, s0, oz, g.nz, N);
          T w1[3][N];
#pragma unroll
          for (int t = 0; t < N; ++t) {
            w1[0][t] = wr[2 * N + t];
            w1[1][t] = wr[t];
            w1[2][t] = wr[N + t];
          }
          // (forward spread of single-channel charges: the binning pass left them by slot -- no load that waits for the record)
          const T v = (args.qs ? args.qs[si] : val[int64_t(orig) * C + c]) * scale;
          const int r3[3] = {rz, rx, ry};
#if MIPME_STAGE_SELECT
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
            for (int k = 0; k < BRICK; ++k) {
              T w = T(0);
#pragma unroll
              for (int t = 0; t < N; ++t) w = (k - r3[ax] == t) ? w1[ax][t] : w;
              dst[ax * BRICK + k] = ax == 1 ? w * v : w;
            }
          }
#else
          // zeros, then the stencil's weights at their places (LDS stores of one lane land in program order): 3 x N conditional
          // stores instead of 3 x 8 x N selects -- the launch is bound by vector issue and this loop was ~240 instructions per
          // survivor
#pragma unroll
          for (int k = 0; k < SW; ++k) dst[k] = T(0);
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
            for (int t = 0; t < N; ++t) {
              const int k = r3[ax] + t;
              if (unsigned(k) < unsigned(BRICK)) dst[ax * BRICK + k] = ax == 1 ? w1[ax][t] * v : w1[ax][t];
            }
          }
#endif
        } else if (PAD_ROWS && tid < nst + PAD_ROWS) {  // zero rows: the accumulation loop runs over them instead of testing for the end
          T* dst = stage + tid * SW;
#pragma unroll
          for (int k = 0; k < SW; ++k) dst[k] = T(0);
        }
        __syncthreads();
        MIPME_WG_PHASE(3);
        // C: register accumulation; wave w takes survivors w, w+W, ...; UC survivors per iteration so that their LDS reads
        // overlap (the loop is a chain of dependent LDS reads otherwise).  Per survivor and wave: the z row (wave-uniform
        // address: LDS broadcast), the lane's x and y entries, one product, four packed FMAs -- about ten vector
        // instructions; the loop shares the SIMDs with the pair sum's row workgroups, which are bound by the same issue slots.
        constexpr int UC = MIPME_SPREAD_UC;
        const int nstc = __builtin_amdgcn_readfirstlane(nst);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        for (int sv0 = wave_u; sv0 < nstc; sv0 += WAVES * UC) {
#if MIPME_SPREAD_PADROWS
          // (rows sv0 + u * WAVES beyond the last survivor are zero rows: constant offsets from one address, no selects)
          T wz[UC][BRICK], fx[UC], fy[UC];
          const T* sw = stage + sv0 * SW;
#pragma unroll
          for (int u = 0; u < UC; ++u) {
            const T* su = sw + u * WAVES * SW;
            fx[u] = su[BRICK + px];
            fy[u] = su[2 * BRICK + py];
            load_row8<T>(su, wz[u]);  // wave-uniform address: LDS broadcast
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
            load_row8<T>(sw, wz[u]);  // wave-uniform address: LDS broadcast
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
    }
    MIPME_WG_PHASE(4);
    // R: sum the waves' partial bricks and write the owned points (the stage is free again: last sync above)
#pragma unroll
    for (int pz = 0; pz < BRICK; ++pz) part[wave * BRICK_PTS + (px * BRICK + py) * BRICK + pz] = acc[pz];
    __syncthreads();
    for (int k = tid; k < BRICK_PTS; k += THREADS) {
      T v = T(0);
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += part[w * BRICK_PTS + k];
      const int qx = k / (BRICK * BRICK), qy = (k / BRICK) % BRICK, qz = k % BRICK;
      const int gx = ox + qx, gy = oy + qy, gz = oz + qz;
      if (gx < g.nx && gy < g.ny && gz < g.nz) mesh[c * M + gx * plane + int64_t(gy) * g.nz + gz] = v;
    }
    __syncthreads();
    MIPME_WG_PHASE(5);
  }
}

template <int N, typename T>
__global__ __launch_bounds__(SPREAD_THREADS) void spread_brick_kernel(SpreadArgs<T> a) {
  MIPME_SKIP_IF_SET(a.skip);
  const unsigned b = brick_of(a.bg, blockIdx.x);
  if (b < unsigned(a.bg.nb)) spread_brick_body<N, T>(a, b);
}
template <int N, typename T>
__global__ __launch_bounds__(SPREAD_THREADS_SPARSE) void spread_brick_sparse_kernel(SpreadArgs<T> a) {
  MIPME_SKIP_IF_SET(a.skip);
  const unsigned b = brick_of(a.bg, blockIdx.x);
  if (b < unsigned(a.bg.nb)) spread_brick_body<N, T, SPREAD_THREADS_SPARSE>(a, b);
}

// Horizontal fusion of the spread with the fused distance + pair-sum row kernel (rows_body.h): the first `n_spread`
// workgroups are bricks of the spread -- a chain of dependent phases that leaves the vector units idle most of the time --
// and the remaining ones are row workgroups of the VALU-bound pair sum, which fill those issue slots.  The two parts are
// independent (the pair sum reads the atom records that the binning pass emitted, not the mesh); the gather adds the mesh
// part to the potentials the pair sum wrote.

// Register budget of the co-scheduled kernel: since the survivor lists lost their uint16 array the launch's LDS (35 KB) allows
// FOUR workgroups per CU = 8 waves per SIMD, which needs <= 64 VGPRs.  The fp32 / 4-byte-entry kernels use 58 (asked for "at
// least 6 waves", i.e. <= 80); one register more than 64 costs a workgroup per CU (measured with a 65-register build: cfg3 20.7
// -> 22.1 us), and ASKING for 8 waves makes the compiler schedule the bodies more tightly than it has to (20.6 -> 20.8 us,
// cfg5 174.9 -> 175.2 us): the bound stays at 6, `make vgpr-check` style vigilance is on whoever touches the bodies
// (hipcc -S, .vgpr_count of spread_rows_kernel<5, float, 1, true, false>).  The variants with the cell sums use 68-72
// registers (three workgroups per CU).  Other instantiations are left alone.
// CELL: the row workgroups also form the per-wave cell-gradient sums of the energy step (FusedRowsArgs::cpart; packed fp32 body
// and fp64 Coulomb body only: rows_cell_supported below)
// Block order of the co-scheduled launch.  pattern == 0: all bricks, then all row blocks.  pattern == a > 0 (XCD mapping only):
// periods of 8 bricks + 8 a row blocks (one brick and a row blocks per XCD) until one kind runs out, then the rest of the other --
// both kinds keep blockIdx % 8 = slot % 8, so the XCD-contiguous mappings of bricks and rows hold.
static constexpr unsigned kBrickPatternMin = 2048;  // bricks of a launch from which the interleaved block order is used (two generations)
struct CoSlot {
  bool brick;
  unsigned slot;  // index among the (padded) bricks / row blocks
};
__host__ __device__ inline unsigned cosched_periods(unsigned n_pad, unsigned n_rows_pad, unsigned a) {
  const unsigned nb8 = n_pad >> 3, nr8 = n_rows_pad >> 3, need = (nr8 + a - 1) / a;
  return nb8 < need ? nb8 : need;
}
__host__ __device__ inline unsigned cosched_grid(unsigned n_pad, unsigned n_rows_pad, unsigned a) {
  if (a == 0) return n_pad + n_rows_pad;
  const unsigned P = cosched_periods(n_pad, n_rows_pad, a);
  const unsigned left_b = n_pad - 8u * P, rows_done = 8u * a * P;
  return P * 8u * (1u + a) + left_b + (n_rows_pad > rows_done ? n_rows_pad - rows_done : 0u);
}
__device__ __forceinline__ CoSlot cosched_slot(unsigned b, unsigned n_pad, unsigned n_rows_pad, unsigned a) {
  if (a == 0) return CoSlot{b < n_pad, b < n_pad ? b : b - n_pad};
  const unsigned P = cosched_periods(n_pad, n_rows_pad, a), period = 8u * (1u + a);
  if (b < P * period) {
    const unsigned p = b / period, r = b - p * period;
    return r < 8u ? CoSlot{true, 8u * p + r} : CoSlot{false, 8u * a * p + (r - 8u)};
  }
  const unsigned rest = b - P * period;
  return (n_pad >> 3) > P ? CoSlot{true, 8u * P + rest} : CoSlot{false, 8u * a * P + rest};
}

// Which block order a co-scheduled launch of n_spread bricks and n_row_blocks row blocks gets (host side).  Launches of one or
// two generations (cfg3: 512 bricks + 999 row blocks on 1 024 slots) run best with all bricks first; launches of many
// generations (cfg5: 4 096 + 8 192) with one brick per `a` row blocks and XCD, a = the ratio of the two counts, so that the
// bricks' idle vector slots are filled from the start and neither kind is left over as a tail: cfg5 175.2 -> 168.5 us, with
// a = 1 or 3 there 196 / 187 us, any pattern at cfg3 +0.9 us (profiles/r04_k_ab_pattern.txt).  MIPME_BRICK_PATTERN = 0 (bricks
// first) or a > 0 overrides.  (The frames launch -- blockIdx.y = frame, bricks first inside every frame -- is interleaved at
// frame granularity as it is; one brick per two row blocks inside the frames measured 1 % slower at 4 and 8 headline frames,
// profiles/r04_m_ab_pattern_frames.txt.)
static inline unsigned brick_pattern(const BrickGeom& bg, unsigned n_spread, unsigned n_row_blocks, bool fp32) {
  static const int pattern_env = [] { const char* e = getenv("MIPME_BRICK_PATTERN"); return e ? atoi(e) : -1; }();
  unsigned pattern = 0;
  if (bg.xcd && n_spread > 0) {
    if (pattern_env >= 0)
      pattern = unsigned(pattern_env);
    else if (fp32 && n_spread >= kBrickPatternMin) {  // (measured for the fp32 kernels, four workgroups per CU; fp64 keeps bricks first)
      const double ratio = double(pad8(n_row_blocks)) / double(pad8(n_spread));
      const unsigned a = unsigned(ratio + 0.5);
      if (a >= 1 && std::fabs(ratio - double(a)) <= 0.15 * double(a)) pattern = a;
    }
  }
  return pattern;
}

// the row workgroup of a co-scheduled launch (shared by spread_rows_kernel and plane_rows_kernel)
template <typename T, int PFAST, bool COMPACT, bool CELL, int BS = SPREAD_THREADS>
__device__ __forceinline__ void cosched_row_block(const FusedRowsArgs<T>& ra, unsigned r, char* smem_rows) {
  AtomRecord<T>* tab = reinterpret_cast<AtomRecord<T>*>(smem_rows);
  bool done = false;
  if constexpr (COMPACT && std::is_same<T, float>::value) {
    if (CELL || !ra.dist_out) {  // uniform: nobody asked for the distances -> the packed body (rows_body.h)
      sr_rows_pk_body<PFAST, BS, CELL>(ra, r, tab);
      done = true;
    }
  }
#if MIPME_ROW_LANES == 16
  if constexpr (COMPACT && std::is_same<T, double>::value && (PFAST == 1 || PFAST == 6)) {
    if (CELL || !ra.dist_out) {  // ... and its fp64 counterpart (erfc from the LDS table; 1/r^6: closed form)
      sr_rows_f64_body<BS, CELL, PFAST>(ra, r, smem_rows);
      done = true;
    }
  }
#endif
  if constexpr (!CELL) {
    if (!done) sr_fused_rows_body<T, kPotForce, false, PFAST, false, true, BS, 0, COMPACT>(ra, r, tab);
  }
}

template <int N, typename T, int PFAST, bool COMPACT, bool CELL>
__device__ __forceinline__ void spread_rows_body(const SpreadArgs<T>& sa, const FusedRowsArgs<T>& ra, unsigned n_spread,
                                                 unsigned pattern) {
  MIPME_WG_STAMP(0);
  // n_spread bricks (0: a rows-only launch) and the row blocks; both through the XCD-contiguous mapping when sa.bg.xcd
  const unsigned n_pad = sa.bg.xcd ? pad8(n_spread) : n_spread;
  const unsigned n_row_blocks = unsigned((ra.N + kRowsPerSpreadBlock - 1) / kRowsPerSpreadBlock);
  const unsigned n_rows_pad = sa.bg.xcd ? pad8(n_row_blocks) : n_row_blocks;
  const CoSlot cs = cosched_slot(blockIdx.x, n_pad, n_rows_pad, pattern);
  extern __shared__ __attribute__((aligned(16))) char smem_rows[];
  if (cs.brick) {
    const unsigned b = brick_of(sa.bg, cs.slot);
    if (cs.slot < n_pad && b < n_spread) spread_brick_body<N, T>(sa, b);
  } else if (cs.slot < n_rows_pad) {
    const unsigned r = sa.bg.xcd ? xcd_contiguous(cs.slot, n_row_blocks) : cs.slot;
    // row workgroups keep their shift table in the launch's dynamic LDS (the spread's staging area, which they do not use)
    if (r < n_row_blocks) cosched_row_block<T, PFAST, COMPACT, CELL>(ra, r, smem_rows);
  }
#ifdef MIPME_WG_TIMELINE
  __syncthreads();
#endif
  MIPME_WG_STAMP(1);
}

template <int N, typename T, int PFAST, bool COMPACT, bool CELL = false>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? (CELL ? MIPME_CELL_WAVES : 6) : 1) void spread_rows_kernel(
    SpreadArgs<T> sa, FusedRowsArgs<T> ra, unsigned n_spread, unsigned pattern) {
  spread_rows_body<N, T, PFAST, COMPACT, CELL>(sa, ra, n_spread, pattern);
}
// ... the same kernel held to 80 scalar registers (MIPME_SGPR_CAP), for the instantiations whose VECTOR registers admit four
// workgroups per CU and that take the cap without spilling -- spread_rows_sgpr_capped() names them.  Left to itself the compiler
// gives them 88-94, i.e. three workgroups per CU: cfg5 (1/r^6, 262 144 atoms, 12 generations of workgroups) 0.280 -> 0.270 ms,
// launch 180 -> 169 us; cfg3 on the bricks 0.0631 -> 0.0624 ms (profiles/r05_experiments.txt item 8).
template <int N, typename T, int PFAST, bool COMPACT, bool CELL = false>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? (CELL ? MIPME_CELL_WAVES : 6) : 1) MIPME_SGPR_CAP void spread_rows_capped_kernel(
    SpreadArgs<T> sa, FusedRowsArgs<T> ra, unsigned n_spread, unsigned pattern) {
  spread_rows_body<N, T, PFAST, COMPACT, CELL>(sa, ra, n_spread, pattern);
}
template <int N, typename T, bool CELL>
constexpr bool spread_rows_sgpr_capped() {
#ifdef MIPME_SGPR_CAP_OFF
  return false;
#else
  return sizeof(T) == 4 && N == 5 && !CELL;  // (N = 4 fp32 and every fp64 instantiation answer the cap with scratch or more VGPRs)
#endif
}

// The pair sum alone (sparse-brick path: the bricks ran in a launch of their own): 256-thread workgroups with nothing but the
// shift table in LDS and no register bound, i.e. full occupancy -- the same bodies, the same per-wave energy partial sums.
template <typename T, int PFAST, bool COMPACT, bool CELL = false>
__global__ __launch_bounds__(256) void rows_only_kernel(FusedRowsArgs<T> ra, int xcd) {
  constexpr bool F64_BODY = COMPACT && std::is_same<T, double>::value && (PFAST == 1 || PFAST == 6) && kRowLanes == 16;
  __shared__ __attribute__((aligned(16))) char tab_raw[F64_BODY ? kRowsF64LdsBytes : sizeof(AtomRecord<T>) * kShiftTableSize];
  AtomRecord<T>* tab = reinterpret_cast<AtomRecord<T>*>(tab_raw);
  constexpr int BS = 256;
  const unsigned n_row_blocks = unsigned((ra.N + BS / kRowLanes - 1) / (BS / kRowLanes));
  const unsigned r = xcd ? xcd_contiguous(blockIdx.x, n_row_blocks) : blockIdx.x;
  if (r >= n_row_blocks) return;
  if constexpr (COMPACT && std::is_same<T, float>::value) {
    if (CELL || !ra.dist_out) {
      sr_rows_pk_body<PFAST, BS, CELL>(ra, r, tab);
      return;
    }
  }
#if MIPME_ROW_LANES == 16
  if constexpr (F64_BODY) {
    if (CELL || !ra.dist_out) {
      sr_rows_f64_body<BS, CELL, PFAST>(ra, r, tab_raw);
      return;
    }
  }
#endif
  if constexpr (!CELL) sr_fused_rows_body<T, kPotForce, false, PFAST, false, true, BS, 0, COMPACT>(ra, r, tab);
}

// the pair bodies that can form the cell sums: 4-byte entries, fp32 (1/r, 1/r^6) or fp64 Coulomb, no distance by-product
template <typename T>
static inline bool rows_cell_supported(int pfast, int shift_format, const void* dist_out) {
  if ((shift_format & kShiftFormatMask) != kShiftTable32 || dist_out || kRowLanes != 16) return false;
  return pfast == 1 || pfast == 6;
}

// ---- plane spread: the charges scattered straight into a (y,z) plane's transform tile -------------------------------------
// Round 5.  The owner-computes bricks above cost a third of the co-scheduled launch's vector instructions (2.6 M of 7.8 M at
// cfg3: every survivor is a 512-point rank-1 update of which 7 % is not zero), and the plane transform that follows re-reads
// the mesh they wrote in a launch of its own (6.7 us of pure latency).  Here `parts` workgroups per x plane of the mesh each
//   A  walk their slice of the plane's atoms -- the PLANE LISTS the binning pass leaves (bin slots by reference point m_x; a
//      plane takes the lists of m_x = x - s0 - t, t < N) as one sequence, in batches of blockDim atoms, a lane per atom, the next
//      batch's record / weights / charge in flight while the current one is scattered,
//   B  add every atom's N x N (y,z) stencil points, times its x weight and charge, to the plane in LDS with 64-bit LDS atomics --
//      N^2 per atom and plane, N^3 per atom in all, nothing is computed that is zero: fp32 meshes in 64-bit fixed point
//      (ds_add_u64, plane_item_scatter), fp64 meshes with ds_add_f64,
//   C  convert the plane to the working precision in the layout the forward transform starts from (rows as bit-reversed complex
//      pairs), transform it in place (yz_forward_finish) and store their block of a half-complex mesh: part 0 into the
//      convolution's buffer, the others into the plan's part buffers -- the transform is linear, and the x stage of the convolution
//      adds the parts on load (kfilter.hip XCellExtra::hat_more).  The convolution's forward (y,z) launch is gone
//      (fft_plan_forward_done), and so is the real charge mesh (callers say they do not read it: MIPME_FWD_RHO_MESH_UNUSED).
// Why 64-bit atomics for fp32 meshes: tools/r05/lds_atomic_bench.hip (profiles/r05_b_lds_atomic.txt) -- one 2 500-atom plane pass
// is 80 us with ds_add_f32 (0.4 lanes per clock and CU, whatever the denormal mode: round 1's "LDS float atomics are the
// limiter"), 13 us with ds_add_f64, 8 us with ds_add_u64, 3.7 us with plain stores; the cost is per wave-level instruction, which
// is why the lanes must be dense (the lists).  Why several workgroups per plane: a plane's atomics go through ONE CU's LDS pipe
// (18 us of a 25 us workgroup with one per plane, profiles/r05_g_plane_timeline.txt).  Integer sums do not depend on the order
// of arrival: with one workgroup per plane the fp32 mesh is bit-reproducible.  MIPME_DETERMINISTIC=1 keeps the bricks (its slots
// come from a sort).  Single channel, planes whose accumulation tile fits the co-scheduled launch's LDS budget (64 x 64); larger
// meshes keep the bricks.  Numbers: profiles/r05_experiments.txt item 3.
template <typename T>
struct PlaneArgs {
  Cplx<T>* hat = nullptr;  // (nx, ny, nz/2 + 1): receives the (y,z)-transformed planes; nullptr: no plane spread in this launch
  int logny = 0, loglz = 0;
  int tile_off = 0, tw_off = 0;  // byte offsets of the transform tile (0: it aliases the accumulation tile) and of the twiddles
  int misc_off = 0;              // ... and of 16 ints of bookkeeping (list lengths)
  // `parts` workgroups per plane: part k takes the k-th slice of every plane list and transforms its own partial plane into
  // hat (k = 0) / hat_more + (k - 1) * more_stride; the x stage of the convolution adds the transforms up (XCellExtra::hat_more)
  int parts = 1;
  Cplx<T>* hat_more = nullptr;
  int64_t more_stride = 0;
  // Planes whose accumulation tile does not fit the co-scheduled launch's LDS (128 x 128: round 6) are spread in BANDS of
  // band_rows rows (a power of two, ny = bands * band_rows; bands == 1: the whole plane): a workgroup per (plane, band) accumulates
  // its rows, transforms them along z only and stores them; the y columns follow as a launch of their own (kfilter.hip ycols,
  // fft_plan_set_forward_ycols).  The plane lists are then keyed by y slice (BinIndex::pband_shift) and a band reads the sub-lists
  // of its own slices and of the two neighbouring ones (an atom's stencil reaches at most one slice beyond its own).
  int band_rows = 0, bands = 1;
};

// LDS of a launch with planes: co-scheduled with the row blocks, four workgroups per CU must fit 160 KB (and the rows need their
// shift / erfcx tables: <= 32 KB); alone, the default dynamic limit
static constexpr size_t kPlaneLdsCosched = 39 * 1024;
static inline size_t plane_tile_bytes(int ny, int nz, size_t real_bytes) { return 2 * real_bytes * size_t(ny) * (size_t(nz / 2) + 1); }
static inline size_t plane_tw_bytes(int ny, int nz, size_t real_bytes) {  // (ny = 0: no y stage, z twiddles only)
  const size_t Lz = size_t(nz / 2), Ltab = size_t(ny) > Lz ? size_t(ny) : Lz;
  return 2 * real_bytes * (Ltab / 2 + (Lz + 1));
}
// accumulation tile (rows x nz doubles); the fp32 transform tile is smaller and takes its place, the fp64 one sits behind it.
// rows = ny: a whole plane (y stage in the tile); rows < ny: a band (z stage only)
template <typename T>
static inline void plane_lds_layout(int rows, int nz, PlaneArgs<T>& pa, size_t& total, bool ystage = true) {
  const size_t acc = sizeof(double) * size_t(rows) * nz, tile = plane_tile_bytes(rows, nz, sizeof(T));
  const size_t tile_off = sizeof(T) == 4 ? 0 : acc;
  const size_t tw_off = sizeof(T) == 4 ? (acc > tile ? acc : tile) : acc + tile;
  pa.tile_off = int(tile_off);
  pa.tw_off = int(tw_off);
  pa.misc_off = int(tw_off + plane_tw_bytes(ystage ? rows : 0, nz, sizeof(T)));
  total = size_t(pa.misc_off) + 640;
}
// rows of a plane one workgroup of the co-scheduled launch accumulates: the whole plane if its tile fits, else the largest band
// of ny / 2, ny / 4, ny / 8 rows that does (at least as many rows as a y slice of the plane lists: ny / kPlaneSub); 0: none fits
static inline int plane_band_rows(const mipme_mesh_t* m, int dtype) {
  static const bool bands_env = env_flag("MIPME_PLANE_BANDS", true);
  for (int rows = m->ny; rows >= 4 && rows * kPlaneSub >= m->ny; rows >>= 1) {
    size_t need = 0;
    if (dtype == MIPME_F32) {
      PlaneArgs<float> pa;
      plane_lds_layout<float>(rows, m->nz, pa, need, rows == m->ny);
    } else {
      PlaneArgs<double> pa;
      plane_lds_layout<double>(rows, m->nz, pa, need, rows == m->ny);
    }
    if (need <= kPlaneLdsCosched) return rows;
    if (!bands_env) break;
  }
  return 0;
}
// whole planes in the co-scheduled launch (the frame batches' condition: no bands there)?
static inline bool plane_fits_cosched(const mipme_mesh_t* m, int dtype) { return plane_band_rows(m, dtype) == m->ny; }

static inline bool sparse_bricks(int64_t n_atoms, int nb);
// Entries per plane list of the bins, 0 if this mesh / system does not use the plane spread: single channel, power-of-two planes
// whose tiles fit the co-scheduled launch's LDS, dense bricks, not the deterministic mode (its slots come from a sort).
// Per SUB-list (kPlaneSub of them per plane): 4 x its mean occupancy + 32; the rest goes to the plane overflow list.
static int plane_list_capacity(const mipme_mesh_t* m, int64_t N, int dtype) {
  static const bool plane_env = env_flag("MIPME_PLANE_SPREAD", true);
  if (!plane_env || deterministic_mode() || m->n_channels != 1 || N <= 0) return 0;
  const bool pow2 = (m->ny & (m->ny - 1)) == 0 && (m->nz & (m->nz - 1)) == 0 && m->nz >= 4 && m->ny >= 2;
  if (!pow2 || m->nx < 2 * BRICK) return 0;
  // (the entries pack (m_x, m_y, m_z) into 11 + 10 + 10 bits of a non-negative int)
  if (m->nx > (1 << 11) || m->ny > (1 << kPlanePackBits) || m->nz > (1 << kPlanePackBits)) return 0;
  if (sparse_bricks(N, make_brick_geom(m).nb)) return 0;
  const int band_rows = plane_band_rows(m, dtype);
  if (band_rows == 0) return 0;
  if (band_rows < m->ny && m->ny / kPlaneSub < 8) return 0;  // (bands: a y slice must hold a stencil's reach, N - 1 <= 6 rows)
  const int64_t lists = int64_t(m->nx) * kPlaneSub, mean = (N + lists - 1) / lists, all = (N + 15) / 16 * 16;
  int64_t cap = (4 * mean + 32 + 15) / 16 * 16;
  if (cap > all) cap = all;
  return int(cap);
}

// what a lane holds of one atom when it scatters it
template <int N, typename T>
struct PlaneItem {
  T vx;      // charge * scale * x weight of this plane; 0 for lanes without an atom
  int my, mz;
  T wy[N], wz[N];
};

// a plane-list entry as loaded (plane_entry_words): what a lane holds of the NEXT batches while the current one is scattered
template <int N, typename T>
struct PlaneRaw {
  T w[plane_entry_words<N, T>()];
};

// entry `idx` of the entry array `base` (lanes without an atom read entry 0 -- always inside the buffer -- and get zero products)
template <int N, typename T>
__device__ __forceinline__ void plane_raw_load(PlaneRaw<N, T>& r, bool ok, const T* __restrict__ base, int64_t idx) {
  constexpr int EW = plane_entry_words<N, T>(), V = 16 / int(sizeof(T));
  struct alignas(16) Chunk {
    T e[V];
  };
  const Chunk* src = reinterpret_cast<const Chunk*>(base + (ok ? idx : 0) * EW);
#pragma unroll
  for (int k = 0; k < EW / V; ++k) {
    const Chunk c = src[k];
#pragma unroll
    for (int u = 0; u < V; ++u) r.w[k * V + u] = c.e[u];
  }
#pragma unroll
  for (int t = 3; t < EW; ++t) r.w[t] = ok ? r.w[t] : T(0);
}
template <typename T>
__device__ __forceinline__ int plane_raw_packed(T w0) {
  if constexpr (sizeof(T) == 4)
    return __float_as_int(w0);
  else
    return int(__double_as_longlong(w0));
}

// entry -> item of stencil row tt: the y / z weights from their offsets (the same weights_1d the binning pass evaluates for the
// gather's rows), the product charge * w_x[tt] picked from the entry
template <int SCHEME, int N, typename T>
__device__ __forceinline__ void plane_item_make(PlaneItem<N, T>& it, const PlaneRaw<N, T>& r, int tt, T scale) {
  const int packed = plane_raw_packed(r.w[0]);
  it.my = (packed >> kPlanePackBits) & ((1 << kPlanePackBits) - 1);
  it.mz = packed & ((1 << kPlanePackBits) - 1);
  T qwx[N], unused[N];
#pragma unroll
  for (int t = 0; t < N; ++t) qwx[t] = r.w[3 + t];
  it.vx = pick<N, T>(qwx, tt) * scale;
  weights_1d<SCHEME, N, false, T>(r.w[1], it.wy, unused);
  weights_1d<SCHEME, N, false, T>(r.w[2], it.wz, unused);
}

// one atom's N x N points of the plane (natural layout acc[y * nz + z]): the products in the working precision (as the bricks
// form them).  The sums: fp64 meshes in double with ds_add_f64; fp32 meshes in 64-bit FIXED POINT with ds_add_u64 (8.0 against
// 13.1 us per 2 500-atom pass, tools/r05/lds_atomic_bench.hip) -- value * fx_scale (a power of two chosen so that the sum of
// ALL |contributions| of the plane stays below 2^50: the atoms of its lists x the largest |charge|, BinIndex::wmax) rounded to an integer by the add-a-magic-number conversion
// (x + 1.5 * 2^52 holds round(x) in its low 52 bits, two's complement; the constant's bit pattern has a zero low word, so taking
// it off is one 32-bit subtraction), and integer sums do not depend on the order of arrival: the mesh is bit-reproducible.
static constexpr unsigned kFxMagicHi = 0x43380000u;  // high word of the bit pattern of 1.5 * 2^52
template <int N, typename T, bool BANDED>
__device__ __forceinline__ void plane_item_scatter(double* __restrict__ acc, const Geom& g, const PlaneItem<N, T>& it, double fx_scale,
                                                   int row_lo, int rows) {
  // (row_lo, rows): the rows of the plane this tile holds -- (0, ny) for a whole plane, a band's otherwise: stencil rows outside
  // it belong to another workgroup
  constexpr int s0 = stencil_start<N>();
  int zo[N];
#pragma unroll
  for (int k = 0; k < N; ++k) zo[k] = wrap1(it.mz + s0 + k, g.nz);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    int r = wrap1(it.my + s0 + j, g.ny);
    if constexpr (BANDED) {
      r -= row_lo;
      if (unsigned(r) >= unsigned(rows)) continue;
    }
    const int row = r * g.nz;
    const T ay = it.vx * it.wy[j];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if constexpr (sizeof(T) == 4) {
        const double t = __builtin_fma(double(ay * it.wz[k]), fx_scale, 6755399441055744.0);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(t) - ((unsigned long long)kFxMagicHi << 32);
        atomicAdd(reinterpret_cast<unsigned long long*>(&acc[row + zo[k]]), bits);
      } else {
        atomicAdd(&acc[row + zo[k]], double(ay * it.wz[k]));
      }
    }
  }
}

template <int SCHEME, int N, typename T, bool BANDED = false>
__device__ __forceinline__ void plane_spread_yz_body(const SpreadArgs<T>& args, const PlaneArgs<T>& pa, unsigned item,
                                                     char* smem) {
  // item = (plane * bands + band) * parts + part: the workgroups of one plane are neighbours in the launch (same XCD: they read the
  // same lists).  bands == 1: the tile is the whole plane; else band_rows rows of it (parts == 1 then)
  // (BANDED is a template argument: the whole-plane instantiation is round 5's code, constants and all)
  constexpr bool banded = BANDED;
  const int n_bands = banded ? pa.bands : 1;
  const unsigned unit = item / unsigned(pa.parts);
  const int part = int(item - unit * unsigned(pa.parts));
  const unsigned plane = banded ? unit / unsigned(n_bands) : unit;
  const int band = banded ? int(unit - plane * unsigned(n_bands)) : 0;
  const Geom& g = args.g;
  const int rows = banded ? pa.band_rows : g.ny, row_lo = band * rows;
  const BinIndex& bins = args.bins;
  const T* __restrict__ plist = (const T*)bins.plist;
  const T* __restrict__ pover = (const T*)bins.pover;
  const int tid = threadIdx.x, nthr = blockDim.x;
  constexpr int s0 = stencil_start<N>();
  MIPME_WG_PHASE(0);
  double* acc = reinterpret_cast<double*>(smem);
  const int npts = rows * g.nz;
  const int x0 = int(plane);
  // Lists of this workgroup: list l = sub-list sub0 + (l % SUBS) (mod kPlaneSub) of plane x0 - s0 - l / SUBS.  Whole planes read
  // all kPlaneSub sub-lists (keyed by wavefront); a band reads the y slices of its rows and one more on either side.
  const int spb = kPlaneSub / n_bands;                        // y slices per band
  const int SUBS = banded ? min(spb + 2, kPlaneSub) : kPlaneSub;
  const int sub0 = banded ? band * spb - 1 : 0;
  const int NL = N * SUBS;
  auto list_of = [&](int l) __attribute__((always_inline)) {
    return posmod(x0 - s0 - l / SUBS, g.nx) * kPlaneSub + ((sub0 + (l % SUBS)) & (kPlaneSub - 1));
  };
  // the prologue's global loads first (list lengths, this lane's share of the per-wavefront charge maxima), the tile zeroing and
  // the twiddles while they are in flight: the prologue was 3 us of a 17 us workgroup
  int my_count = 0;
  if (tid < NL) my_count = bins.plive[list_of(tid)];
  float my_wmax = 0.f;
  if (sizeof(T) == 4 && tid < bins.n_wmax) my_wmax = bins.wmax[tid];
  for (int i = tid; i < npts; i += nthr) acc[i] = 0.0;
  const YzTile<T> yt = banded ? yz_tile_setup<T, false>(rows, g.nz, smem + pa.tile_off, smem + pa.tw_off)
                              : yz_tile_setup<T, true>(g.ny, g.nz, smem + pa.tile_off, smem + pa.tw_off);
  if (args.from_live) {  // per-call snapshot of the brick counts (see spread_brick_body): the workgroups share the bricks among them
    const int n_items = g.nx * n_bands * pa.parts;
    for (int b = int(item) + tid * n_items; b <= bins.nb; b += n_items * nthr) bins.snap[b] = bin_count_of(bins, b, true);
  }
  // The atoms of this tile: the lists above, taken as ONE sequence (list 0, then list 1, ...) of which this part owns an even slice,
  // walked in batches of blockDim atoms -- every lane of a batch but the last holds an atom, so the N^2 LDS atomics of a batch are
  // dense.  lst[l] = entries of the sequence before list l (lst[NL] = all), lst[64 + l] = the length of list l  (LDS, read back
  // with a per-lane index: kept in registers and picked by tt the compiler spills them to a stack array)
  int* lst = reinterpret_cast<int*>(smem + pa.misc_off);
  static_assert(N * kPlaneSub < 64, "bookkeeping of the plane lists");
  double* fxs = reinterpret_cast<double*>(lst + 128);  // fixed-point scale and its inverse (fp32 meshes)
  float* wred = reinterpret_cast<float*>(lst + 132);   // per-wavefront maxima of the largest |charge| (<= 16 wavefronts)
  if constexpr (sizeof(T) == 4) {
    // the largest |charge| of the system: max over the binning pass's per-wavefront maxima (NaN-propagating: a NaN charge must
    // reach the result)
    float a = my_wmax;
    bool bad = !(my_wmax == my_wmax);
    for (int i = tid + nthr; i < bins.n_wmax; i += nthr) {
      const float w = bins.wmax[i];
      bad |= !(w == w);
      a = fmaxf(a, w);
    }
    a = fmaxf(a, dpp_mov<0xB1>(a));
    a = fmaxf(a, dpp_mov<0x4E>(a));
    a = fmaxf(a, dpp_mov<0x141>(a));
    a = fmaxf(a, dpp_mov<0x140>(a));
    a = fmaxf(fmaxf(read_lane(a, 0), read_lane(a, 16)), fmaxf(read_lane(a, 32), read_lane(a, 48)));
    if (__builtin_amdgcn_ballot_w64(bad) != 0) a = __builtin_nanf("");
    if ((tid & 63) == 0) wred[tid >> 6] = a;
  }
  if (tid < NL) lst[64 + tid] = min(my_count, bins.pcap);
  __syncthreads();
  // prefix sums by NL + 1 lanes in parallel (a single lane walking 40 LDS reads one after the other was 1 us of every plane
  // workgroup), the fixed-point scale by a lane of another wavefront
  if (tid <= NL) {
    int run = 0;
    for (int u = 0; u < tid; ++u) run += lst[64 + u];
    lst[tid] = run;
  }
  if constexpr (sizeof(T) == 4) {
    if (tid == 64) {
      // |sum over the plane's atoms of q w| <= (atoms of the plane's lists + overflow list) x max |q| x |scale| < 2^e  ->  times
      // 2^(50 - e) every sum stays below 2^50.  A bound that is not finite (NaN / inf charges) makes the scale NaN, and with it
      // every point of the plane: the NaN guard of the gather then sees what the reference's would
      float qmax = 0.f;
      bool bad = false;
      for (int w = 0; w < (nthr >> 6); ++w) {
        bad |= !(wred[w] == wred[w]);
        qmax = fmaxf(qmax, wred[w]);
      }
      int n_atoms = bins.plive[g.nx * kPlaneSub] + 1;
      for (int u = 0; u < NL; ++u) n_atoms += lst[64 + u];
      int e = 0;
      const double bd = double(qmax) * 1.00001 * fabs(double(args.scale)) * double(n_atoms);
      (void)frexp(bd, &e);
      const bool finite = !bad && bd == bd && bd < 1e300;
      fxs[0] = finite ? ldexp(1.0, 50 - e) : __builtin_nan("");
      fxs[1] = finite ? ldexp(1.0, e - 50) : __builtin_nan("");
    }
  }
  __syncthreads();
  MIPME_WG_PHASE(1);
  const double fx_scale = sizeof(T) == 4 ? fxs[0] : 1.0;
  // A product that is not finite (NaN / inf weights: positions that are not finite) would become a large finite integer in the
  // magic-number conversion below; the lane that meets one poisons the plane's inverse scale instead, so that the plane comes out
  // NaN as it does with float sums (one weight per axis decides: all N weights of an axis come from the same coordinate)
  auto guard_item = [&](const PlaneItem<N, T>& it) __attribute__((always_inline)) {
    if constexpr (sizeof(T) == 4) {
      const T chk = it.vx * it.wy[0] * it.wz[0];
      if (!(__builtin_fabsf(chk) <= 3.0e38f)) fxs[1] = __builtin_nan("");
    }
  };
  const int total = lst[NL];
  const int lo = int(int64_t(total) * part / pa.parts), hi = int(int64_t(total) * (part + 1) / pa.parts);
  const int n_batches = (hi - lo + nthr - 1) / nthr;
  // this lane's entry of batch b: its stencil row tt and the entry's index in the list array (-1: none) -- arithmetic on the
  // prologue's list lengths, no load: the entries of batches b + 1 and b + 2 are in flight while batch b is scattered
  auto entry_of = [&](int b, int& tt) __attribute__((always_inline)) -> int64_t {
    const int gidx = lo + b * nthr + tid;
    int l = 0;
    for (int u = 1; u < NL; ++u) l += gidx >= lst[u] ? 1 : 0;
    tt = l / SUBS;
    if (b >= n_batches || gidx >= hi) return -1;
    return int64_t(list_of(l)) * bins.pcap + (gidx - lst[l]);
  };
  int tt0 = 0, tt1 = 0;
  PlaneRaw<N, T> r0, r1;
  {
    const int64_t e0 = entry_of(0, tt0), e1 = entry_of(1, tt1);
    plane_raw_load<N, T>(r0, e0 >= 0, plist, e0);
    plane_raw_load<N, T>(r1, e1 >= 0, plist, e1);
  }
  for (int b = 0; b < n_batches; ++b) {
    int tt2 = 0;
    const int64_t e2 = entry_of(b + 2, tt2);
    PlaneRaw<N, T> r2;
    plane_raw_load<N, T>(r2, e2 >= 0, plist, e2);  // (past the last batch: entry 0, never used)
    PlaneItem<N, T> cur;
    plane_item_make<SCHEME, N, T>(cur, r0, tt0, args.scale);
    if (cur.vx != T(0)) {
      guard_item(cur);
      plane_item_scatter<N, T, BANDED>(acc, g, cur, fx_scale, row_lo, rows);
    }
    r0 = r1;
    tt0 = tt1;
    r1 = r2;
    tt1 = tt2;
  }
  if (part == 0) {  // the plane overflow list (atoms whose plane list was full: normally none)
    const int oc = bins.plive[g.nx * kPlaneSub];
    for (int i = tid; i < oc; i += nthr) {
      PlaneRaw<N, T> r;
      plane_raw_load<N, T>(r, true, pover, i);
      int d = x0 - (plane_raw_packed(r.w[0]) >> (2 * kPlanePackBits)) - s0;
      d += d < 0 ? g.nx : 0;
      d -= d >= g.nx ? g.nx : 0;
      if (d < N) {
        PlaneItem<N, T> it;
        plane_item_make<SCHEME, N, T>(it, r, d, args.scale);
        guard_item(it);
        plane_item_scatter<N, T, BANDED>(acc, g, it, fx_scale, row_lo, rows);
      }
    }
  }
  __syncthreads();
  MIPME_WG_PHASE(2);
  const double fx_inv = sizeof(T) == 4 ? fxs[1] : 1.0;  // (read AFTER the scatter: a lane may have poisoned it, guard_item)
  // C: to the working precision and the transform's layout (rows as complex sequences c_j = a_2j + i a_2j+1, bit-reversed for
  // the DIT z transform); the real plane itself for callers that keep the charge mesh.  The fp32 tile aliases the accumulation
  // tile: a chunk's values travel through registers, and rows are written in the order they were read (a tile row is shorter
  // than an accumulation row, so the writes never reach rows that are still to be read).
  {
    const int Lz = yt.Lz, RZ = yt.RZ, npairs = rows * Lz;
    constexpr int CH = 4;
    for (int base = 0; base < npairs; base += CH * nthr) {
      Cplx<T> v[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int idx = base + u * nthr + tid;
        if constexpr (sizeof(T) == 4) {
          const long long* ai = reinterpret_cast<const long long*>(acc);
          v[u] = idx < npairs ? Cplx<T>{T(double(ai[2 * idx]) * fx_inv), T(double(ai[2 * idx + 1]) * fx_inv)} : Cplx<T>{T(0), T(0)};
        } else {
          v[u] = idx < npairs ? Cplx<T>{T(acc[2 * idx]), T(acc[2 * idx + 1])} : Cplx<T>{T(0), T(0)};
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int idx = base + u * nthr + tid;
        if (idx < npairs) {
          const int y = idx / Lz, j = idx - y * Lz;
          const int jr = pa.loglz ? int(__brev(unsigned(j)) >> (32 - pa.loglz)) : 0;
          yt.tile[y * RZ + jr] = v[u];
        }
      }
      __syncthreads();
    }
  }
  MIPME_WG_PHASE(3);
  Cplx<T>* dst = part == 0 ? pa.hat : pa.hat_more + int64_t(part - 1) * pa.more_stride;
  if (banded)  // z rows only, y in natural order: the y columns are a launch of their own (kfilter.hip ycols)
    yz_forward_finish<T, false>(yt, rows, g.nz, 0, pa.loglz, dst + (int64_t(plane) * g.ny + row_lo) * yt.RZ);
  else
    yz_forward_finish<T, true>(yt, g.ny, g.nz, pa.logny, pa.loglz, dst + int64_t(plane) * g.ny * yt.RZ);
  MIPME_WG_PHASE(4);
}

template <int SCHEME, int N, typename T, bool BANDED = false>
__global__ __launch_bounds__(1024) void plane_spread_kernel(SpreadArgs<T> sa, PlaneArgs<T> pa) {
  MIPME_SKIP_IF_SET(sa.skip);
  extern __shared__ __attribute__((aligned(16))) char smem_plane[];
  plane_spread_yz_body<SCHEME, N, T, BANDED>(sa, pa, blockIdx.x, smem_plane);
}

// planes first, then the row blocks of the pair sum (the planes are few -- nx -- and long: they must start at once)
template <int SCHEME, int N, typename T, int PFAST, bool COMPACT, bool CELL, bool BANDED>
__device__ __forceinline__ void plane_rows_body(const SpreadArgs<T>& sa, const PlaneArgs<T>& pa, const FusedRowsArgs<T>& ra,
                                                unsigned n_planes, unsigned n_row_blocks) {
  MIPME_WG_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_pr[];
  const unsigned n_pad = pad8(n_planes);
  if (blockIdx.x < n_pad) {
    const unsigned p = xcd_contiguous(blockIdx.x, n_planes);
    if (p < n_planes) plane_spread_yz_body<SCHEME, N, T, BANDED>(sa, pa, p, smem_pr);
  } else {
    const unsigned r = xcd_contiguous(blockIdx.x - n_pad, n_row_blocks);
    if (r < n_row_blocks) cosched_row_block<T, PFAST, COMPACT, CELL>(ra, r, smem_pr);
  }
#ifdef MIPME_WG_TIMELINE
  __syncthreads();
#endif
  MIPME_WG_STAMP(1);
}
template <int SCHEME, int N, typename T, int PFAST, bool COMPACT, bool CELL = false, bool BANDED = false>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? (CELL ? MIPME_CELL_WAVES : 6) : 1) void plane_rows_kernel(
    SpreadArgs<T> sa, PlaneArgs<T> pa, FusedRowsArgs<T> ra, unsigned n_planes /* plane workgroups: nx * bands * parts */,
    unsigned n_row_blocks) {
  plane_rows_body<SCHEME, N, T, PFAST, COMPACT, CELL, BANDED>(sa, pa, ra, n_planes, n_row_blocks);
}
// ... held to 80 scalar registers for the fp32 instantiations (four workgroups per CU: 82 would admit three -- see
// spread_rows_capped_kernel; the fp64 ones are bound by their vector registers and answer the cap with scratch)
template <int SCHEME, int N, typename T, int PFAST, bool COMPACT, bool CELL = false, bool BANDED = false>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? (CELL ? MIPME_CELL_WAVES : 6) : 1) MIPME_SGPR_CAP void plane_rows_capped_kernel(
    SpreadArgs<T> sa, PlaneArgs<T> pa, FusedRowsArgs<T> ra, unsigned n_planes, unsigned n_row_blocks) {
  plane_rows_body<SCHEME, N, T, PFAST, COMPACT, CELL, BANDED>(sa, pa, ra, n_planes, n_row_blocks);
}

// ---- gather with an LDS halo tile ----------------------------------------------------------------
// NT = number of meshes staged (1: potential gather, 2: phi and chi for the gradient gather)
static constexpr int GATHER_THREADS = 512;

// sparse bricks (a 1 A mesh over a dilute system, or 256^3 meshes at water density: 16 atoms per brick): a quarter-size
// workgroup keeps the gather's lanes busy and four times as many bricks in flight per CU (the kernel is a chain of memory
// round trips per brick: 526 848 atoms on 256^3, 32 768 bricks: 237 us with 512 threads per brick)
static constexpr int GATHER_THREADS_SPARSE = 128;
static constexpr int kSparseBrickAtoms = 40;  // mean atoms per brick at or below which the sparse variants are launched ...
static constexpr int kSparseMinBricks = 4096;  // ... on meshes with many generations of bricks (fewer: latency matters, not slots)
// MIPME_SPARSE_FORCE=1: the sparse variants for every brick mesh (they are correct at any occupancy) -- how the random sweeps
// of tests/test_gpu_fuzz.py, whose meshes are small, are run through them (tests/test_gpu_parity.py)
static inline bool sparse_bricks(int64_t n_atoms, int nb) {
  static const bool force = env_flag("MIPME_SPARSE_FORCE", false);
  return force || (n_atoms <= int64_t(kSparseBrickAtoms) * nb && nb >= kSparseMinBricks);
}

template <int N, int NT, typename T, int THREADS = GATHER_THREADS>
__device__ __forceinline__ void load_tiles(const Geom& g, int ox, int oy, int oz, const T* __restrict__ m0,
                                           const T* __restrict__ m1, T* tile) {
  constexpr int TL = BRICK + N - 1;
  constexpr int s0 = stencil_start<N>();
  const int64_t plane = int64_t(g.ny) * g.nz;
  for (int k = threadIdx.x; k < TL * TL * TL; k += THREADS) {
    const int tx = k / (TL * TL), ty = (k / TL) % TL, tz = k % TL;
    // (o + s0 + t lies in (-n, 2n): bricks_supported guarantees n > 2 BRICK >= BRICK + N; one conditional add / subtract
    // instead of an integer division per coordinate -- the staging loop was ~120 instructions per element)
    const int gx = wrap1(ox + s0 + tx, g.nx), gy = wrap1(oy + s0 + ty, g.ny), gz = wrap1(oz + s0 + tz, g.nz);
    const int64_t gi = gx * plane + int64_t(gy) * g.nz + gz;
    tile[k] = m0[gi];
    if constexpr (NT == 2) tile[TL * TL * TL + k] = m1[gi];
  }
}

// FIELD: also write field[a] = (1/V) sum_g mesh(g) grad W_a(g) (Cartesian), single channel.  When the backward pass turns out
// to be in energy mode (g = gE * charges) the mesh force is gE q_a field[a] and no gradient gather is needed at all.
//
// Mapping: 8 lanes per atom, lane = t_z (N <= 8), each lane walks the N x N (t_x, t_y) points of its z column of the LDS
// halo tile.  64 atoms per pass cover a whole brick (~60 atoms at 1 A spacing) in ONE iteration, and the reductions are
// three xor steps over 8 lanes.  (The earlier (t_y,t_z)-per-lane mapping needed 4 passes of 16 atoms, each ending in 20
// dependent 32-lane shuffle steps -- measured: 1.5 us per pass, not hidden by prefetching.)
static constexpr int kGatherLanes = 8;

// ---- tail of the energy + forces step, folded into the gather (TAIL = true) ------------------------------------------
// When the potentials the gather completes are final (the pair sum ran before it: co-scheduled launch) the same kernel forms
//   grad_positions  = s q_a (c F_a + field_a)            (what the energy-mode backward computes; F = pair force sums,
//                                                         c = 1/2 for a full list, s = seed[0] or 1), and
//   energy          = sum_a q_a V_a                      (what the caller's (q * V).sum() / weighted_sum computes),
// which removes the energy-reduction and force-assembly launches of a step (4.8 + 4.2 us of 84 at cfg3, both pure launch
// latency).  The energy needs no reduction ACROSS the gather's workgroups (a last-arrival ticket costs 6-8 us of serial
// memory-side atomics at the very end of the step -- measured): it is assembled from partial sums that EARLIER kernels of the
// step left behind, by workgroup 0 while it waits for its mesh tile,
//   E = sum_a q_a V_sr,a                                  per-workgroup sums of the co-scheduled pair kernel (epart_sr[2w])
//     + (1/2V) sum_k mu_k G_k |rho^_k|^2                  per-workgroup sums of the x stage of the convolution (epart_k)
//     - (self/2) sum_a q_a^2 - bg Q^2 / V                 (epart_sr[2w + 1]; Q = Re rho^(0))
// using sum_a q_a gather(phi)_a = <spread(q), phi> = sum_k mu_k G_k |rho^_k|^2 (the gather is the adjoint of the spread;
// un-normalised transforms, mu = multiplicity of a half-grid point).  Sums in fp64, fixed order: deterministic.
template <typename T>
struct GatherTail {
  const T* force;     // (N,3) pair force sums
  T force_scale;      // c
  const T* seed;      // device scalar, nullable (= 1)
  T* grad_pos;        // (N,3)
  T* energy;          // 1
  const double* epart_sr;  // [2 * n_sr]
  const double* epart_k;   // [n_k]
  int n_sr, n_k;
  // the rest of the autograd contract of E = sum q V (nullable): s dE/dq_a = 2 s V_a (V is a symmetric bilinear form of the
  // charges), and per brick the nine sums  R[c][e] = sum_a r_{a,c} (s q_a field_{a,e})  of the cell gradient's atom part
  T* grad_q;
  double* rpart;               // [9 * bricks]
  const AtomRecord<T>* rec4;   // (x, y, z, q) per atom: positions for rpart
  const T* aux_seed;           // factor of grad_q and rpart (nullable: the seed of the positions)
  // live-bin step (nullable): the pinned flag word of the step; if the spread of THIS step has flagged an atom beyond the margin
  // (bit 1) the energy is written as NaN -- a step whose results are invalid says so in what it returns, not only at the next call
  const int* live_flags;
  // frame farm: the energy also goes to a float64 log (mipme.h, energy_log) -- elog[(cursor mod cap) * stride] with elog / cursor
  // already offset by the frame's index in its batch (a cursor per frame: every frame's writer owns one)
  double* elog = nullptr;
  int* elog_cursor = nullptr;
  int elog_cap = 0, elog_stride = 1;
};

// R sums of a workgroup -> rpart[9 * block ...].  r3: lanes 0..2 of every 8-lane atom group hold r_c * gp_l for c = 0..2 (l = the
// lane's Cartesian component of the gradient), zero elsewhere.  Uniform call (barrier inside).
template <int THREADS>
__device__ __forceinline__ void tail_rpart(double (&r3)[3], double* __restrict__ rpart, unsigned block) {
  __shared__ double rred[THREADS / 64][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double v = r3[c];
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 3) rred[wave][3 * c + lane] = v;  // lane = e
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    double v = 0.0;
    for (int w = 0; w < THREADS / 64; ++w) v += rred[w][threadIdx.x];
    rpart[9 * int64_t(block) + threadIdx.x] = v;
  }
}

template <typename T, int THREADS>
__device__ __forceinline__ void tail_energy(const GatherTail<T>& tail, const T* __restrict__ qsum, T inv_vol, T self_c,
                                            T bg_c) {
  __shared__ double tred[THREADS / 64][3];
  double v[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < tail.n_sr; i += THREADS) {
    v[0] += tail.epart_sr[2 * i];
    v[1] += tail.epart_sr[2 * i + 1];
  }
  for (int i = threadIdx.x; i < tail.n_k; i += THREADS) v[2] += tail.epart_k[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double x = v[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if (lane == 0) tred[wave][k] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[3] = {0.0, 0.0, 0.0};
    for (int w = 0; w < THREADS / 64; ++w)
      for (int k = 0; k < 3; ++k) t[k] += tred[w][k];
    const double Q = double(qsum[0]);
    double e = t[0] + 0.5 * double(inv_vol) * t[2] - 0.5 * double(self_c) * t[1] - double(bg_c) * double(inv_vol) * Q * Q;
    if (tail.live_flags && (__hip_atomic_load(tail.live_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & 2)) e = __builtin_nan("");
    tail.energy[0] = T(e);
    if (tail.elog) {
      const int k = tail.elog_cursor[0];
      tail.elog[int64_t(unsigned(k) % unsigned(tail.elog_cap)) * tail.elog_stride] = double(T(e));
      tail.elog_cursor[0] = k + 1;
    }
  }
}

template <int N, bool FIELD, typename T, bool TAIL = false, int THREADS = GATHER_THREADS>
__device__ __forceinline__ void gather_brick_body(const Geom& g, const BrickGeom& bg, int C, const BinIndex& bins,
                                                  const int4* __restrict__ rec, const T* __restrict__ wts,
                                                  const T* __restrict__ mesh, const T* __restrict__ q,
                                                  const T* __restrict__ qsum, T inv_vol, T self_c, T bg_c, bool accumulate,
                                                  T* __restrict__ out, T* __restrict__ raw, T* __restrict__ field,
                                                  unsigned block, const GatherTail<T>* tail = nullptr,
                                                  int* __restrict__ nan_flag = nullptr) {
  static_assert(N <= kGatherLanes, "one lane per z point of the stencil");
  static_assert(!TAIL || FIELD, "the tail needs the mesh field");
  constexpr int LANES = kGatherLanes;
  constexpr int GROUPS = THREADS / LANES;
  constexpr int TL = BRICK + N - 1;
  __shared__ T tile[TL * TL * TL];
  int bx, by, bz;
  brick_coords(bg, block, bx, by, bz);
  const int ox = bx * BRICK, oy = by * BRICK, oz = bz * BRICK;
  // this brick's atoms: its own slots, then -- normally none -- the atoms of the overflow region whose home brick it is
  const int beg = int(block) * bins.cap, end = beg + bins.snap[block];
  const int n_over = bins.snap[bins.nb];
  if (bins.live && threadIdx.x == 0) {  // forward pass, last consumer of the live counters: leave them zero for the next call
    bins.live[block] = 0;
    if (block == 0) bins.live[bins.nb] = 0;
    if (bins.plive) {  // the plane lists' counters (nx * kPlaneSub + the overflow counter), shared out among the bricks
      for (int i = int(block); i <= g.nx * kPlaneSub; i += bins.nb) bins.plive[i] = 0;
    }
  }
  T seed = T(1), seed_aux = T(1);
  if constexpr (TAIL) {
    if (tail->seed) seed = tail->seed[0];
    seed_aux = tail->aux_seed ? tail->aux_seed[0] : seed;
    if (block == 0) tail_energy<T, THREADS>(*tail, qsum, inv_vol, self_c, bg_c);  // uniform per workgroup
  }
  double r3[3] = {0.0, 0.0, 0.0};
  if (beg == end && n_over == 0) {
    if constexpr (TAIL) {
      if (tail->rpart && threadIdx.x < 9) tail->rpart[9 * int64_t(block) + threadIdx.x] = 0.0;
    }
    return;
  }
  const int main_iters = (end - beg + GROUPS - 1) / GROUPS, over_iters = (n_over + GROUPS - 1) / GROUPS;
  const int64_t M = int64_t(g.nx) * g.ny * g.nz;
  const int l = threadIdx.x % LANES, grp = threadIdx.x / LANES;
  const bool lane_active = l < N;
  const int tz = lane_active ? l : 0;
  for (int c = 0; c < C; ++c) {
    if (c > 0) __syncthreads();
    // The loads of a pass are issued in the order of their dependencies and BEFORE the tile is staged, so that the kernel is
    // three dependent memory round trips (brick range; record + weights + tile; charge + potential of the atom) of which
    // the third overlaps the stencil arithmetic -- not five in series (each is a trip to the Infinity Cache: the inputs
    // were written by other XCDs in the previous kernels).
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
      int4 a = rec[id];
      if (!valid) a = make_int4(ox, oy, oz, 0);  // a slot that may never have been written: keep every index derived from it in range
      const T* wr = wts + int64_t(id) * wts_stride<N, T>();
      T wx[N], wy[N], dwx[FIELD ? N : 1], dwy[FIELD ? N : 1];
#pragma unroll
      for (int t = 0; t < N; ++t) {
        wx[t] = wr[t];
        wy[t] = wr[N + t];
        if constexpr (FIELD) {
          dwx[t] = wr[3 * N + t];
          dwy[t] = wr[4 * N + t];
        }
      }
      const T wzv = lane_active ? wr[2 * N + tz] : T(0);
      const T dwzv = (FIELD && lane_active) ? wr[5 * N + tz] : T(0);
      if (!staged) {
        load_tiles<N, 1, T, THREADS>(g, ox, oy, oz, mesh + c * M, nullptr, tile);
        staged = true;
      }
      // the atom's charge and (accumulate) its potential so far: needed only at the end of the pass
      const int64_t o_early = int64_t(a.w) * C + c;
      T q_early = T(0), out_early = T(0);
      if (q) {
        q_early = q[o_early];
        if (accumulate) out_early = out[o_early];
      }
      T f_early = T(0);  // TAIL: lanes 0..2 hold the x, y, z components of the atom's pair force sum
      AtomRecord<T> r_early{T(0), T(0), T(0), T(0)};
      if constexpr (TAIL) {
        f_early = tail->force[3 * int64_t(a.w) + (l < 3 ? l : 0)];
        if (tail->rpart) r_early = tail->rec4[a.w];
      }
      if (it == 0) __syncthreads();  // tile staged (uniform: every thread runs the first pass)
      const int rx = a.x - ox, ry = a.y - oy, rz = a.z - oz;
      const T* tp = tile + ry * TL + (rz + tz);
      T sA = T(0), sB = T(0), sC = T(0);  // sum wx wy M,  sum dwx wy M,  sum wx dwy M   over (t_x, t_y) of this z column
#pragma unroll
      for (int ty = 0; ty < N; ++ty) {
        T sx = T(0), sdx = T(0);
#pragma unroll
        for (int tx = 0; tx < N; ++tx) {
          const T v = tp[(rx + tx) * TL * TL + ty * TL];
          sx += v * wx[tx];
          if constexpr (FIELD) sdx += v * dwx[tx];
        }
        sA += sx * wy[ty];
        if constexpr (FIELD) {
          sB += sdx * wy[ty];
          sC += sx * dwy[ty];
        }
      }
      T acc = sA * wzv;
      if constexpr (FIELD) {
        const T fx = group_sum_b<LANES, T>(sB * wzv) * T(g.nx) * inv_vol;
        const T fy = group_sum_b<LANES, T>(sC * wzv) * T(g.ny) * inv_vol;
        const T fz = group_sum_b<LANES, T>(sA * dwzv) * T(g.nz) * inv_vol;
        if constexpr (TAIL) {
          // lanes 0..2 own one Cartesian component each: field (kept for other consumers) and the assembled gradient
          const int k3 = l < 3 ? l : 0;
          const T fc = T(g.inv[3 * k3]) * fx + T(g.inv[3 * k3 + 1]) * fy + T(g.inv[3 * k3 + 2]) * fz;
          if (l < 3 && valid) {
            const int64_t o = int64_t(a.w);
            field[3 * o + l] = fc;
            tail->grad_pos[3 * o + l] = seed * q_early * (tail->force_scale * f_early + fc);
            if (tail->rpart) {
              const double gp = double(seed_aux * q_early * fc);
              r3[0] += double(r_early.x) * gp;
              r3[1] += double(r_early.y) * gp;
              r3[2] += double(r_early.z) * gp;
            }
          }
        } else if (l == 0 && valid) {
          const int64_t o = int64_t(a.w);
          field[3 * o + 0] = T(g.inv[0]) * fx + T(g.inv[1]) * fy + T(g.inv[2]) * fz;
          field[3 * o + 1] = T(g.inv[3]) * fx + T(g.inv[4]) * fy + T(g.inv[5]) * fz;
          field[3 * o + 2] = T(g.inv[6]) * fx + T(g.inv[7]) * fy + T(g.inv[8]) * fz;
        }
      }
      acc = group_sum_b<LANES, T>(acc);
      if (l == 0 && valid) {
        const int64_t o = int64_t(a.w) * C + c;
        if (q) {
          const T phi = acc * inv_vol;
          const T lr = T(0.5) * (phi - self_c * q_early - T(2) * bg_c * inv_vol * qsum[c]);
          const T v_final = accumulate ? out_early + lr : lr;
          out[o] = v_final;
          if constexpr (TAIL) {
            if (tail->grad_q) tail->grad_q[o] = T(2) * seed_aux * v_final;
          }
          if (nan_flag && lr != lr) *nan_flag = 1;  // NaN guard of kspace_filter.py:189-195 (see mipme.h, nan_flag)
          if (raw) raw[o] = phi;
        } else {
          out[o] = acc;
        }
      }
    }
  }
  if constexpr (TAIL) {
    if (tail->rpart) tail_rpart<THREADS>(r3, tail->rpart, block);  // uniform
  }
}

template <int N, bool FIELD, typename T>
__global__ __launch_bounds__(GATHER_THREADS) void gather_brick_kernel(Geom g, BrickGeom bg, int C, BinIndex bins,
                                                                     const int4* __restrict__ rec,
                                                                     const T* __restrict__ wts,
                                                                     const T* __restrict__ mesh, const T* __restrict__ q,
                                                                     const T* __restrict__ qsum, T inv_vol, T self_c,
                                                                     T bg_c, bool accumulate, T* __restrict__ out,
                                                                     T* __restrict__ raw, T* __restrict__ field,
                                                                     int* __restrict__ nan_flag) {
  const unsigned b = brick_of(bg, blockIdx.x);
  if (b < unsigned(bg.nb))
    gather_brick_body<N, FIELD, T>(g, bg, C, bins, rec, wts, mesh, q, qsum, inv_vol, self_c, bg_c, accumulate, out, raw, field, b,
                                   nullptr, nan_flag);
}

// gather + energy + force assembly (see GatherTail)
// Waves per SIMD the fp32 gather + tail kernels of launches with MANY generations of workgroups are compiled for.  Left alone
// they take 84-93 vector registers at order 5 = 5 waves = TWO 512-thread workgroups per CU; told to fit 6 waves (80 registers)
// they admit THREE at the price of 0 (live bins) or 8 (binned) bytes of scratch: cfg5 (4 096 bricks) gather 39.7 -> 33.2 us,
// step 0.2705 -> 0.2643 ms binned, 0.2727 -> 0.2635 ms with live bins, one box (profiles/r05_experiments.txt item 10).  A
// launch of one generation (cfg3: 512 bricks) only pays for the spill (+0.2 us): the binned gather keeps both builds and picks
// by the number of bricks (DENSE); the frame batches' gather (20 bytes of scratch: no gain on 8 x 512 bricks) is left alone, as
// are orders above 5 (24-92 bytes).  -DMIPME_GATHER_TAIL_WAVES=1: the compiler's own choice everywhere (A/B builds).
#ifndef MIPME_GATHER_TAIL_WAVES
#define MIPME_GATHER_TAIL_WAVES 6
#endif
template <int N, typename T, int THREADS = GATHER_THREADS, bool DENSE = false>
__global__ __launch_bounds__(THREADS, DENSE ? MIPME_GATHER_TAIL_WAVES : 1) void gather_tail_kernel(Geom g, BrickGeom bg, BinIndex bins,
                                                             const int4* __restrict__ rec, const T* __restrict__ wts,
                                                             const T* __restrict__ mesh, const T* __restrict__ q,
                                                             const T* __restrict__ qsum, T inv_vol, T self_c, T bg_c,
                                                             T* __restrict__ out, T* __restrict__ raw, T* __restrict__ field,
                                                             GatherTail<T> tail, int* __restrict__ nan_flag) {
  MIPME_WG_STAMP_GATHER(0);
  const unsigned b = brick_of(bg, blockIdx.x);
  if (b < unsigned(bg.nb))
    gather_brick_body<N, true, T, true, THREADS>(g, bg, 1, bins, rec, wts, mesh, q, qsum, inv_vol, self_c, bg_c, true, out, raw,
                                                 field, b, &tail, nan_flag);
#if MIPME_WG_TIMELINE_GATHER
  __syncthreads();
#endif
  MIPME_WG_STAMP_GATHER(1);
}

// Same lane mapping as gather_brick_kernel (8 lanes per atom, lane = t_z, N x N points per lane).
template <int N, typename T>
__global__ __launch_bounds__(GATHER_THREADS) void gather_grad_brick_kernel(
    Geom g, BrickGeom bg, int C, BinIndex bins, const int4* __restrict__ rec, const T* __restrict__ wts,
    const T* __restrict__ q, const T* __restrict__ gout, const T* __restrict__ phi, const T* __restrict__ chi,
    const T* __restrict__ psi_dc, const T* __restrict__ gscale, T half_inv_vol, T self_c, T bg_c,
    T* __restrict__ grad_pos, T* __restrict__ grad_q, const int* __restrict__ skip) {
  static_assert(N <= kGatherLanes, "one lane per z point of the stencil");
  MIPME_SKIP_IF_SET(skip);
  constexpr int LANES = kGatherLanes;
  constexpr int GROUPS = GATHER_THREADS / LANES;
  constexpr int TL = BRICK + N - 1;
  constexpr int TV = TL * TL * TL;
  // energy mode (gscale != NULL): the upstream gradient is gscale * charges, hence chi = (gscale / 2V) * phi and
  // dc(psi) = (gscale / 2V) * dc(rho): `chi` / `psi_dc` then alias phi / dc(rho) and are scaled on the fly
  const T cs = gscale ? gscale[0] * half_inv_vol : T(1);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);  // [C][2][TV]: phi, chi
  int bx, by, bz;
  brick_coords(bg, blockIdx.x, bx, by, bz);
  const int ox = bx * BRICK, oy = by * BRICK, oz = bz * BRICK;
  const int block = int(blockIdx.x);
  const int beg = block * bins.cap, end = beg + bins.snap[block];
  const int n_over = bins.snap[bins.nb];
  if (beg == end && n_over == 0) return;
  const int main_iters = (end - beg + GROUPS - 1) / GROUPS, over_iters = (n_over + GROUPS - 1) / GROUPS;
  const int64_t M = int64_t(g.nx) * g.ny * g.nz;
  const int l = threadIdx.x % LANES, grp = threadIdx.x / LANES;
  const bool lane_active = l < N;
  const int tz = lane_active ? l : 0;
  // stage phi and chi of every channel once: tile[(2c + {0: phi, 1: chi}) * TV + k]
  for (int c = 0; c < C; ++c) load_tiles<N, 2, T>(g, ox, oy, oz, phi + c * M, chi + c * M, tile + 2 * c * TV);
  __syncthreads();
  for (int it = 0; it < main_iters + over_iters; ++it) {
    bool valid;
    int id;
    if (it < main_iters) {
      const int idx = beg + it * GROUPS + grp;
      valid = idx < end;
      id = valid ? idx : beg;
    } else {  // overflow region: the atoms whose home brick this is
      const int k = (it - main_iters) * GROUPS + grp;
      valid = k < n_over && bins.over_brick[k < n_over ? k : 0] == block;
      id = int(bins.over_base) + (k < n_over ? k : 0);
    }
    int4 a = rec[id];
    if (!valid) a = make_int4(ox, oy, oz, 0);
    const T* wr = wts + int64_t(id) * wts_stride<N, T>();
    T wx[N], wy[N], dwx[N], dwy[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
      wx[t] = wr[t];
      wy[t] = wr[N + t];
      dwx[t] = wr[3 * N + t];
      dwy[t] = wr[4 * N + t];
    }
    const T wzv = lane_active ? wr[2 * N + tz] : T(0), dwzv = lane_active ? wr[5 * N + tz] : T(0);
    const int rx = a.x - ox, ry = a.y - oy, rz = a.z - oz;
    T sA = T(0), sB = T(0), sC = T(0);  // over (t_x, t_y), summed over channels: wx wy v, dwx wy v, wx dwy v
    for (int c = 0; c < C; ++c) {
      const int64_t o = int64_t(a.w) * C + c;
      const T hc = gout[o] * half_inv_vol;
      const T qc = q[o];
      const T* tp = tile + 2 * c * TV + ry * TL + (rz + tz);
      T schi = T(0);
#pragma unroll
      for (int ty = 0; ty < N; ++ty) {
        T sx = T(0), sdx = T(0), sch = T(0);
#pragma unroll
        for (int tx = 0; tx < N; ++tx) {
          const int off = (rx + tx) * TL * TL + ty * TL;
          const T vphi = tp[off];
          const T vchi = tp[TV + off] * cs;
          const T v = hc * vphi + qc * vchi;
          sx += v * wx[tx];
          sdx += v * dwx[tx];
          sch += vchi * wx[tx];
        }
        sA += sx * wy[ty];
        sB += sdx * wy[ty];
        sC += sx * dwy[ty];
        schi += sch * wy[ty];
      }
      if (grad_q) {
        schi = group_sum_b<LANES, T>(schi * wzv);
        if (l == 0 && valid) grad_q[o] = schi - T(0.5) * self_c * gout[o] - T(2) * bg_c * psi_dc[c] * cs;
      }
    }
    if (grad_pos) {
      const T ax = group_sum_b<LANES, T>(sB * wzv) * T(g.nx);
      const T ay = group_sum_b<LANES, T>(sC * wzv) * T(g.ny);
      const T az = group_sum_b<LANES, T>(sA * dwzv) * T(g.nz);
      if (l == 0 && valid) {
        const int64_t o = int64_t(a.w);
        grad_pos[3 * o + 0] = T(g.inv[0]) * ax + T(g.inv[1]) * ay + T(g.inv[2]) * az;
        grad_pos[3 * o + 1] = T(g.inv[3]) * ax + T(g.inv[4]) * ay + T(g.inv[5]) * az;
        grad_pos[3 * o + 2] = T(g.inv[6]) * ax + T(g.inv[7]) * ay + T(g.inv[8]) * az;
      }
    }
  }
}

// ---- host wrappers -----------------------------------------------------------------------------
#define MIPME_DISPATCH_STENCIL_B(SCHEME_V, ORDER_V, BODY)                                 \
  do {                                                                                    \
    bool _done = true;                                                                    \
    if ((SCHEME_V) == MIPME_P3M) {                                                        \
      switch (ORDER_V) {                                                                  \
        case 1: { constexpr int S = MIPME_P3M, N = 1; BODY; } break;                      \
        case 2: { constexpr int S = MIPME_P3M, N = 2; BODY; } break;                      \
        case 3: { constexpr int S = MIPME_P3M, N = 3; BODY; } break;                      \
        case 4: { constexpr int S = MIPME_P3M, N = 4; BODY; } break;                      \
        case 5: { constexpr int S = MIPME_P3M, N = 5; BODY; } break;                      \
        default: _done = false;                                                           \
      }                                                                                   \
    } else {                                                                              \
      switch (ORDER_V) {                                                                  \
        case 3: { constexpr int S = MIPME_LAGRANGE, N = 3; BODY; } break;                 \
        case 4: { constexpr int S = MIPME_LAGRANGE, N = 4; BODY; } break;                 \
        case 5: { constexpr int S = MIPME_LAGRANGE, N = 5; BODY; } break;                 \
        case 6: { constexpr int S = MIPME_LAGRANGE, N = 6; BODY; } break;                 \
        case 7: { constexpr int S = MIPME_LAGRANGE, N = 7; BODY; } break;                 \
        default: _done = false;                                                           \
      }                                                                                   \
    }                                                                                     \
    if (!_done) {                                                                         \
      set_error("unsupported scheme/order %d/%d", int(SCHEME_V), int(ORDER_V));           \
      return MIPME_EINVAL;                                                                \
    }                                                                                     \
  } while (0)

struct BinsView {
  BinIndex idx;  // live = NULL: set by the caller for forward passes
  int* over_brick;
  int4* rec;
  void* wts;
  void* qs;  // per-slot charge (written by the binning pass for single-channel charges)
  double* epart;
};

static inline BinsView bins_view(const mipme_mesh_t* m, int64_t N, int dtype, void* bins) {
  const BinsLayout l = bins_layout(m, N, dtype);
  const BrickGeom bg = make_brick_geom(m);
  char* b = (char*)bins;
  BinsView v;
  v.over_brick = (int*)(b + l.over_brick);
  v.idx = BinIndex{nullptr, (int*)(b + l.snap), v.over_brick, bg.nb, l.cap, int64_t(bg.nb) * l.cap};
  v.rec = (int4*)(b + l.rec);
  v.wts = (void*)(b + l.wts);
  v.idx.codes = (unsigned char*)(b + l.codes);
  v.qs = (void*)(b + l.qs);
  v.idx.pcap = l.pcap;
  if (l.pcap) {
    const int rows = plane_band_rows(m, dtype);
    if (rows > 0 && rows < m->ny) {
      int sh = 0;
      while ((kPlaneSub << sh) < m->ny) ++sh;
      v.idx.pband_shift = sh;
    }
  }
  v.idx.plist = l.pcap ? (void*)(b + l.plist) : nullptr;
  v.idx.pover = l.pcap ? (void*)(b + l.pover) : nullptr;
  v.idx.wmax = l.pcap ? (float*)(b + l.wmax) : nullptr;
  v.idx.n_wmax = l.pcap ? int((N + 63) / 64) : 0;
  v.epart = (double*)(b + l.epart);
  return v;
}

}  // namespace mipme
#endif  // MIPME_BRICKS_DEVICE_H
