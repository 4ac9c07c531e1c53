// Brick-binned particle <-> mesh kernels (the fast path of csrc/mesh.hip).
//
// The mesh is cut into bricks of 8x8x8 points and the atoms are counting-sorted by the brick that
// holds their stencil base point (three tiny kernels per forward; the backward reuses the bins).
//   spread : one workgroup per brick OWNS its 512 mesh points.  It scans the atoms of the 27 surrounding
//            bricks, keeps those whose n^3 stencil overlaps the brick, accumulates them into an LDS tile
//            with ds_add_f32 and writes the tile with plain coalesced stores: no global atomics (the
//            slowest operation on MI355X, ~21 G/s), no mesh memset, each mesh point written exactly once.
//   gather : one workgroup per brick stages the (8+n-1)^3 mesh tile of its own atoms into LDS with
//            coalesced loads; the n^3 stencil reads of every atom then hit LDS instead of L2.
// Replaces, like mesh.hip, MeshInterpolator.compute_weights / points_to_mesh / mesh_to_points
// (reference lib/mesh_interpolator.py:303-457).
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
  static const bool xcd_map = env_flag("MIPME_XCD_MAP", true);
  b.xcd = xcd_map ? 1 : 0;
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
static constexpr int kPlanePackBits = 10;  // my, mz < 1024 (the plane tiles are far smaller), mx < 4096

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

int plane_bins_capacity(const mipme_mesh_t* m, int64_t N, int dtype) { return plane_list_capacity(m, N, dtype); }

int64_t bins_bytes(const mipme_mesh_t* m, int64_t N, int dtype) {
  if (!bricks_supported(m, dtype)) return 0;
  return int64_t(bins_layout(m, N, dtype).total);
}

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
  const int pl_key = int(((int64_t(block) * blockDim.x + threadIdx.x) >> 6) & (kPlaneSub - 1));  // (uniform) this wavefront's sub-list
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
        const int p0 = __shfl(m[0], leader, 64);
        const unsigned long long peers = __ballot(valid && m[0] == p0);
        if (valid && m[0] == p0) {
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
    if (bi.plive && valid && pl_leader == lane) pbase = atomicAdd(&bi.plive[m[0] * kPlaneSub + pl_key], pl_count);
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
      e = (T*)bi.plist + ((int64_t(m[0]) * kPlaneSub + pl_key) * bi.pcap + pl_slot) * EW;
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
__global__ __launch_bounds__(256) void det_slots_kernel(int64_t Natoms, int cap, const unsigned* __restrict__ keys2,
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
__global__ __launch_bounds__(1024) void det_overflow_kernel(int64_t Natoms, int cap, int nb, const unsigned* __restrict__ vals2,
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
__device__ long long g_wg_timeline[4 * 16384];
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
__device__ long long g_wg_phase[8 * 1024];
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
};

// LDS of a launch with planes: co-scheduled with the row blocks, four workgroups per CU must fit 160 KB (and the rows need their
// shift / erfcx tables: <= 32 KB); alone, the default dynamic limit
static constexpr size_t kPlaneLdsCosched = 39 * 1024;
static inline size_t plane_tile_bytes(int ny, int nz, size_t real_bytes) { return 2 * real_bytes * size_t(ny) * (size_t(nz / 2) + 1); }
static inline size_t plane_tw_bytes(int ny, int nz, size_t real_bytes) {
  const size_t Lz = size_t(nz / 2), Ltab = size_t(ny) > Lz ? size_t(ny) : Lz;
  return 2 * real_bytes * (Ltab / 2 + (Lz + 1));
}
// accumulation tile (ny x nz doubles); the fp32 transform tile is smaller and takes its place, the fp64 one sits behind it
template <typename T>
static inline void plane_lds_layout(int ny, int nz, PlaneArgs<T>& pa, size_t& total) {
  const size_t acc = sizeof(double) * size_t(ny) * nz, tile = plane_tile_bytes(ny, nz, sizeof(T));
  const size_t tile_off = sizeof(T) == 4 ? 0 : acc;
  const size_t tw_off = sizeof(T) == 4 ? (acc > tile ? acc : tile) : acc + tile;
  pa.tile_off = int(tile_off);
  pa.tw_off = int(tw_off);
  pa.misc_off = int(tw_off + plane_tw_bytes(ny, nz, sizeof(T)));
  total = size_t(pa.misc_off) + 640;
}

static inline bool sparse_bricks(int64_t n_atoms, int nb);
// Entries per plane list of the bins, 0 if this mesh / system does not use the plane spread: single channel, power-of-two planes
// whose tiles fit the co-scheduled launch's LDS, dense bricks, not the deterministic mode (its slots come from a sort).
// Per SUB-list (kPlaneSub of them per plane): 4 x its mean occupancy + 32; the rest goes to the plane overflow list.
static int plane_list_capacity(const mipme_mesh_t* m, int64_t N, int dtype) {
  static const bool plane_env = env_flag("MIPME_PLANE_SPREAD", true);
  if (!plane_env || deterministic_mode() || m->n_channels != 1 || N <= 0) return 0;
  const bool pow2 = (m->ny & (m->ny - 1)) == 0 && (m->nz & (m->nz - 1)) == 0 && m->nz >= 4 && m->ny >= 2;
  if (!pow2 || m->nx < 2 * BRICK) return 0;
  if (sparse_bricks(N, make_brick_geom(m).nb)) return 0;
  size_t need = 0;
  if (dtype == MIPME_F32) {
    PlaneArgs<float> pa;
    plane_lds_layout<float>(m->ny, m->nz, pa, need);
  } else {
    PlaneArgs<double> pa;
    plane_lds_layout<double>(m->ny, m->nz, pa, need);
  }
  if (need > kPlaneLdsCosched) return 0;
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
template <int N, typename T>
__device__ __forceinline__ void plane_item_scatter(double* __restrict__ acc, const Geom& g, const PlaneItem<N, T>& it, double fx_scale) {
  constexpr int s0 = stencil_start<N>();
  int zo[N];
#pragma unroll
  for (int k = 0; k < N; ++k) zo[k] = wrap1(it.mz + s0 + k, g.nz);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int row = wrap1(it.my + s0 + j, g.ny) * g.nz;
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

template <int SCHEME, int N, typename T>
__device__ __forceinline__ void plane_spread_yz_body(const SpreadArgs<T>& args, const PlaneArgs<T>& pa, unsigned item,
                                                     char* smem) {
  // item = plane * parts + part: the parts of one plane are neighbours in the launch (same XCD: they read the same bins)
  const unsigned plane = item / unsigned(pa.parts);
  const int part = int(item - plane * unsigned(pa.parts));
  const Geom& g = args.g;
  const BinIndex& bins = args.bins;
  const T* __restrict__ plist = (const T*)bins.plist;
  const T* __restrict__ pover = (const T*)bins.pover;
  const int tid = threadIdx.x, nthr = blockDim.x;
  constexpr int s0 = stencil_start<N>();
  MIPME_WG_PHASE(0);
  double* acc = reinterpret_cast<double*>(smem);
  const int npts = g.ny * g.nz;
  const int x0 = int(plane);
  constexpr int NL = N * kPlaneSub;  // lists of this plane: list l = sub-list (l % kPlaneSub) of plane x0 - s0 - l / kPlaneSub
  // the prologue's global loads first (list lengths, this lane's share of the per-wavefront charge maxima), the tile zeroing and
  // the twiddles while they are in flight: the prologue was 3 us of a 17 us workgroup
  int my_count = 0;
  if (tid < NL) my_count = bins.plive[posmod(x0 - s0 - tid / kPlaneSub, g.nx) * kPlaneSub + (tid % kPlaneSub)];
  float my_wmax = 0.f;
  if (sizeof(T) == 4 && tid < bins.n_wmax) my_wmax = bins.wmax[tid];
  for (int i = tid; i < npts; i += nthr) acc[i] = 0.0;
  const YzTile<T> yt = yz_tile_setup<T, true>(g.ny, g.nz, smem + pa.tile_off, smem + pa.tw_off);
  if (args.from_live) {  // per-call snapshot of the brick counts (see spread_brick_body): the workgroups share the bricks among them
    const int n_items = g.nx * pa.parts;
    for (int b = int(item) + tid * n_items; b <= bins.nb; b += n_items * nthr) bins.snap[b] = bin_count_of(bins, b, true);
  }
  // The atoms of this plane: the plane lists of m_x = x0 - s0 - tt, tt = 0 .. N-1 (stencil row tt of those atoms is this plane),
  // taken as ONE sequence (list 0, then list 1, ...) of which this part owns an even slice, walked in batches of blockDim atoms --
  // every lane of a batch but the last holds an atom, so the N^2 LDS atomics of a batch are dense; the next batch's record,
  // weights and charge are in flight while the current one is scattered, its list entry one batch further ahead.
  // lst[l] = entries of the sequence before list l (lst[NL] = all), lst[64 + l] = the length of list l  (LDS, read back with
  // a per-lane index: kept in registers and picked by tt the compiler spills them to a stack array)
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
    tt = l / kPlaneSub;
    if (b >= n_batches || gidx >= hi) return -1;
    const int list_id = posmod(x0 - s0 - tt, g.nx) * kPlaneSub + (l % kPlaneSub);
    return int64_t(list_id) * bins.pcap + (gidx - lst[l]);
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
      plane_item_scatter<N, T>(acc, g, cur, fx_scale);
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
        plane_item_scatter<N, T>(acc, g, it, fx_scale);
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
    const int Lz = yt.Lz, RZ = yt.RZ, npairs = g.ny * Lz;
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
  yz_forward_finish<T, true>(yt, g.ny, g.nz, pa.logny, pa.loglz, dst + int64_t(plane) * g.ny * yt.RZ);
  MIPME_WG_PHASE(4);
}

template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(1024) void plane_spread_kernel(SpreadArgs<T> sa, PlaneArgs<T> pa) {
  MIPME_SKIP_IF_SET(sa.skip);
  extern __shared__ __attribute__((aligned(16))) char smem_plane[];
  plane_spread_yz_body<SCHEME, N, T>(sa, pa, blockIdx.x, smem_plane);
}

// planes first, then the row blocks of the pair sum (the planes are few -- nx -- and long: they must start at once)
template <int SCHEME, int N, typename T, int PFAST, bool COMPACT, bool CELL = false>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? (CELL ? MIPME_CELL_WAVES : 6) : 1) void plane_rows_kernel(
    SpreadArgs<T> sa, PlaneArgs<T> pa, FusedRowsArgs<T> ra, unsigned n_planes /* plane workgroups: nx * parts */,
    unsigned n_row_blocks) {
  MIPME_WG_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_pr[];
  const unsigned n_pad = pad8(n_planes);
  if (blockIdx.x < n_pad) {
    const unsigned p = xcd_contiguous(blockIdx.x, n_planes);
    if (p < n_planes) plane_spread_yz_body<SCHEME, N, T>(sa, pa, p, smem_pr);
  } else {
    const unsigned r = xcd_contiguous(blockIdx.x - n_pad, n_row_blocks);
    if (r < n_row_blocks) cosched_row_block<T, PFAST, COMPACT, CELL>(ra, r, smem_pr);
  }
#ifdef MIPME_WG_TIMELINE
  __syncthreads();
#endif
  MIPME_WG_STAMP(1);
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
  v.idx.plist = l.pcap ? (void*)(b + l.plist) : nullptr;
  v.idx.pover = l.pcap ? (void*)(b + l.pover) : nullptr;
  v.idx.wmax = l.pcap ? (float*)(b + l.wmax) : nullptr;
  v.idx.n_wmax = l.pcap ? int((N + 63) / 64) : 0;
  v.epart = (double*)(b + l.epart);
  return v;
}

// the per-wave energy partial sums of the co-scheduled pair sum inside the bins buffer (n = number of {e, q^2} pairs)
const void* bins_epart(const mipme_mesh_t* m, int64_t N, int dtype, void* bins, int64_t* n) {
  *n = (N + 64 / kRowLanes - 1) / (64 / kRowLanes);  // waves that hold a valid row
  return bins_view(m, N, dtype, bins).epart;
}

// live: the int[nb + 1] counters of the binning pass, ZERO on entry (plan / frame owned); the forward gather zeroes them
// again.  q + atom_rec (nullable, single channel): also emit the (position, charge) records.  One launch.
template <typename T>
int bins_build(hipStream_t st, const mipme_mesh_t* m, int64_t n_atoms, const void* pos, void* bins, int* live,
               const void* q, void* atom_rec, bool plane_lists) {
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  const Geom g = make_geom(m);
  const BrickGeom bg = make_brick_geom(m);
  BinsView v = bins_view(m, n_atoms, dtype, bins);
  MIPME_REQUIRE(live, "the binning pass needs the live brick counters");
  // plane lists: counters behind the brick counters of the plan (plan_counter_words); not in deterministic mode (slots from a sort)
  if (plane_lists && v.idx.pcap > 0 && !deterministic_mode()) {
    v.idx.plive = live + bg.nb + 1;
  } else {
    v.idx.pcap = 0;
    v.idx.wmax = nullptr;
  }
  MIPME_REQUIRE(bins_layout(m, n_atoms, dtype).slots < (int64_t(1) << 31), "too many bin slots for 32-bit slot indices");
  v.idx.live = live;
  const unsigned blocks = unsigned((n_atoms + 255) / 256);
  const int* slot_of = nullptr;
  if (n_atoms > 0 && deterministic_mode()) {
    const BinsLayout l = bins_layout(m, n_atoms, dtype);
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t arr = al(sizeof(int) * size_t(n_atoms));
    char* base = (char*)bins + l.det;
    unsigned *keys = (unsigned*)base, *vals = (unsigned*)(base + arr), *keys2 = (unsigned*)(base + 2 * arr),
             *vals2 = (unsigned*)(base + 3 * arr);
    int *slots = (int*)(base + 4 * arr), *over_flag = (int*)(base + 5 * arr);
    det_keys_kernel<T><<<blocks, 256, 0, st>>>(g, bg, (m->order % 2) == 0, n_atoms, (const T*)pos, keys, vals);
    MIPME_LAUNCH_CHECK();
    unsigned bits = 1;
    while ((int64_t(1) << bits) < bg.nb) ++bits;
    size_t sb = l.det_sort_bytes;
    MIPME_CHECK_HIP(rocprim::radix_sort_pairs(base + 6 * arr, sb, keys, keys2, vals, vals2, size_t(n_atoms), 0u, bits, st, false));
    det_slots_kernel<<<blocks, 256, 0, st>>>(n_atoms, v.idx.cap, keys2, vals2, slots, over_flag, live);
    MIPME_LAUNCH_CHECK();
    det_overflow_kernel<<<1, 1024, 0, st>>>(n_atoms, v.idx.cap, bg.nb, vals2, over_flag, slots, live);
    MIPME_LAUNCH_CHECK();
    slot_of = slots;
  }
  T* qs = (q && m->n_channels == 1) ? (T*)v.qs : nullptr;  // the plane spread reads the charge by slot
  if (n_atoms > 0) {
    if (n_atoms >= kCoalescedBinAtoms)
      MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                               (bin_atoms_kernel<S, N, T, true><<<blocks, 256, 0, st>>>(g, bg, v.idx, n_atoms, (const T*)pos, v.over_brick,
                                                                                       v.rec, (T*)v.wts, (const T*)q,
                                                                                       (AtomRecord<T>*)atom_rec, slot_of, qs)));
    else
      MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                               (bin_atoms_kernel<S, N, T, false><<<blocks, 256, 0, st>>>(g, bg, v.idx, n_atoms, (const T*)pos, v.over_brick,
                                                                                        v.rec, (T*)v.wts, (const T*)q,
                                                                                        (AtomRecord<T>*)atom_rec, slot_of, qs)));
    MIPME_LAUNCH_CHECK();
  }
  return MIPME_OK;
}

template <typename T>
int spread_bricks(hipStream_t st, const mipme_mesh_t* m, int64_t N, void* bins, const void* val, double scale, void* mesh,
                  int* clear_count, const mipme_sr_job_t* job, bool want_epart, double* cpart, const PlaneHost* ph,
                  bool* used_planes) {
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  const BrickGeom bg = make_brick_geom(m);
  const BinsView v = bins_view(m, N, dtype, bins);
  // sparse bricks: quarter-size brick workgroups in a launch of their own, the pair sum (if any) in a second launch -- its row
  // workgroups then carry only their shift table in LDS; nothing to co-schedule: the bricks alone are many generations
  const bool sparse = sparse_bricks(N, bg.nb);
  const int stage_rows = sparse ? kSpreadStageRowsSparse : spread_stage_rows(m->order, sizeof(T));
  const size_t lds = spread_lds_bytes(m->order, sizeof(T), stage_rows, sparse);
  SpreadArgs<T> sa;
  sa.g = make_geom(m);
  sa.bg = bg;
  sa.C = m->n_channels;
  sa.bins = v.idx;
  sa.bins.live = clear_count;  // forward pass: the live counters (counts are read from them and snapshot); else NULL
  sa.from_live = clear_count != nullptr;
  sa.rec = v.rec;
  sa.wts = (const T*)v.wts;
  sa.val = (const T*)val;
  // the values by bin slot, if the binning pass of this call wrote them (forward spread of single-channel charges)
  sa.qs = (ph && ph->slot_values && clear_count && m->n_channels == 1) ? (const T*)v.qs : nullptr;
  sa.scale = T(scale);
  sa.mesh = (T*)mesh;
  sa.stage_rows = stage_rows;
  sa.det = deterministic_mode();
  sa.skip = job ? nullptr : skip_flag_slot();  // (the co-scheduled forward launch is never conditional)
  // plane spread (see plane_spread_yz_body): the charges go straight into the forward (y,z) transform's tiles; the binning pass
  // of this call has left the plane lists (bins_build(plane_lists = true), same conditions: plane_list_capacity)
  PlaneArgs<T> pa;
  size_t plane_lds = 0;
  if (used_planes) *used_planes = false;
  if (ph && ph->hat && ph->slot_values && !ph->keep_mesh && used_planes && clear_count && v.idx.pcap > 0 && sa.C == 1 && !sparse &&
      !sa.det && N > 0) {
    size_t need = 0;
    plane_lds_layout<T>(m->ny, m->nz, pa, need);
    pa.hat = (Cplx<T>*)ph->hat;
    pa.parts = (ph->parts > 1 && ph->hat_more) ? ph->parts : 1;
    pa.hat_more = (Cplx<T>*)ph->hat_more;
    pa.more_stride = ph->more_stride;
    while ((1 << pa.logny) < m->ny) ++pa.logny;
    while ((1 << pa.loglz) < m->nz / 2) ++pa.loglz;
    // (the row blocks of a co-scheduled launch keep their shift / erfcx tables in the same dynamic region)
    const size_t rows_lds = job ? sizeof(T) * size_t(SPREAD_WAVES) * BRICK_PTS : 0;
    plane_lds = need > rows_lds ? need : rows_lds;
    sa.qs = (const T*)v.qs;
    sa.bins.plive = clear_count + bg.nb + 1;
    *used_planes = true;
  }
  if (job) {
    // co-scheduled pair sum (sr_job_fusable() holds): potentials + speculative force sums (+ distances) of the fused row kernel
    SRPot s;
    int rc = make_srpot(job->pot, s);
    if (rc) return rc;
    const FastRS cf = make_fast_rs(s);
    const int pfast = fast_rs_exponent(s);
    const int lo = 0, hi = job->full_list ? 0 : 1;  // roles that feed the potential (as mipme_sr_rows_fused, transpose = 0)
    const FusedRowsArgs<T> ra = make_fused_rows_args<T>(
        s, cf, job->n_atoms, job->row_ptr, job->entries_shift, job->entries, nullptr, job->positions, job->records,
        job->cell, job->charges, nullptr, lo, hi, job->full_list, 0, job->out, job->force, nullptr, job->dist_out,
        job->shift_format);
    static_assert(kRowsPerSpreadBlock == SPREAD_THREADS / kRowLanes, "epart layout");
    FusedRowsArgs<T> ra_e = ra;
    ra_e.epart = want_epart ? v.epart : nullptr;
    ra_e.cpart = cpart;
    MIPME_REQUIRE(!cpart || rows_cell_supported<T>(pfast, job->shift_format, job->dist_out),
                  "the cell sums of the pair kernel need 4-byte entries, 1/r or 1/r^6 and no distance by-product");
    const unsigned rows_per_block = SPREAD_THREADS / kRowLanes;
    const unsigned n_rows_blocks = unsigned((job->n_atoms + rows_per_block - 1) / rows_per_block);
    const unsigned n_spread = unsigned(bg.nb);
    const size_t lds_k = lds;
    if (sparse) {  // the bricks first, by themselves; then the pair sum in a launch of its own (rows_only_kernel)
      MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                               ((void)S, spread_brick_sparse_kernel<N, T><<<brick_grid(bg), SPREAD_THREADS_SPARSE, lds, st>>>(sa)));
      MIPME_LAUNCH_CHECK();
      const unsigned nrb = unsigned((job->n_atoms + 256 / kRowLanes - 1) / (256 / kRowLanes));
      const unsigned rgrid = bg.xcd ? pad8(nrb) : nrb;
      const bool compact_r = (job->shift_format & kShiftFormatMask) == kShiftTable32;
      if (cpart && pfast == 1)
        rows_only_kernel<T, 1, true, true><<<rgrid, 256, 0, st>>>(ra_e, bg.xcd);
      else if (cpart) {
        rows_only_kernel<T, 6, true, true><<<rgrid, 256, 0, st>>>(ra_e, bg.xcd);
      } else if (pfast == 1 && compact_r)
        rows_only_kernel<T, 1, true><<<rgrid, 256, 0, st>>>(ra_e, bg.xcd);
      else if (pfast == 1)
        rows_only_kernel<T, 1, false><<<rgrid, 256, 0, st>>>(ra_e, bg.xcd);
      else if (compact_r)
        rows_only_kernel<T, 6, true><<<rgrid, 256, 0, st>>>(ra_e, bg.xcd);
      else
        rows_only_kernel<T, 6, false><<<rgrid, 256, 0, st>>>(ra_e, bg.xcd);
      MIPME_LAUNCH_CHECK();
      return MIPME_OK;
    }
    if (pa.hat) {  // planes + row blocks
      const unsigned n_planes = unsigned(m->nx) * unsigned(pa.parts);
      const bool compact_p = (job->shift_format & kShiftFormatMask) == kShiftTable32;
      const unsigned n_here = n_rows_blocks;
      const unsigned pgrid = pad8(n_planes) + pad8(n_here);
#define MIPME_PLANE_ROWS(PF, CO, CE) \
  MIPME_DISPATCH_STENCIL_B(m->scheme, m->order, (plane_rows_kernel<S, N, T, PF, CO, CE><<<pgrid, SPREAD_THREADS, plane_lds, st>>>(sa, pa, ra_e, n_planes, n_here)))
      note_cosched_kernel("plane_rows_kernel");
      if (cpart && pfast == 1)
        MIPME_PLANE_ROWS(1, true, true);
      else if (cpart) {
        MIPME_PLANE_ROWS(6, true, true);
      } else if (pfast == 1 && compact_p)
        MIPME_PLANE_ROWS(1, true, false);
      else if (pfast == 1)
        MIPME_PLANE_ROWS(1, false, false);
      else if (compact_p)
        MIPME_PLANE_ROWS(6, true, false);
      else
        MIPME_PLANE_ROWS(6, false, false);
#undef MIPME_PLANE_ROWS
      MIPME_LAUNCH_CHECK();
      return MIPME_OK;
    }
    const unsigned pattern = brick_pattern(bg, n_spread, n_rows_blocks, sizeof(T) == 4);
    const bool compact = (job->shift_format & kShiftFormatMask) == kShiftTable32;
#define MIPME_SPREAD_ROWS(PF, CO, CE)                                                                                          \
  MIPME_DISPATCH_STENCIL_B(m->scheme, m->order, ((void)S, [&] {                                                                \
    constexpr bool capped = spread_rows_sgpr_capped<N, T, CE>();                                                               \
    const unsigned n_rp = bg.xcd ? pad8(n_rows_blocks) : n_rows_blocks;                                                        \
    const unsigned grid = bg.xcd ? cosched_grid(pad8(n_spread), n_rp, pattern) : n_spread + n_rp;                              \
    note_cosched_kernel(capped ? "spread_rows_capped_kernel" : "spread_rows_kernel");                                          \
    if constexpr (capped)                                                                                                      \
      spread_rows_capped_kernel<N, T, PF, CO, CE><<<grid, SPREAD_THREADS, lds_k, st>>>(sa, ra_e, n_spread, pattern);            \
    else                                                                                                                       \
      spread_rows_kernel<N, T, PF, CO, CE><<<grid, SPREAD_THREADS, lds_k, st>>>(sa, ra_e, n_spread, pattern);                   \
  }()))
    if (cpart && pfast == 1)
      MIPME_SPREAD_ROWS(1, true, true);
    else if (cpart)
      MIPME_SPREAD_ROWS(6, true, true);
    else if (pfast == 1 && compact)
      MIPME_SPREAD_ROWS(1, true, false);
    else if (pfast == 1)
      MIPME_SPREAD_ROWS(1, false, false);
    else if (compact)
      MIPME_SPREAD_ROWS(6, true, false);
    else
      MIPME_SPREAD_ROWS(6, false, false);
#undef MIPME_SPREAD_ROWS
    MIPME_LAUNCH_CHECK();
    return MIPME_OK;
  }
  if (pa.hat)
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             (plane_spread_kernel<S, N, T><<<unsigned(m->nx) * unsigned(pa.parts), 1024, plane_lds, st>>>(sa, pa)));
  else if (sparse)
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, spread_brick_sparse_kernel<N, T><<<brick_grid(bg), SPREAD_THREADS_SPARSE, lds, st>>>(sa)));
  else
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, spread_brick_kernel<N, T><<<brick_grid(bg), SPREAD_THREADS, lds, st>>>(sa)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// the co-scheduled launch exists for the potential + force-sum mode of the fast range-separated potentials (1/r, 1/r^6)
// with table shift codes
bool sr_job_fusable(const mipme_sr_job_t* job) {
  const int fmt = job ? (job->shift_format & kShiftFormatMask) : -1;
  if (!job || !job->pot || !job->force || (fmt != kShiftTable && fmt != kShiftTable32) || job->n_atoms <= 0) return false;
  if ((job->shift_format & kRowsPadded) && job->dist_out) return false;
  SRPot s;
  if (make_srpot(job->pot, s)) return false;
  const int pfast = fast_rs_exponent(s);
  return pfast == 1 || pfast == 6;
}

// tail (nullable): energy + force assembly in the same launch (needs field, accumulate, a single channel)
template <typename T>
int gather_bricks(hipStream_t st, const mipme_mesh_t* m, int64_t N, void* bins, const void* mesh, const void* q,
                  const void* qsum, double self_c, double bg_c, void* out, void* raw, int accumulate, void* field,
                  const GatherTailHost* th, void* nan_flag, int* live) {
  if (N == 0) return MIPME_OK;
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  const Geom g = make_geom(m);
  const BrickGeom bg = make_brick_geom(m);
  BinsView v = bins_view(m, N, dtype, bins);
  v.idx.live = live;  // forward pass: this gather is the last consumer of the live counters and zeroes them
  v.idx.plive = (live && v.idx.pcap > 0) ? live + bg.nb + 1 : nullptr;  // (zero already if the binning pass left them alone)
  MIPME_REQUIRE(!field || m->n_channels == 1, "the field output of the gather is single-channel");
  if (th) {
    MIPME_REQUIRE(field && accumulate && q && qsum && th->force && th->grad_pos && th->energy && th->epart_k,
                  "the gather tail needs the field output, accumulate = 1, pair force sums and output buffers");
    GatherTail<T> tail;
    tail.force = (const T*)th->force;
    tail.force_scale = T(th->force_scale);
    tail.seed = (const T*)th->seed;
    tail.grad_pos = (T*)th->grad_pos;
    tail.energy = (T*)th->energy;
    tail.epart_sr = (const double*)v.epart;
    tail.n_sr = int((N + 64 / kRowLanes - 1) / (64 / kRowLanes));  // waves that hold a valid row (slot = first row / rows per wave)
    tail.epart_k = (const double*)th->epart_k;
    tail.n_k = int(th->n_k);
    if (th->sr_reduced) {  // pre-reduced by the x stage of the convolution (kfilter.hip xconv_kernel, sr_part)
      tail.epart_sr = tail.epart_k + tail.n_k;
      tail.n_sr = tail.n_k;
    }
    tail.grad_q = (T*)th->grad_q;
    tail.rpart = th->rpart;
    tail.rec4 = (const AtomRecord<T>*)th->records;
    tail.aux_seed = (const T*)th->aux_seed;
    tail.live_flags = nullptr;
    tail.elog = th->elog;
    tail.elog_cursor = th->elog_cursor;
    tail.elog_cap = th->elog_cap;
    MIPME_REQUIRE(!tail.rpart || tail.rec4, "the cell sums of the gather need the atom records");
    if (sparse_bricks(N, bg.nb))
      MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                               ((void)S, gather_tail_kernel<N, T, GATHER_THREADS_SPARSE><<<brick_grid(bg), GATHER_THREADS_SPARSE, 0, st>>>(
                                   g, bg, v.idx, v.rec, (const T*)v.wts, (const T*)mesh, (const T*)q, (const T*)qsum,
                                   T(1.0 / m->volume), T(self_c), T(bg_c), (T*)out, (T*)raw, (T*)field, tail, (int*)nan_flag)));
    else
      MIPME_DISPATCH_STENCIL_B(m->scheme, m->order, ((void)S, [&] {
        // more bricks than two workgroups per CU hold at once: the build that admits three (see MIPME_GATHER_TAIL_WAVES)
        if constexpr (sizeof(T) == 4 && N <= 5) {
          if (bg.nb > 2 * 256) {
            gather_tail_kernel<N, T, GATHER_THREADS, true><<<brick_grid(bg), GATHER_THREADS, 0, st>>>(
                g, bg, v.idx, v.rec, (const T*)v.wts, (const T*)mesh, (const T*)q, (const T*)qsum, T(1.0 / m->volume), T(self_c),
                T(bg_c), (T*)out, (T*)raw, (T*)field, tail, (int*)nan_flag);
            return;
          }
        }
        gather_tail_kernel<N, T><<<brick_grid(bg), GATHER_THREADS, 0, st>>>(
            g, bg, v.idx, v.rec, (const T*)v.wts, (const T*)mesh, (const T*)q, (const T*)qsum, T(1.0 / m->volume), T(self_c),
            T(bg_c), (T*)out, (T*)raw, (T*)field, tail, (int*)nan_flag);
      }()));
    MIPME_LAUNCH_CHECK();
    return MIPME_OK;
  }
  if (field)
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, gather_brick_kernel<N, true, T><<<brick_grid(bg), GATHER_THREADS, 0, st>>>(
                                 g, bg, m->n_channels, v.idx, v.rec, (const T*)v.wts, (const T*)mesh, (const T*)q,
                                 (const T*)qsum, T(1.0 / m->volume), T(self_c), T(bg_c), accumulate != 0, (T*)out,
                                 (T*)raw, (T*)field, (int*)nan_flag)));
  else
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, gather_brick_kernel<N, false, T><<<brick_grid(bg), GATHER_THREADS, 0, st>>>(
                                 g, bg, m->n_channels, v.idx, v.rec, (const T*)v.wts, (const T*)mesh, (const T*)q,
                                 (const T*)qsum, T(1.0 / m->volume), T(self_c), T(bg_c), accumulate != 0, (T*)out,
                                 (T*)raw, nullptr, (int*)nan_flag)));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
int gather_grad_bricks(hipStream_t st, const mipme_mesh_t* m, int64_t N, void* bins, const void* q, const void* gout,
                       const void* phi, const void* chi, const void* psi_dc, const void* gscale, double self_c,
                       double bg_c, void* grad_pos, void* grad_q) {
  if (N == 0) return MIPME_OK;
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  const Geom g = make_geom(m);
  const BrickGeom bg = make_brick_geom(m);
  const BinsView v = bins_view(m, N, dtype, bins);
  const int tl = BRICK + m->order - 1;
  const size_t lds = 2 * sizeof(T) * size_t(m->n_channels) * tl * tl * tl;
  MIPME_DISPATCH_STENCIL_B(
      m->scheme, m->order,
      ((void)S, gather_grad_brick_kernel<N, T><<<unsigned(bg.nb), GATHER_THREADS, lds, st>>>(
          g, bg, m->n_channels, v.idx, v.rec, (const T*)v.wts, (const T*)q, (const T*)gout, (const T*)phi,
          (const T*)chi, (const T*)psi_dc, (const T*)gscale, T(0.5 / m->volume), T(self_c), T(bg_c), (T*)grad_pos,
          (T*)grad_q, skip_flag_slot())));
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

// ---- independent frames in one launch (include/mipme.h: mipme_frames_*) -----------------------------------------
// blockIdx.y = frame; every kernel reads its frame's arguments from a device-resident table (built once per batch), so a
// step of F frames is as many launches as a step of one frame.  The bodies are the single-frame kernels' bodies.
template <typename T>
struct FrameDev {
  // binning
  Geom g;
  BrickGeom bg;
  int64_t N;
  const T* pos;
  const T* q;
  BinIndex bins;  // live = the frame's brick counters
  int* over_brick;
  int4* rec;
  T* wts;
  AtomRecord<T>* atom_rec;
  int even;
  // spread + pair sum
  SpreadArgs<T> spread;
  FusedRowsArgs<T> rows;
  unsigned n_row_blocks;
  // gather
  const T* phi_mesh;
  const T* dc;
  T inv_vol, self_c, bg_c;
  T* out;
  T* field;
  // energy, forces
  T* energy;
  const T* force;
  T* grad_pos;
  T force_scale;  // 1/2 for a full list
  // gather tail (energy + forces in the gather launch)
  GatherTail<T> tail;
  bool use_tail;
};

template <int SCHEME, int N, typename T>
__global__ __launch_bounds__(256) void frames_bin_atoms_kernel(const FrameDev<T>* __restrict__ table) {
  const FrameDev<T>& f = table[blockIdx.y];
  if (int64_t(blockIdx.x) * 256 >= f.N) return;
  bin_atoms_body<SCHEME, N, T>(f.g, f.bg, f.bins, f.N, f.pos, f.over_brick, f.rec, f.wts, f.q, f.atom_rec, blockIdx.x, nullptr,
                               const_cast<T*>(f.spread.qs));
}

template <int N, typename T, int PFAST, bool COMPACT>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? 6 : 1) void frames_spread_rows_kernel(const FrameDev<T>* __restrict__ table) {
  const FrameDev<T>& f = table[blockIdx.y];
  const unsigned n_spread = unsigned(f.bg.nb);
  if (blockIdx.x < n_spread)
    spread_brick_body<N, T>(f.spread, blockIdx.x);
  else if (blockIdx.x - n_spread < f.n_row_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem_rows[];  // see spread_rows_kernel
    AtomRecord<T>* tab = reinterpret_cast<AtomRecord<T>*>(smem_rows);
    if constexpr (COMPACT && std::is_same<T, float>::value) {
      if (!f.rows.dist_out) {
        sr_rows_pk_body<PFAST, SPREAD_THREADS>(f.rows, blockIdx.x - n_spread, tab);
        return;
      }
    }
#if MIPME_ROW_LANES == 16
    if constexpr (COMPACT && std::is_same<T, double>::value && (PFAST == 1 || PFAST == 6)) {
      if (!f.rows.dist_out) {
        sr_rows_f64_body<SPREAD_THREADS, false, PFAST>(f.rows, blockIdx.x - n_spread, smem_rows);
        return;
      }
    }
#endif
    sr_fused_rows_body<T, kPotForce, false, PFAST, false, true, SPREAD_THREADS, 2, COMPACT>(f.rows, blockIdx.x - n_spread, tab);
  }
}

// Plane spread for frame batches (round 5): blockIdx.y = frame; the first nx * parts workgroups of a frame are plane workgroups
// (plane_spread_yz_body: part 0 of frame f into its block of the batched half-complex mesh, the other parts into the plan's part
// buffers), the rest its row blocks.  `pa` holds the frame-independent fields; frame_stride = complex values per frame.
template <int SCHEME, int N, typename T, int PFAST, bool COMPACT>
__global__ __launch_bounds__(SPREAD_THREADS, (sizeof(T) == 4 && COMPACT) ? 6 : 1) void frames_plane_rows_kernel(const FrameDev<T>* __restrict__ table,
                                                                                                        PlaneArgs<T> pa, int64_t frame_stride) {
  const FrameDev<T>& f = table[blockIdx.y];
  const unsigned n_items = unsigned(f.g.nx) * unsigned(pa.parts);
  extern __shared__ __attribute__((aligned(16))) char smem_fp[];
  if (blockIdx.x < n_items) {
    pa.hat += int64_t(blockIdx.y) * frame_stride;
    if (pa.hat_more) pa.hat_more += int64_t(blockIdx.y) * frame_stride;
    plane_spread_yz_body<SCHEME, N, T>(f.spread, pa, blockIdx.x, smem_fp);
  } else if (blockIdx.x - n_items < f.n_row_blocks) {
    cosched_row_block<T, PFAST, COMPACT, false>(f.rows, blockIdx.x - n_items, smem_fp);
  }
}

template <int N, typename T>
__global__ __launch_bounds__(GATHER_THREADS) void frames_gather_kernel(const FrameDev<T>* __restrict__ table) {
  const FrameDev<T>& f = table[blockIdx.y];
  gather_brick_body<N, true, T>(f.g, f.bg, 1, f.bins, f.rec, f.wts, f.phi_mesh, f.q, f.dc, f.inv_vol, f.self_c, f.bg_c, true,
                                f.out, nullptr, f.field, blockIdx.x);
}

// the same with the tail: every frame of the batch carries tail scratch
template <int N, typename T>
__global__ __launch_bounds__(GATHER_THREADS) void frames_gather_tail_kernel(const FrameDev<T>* __restrict__ table,
                                                                           const double* __restrict__ epart_k, int n_k) {
  const FrameDev<T>& f = table[blockIdx.y];
  GatherTail<T> tail = f.tail;
  tail.epart_k = epart_k + int64_t(blockIdx.y) * n_k;  // the x stage writes one block of partial sums per batch entry
  tail.n_k = n_k;
  gather_brick_body<N, true, T, true>(f.g, f.bg, 1, f.bins, f.rec, f.wts, f.phi_mesh, f.q, f.dc, f.inv_vol, f.self_c, f.bg_c,
                                      true, f.out, nullptr, f.field, blockIdx.x, &tail);
}

// energy[f] = sum_a q_a V_a: one workgroup per frame, fixed summation order
template <typename T>
__global__ __launch_bounds__(1024) void frames_energy_kernel(const FrameDev<T>* __restrict__ table) {
  const FrameDev<T>& f = table[blockIdx.x];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < f.N; i += 1024) acc += double(f.q[i]) * double(f.out[i]);
  __shared__ double red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += red[w];
    f.energy[0] = T(tot);
  }
}

// grad_positions[f][a] = gscale[f] q_a (c force_a + field_a)
template <typename T>
__global__ __launch_bounds__(256) void frames_finalize_kernel(const FrameDev<T>* __restrict__ table,
                                                             const T* __restrict__ gscale) {
  const FrameDev<T>& f = table[blockIdx.y];
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (t >= 3 * f.N) return;
  f.grad_pos[t] = gscale[blockIdx.y] * f.q[t / 3] * (f.force_scale * f.force[t] + f.field[t]);
}

static void frame_correction_terms(const mipme_potential_t* pot, double& self_c, double& bg_c) {
  // potentials/coulomb.py:144-158, potentials/inversepowerlaw.py:143-166 (as correction_terms in api.hip)
  const int p = pot->kind == MIPME_COULOMB ? 1 : pot->exponent;
  const double two_s2 = 2.0 * pot->smearing * pot->smearing;
  self_c = pot->prefactor / std::tgamma(0.5 * p + 1.0) / std::pow(two_s2, 0.5 * p);
  bg_c = p >= 3 ? 0.0
                : pot->prefactor * std::pow(3.14159265358979323846, 1.5) * std::pow(two_s2, 0.5 * (3 - p)) /
                      ((3 - p) * std::tgamma(0.5 * p));
}

// int32 words of a frame's counter buffer: brick counters + overflow counter, and -- when the plane spread applies to the frame
// (plane_list_capacity) -- the plane lists' counters + their overflow counter
static int64_t frame_counter_ints(const mipme_mesh_t* m, int64_t n_atoms, int dtype) {
  const BrickGeom bg = make_brick_geom(m);
  int64_t n = int64_t(bg.nb) + 1;
  if (plane_list_capacity(m, n_atoms, dtype) > 0) n += int64_t(m->nx) * kPlaneSub + 1;
  return n;
}
static bool frame_plane_lists(const mipme_frame_t& f, int dtype) {
  return plane_list_capacity(&f.mesh, f.n_atoms, dtype) > 0 && int64_t(f.counter_ints) == frame_counter_ints(&f.mesh, f.n_atoms, dtype);
}
int64_t frames_counter_ints(const mipme_mesh_t* m, int64_t n_atoms, int dtype) { return frame_counter_ints(m, n_atoms, dtype); }

static int frames_check(int dtype, int n_frames, const mipme_frame_t* fr) {
  MIPME_REQUIRE(n_frames > 0 && fr, "no frames");
  MIPME_REQUIRE(dtype == MIPME_F32 || dtype == MIPME_F64, "invalid dtype %d", dtype);
  const mipme_mesh_t& m0 = fr[0].mesh;
  for (int k = 0; k < n_frames; ++k) {
    const mipme_frame_t& f = fr[k];
    int rc = validate_mesh(&f.mesh);
    if (rc) return rc;
    MIPME_REQUIRE(f.mesh.nx == m0.nx && f.mesh.ny == m0.ny && f.mesh.nz == m0.nz && f.mesh.scheme == m0.scheme &&
                      f.mesh.order == m0.order && f.mesh.n_channels == 1,
                  "frame %d: all frames need the same mesh, scheme and order and a single channel", k);
    MIPME_REQUIRE(bricks_supported(&f.mesh, dtype) && make_brick_geom(&f.mesh).nb <= 1024,
                  "frame %d: mesh %d x %d x %d is outside the brick kernels' range", k, f.mesh.nx, f.mesh.ny, f.mesh.nz);
    MIPME_REQUIRE(f.n_atoms > 0 && f.positions && f.charges && f.cell && f.atom_bins && f.brick_counters && f.row_ptr &&
                      f.entries_shift && f.entries && f.records && f.rho_mesh && f.phi_mesh && f.dc && f.out && f.force &&
                      f.field && f.energy && f.grad_positions,
                  "frame %d: NULL buffer or no atoms", k);
    // counter_ints was padding before round 5: a caller built against the old header may pass garbage.  Only the three
    // legal values are accepted, so that garbage cannot switch the plane lists on (they write behind the brick counters)
    MIPME_REQUIRE(f.counter_ints == 0 || int64_t(f.counter_ints) == int64_t(make_brick_geom(&f.mesh).nb) + 1 ||
                      int64_t(f.counter_ints) == frame_counter_ints(&f.mesh, f.n_atoms, dtype),
                  "frame %d: counter_ints = %d is neither 0, bricks + 1 = %d nor mipme_frames_counter_ints() = %lld (zero-initialise "
                  "mipme_frame_t)", k, f.counter_ints, make_brick_geom(&f.mesh).nb + 1,
                  (long long)frame_counter_ints(&f.mesh, f.n_atoms, dtype));
    MIPME_REQUIRE((f.shift_format == kShiftTable || f.shift_format == kShiftTable32) && f.shift_format == fr[0].shift_format,
                  "frame %d: the frames path needs the table shift format (1 or 2), the same for every frame", k);
  }
  return MIPME_OK;
}

template <typename T>
static int frames_table_build_t(int n_frames, const mipme_frame_t* fr, const mipme_potential_t* pot, void* host_table) {
  SRPot s;
  int rc = make_srpot(pot, s);
  if (rc) return rc;
  const int pfast = fast_rs_exponent(s);
  MIPME_REQUIRE(pot->smearing > 0 && (pfast == 1 || pfast == 6), "the frames path covers 1/r and 1/r^6 with a smearing");
  const FastRS cf = make_fast_rs(s);
  double self_c, bg_c;
  frame_correction_terms(pot, self_c, bg_c);
  const int dtype = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  FrameDev<T>* out = (FrameDev<T>*)host_table;
  for (int k = 0; k < n_frames; ++k) {
    const mipme_frame_t& f = fr[k];
    const mipme_mesh_t* m = &f.mesh;
    const BinsView v = bins_view(m, f.n_atoms, dtype, f.atom_bins);
    FrameDev<T> d;
    d.g = make_geom(m);
    d.bg = make_brick_geom(m);
    d.N = f.n_atoms;
    d.pos = (const T*)f.positions;
    d.q = (const T*)f.charges;
    d.bins = v.idx;
    d.bins.live = (int*)f.brick_counters;
    d.over_brick = v.over_brick;
    d.rec = v.rec;
    d.wts = (T*)v.wts;
    d.atom_rec = (AtomRecord<T>*)f.records;
    d.even = (m->order % 2) == 0;
    d.spread.g = d.g;
    d.spread.bg = d.bg;
    d.spread.C = 1;
    d.spread.bins = d.bins;
    d.spread.from_live = true;
    d.spread.rec = v.rec;
    d.spread.wts = (const T*)v.wts;
    d.spread.val = (const T*)f.charges;
    d.spread.scale = T(1);
    d.spread.mesh = (T*)f.rho_mesh;
    d.spread.stage_rows = spread_stage_rows(m->order, sizeof(T));
    d.spread.skip = nullptr;
    d.spread.det = false;  // (the frames path keeps the one-pass binning: MIPME_DETERMINISTIC covers single-frame evaluations)
    d.spread.qs = (const T*)v.qs;  // the charge by bin slot (written by the binning pass: bricks' staging and the plane spread)
    if (frame_plane_lists(f, dtype)) {  // plane lists: counters behind the brick counters (mipme_frames_counter_ints)
      d.bins.plive = d.bins.live + d.bg.nb + 1;
    } else {
      d.bins.pcap = 0;
      d.bins.wmax = nullptr;
    }
    d.spread.bins = d.bins;
    d.rows = make_fused_rows_args<T>(s, cf, f.n_atoms, f.row_ptr, f.entries_shift, f.entries, nullptr, f.positions, f.records,
                                     f.cell, f.charges, nullptr, 0, f.full_list ? 0 : 1, f.full_list, 0, f.out, f.force, nullptr,
                                     f.dist_out);
    const int64_t rpb = SPREAD_THREADS / kRowLanes;
    d.n_row_blocks = unsigned((f.n_atoms + rpb - 1) / rpb);
    d.phi_mesh = (const T*)f.phi_mesh;
    d.dc = (const T*)f.dc;
    d.inv_vol = T(1.0 / m->volume);
    d.self_c = T(self_c);
    d.bg_c = T(bg_c);
    d.out = (T*)f.out;
    d.field = (T*)f.field;
    d.energy = (T*)f.energy;
    d.force = (const T*)f.force;
    d.grad_pos = (T*)f.grad_positions;
    d.force_scale = f.full_list ? T(0.5) : T(1);
    d.tail.force = d.force;
    d.tail.force_scale = d.force_scale;
    d.tail.seed = (const T*)f.grad_seed;
    d.tail.grad_pos = d.grad_pos;
    d.tail.energy = d.energy;
    d.tail.epart_sr = v.epart;
    d.tail.n_sr = int((f.n_atoms + 64 / kRowLanes - 1) / (64 / kRowLanes));
    d.tail.epart_k = nullptr;  // per batch entry: set by frames_forward (plan scratch)
    d.tail.n_k = 0;
    d.tail.grad_q = nullptr;   // (the frames path forms energy + forces only)
    d.tail.rpart = nullptr;
    d.tail.rec4 = nullptr;
    d.tail.aux_seed = nullptr;
    d.tail.live_flags = nullptr;
    d.use_tail = f.use_tail != 0;
    d.rows.epart = f.use_tail ? v.epart : nullptr;
    out[k] = d;
  }
  return MIPME_OK;
}

int convolve_xfused(mipme_fft_plan*, hipStream_t, const void*, const void*, void*, void*, void*, int64_t, const mipme_mesh_t*,
                    const mipme_potential_t*, void*, void*, const void*, int64_t, const RowRideHost*, void*, const ConvCell*);
int64_t xconv_blocks(const mipme_fft_plan*);
void* fft_plan_tail_scratch(mipme_fft_plan*, int64_t bytes);
bool fft_plan_xfused(const mipme_fft_plan*);
int fft_plan_batch(const mipme_fft_plan*);
bool fft_plan_plane_forward_ok_batched(const mipme_fft_plan*);
void fft_plan_set_forward_done(mipme_fft_plan*, bool, int);
void* fft_plan_hat_parts(mipme_fft_plan*, hipStream_t, int);
static int plane_parts_setting() {  // (as api.hip plane_spread_parts_setting)
  static const int parts_env = [] { const char* e = getenv("MIPME_PLANE_PARTS"); return e ? atoi(e) : 2; }();
  return parts_env < 1 ? 1 : (parts_env > 8 ? 8 : parts_env);
}

template <typename T>
static int frames_forward_t(mipme_fft_plan* plan, hipStream_t st, int n_frames, const mipme_frame_t* fr, const void* table,
                            const mipme_potential_t* /*unused*/, const void* G, int64_t G_stride, void* rho_all, void* hat_all,
                            void* phi_all, void* dc_all, int pfast) {
  const FrameDev<T>* tb = (const FrameDev<T>*)table;
  const mipme_mesh_t* m = &fr[0].mesh;
  const BrickGeom bg = make_brick_geom(m);
  int64_t max_atoms = 0;
  for (int k = 0; k < n_frames; ++k) max_atoms = std::max<int64_t>(max_atoms, fr[k].n_atoms);
  const unsigned atom_blocks = unsigned((max_atoms + 255) / 256);
  const unsigned F = unsigned(n_frames);
  MIPME_DISPATCH_STENCIL_B(m->scheme, m->order, (frames_bin_atoms_kernel<S, N, T><<<dim3(atom_blocks, F), 256, 0, st>>>(tb)));
  MIPME_LAUNCH_CHECK();
  const int stage_rows = spread_stage_rows(m->order, sizeof(T));
  const size_t lds = spread_lds_bytes(m->order, sizeof(T), stage_rows);
  const int64_t rpb = SPREAD_THREADS / kRowLanes;
  const unsigned grid_x = unsigned(bg.nb) + unsigned((max_atoms + rpb - 1) / rpb);
  const bool compact = fr[0].shift_format == kShiftTable32;  // frames_check: the same format for every frame
  // plane spread (every frame of the batch has its plane lists: frame_plane_lists, decided when the table was built)
  const int dtype_f = sizeof(T) == 4 ? MIPME_F32 : MIPME_F64;
  bool planes = fft_plan_plane_forward_ok_batched(plan);
  for (int k = 0; k < n_frames && planes; ++k) planes = frame_plane_lists(fr[k], dtype_f);
  fft_plan_set_forward_done(plan, false, 1);
  if (!planes) note_cosched_kernel("frames_spread_rows_kernel");
  if (planes) {
    PlaneArgs<T> pa;
    size_t need = 0;
    plane_lds_layout<T>(m->ny, m->nz, pa, need);
    const int64_t Mh = int64_t(m->nx) * m->ny * (m->nz / 2 + 1);
    pa.hat = (Cplx<T>*)hat_all;
    // a batch has its frames for parallelism: as many parts as keep the plane workgroups of the launch at or below 128 (the
    // single-frame optimum at 64^3: 2 x 64); measured on 8 x 8000 ions / 32^3 fp64: 1 part 0.1243, 2 parts 0.1291, 3 parts
    // 0.1322 ms (bricks 0.1336); 16 x 1000 atoms / 32^3 fp32: 0.0569 / 0.0633 / 0.0691 (bricks 0.0600)
    pa.parts = plane_parts_setting();
    while (pa.parts > 1 && int64_t(pa.parts) * m->nx * n_frames > 128) --pa.parts;
    if (pa.parts > 1) {
      pa.hat_more = (Cplx<T>*)fft_plan_hat_parts(plan, st, 7);
      pa.more_stride = Mh * n_frames;
      if (!pa.hat_more) pa.parts = 1;
    }
    while ((1 << pa.logny) < m->ny) ++pa.logny;
    while ((1 << pa.loglz) < m->nz / 2) ++pa.loglz;
    const size_t rows_lds = sizeof(T) * size_t(SPREAD_WAVES) * BRICK_PTS;
    const size_t plds = need > rows_lds ? need : rows_lds;
    const unsigned pgrid_x = unsigned(m->nx) * unsigned(pa.parts) + unsigned((max_atoms + rpb - 1) / rpb);
#define MIPME_FRAMES_PLANES(PF, CO) \
  MIPME_DISPATCH_STENCIL_B(m->scheme, m->order, (frames_plane_rows_kernel<S, N, T, PF, CO><<<dim3(pgrid_x, F), SPREAD_THREADS, plds, st>>>(tb, pa, Mh)))
    note_cosched_kernel("frames_plane_rows_kernel");
    if (pfast == 1 && compact)
      MIPME_FRAMES_PLANES(1, true);
    else if (pfast == 1)
      MIPME_FRAMES_PLANES(1, false);
    else if (compact)
      MIPME_FRAMES_PLANES(6, true);
    else
      MIPME_FRAMES_PLANES(6, false);
#undef MIPME_FRAMES_PLANES
    fft_plan_set_forward_done(plan, true, pa.parts);
  } else if (pfast == 1 && compact)
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, frames_spread_rows_kernel<N, T, 1, true><<<dim3(grid_x, F), SPREAD_THREADS, lds, st>>>(tb)));
  else if (pfast == 1)
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, frames_spread_rows_kernel<N, T, 1, false><<<dim3(grid_x, F), SPREAD_THREADS, lds, st>>>(tb)));
  else if (compact)
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, frames_spread_rows_kernel<N, T, 6, true><<<dim3(grid_x, F), SPREAD_THREADS, lds, st>>>(tb)));
  else
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, frames_spread_rows_kernel<N, T, 6, false><<<dim3(grid_x, F), SPREAD_THREADS, lds, st>>>(tb)));
  MIPME_LAUNCH_CHECK();
  bool all_tail = true;
  for (int k = 0; k < n_frames; ++k) all_tail = all_tail && fr[k].use_tail != 0;
  const int64_t n_k = xconv_blocks(plan) / n_frames;  // blocks of the x stage per batch entry
  double* epart_k = all_tail ? (double*)fft_plan_tail_scratch(plan, int64_t(sizeof(double)) * n_k * n_frames) : nullptr;
  MIPME_REQUIRE(!all_tail || epart_k, "could not allocate the energy partial sums of the plan (not possible during stream "
                                      "capture: run one evaluation before capturing)");
  int rc = convolve_xfused(plan, st, rho_all, G, hat_all, phi_all, dc_all, G_stride, nullptr, nullptr, nullptr, epart_k, nullptr, 0, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (all_tail) {  // energy + forces of every frame in the gather launch
    MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                             ((void)S, frames_gather_tail_kernel<N, T><<<dim3(unsigned(bg.nb), F), GATHER_THREADS, 0, st>>>(
                                 tb, epart_k, int(n_k))));
    MIPME_LAUNCH_CHECK();
    return MIPME_OK;
  }
  MIPME_DISPATCH_STENCIL_B(m->scheme, m->order,
                           ((void)S, frames_gather_kernel<N, T><<<dim3(unsigned(bg.nb), F), GATHER_THREADS, 0, st>>>(tb)));
  MIPME_LAUNCH_CHECK();
  frames_energy_kernel<T><<<F, 1024, 0, st>>>(tb);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

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

template int bins_build<float>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, int*, const void*, void*, bool);
template int bins_build<double>(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, int*, const void*, void*, bool);
template int spread_bricks<float>(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, double, void*, int*,
                                  const mipme_sr_job_t*, bool, double*, const PlaneHost*, bool*);
template int spread_bricks<double>(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, double, void*, int*,
                                   const mipme_sr_job_t*, bool, double*, const PlaneHost*, bool*);
template int gather_bricks<float>(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, const void*, const void*,
                                  double, double, void*, void*, int, void*, const GatherTailHost*, void*, int*);
template int gather_bricks<double>(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, const void*, const void*,
                                   double, double, void*, void*, int, void*, const GatherTailHost*, void*, int*);
template int gather_grad_bricks<float>(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, const void*,
                                       const void*, const void*, const void*, const void*, double, double, void*, void*);
template int gather_grad_bricks<double>(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, const void*,
                                        const void*, const void*, const void*, const void*, double, double, void*, void*);

}  // namespace mipme

using namespace mipme;

template <typename T>
static int frames_table_energy_log_t(int n_frames, void* host_table, void* log, void* cursors, int capacity) {
  FrameDev<T>* d = (FrameDev<T>*)host_table;
  for (int f = 0; f < n_frames; ++f) {
    MIPME_REQUIRE(!log || d[f].use_tail, "the energy log rides on the gather tail (mipme_frame_t.use_tail) of every frame");
    d[f].tail.elog = log ? (double*)log + f : nullptr;
    d[f].tail.elog_cursor = log ? (int*)cursors + f : nullptr;
    d[f].tail.elog_cap = capacity;
    d[f].tail.elog_stride = n_frames;
  }
  return MIPME_OK;
}

extern "C" {

int64_t mipme_frames_table_bytes(int dtype, int n_frames) {
  if (n_frames <= 0) return 0;
  return int64_t(n_frames) * int64_t(dtype == MIPME_F32 ? sizeof(FrameDev<float>) : sizeof(FrameDev<double>));
}

int mipme_frames_table_build(int dtype, int n_frames, const mipme_frame_t* frames, const mipme_potential_t* pot,
                             void* host_table, int64_t host_table_bytes) {
  int rc = frames_check(dtype, n_frames, frames);
  if (rc) return rc;
  MIPME_REQUIRE(pot && host_table && host_table_bytes >= mipme_frames_table_bytes(dtype, n_frames),
                "invalid arguments to mipme_frames_table_build");
  if (dtype == MIPME_F32) return frames_table_build_t<float>(n_frames, frames, pot, host_table);
  return frames_table_build_t<double>(n_frames, frames, pot, host_table);
}

int mipme_frames_table_energy_log(int dtype, int n_frames, void* host_table, int64_t host_table_bytes, void* log, void* cursors,
                                  int capacity) {
  MIPME_REQUIRE((dtype == MIPME_F32 || dtype == MIPME_F64) && n_frames > 0 && host_table &&
                    host_table_bytes >= mipme_frames_table_bytes(dtype, n_frames) && (!log || (cursors && capacity > 0)),
                "invalid arguments to mipme_frames_table_energy_log");
  if (dtype == MIPME_F32) return frames_table_energy_log_t<float>(n_frames, host_table, log, cursors, capacity);
  return frames_table_energy_log_t<double>(n_frames, host_table, log, cursors, capacity);
}

int mipme_frames_forward(mipme_fft_plan* plan, void* stream, int dtype, int n_frames, const mipme_frame_t* frames,
                         const mipme_potential_t* pot, const void* device_table, const void* G, int64_t G_stride,
                         void* rho_mesh_all, void* hat_work_all, void* phi_mesh_all, void* dc_all) {
  int rc = frames_check(dtype, n_frames, frames);
  if (rc) return rc;
  MIPME_REQUIRE(plan && pot && device_table && G && rho_mesh_all && hat_work_all && phi_mesh_all && dc_all && G_stride >= 0,
                "NULL buffer passed to mipme_frames_forward");
  MIPME_REQUIRE(fft_plan_xfused(plan) && fft_plan_batch(plan) == n_frames,
                "mipme_frames_forward needs a plan with batch = n_frames and a power-of-two nx");
  SRPot s;
  if ((rc = make_srpot(pot, s))) return rc;
  const int pfast = fast_rs_exponent(s);
  MIPME_REQUIRE(pfast == 1 || pfast == 6, "the frames path covers 1/r and 1/r^6 with a smearing");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    return frames_forward_t<float>(plan, st, n_frames, frames, device_table, pot, G, G_stride, rho_mesh_all, hat_work_all,
                                   phi_mesh_all, dc_all, pfast);
  return frames_forward_t<double>(plan, st, n_frames, frames, device_table, pot, G, G_stride, rho_mesh_all, hat_work_all,
                                  phi_mesh_all, dc_all, pfast);
}

int64_t mipme_frames_counter_ints(const mipme_mesh_t* mesh, int64_t n_atoms, int dtype) {
  if (!mesh || validate_mesh(mesh) || !bricks_supported(mesh, dtype)) return 0;
  return frames_counter_ints(mesh, n_atoms, dtype);
}

int mipme_frames_backward(void* stream, int dtype, int n_frames, const mipme_frame_t* frames, const void* device_table,
                          const void* grad_scale) {
  int rc = frames_check(dtype, n_frames, frames);
  if (rc) return rc;
  MIPME_REQUIRE(device_table && grad_scale, "NULL buffer passed to mipme_frames_backward");
  int64_t max_atoms = 0;
  for (int k = 0; k < n_frames; ++k) max_atoms = std::max<int64_t>(max_atoms, frames[k].n_atoms);
  const dim3 grid(unsigned((3 * max_atoms + 255) / 256), unsigned(n_frames));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    frames_finalize_kernel<float><<<grid, 256, 0, st>>>((const FrameDev<float>*)device_table, (const float*)grad_scale);
  else
    frames_finalize_kernel<double><<<grid, 256, 0, st>>>((const FrameDev<double>*)device_table, (const double*)grad_scale);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // extern "C"

#ifdef MIPME_WG_TIMELINE
extern "C" int mipme_debug_rows_phase(void* out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mipme::g_rows_phase), size_t(n_words) * 8);
}
extern "C" int mipme_debug_wg_phase(void* out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mipme::g_wg_phase), size_t(n_words) * 8);
}
extern "C" int mipme_debug_wg_timeline(void* out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mipme::g_wg_timeline), size_t(n_words) * 8);
}
#endif
