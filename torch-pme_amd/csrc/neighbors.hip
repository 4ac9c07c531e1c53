// GPU neighbour list (cell list) -- SURVEY.md 8(f) rank 1, the caller-side step that precedes the hot path.
//
// The reference takes (pairs, shifts / distances) from third-party vesin on the host (tests/helpers.py:240-275,
// quantities "PdS") and hands a fresh (P,2) list to every call (examples/02-neighbor-lists-usage.py:97-164).  This
// builder produces, on the device and from the same cell-list traversal,
//   * the reference's quantities: pairs (P,2) int64, integer cell shifts S (P,3) with r_ij = r_j - r_i + S @ cell, |r_ij|,
//     half or full list, strict d < cutoff  (mipme_nl_count -> caller scans -> mipme_nl_fill), and
//   * the ROW STREAM the fused pair kernels read (rows_body.h), directly: for every atom a the 4-byte words
//     other | shift code << 22 of ALL its neighbours, in a row of fixed capacity (mipme_nl_stream) -- no (P,2) int64 round
//     trip, no radix sort, no repacking, every buffer at a fixed address and nothing read back by the host, so that a
//     captured HIP graph refreshes the list of a captured energy + forces step in place.
//
// Any cell (orthorhombic / triclinic), any box size and cutoff: cells are at least cutoff / 2 wide (perpendicular width)
// where the box allows it, an atom's neighbours lie within `reach` cells per axis (reach = ceil(cutoff / cell width): 2,
// or more when the box is smaller than the cutoff), and the walk over cell offsets wraps with the IMAGE shift it crosses,
// so that a small box simply meets the same cell several times with different shifts (the round-2 builder needed three
// cutoff-wide cells per periodic axis and sent everything else to the host).
//
// Phases (all on the caller's stream, no host synchronisation, no memset nodes):
//   bin    count atoms per cell -> exclusive scan -> place (returning atomics) -> order every cell by atom index and write
//          the cell records {wrapped position (fp64), atom, wrap integers}: deterministic whatever the atomics' order
//   walk   one workgroup per cell: the candidates of the (2 reach + 1)^3 surrounding cells (contiguous runs along z) are
//          staged in LDS with their image shift already applied, once for all atoms of the cell; a wavefront per atom then
//          tests 64 candidates per step (all decisions in fp64) and appends the survivors with a ballot compaction.
#include "common.h"
#include "rows_body.h"

namespace mipme {

struct NlGeom {
  double cell[9], inv[9];
  int nc[3];
  int periodic[3];
  int reach[3];
  double cutoff2;
  int full_list;
  int ps;                     // reals per atom in the positions array: 3, or 4 for (x, y, z, charge) records
  double foff[3], fscale[3];  // binning map of the non-periodic axes
};

static inline NlGeom make_nl(const mipme_nl_t* d) {
  NlGeom g;
  for (int k = 0; k < 9; ++k) {
    g.cell[k] = d->cell[k];
    g.inv[k] = d->inv_cell[k];
  }
  for (int k = 0; k < 3; ++k) {
    g.nc[k] = d->n_cells[k];
    g.periodic[k] = d->periodic[k];
    g.reach[k] = d->reach[k];
    g.foff[k] = d->periodic[k] ? 0.0 : d->frac_offset[k];
    g.fscale[k] = d->periodic[k] ? 1.0 : d->frac_scale[k];
  }
  g.cutoff2 = d->cutoff * d->cutoff;
  g.full_list = d->full_list;
  g.ps = d->position_stride > 0 ? d->position_stride : 3;
  return g;
}

// ---- workspace ---------------------------------------------------------------------------------------------------------
// count (zero between calls: the scan clears it after reading, the ordering pass after the placement used it as cursor),
// cell starts, per-atom cell / wrap words, the placement scratch, the cell records, the status words.
struct alignas(32) CellRec {
  double x, y, z;  // wrapped position
  int j, w;        // atom, its wrap word
};

struct NlWorkspace {
  int* count;      // [ncells + 1]
  int* start;      // [ncells + 1]
  int* cell_of;    // [N]
  int* wpack;      // [N]   wrap integers of the atom, 3 x 10 bits biased by 512
  int* tmp;        // [N]   atoms in placement order
  CellRec* rec;    // [N]   cell records, ordered by (cell, atom)
  int* status;     // [8]   0: longest row of the last stream  1: flags  2: refresh counter  (device copy of the report)
};

static constexpr int64_t kNlAlign = 256;
static inline int64_t nl_align(int64_t b) { return (b + kNlAlign - 1) / kNlAlign * kNlAlign; }

static inline int64_t nl_layout(int64_t ncells, int64_t N, char* base, NlWorkspace* w) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + off : nullptr;
    off += nl_align(bytes);
    return p;
  };
  const int64_t n = N > 0 ? N : 1;
  char* count = take(4 * (ncells + 1));
  char* start = take(4 * (ncells + 1));
  char* cell_of = take(4 * n);
  char* wpack = take(4 * n);
  char* tmp = take(4 * n);
  char* rec = take(int64_t(sizeof(CellRec)) * n);
  char* status = take(4 * 8);
  if (w) {
    w->count = (int*)count;
    w->start = (int*)start;
    w->cell_of = (int*)cell_of;
    w->wpack = (int*)wpack;
    w->tmp = (int*)tmp;
    w->rec = (CellRec*)rec;
    w->status = (int*)status;
  }
  return off;
}

// three signed integers in [-512, 511] as 10-bit fields biased by 512
static constexpr int kWrapLimit = 400;   // |wrap integer| of an atom
static constexpr int kImageLimit = 100;  // |image shift| of the cell walk
__host__ __device__ __forceinline__ int pack3(int a, int b, int c) { return (a + 512) | ((b + 512) << 10) | ((c + 512) << 20); }
__device__ __forceinline__ int unpack3(int w, int k) { return ((w >> (10 * k)) & 1023) - 512; }

enum NlFlags {
  kNlRowOverflow = 1,   // a row of the stream is longer than its capacity (entries beyond it were dropped)
  kNlShiftRange = 2,    // a cell shift of the stream is beyond the 7^3 table of the pair kernels (|S| > 3)
  kNlWrapRange = 4,     // an atom lies more than kWrapLimit cells outside the unit cell
};

// wrapped position, wrap integers and cell index of one atom (all decisions in fp64)
template <typename T>
__device__ __forceinline__ void atom_cell(const NlGeom& g, const T* __restrict__ pos, int64_t i, double (&rw)[3],
                                          int (&w)[3], int (&c)[3]) {
  const double r[3] = {double(pos[g.ps * i]), double(pos[g.ps * i + 1]), double(pos[g.ps * i + 2])};
  double f[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f[d] = r[0] * g.inv[d] + r[1] * g.inv[3 + d] + r[2] * g.inv[6 + d];
    const double fl = g.periodic[d] ? floor(f[d]) : 0.0;
    w[d] = int(fl);
    f[d] -= fl;
    int cd = int((f[d] - g.foff[d]) * g.fscale[d] * g.nc[d]);
    cd = cd < 0 ? 0 : (cd >= g.nc[d] ? g.nc[d] - 1 : cd);  // (an atom beyond the binned extent of a non-periodic axis joins
    c[d] = cd;                                              // the edge cell: clamping never widens a cell distance)
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) rw[k] = r[k] - (w[0] * g.cell[k] + w[1] * g.cell[3 + k] + w[2] * g.cell[6 + k]);
}

template <typename T>
__global__ __launch_bounds__(256) void nl_cell_count_kernel(NlGeom g, int64_t N, const T* __restrict__ pos, NlWorkspace ws) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double rw[3];
  int w[3], c[3];
  atom_cell<T>(g, pos, i, rw, w, c);
  const int ci = (c[0] * g.nc[1] + c[1]) * g.nc[2] + c[2];
  ws.cell_of[i] = ci;
  bool far = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    far |= w[d] > kWrapLimit || w[d] < -kWrapLimit;
    w[d] = w[d] > kWrapLimit ? kWrapLimit : (w[d] < -kWrapLimit ? -kWrapLimit : w[d]);
  }
  ws.wpack[i] = pack3(w[0], w[1], w[2]);
  if (far) atomicOr(&ws.status[1], kNlWrapRange);
  atomicAdd(&ws.count[ci], 1);
}

// exclusive scan of the cell counts (one workgroup); leaves the counts at zero for the placement pass, and resets the
// status words of the walk that follows
__global__ __launch_bounds__(1024) void nl_scan_kernel(int n, NlWorkspace ws) {
  __shared__ int part[1024];
  int* __restrict__ count = ws.count;
  int* __restrict__ start = ws.start;
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += count[k];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int k = lo; k < hi; ++k) {
    start[k] = run;
    run += count[k];
    count[k] = 0;
  }
  if (t == 1023) start[n] = part[1023];
  if (t == 0) ws.status[0] = 0;
}

__global__ __launch_bounds__(256) void nl_cell_place_kernel(int64_t N, NlWorkspace ws) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ci = ws.cell_of[i];
  const int slot = atomicAdd(&ws.count[ci], 1);
  ws.tmp[ws.start[ci] + slot] = int(i);
}

// deterministic order: the slot of an atom inside its cell = number of atoms of the cell with a smaller index (rank by
// counting; a cell holds a few atoms).  Writes the cell records and clears the cell's cursor.
template <typename T>
__global__ __launch_bounds__(256) void nl_cell_order_kernel(NlGeom g, int64_t N, const T* __restrict__ pos, NlWorkspace ws) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const int me = ws.tmp[k];
  const int ci = ws.cell_of[me];
  const int beg = ws.start[ci], n = ws.start[ci + 1] - beg;
  int rank = 0;
  for (int m = 0; m < n; ++m) rank += ws.tmp[beg + m] < me;
  double rw[3];
  int w[3], c[3];
  atom_cell<T>(g, pos, me, rw, w, c);
  ws.rec[beg + rank] = CellRec{rw[0], rw[1], rw[2], me, ws.wpack[me]};
  if (rank == 0) ws.count[ci] = 0;
}

// ---- the walk ------------------------------------------------------------------------------------------------------------
enum NlMode { kNlCount = 0, kNlFill = 1, kNlStream = 2 };

static constexpr int kNlCap = 1216;   // candidates staged per round (19 x 64)
static constexpr int kNlBatch = 32;   // atoms of the cell served per staging
static constexpr int kNlList = 512;   // survivors a wavefront gathers before it emits them
static constexpr int kNlSegs = 256;   // runs of cells described per staging group
static constexpr int kNlThreads = 256;
static_assert(kNlSegs == kNlThreads && (kNlSegs & (kNlSegs - 1)) == 0 && kNlCap % 64 == 0, "run table: one run per thread");
// LDS: 32 B per candidate + the survivor lists + run table + the batch's own records = 49 KB: three workgroups per CU

struct NlOut {
  // kNlCount
  int* counts;             // [N] accepted neighbours per atom (half / full rule)
  // kNlFill
  const int64_t* offsets;  // [N + 1]
  int64_t* pairs;
  void* shifts;
  void* dist;              // nullable
  // kNlStream
  int* row_ptr;            // [3 N + 1]: begin, end, end of every row (rows_body.h, kRowsPadded), [3 N] = N * stride
  int* words;              // [N * stride (+ 1)]
  int stride;
};

__device__ __forceinline__ int floordiv(int a, int n) {
  int q = a / n;
  return (a % n < 0) ? q - 1 : q;
}

template <typename T, int MODE>
__global__ __launch_bounds__(kNlThreads) void nl_walk_kernel(NlGeom g, int64_t N, unsigned n_cells, NlWorkspace ws, NlOut o) {
  __shared__ double sX[kNlCap + 64], sY[kNlCap + 64], sZ[kNlCap + 64];
  __shared__ int sJ[kNlCap + 64], sS[kNlCap + 64];
  __shared__ unsigned short sList[kNlThreads / 64][kNlList];
  __shared__ int gBeg[kNlSegs], gOff[kNlSegs + 1], gShift[kNlSegs], gWave[kNlThreads / 64];
  __shared__ double oX[kNlBatch], oY[kNlBatch], oZ[kNlBatch];
  __shared__ int oA[kNlBatch], oW[kNlBatch], oCur[kNlBatch], oBad[kNlBatch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (MODE == kNlStream) {
    if (blockIdx.x == 0 && tid == 0) o.row_ptr[3 * N] = int(N * o.stride);
  }
  const unsigned c = xcd_contiguous(blockIdx.x, n_cells);
  if (c >= n_cells) return;
  const int ncx = g.nc[0], ncy = g.nc[1], ncz = g.nc[2];
  const int cz0 = int(c % unsigned(ncz)), cy0 = int((c / unsigned(ncz)) % unsigned(ncy)), cx0 = int(c / unsigned(ncz * ncy));
  // runs of cells along z that are contiguous in memory and share one image shift: the same for every (x, y) column
  const int zlo = cz0 - g.reach[2], zhi = cz0 + g.reach[2];
  int nzrun = 0;
  for (int z = zlo; z <= zhi;) {
    const int zz = z - floordiv(z, ncz) * ncz;
    z += min(ncz - zz, zhi - z + 1);
    ++nzrun;
  }
  const int nxs = 2 * g.reach[0] + 1, nys = 2 * g.reach[1] + 1;
  const int nseg = nxs * nys * nzrun;
  // the run table of a group of kNlSegs runs: first record, record count, image shift -- one run per thread, so that all
  // the cell-start loads of the group are in flight together (and, for the first group, together with the cell's own bounds)
  auto describe_runs = [&](int seg0) {
    const int ns = min(kNlSegs, nseg - seg0);
    int cnt = 0;
    if (tid < ns) {
      const int sidx = seg0 + tid;
      const int ir = sidx % nzrun, iy = (sidx / nzrun) % nys, ix = sidx / (nzrun * nys);
      const int X = cx0 + ix - g.reach[0], sx = floordiv(X, ncx), xx = X - sx * ncx;
      const int Y = cy0 + iy - g.reach[1], sy = floordiv(Y, ncy), yy = Y - sy * ncy;
      int z = zlo, sz = 0, zz = 0, len = 0;
      for (int r = 0; r <= ir; ++r) {
        sz = floordiv(z, ncz);
        zz = z - sz * ncz;
        len = min(ncz - zz, zhi - z + 1);
        z += len;
      }
      const bool valid = (g.periodic[0] || sx == 0) && (g.periodic[1] || sy == 0) && (g.periodic[2] || sz == 0);
      if (valid) {
        const int cell0 = (xx * ncy + yy) * ncz + zz;
        const int beg = ws.start[cell0];
        cnt = ws.start[cell0 + len] - beg;
        gBeg[tid] = beg;
        gShift[tid] = pack3(sx, sy, sz);
      }
    }
    gOff[tid + 1] = cnt;
    if (tid == 0) gOff[0] = 0;
  };
  describe_runs(0);
  const int cbeg = ws.start[c], na = ws.start[c + 1] - cbeg;
  if (na == 0) return;

  // survivors of `n` staged candidates for the atoms of the batch this wavefront owns: a pass over all candidates that only
  // measures distances and compacts the indices of the close ones (two blocks of 64 per step, all LDS loads of a step in
  // flight together), and a dense pass over the survivors whenever kNlList of them have gathered
  auto process = [&](int n, int nb) {
    unsigned short* __restrict__ list = sList[wave];
    for (int k = wave; k < nb; k += kNlThreads / 64) {
      const double rx = oX[k], ry = oY[k], rz = oZ[k];
      const int a = oA[k], wi = oW[k];
      const int wix = unpack3(wi, 0), wiy = unpack3(wi, 1), wiz = unpack3(wi, 2);
      int cur = oCur[k];
      bool bad = false;
      int cnt = 0;
      auto emit = [&]() {
        __threadfence_block();  // the list is read back by other lanes of this wavefront
        for (int k0 = 0; k0 < cnt; k0 += 64) {
          const bool in = k0 + lane < cnt;
          const int cc = in ? int(list[k0 + lane]) : 0;
          const int j = sJ[cc], sp = sS[cc];
          const int Sx = unpack3(sp, 0) + wix, Sy = unpack3(sp, 1) + wiy, Sz = unpack3(sp, 2) + wiz;  // S = s - w_j + w_i
          bool ok = in;
          if constexpr (MODE != kNlStream) {
            if (!g.full_list) {
              const bool lexpos = Sx > 0 || (Sx == 0 && (Sy > 0 || (Sy == 0 && Sz > 0)));
              ok = ok && ((a < j) || (a == j && lexpos));
            }
          }
          int at = cur + lane, step = min(64, cnt - k0);
          if constexpr (MODE != kNlStream) {
            const unsigned long long m = __ballot(ok);
            at = cur + __popcll(m & ((1ull << lane) - 1ull));
            step = __popcll(m);
          }
          if constexpr (MODE == kNlFill) {
            if (ok) {
              const double vx = sX[cc] - rx, vy = sY[cc] - ry, vz = sZ[cc] - rz;
              const int64_t dst = o.offsets[a] + at;
              o.pairs[2 * dst] = a;
              o.pairs[2 * dst + 1] = j;
              T* __restrict__ sh = (T*)o.shifts;
              sh[3 * dst] = T(Sx);
              sh[3 * dst + 1] = T(Sy);
              sh[3 * dst + 2] = T(Sz);
              if (o.dist) ((T*)o.dist)[dst] = T(sqrt(vx * vx + vy * vy + vz * vz));
            }
          }
          if constexpr (MODE == kNlStream) {
            const unsigned ux = unsigned(Sx + kShiftTableRange), uy = unsigned(Sy + kShiftTableRange),
                           uz = unsigned(Sz + kShiftTableRange);
            const bool fits = max(max(ux, uy), uz) < unsigned(kShiftTableBase);
            bad |= ok && !fits;
            const int code = fits ? int(ux + kShiftTableBase * (uy + kShiftTableBase * uz)) : 0;
            if (ok && at < o.stride) o.words[a * o.stride + at] = j | (code << kCompactAtomBits);
          }
          cur += step;
        }
        cnt = 0;
      };
      // the atom itself (its image with zero shift) is the one candidate at distance exactly 0 that carries its own index
      auto close = [&](int cc, double x, double y, double z, int j) {
        const double vx = x - rx, vy = y - ry, vz = z - rz;
        const double d2 = vx * vx + vy * vy + vz * vz;
        return (cc < n) & (d2 < g.cutoff2) & !((d2 == 0.0) & (j == a));
      };
      for (int base = 0; base < n; base += 128) {
        const int c0 = base + lane, c1 = c0 + 64;  // (the arrays are padded by 64: always inside)
        const double x0 = sX[c0], y0 = sY[c0], z0 = sZ[c0], x1 = sX[c1], y1 = sY[c1], z1 = sZ[c1];
        const int j0 = sJ[c0], j1 = sJ[c1];
        const bool ok0 = close(c0, x0, y0, z0, j0), ok1 = close(c1, x1, y1, z1, j1);
        const unsigned long long m0 = __ballot(ok0), m1 = __ballot(ok1);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (ok0) list[cnt + __popcll(m0 & below)] = (unsigned short)c0;
        cnt += __popcll(m0);
        if (ok1) list[cnt + __popcll(m1 & below)] = (unsigned short)c1;
        cnt += __popcll(m1);
        if (cnt > kNlList - 128) emit();
      }
      emit();
      if (lane == 0) oCur[k] = cur;
      if constexpr (MODE == kNlStream) {
        if (__ballot(bad) != 0ull && lane == 0) oBad[k] = 1;
      }
    }
  };

  for (int b0 = 0; b0 < na; b0 += kNlBatch) {
    const int nb = min(kNlBatch, na - b0);
    __syncthreads();  // the previous batch's epilogue has read the o* arrays
    if (tid < nb) {
      const CellRec r = ws.rec[cbeg + b0 + tid];
      oX[tid] = r.x;
      oY[tid] = r.y;
      oZ[tid] = r.z;
      oA[tid] = r.j;
      oW[tid] = r.w;
      oCur[tid] = 0;
      oBad[tid] = 0;
    }
    for (int seg0 = 0; seg0 < nseg; seg0 += kNlSegs) {
      // (a) the run table (the first group's is already there, and stays valid across batches when it is the only one)
      if (seg0 > 0 || (b0 > 0 && nseg > kNlSegs)) describe_runs(seg0);
      const bool rescan = b0 == 0 || nseg > kNlSegs;
      __syncthreads();
      // (b) inclusive scan of the counts: one value per thread, shuffle scan inside each wavefront + the totals of the
      // wavefronts before it (two barriers instead of sixteen)
      if (rescan) {
        int v = gOff[tid + 1];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int u = __shfl_up(v, off, 64);
          if (lane >= off) v += u;
        }
        if (lane == 63) gWave[wave] = v;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < kNlThreads / 64; ++w) before += (w < wave) ? gWave[w] : 0;
        gOff[tid + 1] = v + before;
        __syncthreads();
      }
      const int total = gOff[kNlSegs];
      // (c) candidates in flattened order, kNlCap at a time: every thread first finds the runs of all its candidates, then
      // issues all its record loads, then writes (a loop of search -> load -> store per candidate spent a memory round trip
      // per candidate and thread)
      for (int f0 = 0; f0 < total; f0 += kNlCap) {
        const int n = min(kNlCap, total - f0);
        constexpr int K = (kNlCap + kNlThreads - 1) / kNlThreads;
        int run[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int f = f0 + tid + u * kNlThreads;
          int lo = 0;  // largest run with gOff[run] <= f
#pragma unroll
          for (int stp = kNlSegs / 2; stp > 0; stp >>= 1) lo += (gOff[lo + stp] <= f) ? stp : 0;
          run[u] = lo;
        }
        CellRec rec[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int t = tid + u * kNlThreads;
          const int src = t < n ? gBeg[run[u]] + (f0 + t - gOff[run[u]]) : cbeg;
          rec[u] = ws.rec[src];
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
          const int t = tid + u * kNlThreads;
          if (t < n) {
            const CellRec r = rec[u];
            const int sp = gShift[run[u]];
            const int sx = unpack3(sp, 0), sy = unpack3(sp, 1), sz = unpack3(sp, 2);
            sX[t] = r.x + (sx * g.cell[0] + sy * g.cell[3] + sz * g.cell[6]);
            sY[t] = r.y + (sx * g.cell[1] + sy * g.cell[4] + sz * g.cell[7]);
            sZ[t] = r.z + (sx * g.cell[2] + sy * g.cell[5] + sz * g.cell[8]);
            sJ[t] = r.j;
            sS[t] = pack3(sx - unpack3(r.w, 0), sy - unpack3(r.w, 1), sz - unpack3(r.w, 2));
          }
        }
        __syncthreads();
        process(n, nb);
        __syncthreads();
      }
    }
    // epilogue of the batch: one lane per atom
    if (tid < 64) {
      int longest = 0, flags = 0;
      if (tid < nb) {
        const int a = oA[tid], cur = oCur[tid];
        if constexpr (MODE == kNlCount) o.counts[a] = cur;
        if constexpr (MODE == kNlStream) {
          const int beg = a * o.stride, end = beg + min(cur, o.stride);
          o.row_ptr[3 * a] = beg;
          o.row_ptr[3 * a + 1] = end;
          o.row_ptr[3 * a + 2] = end;
          longest = cur;
          flags = (cur > o.stride ? kNlRowOverflow : 0) | (oBad[tid] ? kNlShiftRange : 0);
        }
      }
      if constexpr (MODE == kNlStream) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          longest = max(longest, __shfl_xor(longest, off, 64));
          flags |= __shfl_xor(flags, off, 64);
        }
        if (tid == 0) {
          atomicMax(&ws.status[0], longest);
          if (flags) atomicOr(&ws.status[1], flags);
        }
      }
    }
  }
}

// status words of the last refresh to pinned host memory (system-scope stores; the host polls / reads them lazily)
__global__ void nl_report_kernel(NlWorkspace ws, int* __restrict__ host) {
  const int seq = ws.status[2] + 1;
  ws.status[2] = seq;
  const int longest = ws.status[0], flags = ws.status[1];
  ws.status[1] = 0;
  if (host) {
    __hip_atomic_store(&host[0], longest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[1], flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static int validate_nl(const mipme_nl_t* d) {
  MIPME_REQUIRE(d != nullptr, "neighbour-list descriptor is NULL");
  MIPME_REQUIRE(d->cutoff > 0, "cutoff must be positive");
  int64_t ncells = 1;
  for (int k = 0; k < 3; ++k) {
    MIPME_REQUIRE(d->n_cells[k] >= 1 && d->n_cells[k] <= 4096, "invalid cell grid");
    MIPME_REQUIRE(d->reach[k] >= 0 && d->reach[k] <= kImageLimit * d->n_cells[k],
                  "the cell walk along axis %d spans more than %d images of the box (cutoff too large for this cell)", k, kImageLimit);
    MIPME_REQUIRE(d->periodic[k] || d->frac_scale[k] > 0.0, "a non-periodic axis needs a positive frac_scale");
    MIPME_REQUIRE(d->position_stride == 0 || d->position_stride == 3 || d->position_stride == 4, "position_stride must be 3 or 4");
    ncells *= d->n_cells[k];
  }
  MIPME_REQUIRE(ncells < (int64_t(1) << 30), "cell grid too large");
  return MIPME_OK;
}

template <typename T>
static int nl_bin_impl(hipStream_t st, const mipme_nl_t* d, int64_t N, const void* pos, void* workspace) {
  const NlGeom g = make_nl(d);
  const int ncells = g.nc[0] * g.nc[1] * g.nc[2];
  NlWorkspace ws;
  nl_layout(ncells, N, (char*)workspace, &ws);
  const unsigned blocks = unsigned((N + 255) / 256);
  if (N > 0) {
    nl_cell_count_kernel<T><<<blocks, 256, 0, st>>>(g, N, (const T*)pos, ws);
    MIPME_LAUNCH_CHECK();
  }
  nl_scan_kernel<<<1, 1024, 0, st>>>(ncells, ws);
  MIPME_LAUNCH_CHECK();
  if (N > 0) {
    nl_cell_place_kernel<<<blocks, 256, 0, st>>>(N, ws);
    MIPME_LAUNCH_CHECK();
    nl_cell_order_kernel<T><<<blocks, 256, 0, st>>>(g, N, (const T*)pos, ws);
    MIPME_LAUNCH_CHECK();
  }
  return MIPME_OK;
}

template <typename T, int MODE>
static int nl_walk_impl(hipStream_t st, const mipme_nl_t* d, int64_t N, void* workspace, const NlOut& o) {
  const NlGeom g = make_nl(d);
  const int ncells = g.nc[0] * g.nc[1] * g.nc[2];
  NlWorkspace ws;
  nl_layout(ncells, N, (char*)workspace, &ws);
  nl_walk_kernel<T, MODE><<<pad8(unsigned(ncells)), kNlThreads, 0, st>>>(g, N, unsigned(ncells), ws, o);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // namespace mipme

using namespace mipme;

extern "C" {

int64_t mipme_nl_workspace_bytes(const mipme_nl_t* d, int64_t n_atoms) {
  if (!d || n_atoms < 0) return 0;
  const int64_t ncells = int64_t(d->n_cells[0]) * d->n_cells[1] * d->n_cells[2];
  return nl_layout(ncells, n_atoms, nullptr, nullptr);
}

int mipme_nl_bin(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, const void* positions, void* workspace) {
  int rc = validate_nl(d);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && workspace && (n_atoms == 0 || positions), "NULL buffer passed to mipme_nl_bin");
  MIPME_REQUIRE(n_atoms < (int64_t(1) << 31), "too many atoms for the device neighbour list");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return nl_bin_impl<float>(st, d, n_atoms, positions, workspace);
  if (dtype == MIPME_F64) return nl_bin_impl<double>(st, d, n_atoms, positions, workspace);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_nl_count(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, void* workspace, void* counts) {
  int rc = validate_nl(d);
  if (rc) return rc;
  if (n_atoms == 0) return MIPME_OK;
  MIPME_REQUIRE(workspace && counts, "NULL buffer passed to mipme_nl_count");
  NlOut o = {};
  o.counts = (int*)counts;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return nl_walk_impl<float, kNlCount>(st, d, n_atoms, workspace, o);
  if (dtype == MIPME_F64) return nl_walk_impl<double, kNlCount>(st, d, n_atoms, workspace, o);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_nl_fill(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, void* workspace, const void* offsets,
                  void* pairs, void* shifts, void* dist) {
  int rc = validate_nl(d);
  if (rc) return rc;
  if (n_atoms == 0) return MIPME_OK;
  MIPME_REQUIRE(workspace && offsets && pairs && shifts, "NULL buffer passed to mipme_nl_fill");
  NlOut o = {};
  o.offsets = (const int64_t*)offsets;
  o.pairs = (int64_t*)pairs;
  o.shifts = shifts;
  o.dist = dist;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return nl_walk_impl<float, kNlFill>(st, d, n_atoms, workspace, o);
  if (dtype == MIPME_F64) return nl_walk_impl<double, kNlFill>(st, d, n_atoms, workspace, o);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_nl_stream(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, void* workspace, int64_t row_capacity,
                    void* row_ptr, void* words, void* host_status) {
  int rc = validate_nl(d);
  if (rc) return rc;
  MIPME_REQUIRE(workspace && row_ptr && (n_atoms == 0 || words), "NULL buffer passed to mipme_nl_stream");
  MIPME_REQUIRE(row_capacity >= 1 && n_atoms * row_capacity < (int64_t(1) << 31),
                "row capacity %lld x %lld atoms does not fit 32-bit row offsets", (long long)row_capacity, (long long)n_atoms);
  MIPME_REQUIRE(n_atoms <= kCompactMaxAtoms, "the 4-byte entry stream addresses at most 2^22 atoms");
  NlOut o = {};
  o.row_ptr = (int*)row_ptr;
  o.words = (int*)words;
  o.stride = int(row_capacity);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    rc = nl_walk_impl<float, kNlStream>(st, d, n_atoms, workspace, o);
  else if (dtype == MIPME_F64)
    rc = nl_walk_impl<double, kNlStream>(st, d, n_atoms, workspace, o);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  if (rc) return rc;
  NlWorkspace ws;
  nl_layout(int64_t(d->n_cells[0]) * d->n_cells[1] * d->n_cells[2], n_atoms, (char*)workspace, &ws);
  nl_report_kernel<<<1, 1, 0, st>>>(ws, (int*)host_status);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // extern "C"
