// GPU neighbour list (cell list) -- SURVEY.md 8(f) rank 1, the caller-side step that precedes the hot path.
//
// The reference takes (pairs, shifts / distances) from third-party vesin on the host (tests/helpers.py:240-275,
// quantities "PdS").  This builder produces the same quantities on the device: pairs (P,2) int64, integer cell
// shifts S (P,3) with r_ij = r_j - r_i + S @ cell, and |r_ij|, for half or full lists, orthorhombic or triclinic
// cells, strict d < cutoff.  Scope of the device path: at least 3 cells of perpendicular width >= cutoff along every
// periodic axis (so every neighbour lies in the 27 surrounding cells and images do not alias); smaller boxes / larger
// cutoffs keep using the host builder (torch-pme_amd/neighbors.py).
//
// Phases (all on the caller's stream):  bin atoms into cells (wave-aggregated counting sort, then an index sort inside
// each cell so that the output order is deterministic)  ->  count accepted neighbours per atom (one wavefront per
// atom over its 27 cells)  ->  [host: exclusive scan of the counts, allocate P]  ->  fill (same traversal, ballot
// compaction).  Rows come out ordered by the first index i; inside a row the order is the traversal order.
#include "common.h"

namespace mipme {

struct NlGeom {
  double cell[9], inv[9];
  int nc[3];
  int periodic[3];
  double cutoff2;
  int full_list;
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
    g.foff[k] = d->periodic[k] ? 0.0 : d->frac_offset[k];
    g.fscale[k] = d->periodic[k] ? 1.0 : d->frac_scale[k];
  }
  g.cutoff2 = d->cutoff * d->cutoff;
  g.full_list = d->full_list;
  return g;
}

// wrapped position, wrap integers and cell index of one atom (all decisions in fp64)
template <typename T>
__device__ __forceinline__ void atom_cell(const NlGeom& g, const T* __restrict__ pos, int64_t i, double (&rw)[3],
                                          int (&w)[3], int (&c)[3]) {
  const double r[3] = {double(pos[3 * i]), double(pos[3 * i + 1]), double(pos[3 * i + 2])};
  double f[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    f[d] = r[0] * g.inv[d] + r[1] * g.inv[3 + d] + r[2] * g.inv[6 + d];
    const double fl = g.periodic[d] ? floor(f[d]) : 0.0;
    w[d] = int(fl);
    f[d] -= fl;
    int cd = int((f[d] - g.foff[d]) * g.fscale[d] * g.nc[d]);
    cd = cd < 0 ? 0 : (cd >= g.nc[d] ? g.nc[d] - 1 : cd);
    c[d] = cd;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) rw[k] = r[k] - (w[0] * g.cell[k] + w[1] * g.cell[3 + k] + w[2] * g.cell[6 + k]);
}

template <typename T>
__global__ __launch_bounds__(256) void nl_cell_count_kernel(NlGeom g, int64_t N, const T* __restrict__ pos,
                                                           int* __restrict__ cell_of, int* __restrict__ wrap,
                                                           int* __restrict__ count) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double rw[3];
  int w[3], c[3];
  atom_cell<T>(g, pos, i, rw, w, c);
  const int ci = (c[0] * g.nc[1] + c[1]) * g.nc[2] + c[2];
  cell_of[i] = ci;
  wrap[3 * i] = w[0];
  wrap[3 * i + 1] = w[1];
  wrap[3 * i + 2] = w[2];
  atomicAdd(&count[ci], 1);
}

__global__ __launch_bounds__(1024) void nl_scan_kernel(int n, const int* __restrict__ count, int* __restrict__ start) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = t * per, hi = min(lo + per, n);
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
  }
  if (t == 1023) start[n] = part[1023];
}

// deterministic placement: the slot of atom i inside its cell = number of atoms of the same cell with a smaller index
// (rank by counting; cells hold O(100) atoms, and this runs once per list)
__global__ __launch_bounds__(256) void nl_cell_fill_kernel(int64_t N, const int* __restrict__ cell_of,
                                                          const int* __restrict__ start, int* __restrict__ cursor,
                                                          int* __restrict__ cell_atoms) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ci = cell_of[i];
  const int slot = atomicAdd(&cursor[ci], 1);
  cell_atoms[start[ci] + slot] = int(i);
}

__global__ __launch_bounds__(64) void nl_cell_sort_kernel(const int* __restrict__ start, int* __restrict__ cell_atoms,
                                                         int* __restrict__ scratch) {
  const int ci = blockIdx.x;
  const int beg = start[ci], n = start[ci + 1] - beg;
  for (int k = threadIdx.x; k < n; k += 64) {
    const int me = cell_atoms[beg + k];
    int rank = 0;
    for (int m = 0; m < n; ++m) rank += cell_atoms[beg + m] < me;
    scratch[beg + rank] = me;
  }
}

// One wavefront per atom i: walk the atoms of the 27 surrounding cells, accept d < cutoff with the half-list rule
//   (i < j) or (i == j and S lexicographically positive);   S = s - w_j + w_i  (s = image shift of the cell walk).
// FILL = false: counts[i] = accepted;  FILL = true: write pairs / shifts / distances at offsets[i] + running index.
template <typename T, bool FILL>
__global__ __launch_bounds__(256) void nl_rows_kernel(NlGeom g, int64_t N, const T* __restrict__ pos,
                                                     const int* __restrict__ wrap, const int* __restrict__ start,
                                                     const int* __restrict__ cell_atoms, int* __restrict__ counts,
                                                     const int64_t* __restrict__ offsets, int64_t* __restrict__ pairs,
                                                     T* __restrict__ shifts, T* __restrict__ dist) {
  const int lane = threadIdx.x & 63;
  const int64_t i = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  double ri[3];
  int wi[3], ci[3];
  atom_cell<T>(g, pos, i, ri, wi, ci);
  int64_t written = FILL ? offsets[i] : 0;
  int total = 0;
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dz = -1; dz <= 1; ++dz) {
        const int dd[3] = {dx, dy, dz};
        int cc[3], s[3];
        bool skip = false;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          int v = ci[d] + dd[d];
          s[d] = 0;
          if (v < 0) {
            if (!g.periodic[d]) skip = true;
            v += g.nc[d];
            s[d] = -1;
          } else if (v >= g.nc[d]) {
            if (!g.periodic[d]) skip = true;
            v -= g.nc[d];
            s[d] = 1;
          }
          cc[d] = v;
        }
        if (skip) continue;
        const int cj = (cc[0] * g.nc[1] + cc[1]) * g.nc[2] + cc[2];
        const int beg = start[cj], end = start[cj + 1];
        const double sh[3] = {s[0] * g.cell[0] + s[1] * g.cell[3] + s[2] * g.cell[6],
                              s[0] * g.cell[1] + s[1] * g.cell[4] + s[2] * g.cell[7],
                              s[0] * g.cell[2] + s[1] * g.cell[5] + s[2] * g.cell[8]};
        for (int base = beg; base < end; base += 64) {
          const int k = base + lane;
          bool ok = false;
          int j = 0, S[3] = {0, 0, 0};
          double d2 = 0.0;
          if (k < end) {
            j = cell_atoms[k];
            const int wj[3] = {wrap[3 * j], wrap[3 * j + 1], wrap[3 * j + 2]};
            const double rj[3] = {double(pos[3 * int64_t(j)]), double(pos[3 * int64_t(j) + 1]), double(pos[3 * int64_t(j) + 2])};
            double v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const double rjw = rj[c] - (wj[0] * g.cell[c] + wj[1] * g.cell[3 + c] + wj[2] * g.cell[6 + c]);
              v[c] = rjw + sh[c] - ri[c];
              S[c] = s[c] - wj[c] + wi[c];
            }
            d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            const bool self = (j == int(i)) && s[0] == 0 && s[1] == 0 && s[2] == 0;
            ok = d2 < g.cutoff2 && !self;
            if (ok && !g.full_list) {
              const bool lexpos = S[0] > 0 || (S[0] == 0 && (S[1] > 0 || (S[1] == 0 && S[2] > 0)));
              ok = (int(i) < j) || (int(i) == j && lexpos);
            }
          }
          const unsigned long long m = __ballot(ok);
          if constexpr (FILL) {
            if (ok) {
              const int64_t dst = written + __popcll(m & ((1ull << lane) - 1ull));
              pairs[2 * dst] = i;
              pairs[2 * dst + 1] = j;
              shifts[3 * dst] = T(S[0]);
              shifts[3 * dst + 1] = T(S[1]);
              shifts[3 * dst + 2] = T(S[2]);
              if (dist) dist[dst] = T(sqrt(d2));
            }
            written += __popcll(m);
          } else {
            total += __popcll(m);
          }
        }
      }
  if constexpr (!FILL)
    if (lane == 0) counts[i] = total;
}

template <typename T>
static int nl_bin_impl(hipStream_t st, const mipme_nl_t* d, int64_t N, const void* pos, void* cell_of, void* wrap,
                       void* cell_start, void* cell_atoms, void* scratch) {
  const NlGeom g = make_nl(d);
  const int ncells = g.nc[0] * g.nc[1] * g.nc[2];
  int* count = (int*)scratch;               // [ncells + 1]
  int* cursor = count + (ncells + 1);       // [ncells + 1]
  int* tmp_atoms = cursor + (ncells + 1);   // [N]
  MIPME_CHECK_HIP(zero_async(scratch, sizeof(int) * size_t(2 * (ncells + 1)), st));
  if (N == 0) {
    MIPME_CHECK_HIP(zero_async(cell_start, sizeof(int) * size_t(ncells + 1), st));
    return MIPME_OK;
  }
  const unsigned blocks = unsigned((N + 255) / 256);
  nl_cell_count_kernel<T><<<blocks, 256, 0, st>>>(g, N, (const T*)pos, (int*)cell_of, (int*)wrap, count);
  MIPME_LAUNCH_CHECK();
  nl_scan_kernel<<<1, 1024, 0, st>>>(ncells, count, (int*)cell_start);
  MIPME_LAUNCH_CHECK();
  nl_cell_fill_kernel<<<blocks, 256, 0, st>>>(N, (const int*)cell_of, (const int*)cell_start, cursor, tmp_atoms);
  MIPME_LAUNCH_CHECK();
  nl_cell_sort_kernel<<<unsigned(ncells), 64, 0, st>>>((const int*)cell_start, tmp_atoms, (int*)cell_atoms);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

static int validate_nl(const mipme_nl_t* d) {
  MIPME_REQUIRE(d != nullptr, "neighbour-list descriptor is NULL");
  MIPME_REQUIRE(d->cutoff > 0, "cutoff must be positive");
  for (int k = 0; k < 3; ++k) {
    MIPME_REQUIRE(d->n_cells[k] >= 1, "invalid cell grid");
    MIPME_REQUIRE(d->periodic[k] || (d->n_cells[k] >= 1 && d->frac_scale[k] > 0.0),
                  "a non-periodic axis needs n_cells >= 1 and a positive frac_scale");
    MIPME_REQUIRE(!d->periodic[k] || d->n_cells[k] >= 3,
                  "the device neighbour list needs >= 3 cells of width >= cutoff along every periodic axis");
  }
  return MIPME_OK;
}

}  // namespace mipme

using namespace mipme;

extern "C" {

int64_t mipme_nl_scratch_ints(const mipme_nl_t* d, int64_t n_atoms) {
  if (!d) return 0;
  const int64_t ncells = int64_t(d->n_cells[0]) * d->n_cells[1] * d->n_cells[2];
  return 2 * (ncells + 1) + n_atoms;
}

int mipme_nl_bin(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, const void* positions, void* cell_of,
                 void* wrap, void* cell_start, void* cell_atoms, void* scratch) {
  int rc = validate_nl(d);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && cell_start && scratch && (n_atoms == 0 || (positions && cell_of && wrap && cell_atoms)),
                "NULL buffer passed to mipme_nl_bin");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return nl_bin_impl<float>(st, d, n_atoms, positions, cell_of, wrap, cell_start, cell_atoms, scratch);
  if (dtype == MIPME_F64) return nl_bin_impl<double>(st, d, n_atoms, positions, cell_of, wrap, cell_start, cell_atoms, scratch);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_nl_count(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, const void* positions, const void* wrap,
                   const void* cell_start, const void* cell_atoms, void* counts) {
  int rc = validate_nl(d);
  if (rc) return rc;
  if (n_atoms == 0) return MIPME_OK;
  MIPME_REQUIRE(positions && wrap && cell_start && cell_atoms && counts, "NULL buffer passed to mipme_nl_count");
  hipStream_t st = (hipStream_t)stream;
  const NlGeom g = make_nl(d);
  const unsigned blocks = unsigned((n_atoms + 3) / 4);
  if (dtype == MIPME_F32)
    nl_rows_kernel<float, false><<<blocks, 256, 0, st>>>(g, n_atoms, (const float*)positions, (const int*)wrap,
                                                         (const int*)cell_start, (const int*)cell_atoms, (int*)counts,
                                                         nullptr, nullptr, nullptr, nullptr);
  else if (dtype == MIPME_F64)
    nl_rows_kernel<double, false><<<blocks, 256, 0, st>>>(g, n_atoms, (const double*)positions, (const int*)wrap,
                                                          (const int*)cell_start, (const int*)cell_atoms, (int*)counts,
                                                          nullptr, nullptr, nullptr, nullptr);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_nl_fill(void* stream, int dtype, const mipme_nl_t* d, int64_t n_atoms, const void* positions, const void* wrap,
                  const void* cell_start, const void* cell_atoms, const void* offsets, void* pairs, void* shifts,
                  void* dist) {
  int rc = validate_nl(d);
  if (rc) return rc;
  if (n_atoms == 0) return MIPME_OK;
  MIPME_REQUIRE(positions && wrap && cell_start && cell_atoms && offsets && pairs && shifts, "NULL buffer passed to mipme_nl_fill");
  hipStream_t st = (hipStream_t)stream;
  const NlGeom g = make_nl(d);
  const unsigned blocks = unsigned((n_atoms + 3) / 4);
  if (dtype == MIPME_F32)
    nl_rows_kernel<float, true><<<blocks, 256, 0, st>>>(g, n_atoms, (const float*)positions, (const int*)wrap,
                                                        (const int*)cell_start, (const int*)cell_atoms, nullptr,
                                                        (const int64_t*)offsets, (int64_t*)pairs, (float*)shifts,
                                                        (float*)dist);
  else if (dtype == MIPME_F64)
    nl_rows_kernel<double, true><<<blocks, 256, 0, st>>>(g, n_atoms, (const double*)positions, (const int*)wrap,
                                                         (const int*)cell_start, (const int*)cell_atoms, nullptr,
                                                         (const int64_t*)offsets, (int64_t*)pairs, (double*)shifts,
                                                         (double*)dist);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // extern "C"
