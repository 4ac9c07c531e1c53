// Independent frames in one launch (include/mipme.h: mipme_frames_*; SURVEY 8(e): the frames a rank owns).
// Device bodies: bricks_device.h (the single-frame kernels' bodies, blockIdx.y = frame).
#include "bricks_device.h"

namespace mipme {
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
  if (plane_list_capacity(m, n_atoms, dtype) > 0 && plane_fits_cosched(m, dtype)) n += int64_t(m->nx) * kPlaneSub + 1;
  return n;
}
static bool frame_plane_lists(const mipme_frame_t& f, int dtype) {
  // (whole planes only: the frame batches have no band variant; larger planes keep the bricks there)
  return plane_list_capacity(&f.mesh, f.n_atoms, dtype) > 0 && plane_fits_cosched(&f.mesh, dtype) &&
         int64_t(f.counter_ints) == frame_counter_ints(&f.mesh, f.n_atoms, dtype);
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
