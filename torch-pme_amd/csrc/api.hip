// extern "C" entry points of libmipme.so (declared in include/mipme.h) and the composite
// k-space forward / backward sequences (reference calculators/pme.py:88-143 and its autograd).
#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace mipme {

static thread_local char g_error[512] = "";

static thread_local const char* g_last_cosched = "";
void note_cosched_kernel(const char* name) { g_last_cosched = name ? name : ""; }
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// implemented in mesh.hip / kfilter.hip / rspace.hip
template <typename T> int spread_impl(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, double, void*);
template <typename T> int gather_impl(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, void*);
template <typename T> int gather_epilogue_impl(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, const void*, const void*, double, double, void*, void*, int, void*);
template <typename T> int gather_grad_impl(hipStream_t, const mipme_mesh_t*, int64_t, const void*, const void*, const void*, const void*, const void*, const void*, const void*, double, double, void*, void*);
template <typename T> int kfilter_build_impl(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, void*);
template <typename T> int apply_filter_impl(hipStream_t, int64_t, int, const void*, const void*, void*, void*);
template <typename T> int apply_filter_cellgrad_impl(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, const void*, const void*, const void*, void*, void*, void*);
template <typename T> int cellgrad_finalize_impl(hipStream_t, const mipme_mesh_t*, double, int64_t, void*, const void*, const void*, const void*, const void*, const void*, const void*, const void*, void*, int64_t, const void*, const void*);
int64_t xconv_blocks(const mipme_fft_plan*);
int64_t cellgrad_scratch_doubles();
int64_t cellgrad_blocks(const mipme_mesh_t*);
int fft_plan_create(int, int, int, int, int, mipme_fft_plan**);
int fft_plan_destroy(mipme_fft_plan*);
int fft_forward(mipme_fft_plan*, hipStream_t, const void*, void*);
int fft_inverse(mipme_fft_plan*, hipStream_t, void*, void*);
bool fft_plan_xfused(const mipme_fft_plan*);
int convolve_xfused(mipme_fft_plan*, hipStream_t, const void*, const void*, void*, void*, void*, int64_t, const mipme_mesh_t*,
                    const mipme_potential_t*, void*, void*, const void*, int64_t, const RowRideHost*, void*, const ConvCell*);
const void* bins_epart(const mipme_mesh_t*, int64_t, int, void*, int64_t*);
template <typename T, typename I> int rspace_forward_impl(hipStream_t, int64_t, int64_t, int, const void*, const void*, const void*, const void*, int, const mipme_potential_t*, int, void*);
template <typename T, typename I> int rspace_backward_impl(hipStream_t, int64_t, int64_t, int, const void*, const void*, const void*, const void*, int, const mipme_potential_t*, const void*, const void*, void*, void*);
template <typename T, typename I> int distance_forward_impl(hipStream_t, int64_t, const void*, const void*, const void*, const void*, void*);
template <typename T> int pack_pair_shifts_impl(hipStream_t, int64_t, const void*, void*, void*);
template <typename T> int distance_forward_packed_impl(hipStream_t, int64_t, const void*, const void*, const void*, const void*, void*);
template <typename T, typename I> int distance_backward_impl(hipStream_t, int64_t, int64_t, const void*, const void*, const void*, const void*, const void*, void*, void*, void*);
int64_t pair_partials_blocks(int64_t);

struct FftDims { int dtype, nx, ny, nz, batch; };
FftDims fft_plan_dims(const mipme_fft_plan*);
// bricks.hip
bool bricks_supported(const mipme_mesh_t*, int dtype);
int64_t bins_bytes(const mipme_mesh_t*, int64_t, int dtype);
template <typename T> int bins_build(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, int*, const void*, void*, bool = false);
template <typename T> int spread_bricks(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, double, void*, int*,
                                        const mipme_sr_job_t*, bool, double*, const PlaneHost* = nullptr, bool* = nullptr);
bool fft_plan_plane_forward_ok(const mipme_fft_plan*);
int plane_bins_capacity(const mipme_mesh_t*, int64_t, int);
int plane_bands(const mipme_mesh_t*, int);
void fft_plan_set_forward_done(mipme_fft_plan*, bool, int);
void fft_plan_set_forward_ycols(mipme_fft_plan*, bool);
void* fft_plan_hat_parts(mipme_fft_plan*, hipStream_t, int);
template <typename T> int kfilter_deriv_impl(hipStream_t, const mipme_mesh_t*, const mipme_potential_t*, void*);
template <typename T> int cell_tail_finalize_impl(hipStream_t, const mipme_mesh_t*, double, double, int64_t, int64_t, const void*,
                                                  const void*, const void*, const void*, void*);
bool sr_job_fusable(const mipme_sr_job_t*);
int* fft_plan_brick_count(const mipme_fft_plan*);
template <typename T> int gather_bricks(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, const void*, const void*, double, double, void*, void*, int, void*, const GatherTailHost*, void*, int*);
template <typename T> int gather_grad_bricks(hipStream_t, const mipme_mesh_t*, int64_t, void*, const void*, const void*, const void*, const void*, const void*, const void*, double, double, void*, void*);

// ---- optional per-stage timing (bench.py): HIP events recorded on the launch stream around every stage ----
struct ProfEntry {
  const char* name;
  hipEvent_t a, b;
  int reps;  // launches between the two events (the report divides): see kProfRepeat
};
// An event pair around ONE 25 us launch reads 3-4 us more than the kernel trace of the same launch (rocprofv3); stages that
// are idempotent -- the co-scheduled spread + pair sum overwrites everything it writes and only reads the binning counters -- are
// therefore launched kProfRepeat times back to back between their two events while the profiler is on.
static constexpr int kProfRepeat = 8;
static bool g_prof_on = false;
static std::vector<ProfEntry> g_prof;

// ---- optional tracing ranges (the reference brackets its pair sum with torch.profiler.record_function,
// calculators/calculator.py:52,72,77): MIPME_ROCTX=1 brackets every composite C-ABI entry point and every stage inside it with
// roctxRangePush / roctxRangePop, so that a rocprofv3 --marker-trace (or rocprof-sys) timeline of a user's model shows named
// ranges around the kernels instead of anonymous launches.  Off by default (one branch per scope); the library is dlopen'ed:
// libmipme does not link against a profiler.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
  Roctx() {
    const char* e = getenv("MIPME_ROCTX");
    if (!e || e[0] == '0') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) {
          on = true;
          return;
        }
      }
    }
  }
};
static const Roctx& roctx() {
  static const Roctx r;
  return r;
}
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(roctx().on) {
    if (on) roctx().push(name);
  }
  ~TraceRange() {
    if (on) roctx().pop();
  }
};

struct ProfScope {
  hipStream_t st;
  ProfEntry e;
  bool on;
  TraceRange range;
  ProfScope(hipStream_t s, const char* name, int reps = 1) : st(s), on(g_prof_on), range(name) {
    if (!on) return;
    e.name = name;
    e.reps = reps;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(e.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e.b, st);
    g_prof.push_back(e);
  }
};
#define STAGE(st, name, call)        \
  do {                               \
    ProfScope _ps(st, name);         \
    if ((rc = (call))) return rc;    \
  } while (0)

// self / background corrections: potentials/coulomb.py:144-158, potentials/inversepowerlaw.py:143-166
static void correction_terms(const mipme_potential_t* pot, double& self_c, double& bg_c) {
  const int p = pot->kind == MIPME_COULOMB ? 1 : pot->exponent;
  const double two_s2 = 2.0 * pot->smearing * pot->smearing;
  self_c = pot->prefactor / std::tgamma(0.5 * p + 1.0) / std::pow(two_s2, 0.5 * p);
  if (p >= 3)
    bg_c = 0.0;
  else
    bg_c = pot->prefactor * std::pow(3.14159265358979323846, 1.5) * std::pow(two_s2, 0.5 * (3 - p)) /
           ((3 - p) * std::tgamma(0.5 * p));
}

static int check_plan(const mipme_fft_plan* plan, int dtype, const mipme_mesh_t* m) {
  MIPME_REQUIRE(plan != nullptr, "FFT plan is NULL");
  const FftDims d = fft_plan_dims(plan);
  MIPME_REQUIRE(d.dtype == dtype && d.nx == m->nx && d.ny == m->ny && d.nz == m->nz && d.batch == m->n_channels,
                "The real-space mesh is inconsistent with the k-space grid.");
  return MIPME_OK;
}

// cell_work of an energy step's cell gradient (mipme_cell_tail_work), in doubles:
// [rows 25 per rider][rpart 9 per brick][cwave 9 per wavefront of the pair kernel][wbuf: one real per half-grid point]
struct CellWork {
  int64_t n_riders, n_bricks, n_waves;
  double *rows, *rpart, *cwave;
  void* wbuf;
  int64_t total;
};
static CellWork cell_work_layout(const mipme_mesh_t* m, int64_t N, void* base) {
  CellWork w;
  const int64_t Mh = int64_t(m->nx) * m->ny * (m->nz / 2 + 1);
  w.n_riders = std::min<int64_t>(256, std::max<int64_t>(8, Mh / 2048));  // ~2 k-points per rider thread (1024 threads)
  w.n_bricks = int64_t((m->nx + 7) / 8) * ((m->ny + 7) / 8) * ((m->nz + 7) / 8);
  w.n_waves = (N + 3) / 4;  // 16 lanes per row: 4 rows per wavefront (rows_body.h)
  double* b = (double*)base;
  w.rows = b;
  w.rpart = w.rows + 25 * w.n_riders;
  w.cwave = w.rpart + 9 * w.n_bricks;
  w.wbuf = w.cwave + 9 * w.n_waves;
  w.total = 25 * w.n_riders + 9 * w.n_bricks + 9 * w.n_waves + Mh;  // (wbuf: Mh reals of <= 8 bytes)
  return w;
}

static constexpr int kPlanePartsMax = 8;
static int plane_spread_parts_setting() {
  static const int parts_env = [] { const char* e = getenv("MIPME_PLANE_PARTS"); return e ? atoi(e) : 2; }();
  return parts_env < 1 ? 1 : (parts_env > kPlanePartsMax ? kPlanePartsMax : parts_env);
}

template <typename T>
static int kspace_forward_t(mipme_fft_plan* plan, hipStream_t st, const mipme_mesh_t* m, const mipme_potential_t* pot,
                            int64_t N, const void* pos, const void* q, const void* G, void* rho_mesh, void* rho_hat,
                            void* hat_work, void* phi_mesh, void* dc, void* out_lr, void* out_phi, void* bins,
                            void* wait_event, int accumulate, void* out_field, void* out_records,
                            const mipme_sr_job_t* job, void* cell_partials, const GatherTailHost* tail, void* nan_flag,
                            void* out_grad_cell = nullptr, const void* G_deriv = nullptr, void* cell_work = nullptr,
                            void* out_rho_hat = nullptr, bool rho_mesh_unused = false) {
  int rc;
  CellWork cw{};
  if (out_grad_cell) cw = cell_work_layout(m, N, cell_work);
  const int64_t Mh = int64_t(m->nx) * m->ny * (m->nz / 2 + 1);
  double self_c, bg_c;
  correction_terms(pot, self_c, bg_c);
  fft_plan_set_forward_done(plan, false, 1);               // (nor the plane spread's "forward planes done": set below, consumed by convolve_xfused)
  // the plan's brick counters are zero here; the binning pass fills them and the gather -- the last consumer -- zeroes them
  // again (no memset launch per call).  If anything in between fails they are cleared explicitly, so that a failed call does
  // not poison the next one.
  struct CounterGuard {
    hipStream_t st;
    int* counters;
    size_t n;
    bool armed;
    ~CounterGuard() {
      if (armed && counters) (void)zero_async(counters, sizeof(int) * n, st);
    }
  } guard{st, nullptr, plan_counter_words(m->nx, m->ny, m->nz), false};
  if (bins) {
    int* counters = fft_plan_brick_count(plan);
    guard.counters = counters;
    guard.armed = true;
    // fused convolution ahead and planes that fit a workgroup: the spread will write the forward (y,z) transform itself, from
    // the plane lists the binning pass leaves (bricks.hip plane_spread_yz_body)
    const bool want_planes = !rho_hat && rho_mesh_unused && fft_plan_plane_forward_ok(plan);
    STAGE(st, "bin_atoms", bins_build<T>(st, m, N, pos, bins, counters, q, out_records, want_planes));
    {
      const bool co = job && sr_job_fusable(job);
      const int reps = (g_prof_on && co) ? kProfRepeat : 1;
      ProfScope _ps(st, co ? "spread+rspace_forward" : "spread", reps);
      // fused convolution ahead and planes that fit a workgroup: the spread writes the forward (y,z) transform itself
      PlaneHost ph;
      bool planes = false;
      ph.slot_values = m->n_channels == 1;  // (bins_build above was handed these very charges)
      if (want_planes) {
        ph.hat = hat_work;
        ph.keep_mesh = false;
        // MIPME_PLANE_PARTS workgroups per plane (default 2, at most 8; measured 1 / 2 / 3 / 4 / 8: profiles/r05_experiments.txt): a plane's LDS atomics are what its workgroup waits for,
        // and they go through ONE CU's LDS pipe
        ph.parts = plane_spread_parts_setting();
        if (ph.parts > 1) {
          ph.hat_more = fft_plan_hat_parts(plan, st, kPlanePartsMax - 1);
          ph.more_stride = Mh * m->n_channels;
          if (!ph.hat_more) ph.parts = 1;
        }
      }
      for (int r = 0; r < reps; ++r)
        if ((rc = spread_bricks<T>(st, m, N, bins, q, 1.0, rho_mesh, counters, co ? job : nullptr, tail != nullptr,
                                   out_grad_cell ? cw.cwave : nullptr, &ph, &planes))) return rc;
      fft_plan_set_forward_done(plan, planes, planes ? ph.parts_used : 1);
      fft_plan_set_forward_ycols(plan, planes && ph.ycols_pending);
    }
    if (job && !sr_job_fusable(job))  // no co-scheduled kernel for this potential / shift format: one after the other
      STAGE(st, "rspace_forward",
            mipme_sr_rows_fused(st, sizeof(T) == 4 ? MIPME_F32 : MIPME_F64, job->n_atoms, job->row_ptr, job->entries_shift,
                                job->entries, nullptr, job->positions, job->cell, job->charges, job->charges, nullptr, 0,
                                job->full_list, job->pot, 0, job->shift_format, job->records, 1, job->out, job->force,
                                nullptr, nullptr, job->dist_out));
  } else {
    STAGE(st, "spread", spread_impl<T>(st, m, N, pos, q, 1.0, rho_mesh));
  }
  if (!rho_hat) {
    // nobody needs rfftn(rho) itself: (y,z) hipFFT planes + one kernel for x-FFT * G * inverse x-FFT
    int64_t n_sr_part = 0;
    const void* sr_part = tail ? bins_epart(m, N, sizeof(T) == 4 ? MIPME_F32 : MIPME_F64, bins, &n_sr_part) : nullptr;
    ConvCell cc{};
    if (out_grad_cell) cc = ConvCell{G_deriv, cw.cwave, cw.n_waves, cw.wbuf, cw.rows, int(cw.n_riders), nullptr, nullptr, nullptr, nullptr};
    cc.rho_hat_out = out_rho_hat;
    STAGE(st, "convolve_xfused", convolve_xfused(plan, st, rho_mesh, G, hat_work, phi_mesh, dc, 0, m, pot, cell_partials,
                                                 tail ? const_cast<void*>(tail->epart_k) : nullptr, sr_part, n_sr_part,
                                                 nullptr, nan_flag, (out_grad_cell || out_rho_hat) ? &cc : nullptr));
  } else {
    STAGE(st, "fft_r2c", fft_forward(plan, st, rho_mesh, rho_hat));
    STAGE(st, "apply_filter", apply_filter_impl<T>(st, Mh, m->n_channels, rho_hat, G, hat_work, dc));
    STAGE(st, "fft_c2r", fft_inverse(plan, st, hat_work, phi_mesh));
  }
  // the short-range sum may be running on another stream into out_lr: join it before the gather adds to it
  if (wait_event) MIPME_CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)wait_event, 0));
  if (bins)
    STAGE(st, tail ? "gather+energy+forces" : "gather",
          gather_bricks<T>(st, m, N, bins, phi_mesh, q, dc, self_c, bg_c, out_lr, out_phi, accumulate, out_field, tail, nan_flag,
                           fft_plan_brick_count(plan)));
  else
    STAGE(st, "gather", gather_epilogue_impl<T>(st, m, N, pos, phi_mesh, q, dc, self_c, bg_c, out_lr, out_phi, accumulate, nan_flag));
  guard.armed = N == 0 && bins;  // the gather has zeroed the counters (it does not run without atoms: nothing was counted either)
  guard.armed = false;
  if (out_grad_cell)
    STAGE(st, "cell_finalize",
          cell_tail_finalize_impl<T>(st, m, bg_c, 0.5 * tail->force_scale, cw.n_riders, cw.n_bricks, cw.rows, cw.rpart, dc,
                                     tail->aux_seed ? tail->aux_seed : tail->seed, out_grad_cell));
  return MIPME_OK;
}

template <typename T>
static int kspace_backward_t(mipme_fft_plan* plan, hipStream_t st, const mipme_mesh_t* m, const mipme_potential_t* pot,
                             int64_t N, const void* pos, const void* q, const void* gout, const void* G,
                             const void* phi_mesh, const void* rho_hat, const void* rho_dc, const void* phi_atoms,
                             void* psi_mesh, void* psi_hat, void* hat_work, void* chi_mesh, void* dc, void* partials,
                             void* grad_pos, void* grad_q, void* grad_cell, void* bins, const void* grad_scale,
                             const void* mesh_field, int64_t kgrid_blocks_ready, const void* G_deriv) {
  int rc;
  double self_c, bg_c;
  correction_terms(pot, self_c, bg_c);
  fft_plan_set_forward_done(plan, false, 1);  // (a forward call that failed after its plane spread must not make this call's convolution skip its forward planes)
  if (grad_scale) {
    // energy mode: grad_out = grad_scale * charges  =>  psi = (grad_scale/2V) rho, chi = (grad_scale/2V) phi:
    // no second spread / FFT / filter / inverse FFT (special case L = sum q V)
    MIPME_REQUIRE(rho_dc, "energy-mode backward needs rho_dc");
    // mesh_field (the forward gather's per-atom field, single channel): the mesh forces are gE q_a field_a -- no gradient
    // gather; the caller assembles them (mipme_sr_rows_finalize) and passes grad_positions = grad_charges = NULL
    const bool from_field = mesh_field != nullptr && !grad_pos && !grad_q;
    if (!from_field) {
      if (bins)
        STAGE(st, "gather_grad", gather_grad_bricks<T>(st, m, N, bins, q, gout, phi_mesh, phi_mesh, rho_dc, grad_scale, self_c, bg_c, grad_pos, grad_q));
      else
        STAGE(st, "gather_grad", gather_grad_impl<T>(st, m, N, pos, q, gout, phi_mesh, phi_mesh, rho_dc, grad_scale, self_c, bg_c, grad_pos, grad_q));
    }
    if (grad_cell) {
      // dL/dG(k) = (gE / 2V) mu(k) |rho^(k)|^2: the 12 k-grid sums from rho^ alone, scaled in the finalisation -- either
      // already in `partials` (kgrid_blocks_ready of them, written by the forward's fused convolution) or formed here from
      // the saved rho^
      MIPME_REQUIRE(phi_atoms && partials && (grad_pos || from_field),
                    "cell gradient needs phi_atoms, partials and grad_positions (or the mesh field) buffers");
      if (kgrid_blocks_ready <= 0) {
        MIPME_REQUIRE(rho_hat, "cell gradient needs rho_hat unless the k-grid sums are ready");
        STAGE(st, "apply_filter_cellgrad", apply_filter_cellgrad_impl<T>(st, m, pot, rho_hat, rho_hat, G, nullptr, nullptr, partials));
      }
      STAGE(st, "cellgrad_finalize",
            cellgrad_finalize_impl<T>(st, m, bg_c, N, partials, pos, from_field ? nullptr : grad_pos, gout, phi_atoms, rho_dc,
                                      rho_dc, grad_scale, grad_cell, kgrid_blocks_ready, mesh_field, q));
    }
    return MIPME_OK;
  }
  // psi = spread(g / 2V); chi = F psi
  if (bins)
    STAGE(st, "spread", spread_bricks<T>(st, m, N, bins, gout, 0.5 / m->volume, psi_mesh, nullptr, nullptr, false, nullptr));
  else
    STAGE(st, "spread", spread_impl<T>(st, m, N, pos, gout, 0.5 / m->volume, psi_mesh));
  const int64_t Mh = int64_t(m->nx) * m->ny * (m->nz / 2 + 1);
  if (grad_cell && !psi_hat) {
    // the fused convolution with the cell sums: psi^ is contracted with the saved rho^ by the x stage, the k-grid sums against
    // the derivative table are formed by the riders of the inverse (y,z) launch -- as 12-value rows for cellgrad_finalize_kernel
    MIPME_REQUIRE(G_deriv && rho_hat && rho_dc && phi_atoms && partials && grad_pos && m->n_channels == 1,
                  "fused cell gradient needs G_deriv, rho_hat, rho_dc, phi_atoms, partials, grad_positions and one channel");
    const int64_t n_riders = std::min<int64_t>(256, std::max<int64_t>(8, Mh / 2048));
    double* kh = (double*)partials;
    double* rows = kh + 12 * n_riders + cellgrad_scratch_doubles();
    void* wbuf = rows + 25 * n_riders;
    ConvCell cc{G_deriv, nullptr, 0, wbuf, rows, int(n_riders), nullptr, rho_hat, kh,
                reinterpret_cast<int*>(kh + 12 * n_riders + cellgrad_scratch_doubles() - 1)};
    STAGE(st, "convolve_xfused", convolve_xfused(plan, st, psi_mesh, G, hat_work, chi_mesh, dc, 0, m, pot, nullptr, nullptr,
                                                 nullptr, 0, nullptr, nullptr, &cc));
    if (bins)
      STAGE(st, "gather_grad", gather_grad_bricks<T>(st, m, N, bins, q, gout, phi_mesh, chi_mesh, dc, nullptr, self_c, bg_c, grad_pos, grad_q));
    else
      STAGE(st, "gather_grad", gather_grad_impl<T>(st, m, N, pos, q, gout, phi_mesh, chi_mesh, dc, nullptr, self_c, bg_c, grad_pos, grad_q));
    STAGE(st, "cellgrad_finalize",
          cellgrad_finalize_impl<T>(st, m, bg_c, N, partials, pos, grad_pos, gout, phi_atoms, rho_dc, dc, nullptr, grad_cell,
                                    n_riders, nullptr, nullptr));
    return MIPME_OK;
  }
  const bool xfused = !grad_cell && !psi_hat;
  if (xfused) {
    STAGE(st, "convolve_xfused", convolve_xfused(plan, st, psi_mesh, G, hat_work, chi_mesh, dc, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr));
  } else {
    STAGE(st, "fft_r2c", fft_forward(plan, st, psi_mesh, psi_hat));
  }
  if (xfused) {
  } else if (grad_cell) {
    MIPME_REQUIRE(rho_hat && rho_dc && phi_atoms && partials && grad_pos,
                  "cell gradient needs rho_hat, rho_dc, phi_atoms, partials and grad_positions buffers");
    STAGE(st, "apply_filter_cellgrad", apply_filter_cellgrad_impl<T>(st, m, pot, psi_hat, rho_hat, G, hat_work, dc, partials));
  } else {
    STAGE(st, "apply_filter", apply_filter_impl<T>(st, Mh, m->n_channels, psi_hat, G, hat_work, dc));
  }
  if (!xfused) STAGE(st, "fft_c2r", fft_inverse(plan, st, hat_work, chi_mesh));
  if (bins)
    STAGE(st, "gather_grad", gather_grad_bricks<T>(st, m, N, bins, q, gout, phi_mesh, chi_mesh, dc, nullptr, self_c, bg_c, grad_pos, grad_q));
  else
    STAGE(st, "gather_grad", gather_grad_impl<T>(st, m, N, pos, q, gout, phi_mesh, chi_mesh, dc, nullptr, self_c, bg_c, grad_pos, grad_q));
  if (grad_cell)
    STAGE(st, "cellgrad_finalize",
          cellgrad_finalize_impl<T>(st, m, bg_c, N, partials, pos, grad_pos, gout, phi_atoms, rho_dc, dc, nullptr, grad_cell, 0,
                                    nullptr, nullptr));
  return MIPME_OK;
}

// ---- slab (2-D periodic) correction: potentials/coulomb.py:6-40 ---------------------------------
// moments[c*6 + {0..5}] = Q, M, M2 (charges) and S0, S1, S2 (g/2) per channel
template <typename T>
__global__ __launch_bounds__(1024) void slab_moments_kernel(int axis, int64_t N, int C, const T* __restrict__ pos,
                                                           const T* __restrict__ q, const T* __restrict__ g,
                                                           double* __restrict__ moments) {
  __shared__ double red[16][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c < C; ++c) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t a = threadIdx.x; a < N; a += blockDim.x) {
      const double z = double(pos[3 * a + axis]);
      const double qc = double(q[a * C + c]);
      acc[0] += qc;
      acc[1] += qc * z;
      acc[2] += qc * z * z;
      if (g) {
        const double gh = 0.5 * double(g[a * C + c]);
        acc[3] += gh;
        acc[4] += gh * z;
        acc[5] += gh * z * z;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double v = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      double v = 0.0;
      for (int w = 0; w < int(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
      moments[c * 6 + threadIdx.x] = v;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void slab_forward_kernel(int axis, int64_t N, int C, double c0, double Lz, const T* __restrict__ pos,
                                    const double* __restrict__ mom, T* __restrict__ pot) {
  const int64_t a = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const double z = double(pos[3 * a + axis]);
  for (int c = 0; c < C; ++c) {
    const double Q = mom[c * 6], M = mom[c * 6 + 1], M2 = mom[c * 6 + 2];
    const double e = c0 * (z * M - 0.5 * (M2 + Q * z * z) - Q * Lz * Lz / 12.0);
    pot[a * C + c] += T(0.5 * e);
  }
}

template <typename T>
__global__ void slab_backward_kernel(int axis, int64_t N, int C, double c0, double Lz, const T* __restrict__ pos,
                                     const T* __restrict__ q, const T* __restrict__ g,
                                     const double* __restrict__ mom, T* __restrict__ grad_pos, T* __restrict__ grad_q) {
  const int64_t a = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const double z = double(pos[3 * a + axis]);
  double dz = 0.0;
  for (int c = 0; c < C; ++c) {
    const double Q = mom[c * 6], M = mom[c * 6 + 1];
    const double S0 = mom[c * 6 + 3], S1 = mom[c * 6 + 4], S2 = mom[c * 6 + 5];
    const double gh = 0.5 * double(g[a * C + c]);
    dz += gh * (M - Q * z) + double(q[a * C + c]) * (S1 - z * S0);
    if (grad_q) grad_q[a * C + c] += T(c0 * (z * S1 - 0.5 * z * z * S0 - 0.5 * S2 - S0 * Lz * Lz / 12.0));
  }
  if (grad_pos) grad_pos[3 * a + axis] += T(c0 * dz);
}

template <typename T>
__global__ void slab_cell_kernel(int axis, int C, mipme_mesh_t m, double c0, double Lz, const double* __restrict__ mom,
                                 T* __restrict__ grad_cell) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double sval = 0.0, dl = 0.0;
  for (int c = 0; c < C; ++c) {
    const double Q = mom[c * 6], M = mom[c * 6 + 1], M2 = mom[c * 6 + 2];
    const double S0 = mom[c * 6 + 3], S1 = mom[c * 6 + 4], S2 = mom[c * 6 + 5];
    sval += c0 * (S1 * M - 0.5 * (S0 * M2 + Q * S2) - Q * S0 * Lz * Lz / 12.0);
    dl += c0 * (-Q * S0 * Lz / 6.0);
  }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double v = -sval * m.inv_cell[3 * b + a];
      if (a == axis) v += dl * m.cell[3 * a + b] / Lz;
      grad_cell[3 * a + b] += T(v);
    }
}

// ---- E = sum_i a_i b_i (the caller's energy reduction, README.rst:112-114) and its adjoint ---------------------
// two stages, both deterministic: kDotBlocks block partials (double), then one wave sums them
static constexpr int kDotBlocks = 64;

// One kernel: per-block partial sums, then the block that draws the last ticket adds them in index order (deterministic)
// and resets the ticket counter.  scratch: kDotBlocks doubles + one int counter that is zero between calls.
template <typename T>
__global__ __launch_bounds__(64) void energy_log_push_kernel(int n, const T* __restrict__ src, double* __restrict__ log,
                                                             int* __restrict__ cursor, int capacity) {
  for (int f = threadIdx.x; f < n; f += 64) {  // a cursor per frame (the gather tails of a batch own one each: mipme.h)
    const int k = cursor[f];
    log[int64_t(unsigned(k) % unsigned(capacity)) * n + f] = double(src[f]);
    cursor[f] = k + 1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(int64_t n, const T* __restrict__ a, const T* __restrict__ b,
                                                 double* partials, int* counter, T* __restrict__ out) {
  double acc = 0.0;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    acc += double(a[i]) * double(b[i]);
  __shared__ double red[4];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], red[0] + red[1] + red[2] + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = ticket == int(gridDim.x) - 1;
  }
  __syncthreads();
  if (!last || wave != 0) return;
  double tot = int(threadIdx.x) < int(gridDim.x)
                   ? __hip_atomic_load(&partials[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                   : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
  if (threadIdx.x == 0) {
    out[0] = T(tot);
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// out_a[i] = g * b[i], out_b[i] = g * a[i]  (g: device scalar)
template <typename T>
__global__ __launch_bounds__(256) void dot_backward_kernel(int64_t n, const T* __restrict__ g, const T* __restrict__ a,
                                                          const T* __restrict__ b, T* __restrict__ out_a,
                                                          T* __restrict__ out_b) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T gv = g[0];
  if (out_a) out_a[i] = gv * b[i];
  if (out_b) out_b[i] = gv * a[i];
}

// Is g == s * q for one scalar s?  (the gradient (q * V).sum() sends back to V: energy mode of the backward passes.)
// One workgroup: s = g[k] / q[k] at the k of the largest |q|, then max_i |g_i - s q_i| <= tol |s q_i|.  result = {s, 0 | 1}.
// Both passes load kMatchUnroll elements per thread before touching any of them: a single workgroup has nothing else to hide
// its memory latency behind (the first version, one dependent load per iteration, took 25 us for 32k values).
static constexpr int kMatchUnroll = 8;

// n <= 1024 * kMatchRegs: every thread loads ALL its elements of q and g at once and keeps them in registers -- one memory
// round trip for the whole kernel (the reference element's g and q reach the other threads through LDS, not through a second
// trip).  The looped version below needs ~9 dependent trips to data other kernels have just written on other XCDs: 24 us at
// 32 000 values in the kernel trace of the reference call sequence, this one 8.9 (profiles/r03_j_kernel_stats_dropin.txt).
static constexpr int kMatchRegs = 32;
// PARTS: block b of many does this for ITS 32 768 values and leaves {scale_b, max |q|_b, bad_b} in parts[3 b ..] for
// scaled_match_combine_kernel (a block without a non-zero q demands g == 0 of its values and lets the others name the scale).
template <typename T, bool PARTS>
__global__ __launch_bounds__(1024) void scaled_match_small_kernel(int64_t n_all, const T* __restrict__ g_all,
                                                                  const T* __restrict__ q_all, T* __restrict__ result,
                                                                  int* __restrict__ host_flag, double* __restrict__ parts) {
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_bad[16];
  __shared__ T s_ref[2];
  const int64_t first = int64_t(blockIdx.x) * (1024 * kMatchRegs);
  const int n = int(n_all - first < 1024 * kMatchRegs ? n_all - first : 1024 * kMatchRegs);
  const T* __restrict__ g = g_all + first;
  const T* __restrict__ q = q_all + first;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T qv[kMatchRegs], gv[kMatchRegs];
#pragma unroll
  for (int u = 0; u < kMatchRegs; ++u) {
    const int i = threadIdx.x + u * 1024;
    qv[u] = i < n ? q[i] : T(0);
    gv[u] = i < n ? g[i] : T(0);
  }
  double best = -1.0;
  int bi = 0;
#pragma unroll
  for (int u = 0; u < kMatchRegs; ++u) {  // ascending index within the thread: the first maximum wins, as below
    const double v = fabs(double(qv[u]));
    if (v > best) {
      best = v;
      bi = threadIdx.x + u * 1024;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (s_val[w] > s_val[0] || (s_val[w] == s_val[0] && s_idx[w] < s_idx[0])) { s_val[0] = s_val[w]; s_idx[0] = s_idx[w]; }
  }
  __syncthreads();
  const int k = s_idx[0];
  const bool usable = s_val[0] > 0.0;
  if (int(threadIdx.x) == (k & 1023)) {  // the owner of element k hands its g and q to everybody
    const int u = k >> 10;
    T gk = T(0), qk = T(1);
#pragma unroll
    for (int t = 0; t < kMatchRegs; ++t)
      if (t == u) { gk = gv[t]; qk = qv[t]; }
    s_ref[0] = gk;
    s_ref[1] = qk;
  }
  __syncthreads();
  const T scale = usable ? s_ref[0] / s_ref[1] : T(0);
  const double sd = double(scale);
  const double tol = 8.0 * (sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16);
  int bad = PARTS ? ((!usable || isfinite(sd)) ? 0 : 1) : ((usable && isfinite(sd)) ? 0 : 1);
#pragma unroll
  for (int u = 0; u < kMatchRegs; ++u) {
    const double e = sd * double(qv[u]);
    if (!(fabs(double(gv[u]) - e) <= tol * fabs(e))) bad = 1;
  }
  bad = __any(bad) ? 1 : 0;
  if (lane == 0) s_bad[wave] = bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    int any = 0;
    for (int w = 0; w < 16; ++w) any |= s_bad[w];
    if constexpr (PARTS) {
      parts[3 * blockIdx.x] = sd;
      parts[3 * blockIdx.x + 1] = s_val[0];
      parts[3 * blockIdx.x + 2] = double(any);
    } else {
      result[0] = scale;
      result[1] = any ? T(0) : T(1);
      if (host_flag) __hip_atomic_store(host_flag, any ? 0 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// the verdict of many blocks: the scale of the block that holds the largest |q| (the first such block), every block's own check
// passed, and every other block's scale equal to it within 4 ulp -- |g - s q| <= ~12 ulp |s q| overall (one block: 8)
template <typename T>
__global__ __launch_bounds__(256) void scaled_match_combine_kernel(int n_blocks, const double* __restrict__ parts,
                                                                   T* __restrict__ result, int* __restrict__ host_flag) {
  __shared__ double s_val[4];
  __shared__ int s_idx[4];
  __shared__ int s_bad[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best = -1.0;
  int bi = 0;
  for (int b = threadIdx.x; b < n_blocks; b += 256) {
    const double v = parts[3 * b + 1];
    if (v > best) {
      best = v;
      bi = b;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_val[w] > s_val[0] || (s_val[w] == s_val[0] && s_idx[w] < s_idx[0])) { s_val[0] = s_val[w]; s_idx[0] = s_idx[w]; }
  }
  __syncthreads();
  const bool usable = s_val[0] > 0.0;
  const double sd = usable ? parts[3 * s_idx[0]] : 0.0;
  const double tol = 4.0 * (sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16);
  int bad = (usable && isfinite(sd)) ? 0 : 1;
  for (int b = threadIdx.x; b < n_blocks; b += 256) {
    if (parts[3 * b + 2] != 0.0) bad = 1;
    if (parts[3 * b + 1] > 0.0 && !(fabs(parts[3 * b] - sd) <= tol * fabs(sd))) bad = 1;
  }
  bad = __any(bad) ? 1 : 0;
  if (lane == 0) s_bad[wave] = bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int any = s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3];
    result[0] = T(sd);
    result[1] = any ? T(0) : T(1);
    if (host_flag) __hip_atomic_store(host_flag, any ? 0 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void scaled_match_kernel(int64_t n, const T* __restrict__ g, const T* __restrict__ q,
                                                            T* __restrict__ result, int* __restrict__ host_flag) {
  __shared__ double s_val[16];
  __shared__ long long s_idx[16];
  __shared__ int s_bad[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = blockDim.x;
  double best = -1.0;
  long long bi = 0;
  for (int64_t base = threadIdx.x; base < n; base += stride * kMatchUnroll) {
    T qv[kMatchUnroll];
#pragma unroll
    for (int u = 0; u < kMatchUnroll; ++u) {
      const int64_t i = base + u * stride;
      qv[u] = i < n ? q[i] : T(0);
    }
#pragma unroll
    for (int u = 0; u < kMatchUnroll; ++u) {
      const double v = fabs(double(qv[u]));
      if (v > best) {
        best = v;
        bi = base + u * stride;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(best, off, 64);
    const long long oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < int(blockDim.x >> 6); ++w)
      if (s_val[w] > s_val[0] || (s_val[w] == s_val[0] && s_idx[w] < s_idx[0])) { s_val[0] = s_val[w]; s_idx[0] = s_idx[w]; }
  }
  __syncthreads();
  const long long k = s_idx[0];
  const bool usable = s_val[0] > 0.0;
  const T scale = usable ? g[k] / q[k] : T(0);
  const double sd = double(scale);
  const double tol = 8.0 * (sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16);
  int bad = (usable && isfinite(sd)) ? 0 : 1;
  for (int64_t base = threadIdx.x; base < n; base += stride * kMatchUnroll) {
    T qv[kMatchUnroll], gv[kMatchUnroll];
#pragma unroll
    for (int u = 0; u < kMatchUnroll; ++u) {
      const int64_t i = base + u * stride;
      qv[u] = i < n ? q[i] : T(0);
      gv[u] = i < n ? g[i] : T(0);
    }
#pragma unroll
    for (int u = 0; u < kMatchUnroll; ++u) {
      const double e = sd * double(qv[u]);
      if (!(fabs(double(gv[u]) - e) <= tol * fabs(e))) bad = 1;
    }
  }
  bad = __any(bad) ? 1 : 0;
  if (lane == 0) s_bad[wave] = bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    int any = 0;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) any |= s_bad[w];
    result[0] = scale;
    result[1] = any ? T(0) : T(1);
    if (host_flag) __hip_atomic_store(host_flag, any ? 0 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// verdict = result of scaled_match_kernel: {s, 1 or 0}.  1: the upstream gradient is s * charges -- overwrite the general path's
// (skipped) outputs with the energy-mode expressions from the forward's per-atom sums; 0: leave them.
template <typename T>
__global__ __launch_bounds__(64) void values_equal_kernel(int n, const T* __restrict__ a, const T* __restrict__ b, int* host_flag) {
  bool same = true;
  for (int i = threadIdx.x; i < n; i += 64) same = same && (a[i] == b[i]);
  const bool all = __builtin_amdgcn_ballot_w64(!same) == 0;
  if (threadIdx.x == 0) __hip_atomic_store(host_flag, all ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Two 64-bit position-keyed sums over the 32-bit words w_j of the buffer: a = sum w_j K1(j), b = sum w_j K2(j) (mod 2^64), with
// K(j) = ((j + 1) C) mod 2^32 for two odd constants C -- bijections of the word index that are never zero below 2^32 - 1 words.
// A change of one word moves both sums (w K < 2^64 cannot wrap to zero), so does a swap of two different words ((w - w')(K - K')),
// and unrelated changes cancel in both with probability 2^-128.  One v_mad_u64_u32 per word and sum: the first version (four
// splitmix64 finalisers per 16 bytes) was bound by its 64-bit multiplies -- 91 us for the 76 MB list of cfg3; this one streams.
__device__ __forceinline__ void checksum_words(const uint4 v, unsigned j, unsigned long long& a, unsigned long long& b) {
  constexpr unsigned C1 = 0x9e3779b1u, C2 = 0x85ebca6bu;
  const unsigned k1 = (j + 1u) * C1, k2 = (j + 1u) * C2;  // keys of word j; the next words' keys follow by adding C
  a += (unsigned long long)v.x * k1 + (unsigned long long)v.y * (k1 + C1) + (unsigned long long)v.z * (k1 + 2u * C1) +
       (unsigned long long)v.w * (k1 + 3u * C1);
  b += (unsigned long long)v.x * k2 + (unsigned long long)v.y * (k2 + C2) + (unsigned long long)v.z * (k2 + 2u * C2) +
       (unsigned long long)v.w * (k2 + 3u * C2);
}

static constexpr int kChecksumBlocks = 256;

// sums: uint64[3 + 2 kChecksumBlocks]: {a, b, ticket, per-block partial sums}.  Every block stores its two partial sums and
// draws a ticket; the last one adds them up in block order.  (The first version added its sums to sums[0..1] with three
// device-scope atomics per block: 2048 blocks x 3 serialised round trips were 60 of the kernel's 83 us at 76 MB.)
__global__ __launch_bounds__(256) void checksum_kernel(const uint4* __restrict__ data, int64_t n_vec, const unsigned* __restrict__ tail,
                                                       int n_tail, unsigned long long* __restrict__ sums,
                                                       const unsigned long long* __restrict__ expect, int* host_flag) {
  unsigned long long a = 0, b = 0;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n_vec; i += 8 * stride) {  // eight loads in flight
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = data[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) checksum_words(v[u], unsigned(4 * (i + u * stride)), a, b);
  }
  for (; i < n_vec; i += stride) checksum_words(data[i], unsigned(4 * i), a, b);
  if (blockIdx.x == 0 && int(threadIdx.x) < n_tail) {
    const unsigned j = unsigned(4 * n_vec) + threadIdx.x;
    a += (unsigned long long)tail[threadIdx.x] * ((j + 1u) * 0x9e3779b1u);
    b += (unsigned long long)tail[threadIdx.x] * ((j + 1u) * 0x85ebca6bu);
  }
  __shared__ unsigned long long red[4][2];
  __shared__ bool last;
  auto block_sum = [&](unsigned long long& x, unsigned long long& y) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      x += __shfl_xor(x, off, 64);
      y += __shfl_xor(y, off, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6][0] = x;
      red[threadIdx.x >> 6][1] = y;
    }
    __syncthreads();
    x = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    y = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  };
  block_sum(a, b);
  unsigned long long* part = sums + 3;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&part[2 * blockIdx.x], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&part[2 * blockIdx.x + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = __hip_atomic_fetch_add(&sums[2], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  unsigned long long ta = 0, tb = 0;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += blockDim.x) {
    ta += __hip_atomic_load(&part[2 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tb += __hip_atomic_load(&part[2 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  block_sum(ta, tb);
  if (threadIdx.x == 0) {
    const unsigned long long s0 = sums[0] + ta, s1 = sums[1] + tb;
    sums[0] = s0;
    sums[1] = s1;
    sums[2] = 0;  // (ready for a further buffer accumulated into the same sums)
    if (host_flag) {
      const int same = expect && s0 == expect[0] && s1 == expect[1];
      __hip_atomic_store(host_flag, same ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void energy_select_kernel(int64_t N, const T* __restrict__ verdict, const T* __restrict__ q,
                                                           const T* __restrict__ force, const T* __restrict__ field, T f,
                                                           T* __restrict__ grad_mesh, T* __restrict__ grad_pair) {
  if (verdict[1] != T(1)) return;
  const T s = verdict[0];
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < 3 * N; t += int64_t(gridDim.x) * blockDim.x) {
    const T sq = s * q[t / 3];
    if (grad_mesh) grad_mesh[t] = sq * field[t];
    if (grad_pair) grad_pair[t] = sq * f * force[t];
  }
}

// ... and the variant for a caller that wants ONE gradient: out = match ? s q (f force + field) : grad_mesh + grad_pair
template <typename T>
__global__ __launch_bounds__(256) void energy_select_sum_kernel(int64_t N, const T* __restrict__ verdict, const T* __restrict__ q,
                                                               const T* __restrict__ force, const T* __restrict__ field, T f,
                                                               const T* grad_mesh, const T* grad_pair, T* out) {
  const bool match = verdict[1] == T(1);
  const T s = verdict[0];
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < 3 * N; t += int64_t(gridDim.x) * blockDim.x)
    out[t] = match ? s * q[t / 3] * (f * force[t] + field[t]) : grad_mesh[t] + grad_pair[t];
}

// ... and for the charge / cell gradients of the same step (mipme.h, mipme_energy_select_contract)
template <typename T>
__global__ __launch_bounds__(256) void energy_select_contract_kernel(int64_t N, const T* __restrict__ verdict,
                                                                    const T* __restrict__ tail_q, T* grad_q,
                                                                    const T* __restrict__ tail_cell, int cell_off,
                                                                    const T* cell_mesh, const T* cell_pair, T* grad_cell) {
  const bool match = verdict[1] == T(1);
  const T s = verdict[0];
  if (grad_cell && blockIdx.x == 0 && threadIdx.x < 9) {
    const int i = threadIdx.x;
    grad_cell[i] = match ? s * tail_cell[cell_off + i] : cell_mesh[i] + (cell_pair ? cell_pair[i] : T(0));
  }
  if (!grad_q || !match) return;  // (the general adjoint's charge gradient is already in place)
  const T h = T(0.5) * s;
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < N; t += int64_t(gridDim.x) * blockDim.x)
    grad_q[t] = h * tail_q[t];
}

static double axis_length(const mipme_mesh_t* m, int axis) {
  const double* a = m->cell + 3 * axis;
  return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
}

template <typename T>
static int slab_forward_t(hipStream_t st, int axis, const mipme_mesh_t* m, double pref, int64_t N, const void* pos,
                          const void* q, void* moments, void* pot) {
  if (N == 0) return MIPME_OK;
  const int C = m->n_channels;
  const double c0 = pref * 4.0 * 3.14159265358979323846 / m->volume;
  slab_moments_kernel<T><<<1, 1024, 0, st>>>(axis, N, C, (const T*)pos, (const T*)q, (const T*)nullptr, (double*)moments);
  MIPME_LAUNCH_CHECK();
  slab_forward_kernel<T><<<unsigned((N + 255) / 256), 256, 0, st>>>(axis, N, C, c0, axis_length(m, axis), (const T*)pos,
                                                                    (const double*)moments, (T*)pot);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int slab_backward_t(hipStream_t st, int axis, const mipme_mesh_t* m, double pref, int64_t N, const void* pos,
                           const void* q, const void* g, void* moments, void* grad_pos, void* grad_q, void* grad_cell) {
  if (N == 0) return MIPME_OK;
  const int C = m->n_channels;
  const double c0 = pref * 4.0 * 3.14159265358979323846 / m->volume;
  const double Lz = axis_length(m, axis);
  slab_moments_kernel<T><<<1, 1024, 0, st>>>(axis, N, C, (const T*)pos, (const T*)q, (const T*)g, (double*)moments);
  MIPME_LAUNCH_CHECK();
  if (grad_pos || grad_q) {
    slab_backward_kernel<T><<<unsigned((N + 255) / 256), 256, 0, st>>>(
        axis, N, C, c0, Lz, (const T*)pos, (const T*)q, (const T*)g, (const double*)moments, (T*)grad_pos, (T*)grad_q);
    MIPME_LAUNCH_CHECK();
  }
  if (grad_cell) {
    slab_cell_kernel<T><<<1, 64, 0, st>>>(axis, C, *m, c0, Lz, (const double*)moments, (T*)grad_cell);
    MIPME_LAUNCH_CHECK();
  }
  return MIPME_OK;
}

}  // namespace mipme

namespace mipme {
void* fft_plan_tail_scratch(mipme_fft_plan*, int64_t bytes);
}
// Copy a caller's versioned argument struct into the library's own layout: only the first `size` bytes the caller
// compiled are read, everything after them stays zero (fields are only ever appended).
template <typename A>
static int load_args(const A* in, A& out, const char* what) {
  MIPME_REQUIRE(in != nullptr, "%s: NULL argument struct", what);
  MIPME_REQUIRE(in->version == MIPME_ARGS_VERSION, "%s: argument struct version %u, library expects %u", what, in->version,
                unsigned(MIPME_ARGS_VERSION));
  MIPME_REQUIRE(in->size >= 16 && in->size <= 4096, "%s: implausible argument struct size %u", what, in->size);
  memset(&out, 0, sizeof(A));
  memcpy(&out, in, in->size < sizeof(A) ? in->size : sizeof(A));
  return MIPME_OK;
}


using namespace mipme;

#define DT_SWITCH(dtype, CALL_F32, CALL_F64)                    \
  do {                                                          \
    if ((dtype) == MIPME_F32) return CALL_F32;                  \
    if ((dtype) == MIPME_F64) return CALL_F64;                  \
    set_error("invalid dtype %d", int(dtype));                  \
    return MIPME_EINVAL;                                        \
  } while (0)

// ---- MD step on live bins (bricks.hip) ---------------------------------------------------------------------------------------
namespace mipme {
bool live_supported(const mipme_mesh_t*, int64_t, int);
int64_t live_lists_ints(const mipme_mesh_t*, int64_t);
template <typename T> int live_rebin(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, void*);
template <typename T> int live_spread(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, void*, const mipme_sr_job_t*, void*, double*);
template <typename T> int live_gather(hipStream_t, const mipme_mesh_t*, int64_t, const void*, void*, void*, const void*, const void*,
                                      double, double, void*, void*, const GatherTailHost*, void*);
}  // namespace mipme

static int md_check(const mipme_md_args_t* in, mipme_md_args_t& a, const char* who) {
  MIPME_REQUIRE(in && in->size >= 16 && in->version == 1, "%s: NULL or unversioned argument struct", who);
  std::memset(&a, 0, sizeof(a));
  std::memcpy(&a, in, std::min<size_t>(in->size, sizeof(a)));
  int rc;
  if ((rc = validate_mesh(a.mesh))) return rc;
  MIPME_REQUIRE(a.dtype == MIPME_F32 || a.dtype == MIPME_F64, "invalid dtype %d", a.dtype);
  MIPME_REQUIRE(a.pot && a.pot->smearing > 0, "Must specify smearing to use a potential with PMECalculator");
  MIPME_REQUIRE(live_supported(a.mesh, a.n_atoms, a.dtype), "%s: mesh / atom count outside the live-bin kernels' range", who);
  MIPME_REQUIRE(a.records && a.atom_bins && a.live_lists, "%s: NULL buffer", who);
  return MIPME_OK;
}

template <typename T>
static int md_step_t(const mipme_md_args_t& a) {
  int rc;
  hipStream_t st = (hipStream_t)a.stream;
  const mipme_mesh_t* m = a.mesh;
  double self_c, bg_c;
  correction_terms(a.pot, self_c, bg_c);
  mipme_sr_job_t job{};
  job.n_atoms = a.n_atoms;
  job.row_ptr = a.row_ptr;
  job.entries_shift = a.words;
  job.entries = a.words;  // (no pair indices behind the rows: never read)
  job.positions = nullptr;
  job.cell = a.cell;
  job.charges = nullptr;
  job.pot = a.pot;
  job.full_list = 0;
  job.shift_format = a.shift_format;
  job.records = const_cast<void*>(a.records);
  job.out = a.potentials;
  job.force = a.pair_force;
  job.dist_out = nullptr;
  GatherTailHost tail{};
  tail.force = a.pair_force;
  tail.force_scale = 1.0;
  tail.seed = a.grad_seed;
  tail.grad_pos = a.grad_positions;
  tail.energy = a.energy;
  tail.n_k = xconv_blocks(a.plan);
  tail.epart_k = fft_plan_tail_scratch(a.plan, 3 * int64_t(sizeof(double)) * tail.n_k);
  tail.sr_reduced = 1;
  MIPME_REQUIRE(tail.epart_k, "could not allocate the energy partial sums of the plan (not possible during stream capture: run "
                              "one evaluation before capturing)");
  CellWork cw{};
  if (a.grad_cell) {
    cw = cell_work_layout(m, a.n_atoms, a.cell_work);
    tail.rpart = cw.rpart;
  }
  tail.grad_q = a.grad_charges;
  tail.aux_seed = a.aux_seed;
  tail.live_flags = a.host_flags;
  if (a.energy_log) {
    MIPME_REQUIRE(a.energy_log_cursor && a.energy_log_capacity > 0, "energy_log needs energy_log_cursor and a capacity > 0");
    tail.elog = (double*)a.energy_log;
    tail.elog_cursor = (int*)a.energy_log_cursor;
    tail.elog_cap = int(a.energy_log_capacity);
  }
  STAGE(st, "spread+rspace_forward", live_spread<T>(st, m, a.n_atoms, a.records, a.atom_bins, a.live_lists, a.rho_mesh, &job, a.host_flags,
                                                     a.grad_cell ? cw.cwave : nullptr));
  int64_t n_sr_part = 0;
  const void* sr_part = bins_epart(m, a.n_atoms, a.dtype, a.atom_bins, &n_sr_part);
  const ConvCell cc{a.G_deriv, cw.cwave, cw.n_waves, cw.wbuf, cw.rows, int(cw.n_riders), nullptr, nullptr, nullptr, nullptr};
  STAGE(st, "convolve_xfused", convolve_xfused(a.plan, st, a.rho_mesh, a.G, a.hat_work, a.phi_mesh, a.dc, 0, m, a.pot, nullptr,
                                               const_cast<void*>(tail.epart_k), sr_part, n_sr_part, nullptr, a.nan_flag,
                                               a.grad_cell ? &cc : nullptr));
  STAGE(st, "gather+energy+forces",
        live_gather<T>(st, m, a.n_atoms, a.records, a.atom_bins, a.live_lists, a.phi_mesh, a.dc, self_c, bg_c, a.potentials, nullptr,
                       &tail, a.nan_flag));
  if (a.grad_cell)
    STAGE(st, "cell_finalize",
          cell_tail_finalize_impl<T>(st, m, bg_c, 0.5, cw.n_riders, cw.n_bricks, cw.rows, cw.rpart, a.dc,
                                     a.aux_seed ? a.aux_seed : a.grad_seed, a.grad_cell));
  return MIPME_OK;
}


template <typename T>
static int scaled_match_wide_t(hipStream_t st, int64_t n, const void* g, const void* q, void* result, void* host_flag, void* work) {
  const int64_t nb = (n + 1024 * kMatchRegs - 1) / (1024 * kMatchRegs);
  MIPME_REQUIRE(nb < (int64_t(1) << 24), "mipme_scaled_match_wide: too many values");
  scaled_match_small_kernel<T, true><<<unsigned(nb), 1024, 0, st>>>(n, (const T*)g, (const T*)q, nullptr, nullptr, (double*)work);
  MIPME_LAUNCH_CHECK();
  scaled_match_combine_kernel<T><<<1, 256, 0, st>>>(int(nb), (const double*)work, (T*)result, (int*)host_flag);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

extern "C" {

const char* mipme_last_error(void) { return g_error; }
const char* mipme_last_cosched_kernel(void) { return g_last_cosched; }
int mipme_version(void) { return MIPME_VERSION; }

int mipme_fft_plan_create(int dtype, int nx, int ny, int nz, int batch, mipme_fft_plan** out) {
  return fft_plan_create(dtype, nx, ny, nz, batch, out);
}
int mipme_fft_plan_destroy(mipme_fft_plan* plan) { return fft_plan_destroy(plan); }

int mipme_kfilter_build(void* stream, int dtype, const mipme_mesh_t* mesh, const mipme_potential_t* pot, void* G) {
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(G != nullptr, "G is NULL");
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, kfilter_build_impl<float>(st, mesh, pot, G), kfilter_build_impl<double>(st, mesh, pot, G));
}

int mipme_kfilter_build_deriv(void* stream, int dtype, const mipme_mesh_t* mesh, const mipme_potential_t* pot, void* G_deriv) {
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(G_deriv != nullptr, "G_deriv is NULL");
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, kfilter_deriv_impl<float>(st, mesh, pot, G_deriv), kfilter_deriv_impl<double>(st, mesh, pot, G_deriv));
}

int64_t mipme_cell_tail_work(const mipme_fft_plan* plan, const mipme_mesh_t* mesh, int64_t n_atoms) {
  if (!plan || !mesh || validate_mesh(mesh) || n_atoms < 0) return 0;
  return cell_work_layout(mesh, n_atoms, nullptr).total;
}

int mipme_convolve(mipme_fft_plan* plan, void* stream, const void* mesh_in, const void* G, void* hat_out, void* hat_work,
                   void* mesh_out, void* dc_out) {
  TraceRange _tr("mipme_convolve");
  MIPME_REQUIRE(plan != nullptr, "FFT plan is NULL");
  MIPME_REQUIRE(mesh_in && G && hat_out && hat_work && mesh_out, "NULL buffer passed to mipme_convolve");
  hipStream_t st = (hipStream_t)stream;
  const FftDims d = fft_plan_dims(plan);
  const int64_t Mh = int64_t(d.nx) * d.ny * (d.nz / 2 + 1);
  int rc;
  if ((rc = fft_forward(plan, st, mesh_in, hat_out))) return rc;
  if (d.dtype == MIPME_F32)
    rc = apply_filter_impl<float>(st, Mh, d.batch, hat_out, G, hat_work, dc_out);
  else
    rc = apply_filter_impl<double>(st, Mh, d.batch, hat_out, G, hat_work, dc_out);
  if (rc) return rc;
  return fft_inverse(plan, st, hat_work, mesh_out);
}

int mipme_spread(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* positions,
                 const void* values, void* mesh_out) {
  TraceRange _tr("mipme_spread");
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && mesh_out && (n_atoms == 0 || (positions && values)), "NULL buffer passed to mipme_spread");
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, spread_impl<float>(st, mesh, n_atoms, positions, values, 1.0, mesh_out),
            spread_impl<double>(st, mesh, n_atoms, positions, values, 1.0, mesh_out));
}

int mipme_gather(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* positions,
                 const void* mesh_in, void* out) {
  TraceRange _tr("mipme_gather");
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  MIPME_REQUIRE(n_atoms >= 0 && mesh_in && (n_atoms == 0 || (positions && out)), "NULL buffer passed to mipme_gather");
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, gather_impl<float>(st, mesh, n_atoms, positions, mesh_in, out),
            gather_impl<double>(st, mesh, n_atoms, positions, mesh_in, out));
}

int mipme_kspace_forward(const mipme_kspace_forward_args_t* args_in) {
  TraceRange _tr("mipme_kspace_forward");
  mipme_kspace_forward_args_t a;
  int rc = load_args(args_in, a, "mipme_kspace_forward");
  if (rc) return rc;
  if ((rc = validate_mesh(a.mesh))) return rc;
  if ((rc = check_plan(a.plan, a.dtype, a.mesh))) return rc;
  const mipme_mesh_t* mesh = a.mesh;
  MIPME_REQUIRE(a.pot && a.pot->smearing > 0, "Must specify smearing to use a potential with PMECalculator");
  MIPME_REQUIRE(a.G && a.rho_mesh && a.hat_work && a.phi_mesh && a.dc, "NULL work buffer passed to mipme_kspace_forward");
  MIPME_REQUIRE(a.rho_hat || fft_plan_xfused(a.plan), "rho_hat may only be NULL for plans with a power-of-two nx");
  MIPME_REQUIRE(!a.out_cell_partials || !a.rho_hat, "out_cell_partials is produced by the fused convolution (rho_hat == NULL)");
  MIPME_REQUIRE(!a.out_rho_hat || (!a.rho_hat && mesh->n_channels == 1),
                "out_rho_hat is written by the fused convolution (rho_hat == NULL) of a single-channel mesh");
  MIPME_REQUIRE(a.n_atoms == 0 || (a.positions && a.charges && a.out_lr), "NULL atom buffer passed to mipme_kspace_forward");
  MIPME_REQUIRE(!a.atom_bins || bricks_supported(mesh, a.dtype), "atom bins passed for a mesh the brick kernels do not support");
  MIPME_REQUIRE(!a.out_field || (a.atom_bins && mesh->n_channels == 1), "out_field needs atom bins and a single channel");
  MIPME_REQUIRE(!a.out_records || (a.atom_bins && mesh->n_channels == 1), "out_records needs atom bins and a single channel");
  if (a.sr_job) {
    MIPME_REQUIRE(a.atom_bins && a.out_records && mesh->n_channels == 1 && a.accumulate_out && !a.gather_wait_event,
                  "sr_job needs atom bins, out_records, a single channel and accumulate_out = 1 without a wait event");
    MIPME_REQUIRE(a.sr_job->n_atoms == a.n_atoms && a.sr_job->records == a.out_records && a.sr_job->out == a.out_lr,
                  "sr_job must describe the atoms, records and output of the same call");
    MIPME_REQUIRE(a.sr_job->row_ptr && a.sr_job->entries_shift && a.sr_job->entries && a.sr_job->positions &&
                      a.sr_job->charges && a.sr_job->pot, "NULL buffer in sr_job");
  }
  GatherTailHost tail{}, *tp = nullptr;
  if (a.out_energy || a.out_grad_positions) {
    MIPME_REQUIRE(a.out_energy && a.out_grad_positions && a.sr_job && a.sr_job->force && a.out_field && a.n_atoms > 0,
                  "the gather tail (out_energy, out_grad_positions) needs both outputs, sr_job with force sums and out_field");
    tail.force = a.sr_job->force;
    tail.force_scale = a.sr_job->full_list ? 0.5 : 1.0;
    tail.seed = a.grad_seed;
    tail.grad_pos = a.out_grad_positions;
    tail.energy = a.out_energy;
    MIPME_REQUIRE(!a.rho_hat && mesh->n_channels == 1 && sr_job_fusable(a.sr_job),
                  "the gather tail needs the fused convolution (rho_hat == NULL), one channel and a co-schedulable sr_job");
    tail.n_k = xconv_blocks(a.plan);
    tail.epart_k = fft_plan_tail_scratch(a.plan, 3 * int64_t(sizeof(double)) * tail.n_k);  // + the reduced pair partials
    tail.sr_reduced = 1;
    MIPME_REQUIRE(tail.epart_k, "could not allocate the energy partial sums of the plan (not possible during stream "
                                "capture: run one evaluation before capturing)");
    tp = &tail;
  }
  if (a.energy_log) {
    MIPME_REQUIRE(tp && a.energy_log_cursor && a.energy_log_capacity > 0,
                  "energy_log rides on the gather tail (out_energy, out_grad_positions) and needs energy_log_cursor and a capacity > 0");
    tail.elog = (double*)a.energy_log;
    tail.elog_cursor = (int*)a.energy_log_cursor;
    tail.elog_cap = int(a.energy_log_capacity);
  }
  if (a.out_grad_charges || a.out_grad_cell) {
    MIPME_REQUIRE(tp, "out_grad_charges / out_grad_cell ride on the gather tail (out_energy, out_grad_positions)");
    MIPME_REQUIRE(!a.out_grad_charges || !a.sr_job->full_list || (a.sr_job->shift_format & MIPME_ROWS_PADDED),
                  "out_grad_charges = 2 s V needs a half list (a full list need not be symmetric)");
    tail.grad_q = a.out_grad_charges;
    tail.aux_seed = a.aux_seed;
    if (a.out_grad_cell) {
      const int p = a.pot->kind == MIPME_COULOMB ? 1 : a.pot->exponent;
      MIPME_REQUIRE(a.G_deriv && a.cell_work, "out_grad_cell needs G_deriv (mipme_kfilter_build_deriv) and cell_work");
      MIPME_REQUIRE((a.sr_job->shift_format & 0xff) == 2 && !a.sr_job->dist_out && (p == 1 || p == 6),
                    "out_grad_cell needs 4-byte entries (shift_format 2), no dist_out, and 1/r or 1/r^6");
      MIPME_REQUIRE(!a.out_cell_partials, "out_grad_cell replaces out_cell_partials");
      tail.rpart = cell_work_layout(mesh, a.n_atoms, a.cell_work).rpart;
      tail.records = a.out_records;
    }
  }
  hipStream_t st = (hipStream_t)a.stream;
  DT_SWITCH(a.dtype,
            kspace_forward_t<float>(a.plan, st, mesh, a.pot, a.n_atoms, a.positions, a.charges, a.G, a.rho_mesh, a.rho_hat,
                                    a.hat_work, a.phi_mesh, a.dc, a.out_lr, a.out_phi, a.atom_bins, a.gather_wait_event,
                                    a.accumulate_out, a.out_field, a.out_records, a.sr_job, a.out_cell_partials, tp, a.nan_flag,
                                    a.out_grad_cell, a.G_deriv, a.cell_work, a.out_rho_hat, (a.flags & MIPME_FWD_RHO_MESH_UNUSED) != 0),
            kspace_forward_t<double>(a.plan, st, mesh, a.pot, a.n_atoms, a.positions, a.charges, a.G, a.rho_mesh, a.rho_hat,
                                     a.hat_work, a.phi_mesh, a.dc, a.out_lr, a.out_phi, a.atom_bins, a.gather_wait_event,
                                     a.accumulate_out, a.out_field, a.out_records, a.sr_job, a.out_cell_partials, tp, a.nan_flag,
                                     a.out_grad_cell, a.G_deriv, a.cell_work, a.out_rho_hat, (a.flags & MIPME_FWD_RHO_MESH_UNUSED) != 0));
}

int mipme_md_supported(const mipme_mesh_t* mesh, const mipme_potential_t* pot, int64_t n_atoms, int dtype) {
  if (!mesh || !pot || validate_mesh(mesh) || pot->smearing <= 0 || pot->exclusion_radius > 0) return 0;
  const int p = pot->kind == MIPME_COULOMB ? 1 : pot->exponent;
  if (p != 1 && p != 6) return 0;
  if (mesh->nx < 2 || (mesh->nx & (mesh->nx - 1))) return 0;  // the fused convolution needs a power-of-two nx
  return live_supported(mesh, n_atoms, dtype) ? 1 : 0;
}

int64_t mipme_md_lists_ints(const mipme_mesh_t* mesh, int64_t n_atoms) {
  if (!mesh || validate_mesh(mesh) || n_atoms <= 0) return 0;
  return live_lists_ints(mesh, n_atoms);
}

int mipme_md_rebin(const mipme_md_args_t* args_in) {
  TraceRange _tr("mipme_md_rebin");
  mipme_md_args_t a;
  int rc = md_check(args_in, a, "mipme_md_rebin");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)a.stream;
  DT_SWITCH(a.dtype, live_rebin<float>(st, a.mesh, a.n_atoms, a.records, a.atom_bins, a.live_lists, a.host_flags),
            live_rebin<double>(st, a.mesh, a.n_atoms, a.records, a.atom_bins, a.live_lists, a.host_flags));
}

int mipme_md_step(const mipme_md_args_t* args_in) {
  TraceRange _tr("mipme_md_step");
  mipme_md_args_t a;
  int rc = md_check(args_in, a, "mipme_md_step");
  if (rc) return rc;
  if ((rc = check_plan(a.plan, a.dtype, a.mesh))) return rc;
  MIPME_REQUIRE(fft_plan_xfused(a.plan), "mipme_md_step needs a plan with a power-of-two nx");
  MIPME_REQUIRE(mipme_md_supported(a.mesh, a.pot, a.n_atoms, a.dtype), "mipme_md_step: potential / mesh outside its range");
  MIPME_REQUIRE(a.cell && a.G && a.rho_mesh && a.hat_work && a.phi_mesh && a.dc && a.row_ptr && a.words && a.potentials &&
                    a.pair_force && a.energy && a.grad_positions, "NULL buffer passed to mipme_md_step");
  MIPME_REQUIRE((a.shift_format & 0xff) == 2, "mipme_md_step reads 4-byte entries (shift_format 2, with or without MIPME_ROWS_PADDED)");
  if (a.grad_cell) {
    const int p = a.pot->kind == MIPME_COULOMB ? 1 : a.pot->exponent;
    MIPME_REQUIRE(a.G_deriv && a.cell_work, "grad_cell needs G_deriv (mipme_kfilter_build_deriv) and cell_work");
    MIPME_REQUIRE(p == 1 || p == 6, "grad_cell: the pair kernels form the cell sums for 1/r and 1/r^6");
  }
  DT_SWITCH(a.dtype, md_step_t<float>(a), md_step_t<double>(a));
}

int mipme_kspace_backward(const mipme_kspace_backward_args_t* args_in) {
  TraceRange _tr("mipme_kspace_backward");
  mipme_kspace_backward_args_t a;
  int rc = load_args(args_in, a, "mipme_kspace_backward");
  if (rc) return rc;
  if ((rc = validate_mesh(a.mesh))) return rc;
  if ((rc = check_plan(a.plan, a.dtype, a.mesh))) return rc;
  MIPME_REQUIRE(a.pot && a.pot->smearing > 0, "Must specify smearing to use a potential with PMECalculator");
  MIPME_REQUIRE(a.G && a.phi_mesh && (a.grad_scale || (a.psi_mesh && a.hat_work && a.chi_mesh && a.dc)),
                "NULL work buffer passed to mipme_kspace_backward");
  MIPME_REQUIRE(a.grad_scale || a.psi_hat || ((!a.grad_cell || a.G_deriv) && fft_plan_xfused(a.plan)),
                "psi_hat may only be NULL for plans with a power-of-two nx, and with a cell gradient only together with G_deriv");
  MIPME_REQUIRE(a.n_atoms == 0 || (a.positions && a.charges && a.grad_out), "NULL atom buffer passed to mipme_kspace_backward");
  MIPME_REQUIRE(!a.atom_bins || bricks_supported(a.mesh, a.dtype), "atom bins passed for a mesh the brick kernels do not support");
  hipStream_t st = (hipStream_t)a.stream;
  DT_SWITCH(a.dtype,
            kspace_backward_t<float>(a.plan, st, a.mesh, a.pot, a.n_atoms, a.positions, a.charges, a.grad_out, a.G, a.phi_mesh,
                                     a.rho_hat, a.rho_dc, a.phi_atoms, a.psi_mesh, a.psi_hat, a.hat_work, a.chi_mesh, a.dc,
                                     a.partials, a.grad_positions, a.grad_charges, a.grad_cell, a.atom_bins, a.grad_scale,
                                     a.mesh_field, a.kgrid_blocks_ready, a.G_deriv),
            kspace_backward_t<double>(a.plan, st, a.mesh, a.pot, a.n_atoms, a.positions, a.charges, a.grad_out, a.G, a.phi_mesh,
                                      a.rho_hat, a.rho_dc, a.phi_atoms, a.psi_mesh, a.psi_hat, a.hat_work, a.chi_mesh, a.dc,
                                      a.partials, a.grad_positions, a.grad_charges, a.grad_cell, a.atom_bins, a.grad_scale,
                                      a.mesh_field, a.kgrid_blocks_ready, a.G_deriv));
}

int mipme_fft_plan_xfused(const mipme_fft_plan* plan) { return plan && fft_plan_xfused(plan) ? 1 : 0; }

int64_t mipme_fft_plan_kgrid_blocks(const mipme_fft_plan* plan) { return plan ? xconv_blocks(plan) : 0; }

/* hat = rfftn(mesh) over the three mesh dimensions of every channel, un-normalised (the plan's 3-D R2C transform) */
int mipme_fft_r2c(mipme_fft_plan* plan, void* stream, int dtype, const mipme_mesh_t* mesh, const void* mesh_in, void* hat) {
  int rc = validate_mesh(mesh);
  if (rc) return rc;
  if ((rc = check_plan(plan, dtype, mesh))) return rc;
  MIPME_REQUIRE(mesh_in && hat, "NULL buffer passed to mipme_fft_r2c");
  return fft_forward(plan, (hipStream_t)stream, mesh_in, hat);
}

int mipme_plane_spread_parts(const mipme_mesh_t* mesh, int64_t n_atoms, int dtype) {
  if (!mesh || validate_mesh(mesh) || !bricks_supported(mesh, dtype)) return 0;
  if (mesh->nx < 2 || (mesh->nx & (mesh->nx - 1)) || plane_bins_capacity(mesh, n_atoms, dtype) <= 0) return 0;
  return plane_bands(mesh, dtype) > 1 ? 1 : plane_spread_parts_setting();  // (planes spread in bands of rows: one workgroup per band)
}

int64_t mipme_atom_bins_bytes(const mipme_mesh_t* mesh, int64_t n_atoms, int dtype) {
  if (!mesh || n_atoms < 0) return 0;
  return bins_bytes(mesh, n_atoms, dtype);
}

int mipme_profile_enable(int on) {
  for (auto& e : g_prof) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  g_prof.clear();
  g_prof_on = on != 0;
  return MIPME_OK;
}

/* Writes "name calls total_ms\n" lines (NUL terminated) for the stages recorded since mipme_profile_enable(1);
 * synchronises on the recorded events.  Returns the number of bytes needed (excluding the NUL). */
int64_t mipme_profile_report(char* buf, int64_t buflen) {
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& e : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      auto& a = agg[e.name];
      a.first += 1;
      a.second += ms / float(e.reps > 0 ? e.reps : 1);
    }
  }
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && buflen > 0) {
    const size_t n = out.size() < size_t(buflen - 1) ? out.size() : size_t(buflen - 1);
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return int64_t(out.size());
}

int64_t mipme_cellgrad_partials_size(const mipme_mesh_t* mesh, int64_t n_atoms) {
  (void)n_atoms;
  if (!mesh) return 0;
  // k-grid partial sums come either from apply_filter_cellgrad (one block per 256 half-grid points) or from the x stage of
  // the fused convolution (at most one block per (ky, kz) column and channel)
  const int64_t xmax = int64_t(mesh->ny) * (mesh->nz / 2 + 1) * mesh->n_channels;
  // ... or from the riders of the fused general adjoint (G_deriv): 12 + 25 doubles per rider and one real per half-grid point
  const int64_t Mh = int64_t(mesh->nx) * mesh->ny * (mesh->nz / 2 + 1);
  const int64_t fused = 37 * 256 + Mh;
  return std::max<int64_t>(12 * std::max<int64_t>(cellgrad_blocks(mesh), xmax), fused) + cellgrad_scratch_doubles();
}

int mipme_slab_forward(void* stream, int dtype, int axis, const mipme_mesh_t* mesh, double prefactor, int64_t n_atoms,
                       const void* positions, const void* charges, void* moments, void* pot) {
  MIPME_REQUIRE(mesh && axis >= 0 && axis < 3 && moments, "invalid arguments to mipme_slab_forward");
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, slab_forward_t<float>(st, axis, mesh, prefactor, n_atoms, positions, charges, moments, pot),
            slab_forward_t<double>(st, axis, mesh, prefactor, n_atoms, positions, charges, moments, pot));
}

int mipme_slab_backward(void* stream, int dtype, int axis, const mipme_mesh_t* mesh, double prefactor, int64_t n_atoms,
                        const void* positions, const void* charges, const void* grad_out, void* moments,
                        void* grad_positions, void* grad_charges, void* grad_cell) {
  MIPME_REQUIRE(mesh && axis >= 0 && axis < 3 && moments && grad_out, "invalid arguments to mipme_slab_backward");
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype,
            slab_backward_t<float>(st, axis, mesh, prefactor, n_atoms, positions, charges, grad_out, moments,
                                   grad_positions, grad_charges, grad_cell),
            slab_backward_t<double>(st, axis, mesh, prefactor, n_atoms, positions, charges, grad_out, moments,
                                    grad_positions, grad_charges, grad_cell));
}

#define IDX_SWITCH(dtype, idx, FN, ...)                                                   \
  do {                                                                                    \
    if ((dtype) == MIPME_F32 && (idx) == MIPME_I64) return FN<float, int64_t>(__VA_ARGS__);  \
    if ((dtype) == MIPME_F32 && (idx) == MIPME_I32) return FN<float, int32_t>(__VA_ARGS__);  \
    if ((dtype) == MIPME_F64 && (idx) == MIPME_I64) return FN<double, int64_t>(__VA_ARGS__); \
    if ((dtype) == MIPME_F64 && (idx) == MIPME_I32) return FN<double, int32_t>(__VA_ARGS__); \
    set_error("invalid dtype/index dtype %d/%d", int(dtype), int(idx));                   \
    return MIPME_EINVAL;                                                                  \
  } while (0)

int mipme_rspace_forward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels,
                         const void* pairs, const void* dist, const void* charges, const void* pair_mask,
                         int full_list, const mipme_potential_t* pot, int accumulate, void* out_pot) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0 && n_channels > 0, "invalid sizes passed to mipme_rspace_forward");
  MIPME_REQUIRE(n_atoms == 0 || out_pot, "out_pot is NULL");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && dist && charges), "NULL pair buffer passed to mipme_rspace_forward");
  hipStream_t st = (hipStream_t)stream;
  IDX_SWITCH(dtype, idx_dtype, rspace_forward_impl, st, n_pairs, n_atoms, n_channels, pairs, dist, charges, pair_mask,
             full_list, pot, accumulate, out_pot);
}

int mipme_rspace_backward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels,
                          const void* pairs, const void* dist, const void* charges, const void* pair_mask,
                          int full_list, const mipme_potential_t* pot, const void* grad_out, const void* grad_scale,
                          void* grad_dist, void* grad_charges) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0 && n_channels > 0, "invalid sizes passed to mipme_rspace_backward");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && dist && charges && grad_out), "NULL pair buffer passed to mipme_rspace_backward");
  hipStream_t st = (hipStream_t)stream;
  IDX_SWITCH(dtype, idx_dtype, rspace_backward_impl, st, n_pairs, n_atoms, n_channels, pairs, dist, charges, pair_mask,
             full_list, pot, grad_out, grad_scale, grad_dist, grad_charges);
}

int mipme_pair_distance_forward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, const void* pairs,
                                const void* positions, const void* cell, const void* shifts, void* out_dist) {
  MIPME_REQUIRE(n_pairs >= 0, "invalid n_pairs");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && positions && out_dist), "NULL buffer passed to mipme_pair_distance_forward");
  MIPME_REQUIRE((cell == nullptr) == (shifts == nullptr), "`cell` and `shifts` must be given together");
  hipStream_t st = (hipStream_t)stream;
  IDX_SWITCH(dtype, idx_dtype, distance_forward_impl, st, n_pairs, pairs, positions, cell, shifts, out_dist);
}

int mipme_pack_pair_shifts(void* stream, int dtype, int64_t n_pairs, const void* shifts, void* packed, void* flag) {
  MIPME_REQUIRE(n_pairs >= 0 && flag && (n_pairs == 0 || (shifts && packed)), "invalid arguments to mipme_pack_pair_shifts");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return pack_pair_shifts_impl<float>(st, n_pairs, shifts, packed, flag);
  if (dtype == MIPME_F64) return pack_pair_shifts_impl<double>(st, n_pairs, shifts, packed, flag);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_pair_distance_forward_packed(void* stream, int dtype, int64_t n_pairs, const void* pairs32,
                                       const void* packed_shifts, const void* positions, const void* cell,
                                       void* out_dist) {
  MIPME_REQUIRE(n_pairs >= 0, "invalid n_pairs");
  MIPME_REQUIRE(n_pairs == 0 || (pairs32 && packed_shifts && positions && cell && out_dist),
                "NULL buffer passed to mipme_pair_distance_forward_packed");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) return distance_forward_packed_impl<float>(st, n_pairs, pairs32, packed_shifts, positions, cell, out_dist);
  if (dtype == MIPME_F64) return distance_forward_packed_impl<double>(st, n_pairs, pairs32, packed_shifts, positions, cell, out_dist);
  set_error("invalid dtype %d", dtype);
  return MIPME_EINVAL;
}

int mipme_pair_distance_backward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms,
                                 const void* pairs, const void* positions, const void* cell, const void* shifts,
                                 const void* grad_dist, void* partials, void* grad_positions, void* grad_cell) {
  MIPME_REQUIRE(n_pairs >= 0 && n_atoms >= 0, "invalid sizes");
  MIPME_REQUIRE(grad_positions || n_atoms == 0, "grad_positions is NULL");
  MIPME_REQUIRE(n_pairs == 0 || (pairs && positions && grad_dist), "NULL buffer passed to mipme_pair_distance_backward");
  MIPME_REQUIRE((cell == nullptr) == (shifts == nullptr), "`cell` and `shifts` must be given together");
  MIPME_REQUIRE(!grad_cell || cell, "cell gradient requested without a cell");
  hipStream_t st = (hipStream_t)stream;
  IDX_SWITCH(dtype, idx_dtype, distance_backward_impl, st, n_pairs, n_atoms, pairs, positions, cell, shifts, grad_dist,
             partials, grad_positions, grad_cell);
}

int mipme_dot_forward(void* stream, int dtype, int64_t n, const void* a, const void* b, void* scratch, void* out) {
  MIPME_REQUIRE(n >= 0 && out && scratch && (n == 0 || (a && b)), "invalid arguments to mipme_dot_forward");
  hipStream_t st = (hipStream_t)stream;
  double* partials = (double*)scratch;
  int* counter = (int*)(partials + kDotBlocks);
  if (dtype == MIPME_F32) {
    dot_kernel<float><<<kDotBlocks, 256, 0, st>>>(n, (const float*)a, (const float*)b, partials, counter, (float*)out);
  } else if (dtype == MIPME_F64) {
    dot_kernel<double><<<kDotBlocks, 256, 0, st>>>(n, (const double*)a, (const double*)b, partials, counter, (double*)out);
  } else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_dot_backward(void* stream, int dtype, int64_t n, const void* grad, const void* a, const void* b, void* grad_a,
                       void* grad_b) {
  MIPME_REQUIRE(n >= 0 && grad && (n == 0 || (a && b)), "invalid arguments to mipme_dot_backward");
  if (n == 0 || (!grad_a && !grad_b)) return MIPME_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = unsigned((n + 255) / 256);
  if (dtype == MIPME_F32)
    dot_backward_kernel<float><<<blocks, 256, 0, st>>>(n, (const float*)grad, (const float*)a, (const float*)b,
                                                       (float*)grad_a, (float*)grad_b);
  else if (dtype == MIPME_F64)
    dot_backward_kernel<double><<<blocks, 256, 0, st>>>(n, (const double*)grad, (const double*)a, (const double*)b,
                                                        (double*)grad_a, (double*)grad_b);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_energy_log_push(void* stream, int dtype, int n, const void* src, void* log, void* cursor, int capacity) {
  MIPME_REQUIRE(n > 0 && capacity > 0 && src && log && cursor, "invalid arguments to mipme_energy_log_push");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32) {
    energy_log_push_kernel<float><<<1, 64, 0, st>>>(n, (const float*)src, (double*)log, (int*)cursor, capacity);
  } else if (dtype == MIPME_F64) {
    energy_log_push_kernel<double><<<1, 64, 0, st>>>(n, (const double*)src, (double*)log, (int*)cursor, capacity);
  } else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int64_t mipme_scaled_match_work(int64_t n) {
  return n <= 1024 * kMatchRegs ? 0 : 3 * ((n + 1024 * kMatchRegs - 1) / (1024 * kMatchRegs));
}

int mipme_scaled_match_wide(void* stream, int dtype, int64_t n, const void* g, const void* q, void* result, void* host_flag,
                            void* work) {
  MIPME_REQUIRE(n > 0 && g && q && result, "invalid arguments to mipme_scaled_match_wide");
  MIPME_REQUIRE(dtype == MIPME_F32 || dtype == MIPME_F64, "invalid dtype %d", dtype);
  if (!work || n <= 1024 * kMatchRegs) return mipme_scaled_match(stream, dtype, n, g, q, result, host_flag);
  return dtype == MIPME_F32 ? scaled_match_wide_t<float>((hipStream_t)stream, n, g, q, result, host_flag, work)
                            : scaled_match_wide_t<double>((hipStream_t)stream, n, g, q, result, host_flag, work);
}

int mipme_scaled_match(void* stream, int dtype, int64_t n, const void* g, const void* q, void* result, void* host_flag) {
  MIPME_REQUIRE(n > 0 && g && q && result, "invalid arguments to mipme_scaled_match");
  hipStream_t st = (hipStream_t)stream;
  const bool small = n <= 1024 * kMatchRegs;
  if (dtype == MIPME_F32 && small)
    scaled_match_small_kernel<float, false><<<1, 1024, 0, st>>>(n, (const float*)g, (const float*)q, (float*)result, (int*)host_flag,
                                                                nullptr);
  else if (dtype == MIPME_F64 && small)
    scaled_match_small_kernel<double, false><<<1, 1024, 0, st>>>(n, (const double*)g, (const double*)q, (double*)result,
                                                                 (int*)host_flag, nullptr);
  else if (dtype == MIPME_F32)
    scaled_match_kernel<float><<<1, 1024, 0, st>>>(n, (const float*)g, (const float*)q, (float*)result, (int*)host_flag);
  else if (dtype == MIPME_F64)
    scaled_match_kernel<double><<<1, 1024, 0, st>>>(n, (const double*)g, (const double*)q, (double*)result, (int*)host_flag);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int64_t mipme_pair_partials_size(int64_t n_pairs) { return 9 * pair_partials_blocks(n_pairs); }

int64_t mipme_checksum_words(void) { return 3 + 2 * kChecksumBlocks; }

int mipme_checksum(void* stream, const void* data, int64_t n_bytes, void* sums, const void* expect, void* host_flag) {
  MIPME_REQUIRE(n_bytes >= 0 && (n_bytes % 4) == 0 && sums && (n_bytes == 0 || data), "mipme_checksum: a multiple of 4 bytes, non-NULL buffers");
  MIPME_REQUIRE((reinterpret_cast<uintptr_t>(data) & 15) == 0, "mipme_checksum: the buffer must be 16-byte aligned");
  const int64_t n_vec = n_bytes / 16;
  const int n_tail = int((n_bytes % 16) / 4);
  int64_t blocks = (n_vec + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > kChecksumBlocks ? kChecksumBlocks : blocks);
  checksum_kernel<<<unsigned(blocks), 256, 0, (hipStream_t)stream>>>(
      (const uint4*)data, n_vec, (const unsigned*)((const char*)data + 16 * n_vec), n_tail, (unsigned long long*)sums,
      (const unsigned long long*)expect, (int*)host_flag);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_values_equal(void* stream, int dtype, int64_t n, const void* a, const void* b, void* host_flag) {
  MIPME_REQUIRE(n >= 0 && n <= 1024 && a && b && host_flag, "mipme_values_equal: up to 1024 values, non-NULL buffers");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIPME_F32)
    values_equal_kernel<float><<<1, 64, 0, st>>>(int(n), (const float*)a, (const float*)b, (int*)host_flag);
  else if (dtype == MIPME_F64)
    values_equal_kernel<double><<<1, 64, 0, st>>>(int(n), (const double*)a, (const double*)b, (int*)host_flag);
  else
    MIPME_REQUIRE(false, "invalid dtype %d", dtype);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_set_skip_flag(const void* device_flag) {
  skip_flag_slot() = (const int*)device_flag;
  return MIPME_OK;
}

int mipme_energy_select(void* stream, int dtype, int64_t n_atoms, const void* verdict, const void* charges, const void* force,
                        const void* field, int full_list, void* grad_mesh, void* grad_pair) {
  MIPME_REQUIRE(n_atoms >= 0 && verdict && charges, "invalid arguments to mipme_energy_select");
  MIPME_REQUIRE((!grad_mesh || field) && (!grad_pair || force), "mipme_energy_select: an output without its forward sum");
  if (n_atoms == 0) return MIPME_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = unsigned(std::min<int64_t>((3 * n_atoms + 255) / 256, 4096));
  if (dtype == MIPME_F32)
    energy_select_kernel<float><<<blocks, 256, 0, st>>>(n_atoms, (const float*)verdict, (const float*)charges, (const float*)force,
                                                       (const float*)field, full_list ? 0.5f : 1.0f, (float*)grad_mesh, (float*)grad_pair);
  else if (dtype == MIPME_F64)
    energy_select_kernel<double><<<blocks, 256, 0, st>>>(n_atoms, (const double*)verdict, (const double*)charges, (const double*)force,
                                                        (const double*)field, full_list ? 0.5 : 1.0, (double*)grad_mesh, (double*)grad_pair);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_energy_select_sum(void* stream, int dtype, int64_t n_atoms, const void* verdict, const void* charges, const void* force,
                            const void* field, int full_list, const void* grad_mesh, const void* grad_pair, void* out) {
  MIPME_REQUIRE(n_atoms >= 0 && verdict && charges && force && field && grad_mesh && grad_pair && out,
                "invalid arguments to mipme_energy_select_sum");
  if (n_atoms == 0) return MIPME_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = unsigned(std::min<int64_t>((3 * n_atoms + 255) / 256, 4096));
  if (dtype == MIPME_F32)
    energy_select_sum_kernel<float><<<blocks, 256, 0, st>>>(n_atoms, (const float*)verdict, (const float*)charges, (const float*)force,
                                                           (const float*)field, full_list ? 0.5f : 1.0f, (const float*)grad_mesh,
                                                           (const float*)grad_pair, (float*)out);
  else if (dtype == MIPME_F64)
    energy_select_sum_kernel<double><<<blocks, 256, 0, st>>>(n_atoms, (const double*)verdict, (const double*)charges,
                                                            (const double*)force, (const double*)field, full_list ? 0.5 : 1.0,
                                                            (const double*)grad_mesh, (const double*)grad_pair, (double*)out);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

int mipme_energy_select_contract(void* stream, int dtype, int64_t n_atoms, const void* verdict, const void* tail_grad_charges,
                                 void* grad_charges, const void* tail_grad_cell, int pair_through_distances,
                                 const void* cell_mesh, const void* cell_pair, void* grad_cell) {
  MIPME_REQUIRE(n_atoms >= 0 && verdict, "invalid arguments to mipme_energy_select_contract");
  MIPME_REQUIRE(!grad_charges || tail_grad_charges, "grad_charges needs the tail's charge gradient");
  MIPME_REQUIRE(!grad_cell || (tail_grad_cell && cell_mesh), "grad_cell needs the tail's cell gradient and the general mesh part");
  if (!grad_cell && (!grad_charges || n_atoms == 0)) return MIPME_OK;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = grad_charges ? unsigned(std::max<int64_t>(1, std::min<int64_t>((n_atoms + 255) / 256, 2048))) : 1u;
  const int off = pair_through_distances ? 0 : 18;
  if (dtype == MIPME_F32)
    energy_select_contract_kernel<float><<<blocks, 256, 0, st>>>(n_atoms, (const float*)verdict, (const float*)tail_grad_charges,
                                                                (float*)grad_charges, (const float*)tail_grad_cell, off,
                                                                (const float*)cell_mesh, (const float*)cell_pair,
                                                                (float*)grad_cell);
  else if (dtype == MIPME_F64)
    energy_select_contract_kernel<double><<<blocks, 256, 0, st>>>(n_atoms, (const double*)verdict,
                                                                 (const double*)tail_grad_charges, (double*)grad_charges,
                                                                 (const double*)tail_grad_cell, off, (const double*)cell_mesh,
                                                                 (const double*)cell_pair, (double*)grad_cell);
  else {
    set_error("invalid dtype %d", dtype);
    return MIPME_EINVAL;
  }
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

}  // extern "C"
