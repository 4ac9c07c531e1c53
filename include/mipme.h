/*
 * mipme.h -- C-ABI of libmipme.so, the MI355X (gfx950) native PME / P3M hot path.
 *
 * The reference (lab-cosmo/torch-pme) has no FFI of its own: its boundary is the Python
 * nn.Module API, and every "kernel" is a chain of ATen ops.  This header is the boundary a
 * maintainer would bind (ctypes) to replace those ATen chains.  Each
 * entry point cites the reference code it replaces (paths relative to src/torchpme/).
 *
 * Conventions
 *   - C linkage, plain pointers and sizes, no C++/torch types.  Every function returns 0 on
 *     success or a negative MIPME_E* code; mipme_last_error() gives the thread-local message.
 *     Nothing throws, nothing synchronises the device, nothing allocates device memory except
 *     mipme_fft_plan_create (hipFFT work area).
 *   - All array arguments are DEVICE pointers unless marked "host".  The caller owns every
 *     buffer.  `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued on it in order, so calls are HIP-graph capturable.
 *   - `dtype` selects the real type of every floating array (MIPME_F32: float, MIPME_F64:
 *     double).  Index arrays are int64 (MIPME_I64) or int32 (MIPME_I32).
 *   - Layouts follow the reference: positions (N,3) row-major; charges / potentials (N,C)
 *     row-major; cell (3,3) row-major with ROWS = lattice vectors; mesh (C,nx,ny,nz) row-major
 *     (z fastest); half-complex mesh (C,nx,ny,nz/2+1) interleaved re/im; pairs (P,2).
 */
#ifndef MIPME_H
#define MIPME_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPME_VERSION 408

enum { MIPME_F32 = 0, MIPME_F64 = 1 };
enum { MIPME_I64 = 0, MIPME_I32 = 1 };
enum { MIPME_LAGRANGE = 0, MIPME_P3M = 1 };          /* MeshInterpolator(method=...) */
enum { MIPME_COULOMB = 0, MIPME_INVERSE_POWER_LAW = 1 };

enum {
  MIPME_OK = 0,
  MIPME_EINVAL = -1,   /* bad argument (the Python layer raises ValueError) */
  MIPME_EHIP = -2,     /* HIP runtime error */
  MIPME_EFFT = -3,     /* hipFFT error */
  MIPME_EUNSUPPORTED = -4
};

/* Pair potential 1/r^p with Gaussian range separation: potentials/potential.py:31-57,
 * potentials/coulomb.py:71-78, potentials/inversepowerlaw.py:40-52. */
typedef struct {
  int32_t kind;             /* MIPME_COULOMB (p=1) or MIPME_INVERSE_POWER_LAW */
  int32_t exponent;         /* p in 1..6 (ignored for Coulomb) */
  double smearing;          /* sigma; <= 0 means "None": no range separation, no k-space part */
  double prefactor;
  double exclusion_radius;  /* <= 0 means None */
  int32_t exclusion_degree;
  int32_t _pad;
} mipme_potential_t;

/* Mesh geometry for one cell: what MeshInterpolator.update (lib/mesh_interpolator.py:81-125) and
 * KSpaceFilter._prep_kvectors (lib/kspace_filter.py:199-222) derive from (cell, ns_mesh). */
typedef struct {
  int32_t scheme;      /* MIPME_LAGRANGE (orders 3..7) or MIPME_P3M (orders 1..5) */
  int32_t order;       /* interpolation_nodes */
  int32_t nx, ny, nz;  /* ns_mesh */
  int32_t n_channels;  /* C */
  double cell[9];      /* host copy, row-major, rows = lattice vectors */
  double inv_cell[9];  /* inverse of cell */
  double volume;       /* |det cell| */
} mipme_mesh_t;

typedef struct mipme_fft_plan mipme_fft_plan;

const char* mipme_last_error(void);
int mipme_version(void);

/* ---- reciprocal-space convolution: KSpaceFilter.forward, lib/kspace_filter.py:122-197 -------- */

/* Transform plan for a (batch, nx, ny, nz) real mesh on the current device: own (y,z) plane kernels + x stage for the fused
 * convolution, hipFFT R2C / C2R plans for everything else (3-D plans are created on first use when the own kernels cover the
 * fused path).  Every hipFFT plan is self-tested when created (irfftn(rfftn(x)) == M x): MIPME_EFFT on failure. */
int mipme_fft_plan_create(int dtype, int nx, int ny, int nz, int batch, mipme_fft_plan** out);
int mipme_fft_plan_destroy(mipme_fft_plan* plan);

/* G(k) on the rfft half grid, shape (nx,ny,nz/2+1) reals.
 * PME:  G = v_LR^(k^2)                     KSpaceFilter.update        lib/kspace_filter.py:97-120
 * P3M:  G = v_LR^(k^2) / U^2(k)  (mode 0)  P3MKSpaceFilter.update     lib/kspace_filter.py:293-329,349-361
 * k-grid: generate_kvectors_for_mesh      lib/kvectors.py:24-74;  kernels: potentials/coulomb.py:122-142,
 * potentials/inversepowerlaw.py:109-141, lib/math.py:85-104. */
int mipme_kfilter_build(void* stream, int dtype, const mipme_mesh_t* mesh, const mipme_potential_t* pot, void* G);

/* mesh_out = irfftn(rfftn(mesh_in) * G), both transforms unnormalised.
 * hat_out  (C,nx,ny,nz/2+1) complex: receives rfftn(mesh_in) (kept for the cell gradient);
 * hat_work same shape, scratch;  dc_out (C reals, nullable): Re hat[c,0,0,0] = sum of mesh_in[c]. */
int mipme_convolve(mipme_fft_plan* plan, void* stream, const void* mesh_in, const void* G, void* hat_out, void* hat_work,
                   void* mesh_out, void* dc_out);

/* ---- particle <-> mesh: MeshInterpolator, lib/mesh_interpolator.py:303-457 ------------------- */

/* mesh[c,ix,iy,iz] = sum_i values[i,c] * wx*wy*wz  (compute_weights + points_to_mesh, :303-426).
 * The mesh is zeroed by this call. */
int mipme_spread(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* positions,
                 const void* values, void* mesh_out);

/* out[i,c] = sum_stencil mesh[c,ix,iy,iz] * wx*wy*wz  (compute_weights + mesh_to_points, :428-457). */
int mipme_gather(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* positions,
                 const void* mesh_in, void* out);

/* ---- differentiable primitives (second-order route, calculator.double_backward = "analytic") ----
 * The reference's path is ATen ops, so autograd differentiates it to any order (a loss on forces:
 * calculators/calculator.py:43-87,103-189; tests/calculators/test_workflow.py:164-192 for the first order).  These four
 * linear maps are closed under differentiation -- the backward pass of each is made of the same four -- and take the
 * FRACTIONAL mesh coordinates u = ns * (positions @ inv(cell)) (N,3) (lib/mesh_interpolator.py:326-341) as a tensor, so
 * that the chain to positions and cell belongs to the caller's autograd:
 *   mesh[c,m]  = sum_i values[i,c] D^k W_i(m)      (spread; zeroes the mesh first)
 *   out[i,c]   = sum_m mesh[c,m]   D^k W_i(m)      (gather)
 * with W_i(m) = w(x) w(y) w(z) the interpolation weights of compute_weights (:303-377) and D^k = d^kx/du_x^kx d^ky/du_y^ky
 * d^kz/du_z^kz, every order in 0..3 (piecewise polynomials of degree order-1: higher derivatives vanish). */
int mipme_spread_jet(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* u, const void* values,
                     int kx, int ky, int kz, void* mesh_out);
int mipme_gather_jet(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* u, const void* mesh_in,
                     int kx, int ky, int kz, void* out);
/* out[i,c,d] = the gather of orders (kx,ky,kz) + e_d for d = x, y, z in one walk over the stencil (N,C,3): what the derivative
 * of a spread or a gather with respect to u needs; every order + 1 must stay within 0..3. */
int mipme_gather_jet3(void* stream, int dtype, const mipme_mesh_t* mesh, int64_t n_atoms, const void* u, const void* mesh_in,
                      int kx, int ky, int kz, void* out);
/* out[i,c] = sum_p weights[p] x[j_p,c]  (the index_add_ of _compute_rspace, calculators/calculator.py:70-84, with the bare
 * potentials as an input).  mode 0: half list (both directions), 1: full list (i <- j), 2: full list transposed (j <- i, the
 * adjoint of mode 1).  Zeroes `out` (n_atoms, C) first. */
int mipme_pair_sum(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels, const void* pairs,
                   const void* weights, const void* x, int mode, void* out);
/* The same sum from the transposed list of mipme_topology_build (row_ptr int32[2N+1], entries int32[2P][2]): owner-computes
 * rows, no atomics, fixed summation order.  weights (P) stay indexed by pair. */
int mipme_pair_sum_rows(void* stream, int dtype, int64_t n_atoms, int n_channels, const void* row_ptr, const void* entries,
                        const void* weights, const void* x, int mode, void* out);
/* The distance helper compute_distances (tests/helpers.py:278-304) differentiated twice: the pair difference and its adjoint,
 *   mipme_pair_diff     out[p,c] = x[j_p,c] - x[i_p,c]                                       (positions[j] - positions[i])
 *   mipme_pair_scatter  out[a,c] = sum_{p: j_p = a} values[p,c] - sum_{p: i_p = a} values[p,c]   (the two index_add_ calls)
 * each the other's adjoint.  pair_scatter walks the transposed list when row_ptr / entries (mipme_topology_build) are given
 * (pairs may then be NULL: no atomics, fixed order) and uses atomics on the (P,2) list otherwise; `out` (n_atoms, C) is
 * overwritten either way. */
int mipme_pair_diff(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int n_channels, const void* pairs, const void* x,
                    void* out);
int mipme_pair_scatter(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels, const void* pairs,
                       const void* row_ptr, const void* entries, const void* values, void* out);
/* out[p] = sum_c a[i_p,c] b[j_p,c] (+ a[j_p,c] b[i_p,c] when half != 0): the adjoint of mipme_pair_sum w.r.t. its weights. */
int mipme_pair_dot(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int n_channels, const void* pairs, const void* a,
                   const void* b, int half, void* out);

/* ---- PMECalculator._compute_kspace, calculators/pme.py:88-143 (P3M: calculators/p3m.py:45-84) -- */

/* out_lr[i,c] = 1/2 [ gather(convolve(spread(q)))/V - q*self - 2*bg*Q_c/V ]      (no slab term)
 * Work buffers (caller allocated): rho_mesh, phi_mesh (C,nx,ny,nz); rho_hat, hat_work complex half grids;
 * dc (C).  phi_mesh, rho_hat, dc and out_phi (N,C, nullable: the raw gather/V) are what the backward needs.
 * gather_wait_event (hipEvent_t as void*, nullable) + accumulate_out = 1: the final gather first waits for the event
 * and then ADDS the long-range part to out_lr -- the short-range pair sum can then run concurrently on a second
 * stream, writing out_lr with accumulate = 0 and recording that event (bandwidth-bound pair kernels hide under the
 * latency-bound mesh kernels).
 * out_field (N,3), nullable, needs atom_bins and C == 1: field[a] = (1/V) sum_g phi(g) grad W_a(g), written by the same
 * gather.  If the backward pass is in energy mode (g = gE * charges) the mesh force is gE q_a field[a]
 * (mipme_sr_rows_finalize) and mipme_kspace_backward is not needed.
 * out_records (4N reals, 16-byte aligned), nullable, same conditions: (x, y, z, q) per atom for mipme_sr_rows_fused
 * (records_ready = 1), written by the binning pass while the positions are in registers.
 * The plan owns the per-brick atom counters of the binning pass (zero between calls: the gather, their last consumer,
 * clears them): a plan serves one stream at a time.
 * rho_hat == NULL (allowed when mipme_fft_plan_xfused(plan) != 0, i.e. nx is a power of two): rfftn(rho) is not kept and
 * the convolution runs as (y,z) plane transforms (own LDS kernels for power-of-two ny, nz: one launch per direction when a
 * half-complex plane fits 152 KB of LDS, z rows + y columns as two launches otherwise; 2-D hipFFT plans for other sizes) + one kernel doing x-FFT, * G and the inverse x-FFT.
 * sr_job (nullable; needs atom_bins, out_records, a single channel, job->records == out_records, job->out == out_lr and
 * accumulate_out = 1): the short-range pair sum of the same call -- mipme_sr_rows_fused in its potential + force-sum mode
 * (src = charges, no pair mask, no cell partials; the fields mean what the arguments of that function mean) -- run
 * CO-SCHEDULED with the spread in one launch: the workgroups of the spread (a chain of dependent phases that leaves the
 * vector units mostly idle) and of the VALU-bound pair sum share the CUs.  The pair sum overwrites job->out, the gather
 * then adds the mesh part.  Without a co-scheduled kernel for the request (anything but 1/r or 1/r^6 with the
 * table shift format and force sums) the two kernels run one after the other -- same results. */
typedef struct mipme_sr_job {
  int64_t n_atoms;
  const void* row_ptr;        /* int32[2N+1]     (mipme_topology_build) */
  const void* entries_shift;  /* int32[2P+1][2] (shift_format 1) or int32[2P+1] (shift_format 2)  (mipme_topology_pack_entries) */
  const void* entries;        /* int32[2P+1][2]  (mipme_topology_build) */
  const void* positions;      /* (N,3) reals the pair distances are computed from */
  const void* cell;           /* (3,3) */
  const void* charges;        /* (N) */
  const mipme_potential_t* pot;
  int32_t full_list;
  int32_t shift_format;
  void* records;              /* 4N reals: out_records of the same call */
  void* out;                  /* (N)   potentials, overwritten (= out_lr of the same call) */
  void* force;                /* (N,3) speculative force sums, nullable */
  void* dist_out;             /* (P)   pair distances, nullable (see mipme_sr_rows_fused) */
} mipme_sr_job_t;
/* Arguments of mipme_kspace_forward as ONE versioned struct (a 23-pointer positional call is how bindings drift: a missing
 * trailing argument is undefined behaviour through ctypes, not an error).  The caller sets `size = sizeof(struct)` as IT was
 * compiled and `version = MIPME_ARGS_VERSION`; the library rejects a mismatching version, reads only the first `size` bytes
 * and treats every field beyond them as zero / NULL -- fields are only ever appended.  Field meanings: the comment above. */
#define MIPME_ARGS_VERSION 2
typedef struct mipme_kspace_forward_args {
  uint32_t size;
  uint32_t version;
  mipme_fft_plan* plan;
  void* stream;
  int32_t dtype;
  int32_t accumulate_out;
  const mipme_mesh_t* mesh;
  const mipme_potential_t* pot;
  int64_t n_atoms;
  const void* positions;
  const void* charges;
  const void* G;
  void* rho_mesh;
  void* rho_hat;
  void* hat_work;
  void* phi_mesh;
  void* dc;
  void* out_lr;
  void* out_phi;
  void* atom_bins;
  void* gather_wait_event;
  void* out_field;
  void* out_records;
  const mipme_sr_job_t* sr_job;
  void* out_cell_partials;
  /* Tail of an energy + forces step folded into the gather launch (all three nullable together; needs sr_job with force sums,
   * out_field, no slab term): out_energy (1 real) = sum_a charges_a out_lr_a -- the reduction the caller's (q * V).sum()
   * performs (README.rst:112-114) -- and out_grad_positions (N,3) = s q_a (c force_a + field_a), the gradient of that energy
   * w.r.t. the positions times s = grad_seed[0] (device scalar; NULL = 1), c = 1/2 for a full list: what
   * mipme_dot_forward + mipme_sr_rows_finalize would compute in two more launches.  The energy is assembled without any
   * reduction across the gather's workgroups, from partial sums the earlier kernels of the call leave behind:
   * sum_a q_a V_sr,a and sum_a q_a^2 from the co-scheduled pair sum, (1/2V) sum_k mu_k G_k |rho^_k|^2 from the x stage of
   * the convolution (= sum_a q_a gather(phi)_a / 2V, the gather being the adjoint of the spread). */
  void* out_energy;
  void* out_grad_positions;
  const void* grad_seed;
  /* NaN guard of KSpaceFilter.forward (lib/kspace_filter.py:189-195 raises when the filtered mesh holds a NaN, after a
   * device synchronisation): nan_flag (nullable) points to ONE int32 the gather kernel sets to 1 if a long-range potential it
   * writes is NaN -- device memory, or pinned host memory the caller reads without synchronising (at the next call, ...). */
  void* nan_flag;
  /* The rest of the autograd contract of E = sum_a q_a V_a from the same launches (each nullable; they need the gather tail
   * above, i.e. out_energy / out_grad_positions).  The reference obtains these from one backward pass through its ATen graph
   * (tests/calculators/test_workflow.py:164-192; tuning/tuner.py:350-369 times exactly that):
   *   out_grad_charges (N): s dE/dq_a = 2 s V_a, written by the gather (E is a symmetric bilinear form of the charges; half
   *     lists and the rows of mipme_nl_stream -- a "full" list is only symmetric if the caller made it so, hence refused);
   *   out_grad_cell (27 reals): s dE/dcell at fixed Cartesian positions, [0..8] the mesh part (k-grid sums of dG/dcell, the atoms'
   *     r (x) dE/dr term, the 1/V factors), [9..17] the pair part (through d = |r_j - r_i + S cell|), [18..26] their sum (a
   *     caller whose calculator and distances share ONE cell tensor takes the sum, others route the parts).  Needs G_deriv = mipme_kfilter_build_deriv() of the same mesh and
   *     potential, and cell_work = float64[mipme_cell_tail_work()] of scratch.  How: the co-scheduled pair sum also forms
   *     sum q_a q_o v'/d sh (x) u per wave, the x stage the 12 k-grid sums from the derivative table (+ pre-reduces the pair
   *     sums), the gather sum_a r_a (x) dE/dr_a(mesh) per brick, and ONE more launch (a single workgroup) adds everything up.
   *     Supported: 4-byte entries (shift_format 2), 1/r and 1/r^6 in either precision (fp64 1/r^6: round 5), no dist_out
   *     in the job. */
  void* out_grad_charges;
  void* out_grad_cell;
  const void* G_deriv;
  void* cell_work;
  /* aux_seed (device scalar, nullable = grad_seed): the factor s of out_grad_charges / out_grad_cell, when it is not the one of
   * out_grad_positions -- an MD loop seeds the positions with -1 (forces) and wants dE/dq, dE/dcell themselves */
  const void* aux_seed;
  /* out_rho_hat (nullable, with rho_hat == NULL: the fused convolution): rfftn(rho) (C,nx,ny,nz/2+1 complex), stored by the x
   * stage while it holds the values -- what a later mipme_kspace_backward with a cell gradient takes as `rho_hat` (general
   * upstream gradient: fused convolution with G_deriv; energy mode: the k-grid sums) without a 3-D transform of its own. */
  void* out_rho_hat;
  /* flags (bit set, appended in round 5): MIPME_FWD_RHO_MESH_UNUSED -- the caller never reads rho_mesh after the call (it is a work
   * buffer; a caller that transforms it later, e.g. mipme_fft_r2c for a cell gradient, leaves the bit clear).  With it the spread
   * may skip the real charge mesh altogether: for power-of-two planes that fit a workgroup's LDS the charges are added straight
   * into the tiles of the forward (y,z) transform (plane spread, csrc/bricks.hip), and neither the mesh nor the forward plane
   * launch of the convolution exists. */
  int64_t flags;
  /* frame farm (appended in round 6; see mipme_energy_log_push): with the gather tail above, out_energy is also appended to
   * energy_log (float64[energy_log_capacity], slot = energy_log_cursor[0] mod capacity; the cursor, ONE int32 of device
   * memory, is then incremented) by the thread that writes out_energy -- all three NULL / 0 to switch it off. */
  void* energy_log;
  void* energy_log_cursor;
  int64_t energy_log_capacity;
} mipme_kspace_forward_args_t;
#define MIPME_FWD_RHO_MESH_UNUSED 1
int mipme_kspace_forward(const mipme_kspace_forward_args_t* args);
/* Derivative table of G(k) for out_grad_cell: 4 reals per half-grid point, shape (nx,ny,nz/2+1,4) = {alpha, beta_x, beta_y,
 * beta_z} with dG/dk_c = alpha k_c - beta_c h_c and dG/dh_c = -beta_c k_c (k Cartesian, h_c = |a_c| / n_c; beta = 0 for PME).
 * Computed in double precision like G; rebuild it whenever G is rebuilt (new cell values, potential or mesh). */
int mipme_kfilter_build_deriv(void* stream, int dtype, const mipme_mesh_t* mesh, const mipme_potential_t* pot, void* G_deriv);
/* float64 elements of cell_work for (plan, mesh, n_atoms): k-grid sums and pair sums per x-stage tile, atom sums per brick, pair
 * sums per wavefront of the pair kernel. */
int64_t mipme_cell_tail_work(const mipme_fft_plan* plan, const mipme_mesh_t* mesh, int64_t n_atoms);
/* out_cell_partials (nullable, needs rho_hat == NULL; float64[mipme_cellgrad_partials_size]): the x stage of the fused
 * convolution also forms the 12 k-grid sums of the cell gradient for the energy mode (dL/dG(k) = mu(k) |rho^(k)|^2 up to
 * gE / 2V) while rho^ is in LDS -- mipme_fft_plan_kgrid_blocks(plan) partial sums that mipme_kspace_backward takes as
 * kgrid_blocks_ready, so that neither rfftn(rho) nor the 3-D plans are needed for the stress. */
int64_t mipme_fft_plan_kgrid_blocks(const mipme_fft_plan* plan);
/* hat (C,nx,ny,nz/2+1 complex) = rfftn(mesh_in (C,nx,ny,nz)), un-normalised: for a backward pass with a general upstream
 * gradient and a cell gradient after a forward that kept the charge mesh instead of rho^ (see out_cell_partials). */
int mipme_fft_r2c(mipme_fft_plan* plan, void* stream, int dtype, const mipme_mesh_t* mesh, const void* mesh_in, void* hat);

/* ---- independent frames in one launch (the frames a rank owns) -----------------------------------------
 * Energy + forces of n_frames independent frames (own atoms, cell, pair list; same mesh
 * dimensions, interpolation scheme / order, potential and dtype; single channel) with ONE launch per kernel of the
 * pipeline: blockIdx.y = frame.  The per-frame kernel arguments live in a device-resident table built once per batch
 * (mipme_frames_table_build fills a HOST buffer of mipme_frames_table_bytes bytes; the caller copies it to the device);
 * every pointer in it must stay valid while the table is used.
 *   forward : binning -> spread co-scheduled with the fused distance + pair kernel (potentials, speculative force sums,
 *             distances) -> batched (y,z) plane transforms + x stage (plan: batch = n_frames; G: n_frames filter tables,
 *             G_stride reals apart, 0 = shared) -> gather (+ mesh force field) -> energy[f] = sum_a q_a V_a
 *   backward: grad_positions[f] = grad_scale[f] q_a (c force_a + field_a)   (c = 1/2 for a full list)
 * Requirements (checked; MIPME_EINVAL otherwise): brick kernels support the mesh, <= 1024 bricks, potential 1/r or 1/r^6
 * with a smearing, table shift format, power-of-two nx.  The brick counters of a frame must be zero before its first
 * use; every forward leaves them zero again. */
typedef struct mipme_frame {
  int64_t n_atoms;
  const void* positions;      /* (N,3) */
  const void* charges;        /* (N)   */
  const void* cell;           /* (3,3) device copy of mesh.cell in the working dtype */
  mipme_mesh_t mesh;          /* n_channels = 1 */
  void* atom_bins;            /* mipme_atom_bins_bytes(mesh, N, dtype) */
  void* brick_counters;       /* int32[bricks + 1] (or counter_ints of them, below), zero before the first use (every forward leaves them zero) */
  const void* row_ptr;        /* pair topology, see mipme_sr_rows_fused */
  const void* entries_shift;
  const void* entries;
  int32_t full_list;
  int32_t shift_format;       /* 1 (table, int2 entries) or 2 (table, 4-byte entries); the same for every frame */
  void* records;              /* 4N reals */
  void* rho_mesh;             /* (nx,ny,nz): frame f's slice of the batched mesh buffers */
  void* phi_mesh;
  void* dc;                   /* 1 real: slice of the batched dc buffer */
  void* out;                  /* (N)   potentials */
  void* force;                /* (N,3) */
  void* field;                /* (N,3) */
  void* dist_out;             /* (P) nullable */
  void* energy;               /* 1 real */
  void* grad_positions;       /* (N,3) */
  /* gather tail (see mipme_kspace_forward_args_t.out_energy): with use_tail != 0 (for ALL frames of a batch) the gather of
   * the forward call also forms energy and grad_positions = grad_seed[0] q_a (c force_a + field_a) (grad_seed: device
   * scalar, NULL = 1) -- no energy launch, and mipme_frames_backward is only needed for a different seed. */
  int32_t use_tail;
  /* int32 words the brick_counters buffer holds; 0 = bricks + 1 (the field was padding until round 5).  With
   * mipme_frames_counter_ints(mesh, n_atoms, dtype) words (zero before the first use, left zero by every forward) the frame
   * batch uses the plane spread where it applies: the plane lists' counters live behind the brick counters. */
  int32_t counter_ints;
  const void* grad_seed;
} mipme_frame_t;
int64_t mipme_frames_counter_ints(const mipme_mesh_t* mesh, int64_t n_atoms, int dtype);
int64_t mipme_frames_table_bytes(int dtype, int n_frames);
int mipme_frames_table_build(int dtype, int n_frames, const mipme_frame_t* frames, const mipme_potential_t* pot,
                             void* host_table, int64_t host_table_bytes);
int mipme_frames_forward(mipme_fft_plan* plan, void* stream, int dtype, int n_frames, const mipme_frame_t* frames,
                         const mipme_potential_t* pot, const void* device_table, const void* G, int64_t G_stride,
                         void* rho_mesh_all, void* hat_work_all, void* phi_mesh_all, void* dc_all);
int mipme_frames_backward(void* stream, int dtype, int n_frames, const mipme_frame_t* frames, const void* device_table,
                          const void* grad_scale);

int mipme_fft_plan_xfused(const mipme_fft_plan* plan);

/* Adjoint of mipme_kspace_forward for an upstream gradient g = dL/d(out_lr), shape (N,C).
 * (In the reference this is PyTorch autograd through the ATen chain)
 * grad_positions (N,3), grad_charges (N,C), grad_cell (9) are OVERWRITTEN; each may be NULL.
 * Work: psi_mesh, chi_mesh (C,nx,ny,nz); psi_hat, hat_work complex half grids; dc (C);
 * partials: float64 scratch of >= mipme_cellgrad_partials_size() elements (only used when grad_cell != NULL,
 * which also requires rho_hat, rho_dc, phi_atoms and grad_positions).
 * grad_scale (device scalar, nullable): "energy mode" -- promises grad_out == grad_scale[0] * charges (the gradient of
 * E = sum q V).  Then chi = (grad_scale/2V) phi, so the second spread, both FFTs and the filter are skipped and only the
 * gradient gather runs (needs phi_mesh and rho_dc; the work meshes may be NULL).  With grad_cell != NULL the k-grid sums of
 * the cell gradient are formed from the saved rho_hat alone (dL/dG = (gE/2V) mu |rho^|^2): needs rho_hat, phi_atoms,
 * partials and grad_positions, as in the general case.
 * psi_hat == NULL (only without grad_cell, plans with mipme_fft_plan_xfused): fused convolution as in the forward. */
typedef struct mipme_kspace_backward_args {
  uint32_t size;     /* as mipme_kspace_forward_args_t */
  uint32_t version;
  mipme_fft_plan* plan;
  void* stream;
  int32_t dtype;
  int32_t _pad;
  const mipme_mesh_t* mesh;
  const mipme_potential_t* pot;
  int64_t n_atoms;
  const void* positions;
  const void* charges;
  const void* grad_out;
  const void* G;
  const void* phi_mesh;
  const void* rho_hat;
  const void* rho_dc;
  const void* phi_atoms;
  void* psi_mesh;
  void* psi_hat;
  void* hat_work;
  void* chi_mesh;
  void* dc;
  void* partials;
  void* grad_positions;
  void* grad_charges;
  void* grad_cell;
  void* atom_bins;
  const void* grad_scale;
  const void* mesh_field;
  int64_t kgrid_blocks_ready;
  /* G_deriv (nullable; mipme_kfilter_build_deriv): with it a general upstream gradient WITH a cell gradient runs the fused
   * convolution too (psi_hat == NULL allowed; single channel, plans with mipme_fft_plan_xfused): the x stage contracts psi^ with
   * the saved rho_hat, rider workgroups of the inverse (y,z) launch form the k-grid sums against the table -- no 3-D hipFFT
   * plans, no influence-function derivatives evaluated per k-point. */
  const void* G_deriv;
} mipme_kspace_backward_args_t;
int mipme_kspace_backward(const mipme_kspace_backward_args_t* args);
/* Energy mode extras (grad_scale != NULL): mesh_field (nullable; out_field of the forward call) with grad_positions ==
 * grad_charges == NULL -- no gradient gather, the mesh part of dL/dr is grad_scale q_a field_a (the caller assembles the
 * forces with mipme_sr_rows_finalize; the cell gradient uses the same expression); kgrid_blocks_ready > 0 -- `partials`
 * already holds that many k-grid partial sums (out_cell_partials of the forward call) and rho_hat may be NULL. */
int64_t mipme_cellgrad_partials_size(const mipme_mesh_t* mesh, int64_t n_atoms);

/* atom_bins (nullable): device scratch of mipme_atom_bins_bytes() bytes.  When given, the atoms are binned by 8x8x8 mesh
 * brick in the forward call -- ONE pass: every brick owns a fixed number of slots (4 x the mean occupancy + 8), an atom takes
 * the next free slot of its brick with one wave-aggregated atomic and writes its mesh coordinates and 1-D weights there;
 * atoms that find their brick full go to an overflow region that every consumer also walks -- and the particle<->mesh stages
 * run as brick kernels (owner-computes spread: no global atomics, no mesh memset; LDS-tiled gathers).  The SAME buffer must be
 * handed to the backward call (it reuses the bins and the per-call copy of the brick counts kept inside).
 * mipme_atom_bins_bytes returns 0 when the mesh is too small for bricks (< 17 points on an axis, or a last brick narrower
 * than 4 points): pass NULL then (atomic-scatter kernels). */
int64_t mipme_atom_bins_bytes(const mipme_mesh_t* mesh, int64_t n_atoms, int dtype);
/* Name (without template arguments) of the co-scheduled spread + pair-sum kernel the calling thread launched -- or captured
 * into a graph -- last: "plane_rows_capped_kernel", "plane_rows_kernel", "spread_rows_capped_kernel", "spread_rows_kernel", "sparse_spread_rows_kernel",
 * "live_spread_rows_kernel", "frames_plane_rows_kernel", "frames_spread_rows_kernel"; "" before the first one.  What ran, as
 * opposed to mipme_plane_spread_parts (what the geometry allows): a benchmark labels its dominant launch with this. */
const char* mipme_last_cosched_kernel(void);
/* Workgroups per x plane of the PLANE SPREAD that mipme_kspace_forward uses for this mesh / system when the caller sets
 * MIPME_FWD_RHO_MESH_UNUSED and rho_hat == NULL (0: the owner-computes bricks + the forward plane launch): single channel,
 * power-of-two nx, ny, nz, planes whose accumulation tile fits a workgroup's LDS, dense bricks, not MIPME_DETERMINISTIC.  For
 * callers that account for the launches of a step (bench.py); nothing needs it to call the library. */
int mipme_plane_spread_parts(const mipme_mesh_t* mesh, int64_t n_atoms, int dtype);

/* Per-stage timing for benchmarks: HIP events on the launch stream around every stage of the composite calls.
 * mipme_profile_report writes "stage calls total_ms" lines into buf and returns the byte count needed. */
int mipme_profile_enable(int on);
int64_t mipme_profile_report(char* buf, int64_t buflen);

/* 2-D slab correction, potentials/coulomb.py:6-40 (active when exactly two axes are periodic).
 * forward: pot[i,c] += 1/2 * prefactor * E_slab[i,c].  moments: float64 scratch of 6*C elements.
 * backward: ACCUMULATES into grad_positions / grad_charges / grad_cell (each nullable). */
int mipme_slab_forward(void* stream, int dtype, int axis, const mipme_mesh_t* mesh, double prefactor, int64_t n_atoms,
                       const void* positions, const void* charges, void* moments, void* pot);
int mipme_slab_backward(void* stream, int dtype, int axis, const mipme_mesh_t* mesh, double prefactor, int64_t n_atoms,
                        const void* positions, const void* charges, const void* grad_out, void* moments,
                        void* grad_positions, void* grad_charges, void* grad_cell);

/* ---- Calculator._compute_rspace, calculators/calculator.py:43-87 ------------------------------ */

/* pot[i,c] (+)= 1/2 sum_pairs q[j,c] v_SR(d)   (+ the (j,i) direction for a half list, :82-84)
 * v_SR: Potential.sr_from_dist / from_dist / f_cutoff, potentials/potential.py:59-138.
 * pair_mask: nullable uint8/bool (P).  accumulate = 0 zeroes pot first. */
int mipme_rspace_forward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels,
                         const void* pairs, const void* dist, const void* charges, const void* pair_mask,
                         int full_list, const mipme_potential_t* pot, int accumulate, void* out_pot);

/* grad_dist[p] = 1/2 v_SR'(d_p) sum_c (g[i,c] q[j,c] + g[j,c] q[i,c])   (overwritten, nullable)
 * grad_charges (N,C): ACCUMULATED atomically (nullable).
 * grad_scale (device scalar, nullable): energy mode, grad_out == grad_scale[0] * charges (see mipme_kspace_backward);
 * the kernel then needs no gathers of grad_out. */
int mipme_rspace_backward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms, int n_channels,
                          const void* pairs, const void* dist, const void* charges, const void* pair_mask,
                          int full_list, const mipme_potential_t* pot, const void* grad_out, const void* grad_scale,
                          void* grad_dist, void* grad_charges);

/* ---- caller side: the energy reduction E = sum_ic q_ic V_ic (README.rst:112-114, tests/calculators/test_values_ewald.py:306)
 * as one kernel, and its adjoint grad_a = g*b, grad_b = g*a (g: device scalar; grad_a / grad_b nullable).
 * scratch: 65 float64 of PERSISTENT device memory, zero-initialised once by the caller and used by one stream at a time:
 * 64 block sums + a ticket counter that the kernel leaves at zero (the block drawing the last ticket adds the block sums
 * in index order, so the result is deterministic). ---- */
int mipme_dot_forward(void* stream, int dtype, int64_t n, const void* a, const void* b, void* scratch /* 65 float64 */,
                      void* out);
int mipme_dot_backward(void* stream, int dtype, int64_t n, const void* grad, const void* a, const void* b, void* grad_a,
                       void* grad_b);

/* ---- frame farm (the reference loops over frames on the host, calculators/pme.py:102-105): the energy log.
 * A rank evaluates batch after batch of independent frames and keeps every batch's frame energies in a device-resident log
 * that is exchanged ONCE (one all-gather of the whole log) -- no collective between two evaluations.  One tiny launch, meant to
 * be captured as the last node of a step's HIP graph: log[(cursor[f] mod capacity) * n + f] = (double) src[f], cursor[f] += 1
 * for f < n.  src: n reals of `dtype` (the step's energy outputs); log: float64[capacity * n]; cursor: int32[n] of device
 * memory, one counter per frame (all equal: the number of pushes; the caller zeroes them to start a new log -- a counter per
 * frame so that the writers of a batch, one workgroup per frame, share nothing).  Slots wrap around: the log holds the last
 * `capacity` pushes.
 * The steps that form their energy in the gather launch append to the log THERE, at no cost (no extra launch):
 * mipme_kspace_forward_args_t.energy_log / mipme_md_args_t.energy_log (n = 1) and mipme_frames_table_energy_log (n =
 * n_frames: patches the host table between mipme_frames_table_build and the upload; log == NULL switches it off); this entry
 * point serves every other way of producing energies. */
int mipme_energy_log_push(void* stream, int dtype, int n, const void* src, void* log, void* cursor, int capacity);
int mipme_frames_table_energy_log(int dtype, int n_frames, void* host_table, int64_t host_table_bytes, void* log, void* cursors,
                                  int capacity);

/* Energy-mode detection for callers that reduce with plain tensor ops, E = (charges * V).sum() (README.rst:112-114): the
 * gradient arriving at the calculator's backward is then gE * charges.  result[0] = s = g[k] / q[k] at the k of the largest
 * |q|; result[1] = 1 if |g[i] - s q[i]| <= 8 eps |s q[i]| for every i, else 0 (2 reals of `dtype`, device memory).  When it
 * matches, result (the device scalar s) can be passed as grad_scale of the backward entry points.  One workgroup.
 * host_flag (nullable): ONE int32 of pinned host memory that also receives the verdict (0 / 1, system-scope store) -- the
 * caller presets it to -1 and polls it instead of copying `result` back. */
int mipme_scaled_match(void* stream, int dtype, int64_t n, const void* g, const void* q, void* result, void* host_flag);
/* The same for MANY values (one workgroup takes 87 us for 262 144 values and 630 us for a million): blocks of 32 768 values each
 * find their own reference element and check their own values, a second small launch compares the blocks' scales (within 4 ulp
 * of the one that belongs to the largest |q|).  work: float64[mipme_scaled_match_work(n)] of device scratch, 0 elements (NULL)
 * for n <= 32 768, where this is mipme_scaled_match. */
int64_t mipme_scaled_match_work(int64_t n);
int mipme_scaled_match_wide(void* stream, int dtype, int64_t n, const void* g, const void* q, void* result, void* host_flag,
                            void* work);
/* host_flag[0] = 1 if a[i] == b[i] for all i < n (bitwise equal reals of `dtype`, n <= 1024), else 0; host_flag: ONE int32 of
 * pinned host memory the caller presets to -1 and polls.  For callers that receive a NEW cell tensor every call
 * (tuning/tuner.py:350-352 clones its inputs; data loaders do the same): the library's host layer then launches the step with
 * the mesh geometry and filter it cached for the previous cell, this kernel FIRST in the queue, and looks at the verdict after
 * its last launch -- the 9-value device-to-host copy the reference pays before it can size the mesh
 * (lib/kvectors.py:17-21) would make the host wait for everything queued before it, every call. */
int mipme_values_equal(void* stream, int dtype, int64_t n, const void* a, const void* b, void* host_flag);

/* 128-bit order-sensitive checksum of a device buffer (n_bytes a multiple of 4; two position-keyed sums of its 32-bit words):
 * sums[0..1] += the hash.  sums: uint64[mipme_checksum_words()] of device memory, the first three ZERO on entry (sums[2] is the
 * kernel's ticket counter, zero again on exit; the rest per-workgroup partial sums).  expect (nullable, device uint64[2]) + host_flag
 * (pinned int32 preset to -1): the last workgroup also writes 1 / 0 -- equal / different -- to host_flag.  How the host layer
 * recognises a NEW neighbour-list tensor that holds the values of the previous one (the reference's users hand a fresh list to
 * every call, examples/02-neighbor-lists-usage.py:97-164; the per-list structures -- transposition, entry streams -- are then
 * reused on the bet that the checksums agree, verified before any result is handed out). */
int64_t mipme_checksum_words(void);
int mipme_checksum(void* stream, const void* data, int64_t n_bytes, void* sums, const void* expect, void* host_flag);

/* The same decision WITHOUT a host round trip (the poll above makes the host wait for everything queued before it: the eager
 * reference call sequence then runs GPU and host one after the other).  The caller launches mipme_scaled_match with a DEVICE
 * int32 as `host_flag`, announces it with mipme_set_skip_flag(flag) -- kernels launched by THIS THREAD through
 * mipme_kspace_backward (general path) and mipme_sr_rows_fused until mipme_set_skip_flag(NULL) read it first and return at once
 * if it is 1; best effort: kernels without the check (hipFFT plans, atomic mesh kernels) just do their work -- runs the general
 * backward, and finally mipme_energy_select: if the verdict (`result` of mipme_scaled_match) is a match it overwrites
 *   grad_mesh[a] = s q_a field_a            (nullable; field = out_field of the forward)
 *   grad_pair[a] = s q_a f force_a          (nullable; force = pair force sums of the forward; f = 1/2 for a full list)
 * and otherwise leaves the general path's results in place. */
int mipme_set_skip_flag(const void* device_flag);
int mipme_energy_select(void* stream, int dtype, int64_t n_atoms, const void* verdict, const void* charges, const void* force,
                        const void* field, int full_list, void* grad_mesh, void* grad_pair);
/* The same for a caller that wants one gradient: out[a] = match ? s q_a (f force_a + field_a) : grad_mesh[a] + grad_pair[a]
 * (all (N,3); out may alias grad_mesh or grad_pair). */
int mipme_energy_select_sum(void* stream, int dtype, int64_t n_atoms, const void* verdict, const void* charges, const void* force,
                            const void* field, int full_list, const void* grad_mesh, const void* grad_pair, void* out);
/* ... and for the rest of the autograd contract (Calculator.forward's gradients w.r.t. charges and cell,
 * tests/calculators/test_workflow.py:164-192) when the forward's gather tail holds them for the energy mode (out_grad_charges,
 * out_grad_cell of mipme_kspace_forward, per unit seed):
 *   grad_charges[a] = match ? 1/2 s tail_grad_charges[a] : grad_charges[a]            (N; in: the general adjoint's; nullable)
 *   grad_cell[i]    = match ? s tail_grad_cell[(pair_through_distances ? 0 : 18) + i]
 *                           : cell_mesh[i] + cell_pair[i]                             (9; cell_pair nullable; all nullable)
 * 1/2: the tail holds the TOTAL dE/dq = 2 V of E = sum q V, of which the calculator's node owns the half that comes from V's
 * dependence on the charges.  pair_through_distances != 0: the pair part of dE/dcell flows through neighbor_distances (its
 * gradient is observed), so only the mesh part (tail slots 0..8) belongs to the calculator's own cell argument. */
int mipme_energy_select_contract(void* stream, int dtype, int64_t n_atoms, const void* verdict, const void* tail_grad_charges,
                                 void* grad_charges, const void* tail_grad_cell, int pair_through_distances,
                                 const void* cell_mesh, const void* cell_pair, void* grad_cell);

/* ---- caller side: pair distances, tests/helpers.py:278-304 ------------------------------------ */

/* d[p] = | r[j] - r[i] + shifts[p] @ cell |.  cell: DEVICE (9 reals); shifts (P,3) reals (nullable = 0). */
int mipme_pair_distance_forward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, const void* pairs,
                                const void* positions, const void* cell, const void* shifts, void* out_dist);
/* Compressed pair stream for the same op (16 instead of 28 bytes per pair in fp32): pairs32 int32 (P,2), packed_shifts
 * int32 (P) = 3 x int8 from mipme_pack_pair_shifts (flag[0] != 0 -> some shift is not an integer in [-127,127]). */
int mipme_pack_pair_shifts(void* stream, int dtype, int64_t n_pairs, const void* shifts, void* packed, void* flag);
int mipme_pair_distance_forward_packed(void* stream, int dtype, int64_t n_pairs, const void* pairs32,
                                       const void* packed_shifts, const void* positions, const void* cell,
                                       void* out_dist);
/* grad_positions (N,3): zeroed then accumulated.  grad_cell (9, nullable): overwritten; partials: float64 scratch
 * of >= mipme_pair_partials_size(n_pairs) elements, required when grad_cell != NULL. */
int mipme_pair_distance_backward(void* stream, int dtype, int idx_dtype, int64_t n_pairs, int64_t n_atoms,
                                 const void* pairs, const void* positions, const void* cell, const void* shifts,
                                 const void* grad_dist, void* partials, void* grad_positions, void* grad_cell);
int64_t mipme_pair_partials_size(int64_t n_pairs);

/* ---- pair-list topology (build-side helper, no reference counterpart) ---------------------------
 * Scattered float atomics are the slowest thing an MI355X does (~21 G/s); the reference's index_add_
 * (calculators/calculator.py:78-84) and the scatter in the distance gradient are therefore restated as
 * owner-computes row sums over a transposed pair list, built once per neighbour list:
 *   row_ptr  int32[2N+1]: entries row_ptr[2a]..row_ptr[2a+1] have atom a as FIRST index (role i),
 *                         row_ptr[2a+1]..row_ptr[2a+2] as SECOND index (role j); pair index ascending.
 *   entries  int32[2P][2]: { other atom, pair index }.  Allocate the entry tables (entries, packed shifts, entries_shift)
 *                         with ONE extra element: the row kernels prefetch the first entry of a row before testing
 *                         whether the row is empty, which for an empty last row is index 2P (the value is never used).
 * workspace: >= mipme_topology_workspace_bytes(P) bytes of device scratch. */
int64_t mipme_topology_workspace_bytes(int64_t n_pairs);
int mipme_topology_build(void* stream, int idx_dtype, int64_t n_pairs, int64_t n_atoms, const void* pairs,
                         void* workspace, int64_t workspace_bytes, void* row_ptr, void* entries);
/* packed[e] = 3 x int8 cell shift of entry e's pair; flag[0] != 0 if some shift is not an integer in [-127,127]. */
int mipme_topology_pack_shifts(void* stream, int dtype, int64_t n_pairs, const void* entries, const void* shifts,
                               void* packed, void* flag);

/* Row form of mipme_rspace_forward (transpose = 0, src = charges) and of the charge-gradient part of
 * mipme_rspace_backward (transpose = 1, src = upstream gradient):
 *   out[a,c] (+)= 1/2 sum_{entries of a} src[other,c] v_SR(dist[p]).   accumulate = 0 overwrites. */
int mipme_rspace_rows(void* stream, int dtype, int64_t n_atoms, int n_channels, const void* row_ptr, const void* entries,
                      const void* dist, const void* src, const void* pair_mask, int transpose, int full_list,
                      const mipme_potential_t* pot, int accumulate, void* out_pot);

/* CONSTANT distances -- a charge loop over a fixed geometry, and the reference tuner's timing protocol, which calls the calculator
 * with the same neighbor_distances tensor over and over (tuning/tuner.py:337-373): v_SR(dist[p]) is formed ONCE per row entry,
 *   values: mipme_rspace_rows_value_bytes() bytes, one {int32 partner, real v_SR(d)} record per row entry in row order (8 bytes
 *           fp32, 16 bytes fp64; zero for a masked pair),
 *   row_sum_transposed (N, nullable) = 1/2 sum_{roles of the transposed sum} v: the charge gradient of the pair part per unit
 *           of a UNIFORM upstream gradient (result.sum().backward()),
 * and mipme_rspace_rows_tabulated is mipme_rspace_rows (single channel) over those records: a sparse matrix-vector product that
 * streams them and gathers only src[partner] -- no distance gather, no erfc (39.8 -> ~22 us at 4.76 M pairs). */
int64_t mipme_rspace_rows_value_bytes(int dtype, int64_t n_pairs);
int mipme_rspace_rows_tabulate(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* entries, const void* dist,
                               const void* pair_mask, int full_list, const mipme_potential_t* pot, void* values,
                               void* row_sum_transposed);
int mipme_rspace_rows_tabulated(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* values, const void* src,
                                int transpose, int full_list, int accumulate, void* out);

/* Row form of mipme_pair_distance_backward: grad_positions (N,3) OVERWRITTEN.  packed_shifts (from
 * mipme_topology_pack_shifts) or shifts (P,3 reals) supply the cell shifts; partials: float64 scratch of
 * >= mipme_rows_partials_size(N) elements when grad_cell != NULL. */
int mipme_pair_distance_backward_rows(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* entries,
                                      const void* packed_shifts, const void* positions, const void* cell,
                                      const void* shifts, const void* grad_dist, void* partials, void* grad_positions,
                                      void* grad_cell);
int64_t mipme_rows_partials_size(int64_t n_atoms);

/* ---- fused distance + pair-sum row kernels (single channel) --------------------------------------
 * For callers whose distances come from mipme_pair_distance_forward(positions, cell, shifts): the row kernels recompute
 * d_e = |r_other - r_a + S A| from the L2-resident positions instead of gathering dist[p] / grad_dist[p] from P-sized
 * arrays, and apply the chain rule through d in the same pass.  Fuses Calculator._compute_rspace
 * (calculators/calculator.py:43-87) with the caller's compute_distances (tests/helpers.py:278-304) and their adjoints.
 *   entries_shift int32[2P][2] = { other atom, cell-shift code } from mipme_topology_pack_entries (shifts == NULL -> zero
 *                 shifts).  The shift is stored role-adjusted (S for role i, -S for role j).  shift_format 0: 3 x int8;
 *                 1: index into a 7^3 table of Cartesian shift vectors kept in LDS (needs |s| <= 3, no pair mask).
 *                 2: the table code and the partner in ONE int32 per entry, other | code << 22 (int32[2P]; atoms < 2^22,
 *                 |s| <= 3) -- half the entry stream; read by the co-scheduled pair sum of mipme_kspace_forward (sr_job) and
 *                 by mipme_frames_forward only, mipme_sr_rows_fused takes formats 0 and 1.
 *                 flag[0] bit 0: some shift is not an integer in [-127,127] (unusable); bit 1: some |s| > 3 (formats 1, 2
 *                 unusable).
 *   out   (N) nullable: out[a] (+)= 1/2 sum_{potential roles} src[o] v_SR(d_e)          (transpose as mipme_rspace_rows)
 *   force (N,3) nullable, OVERWRITTEN: sum_e sign_e w_e v_SR'(d_e) vec_e / d_e with
 *         w_e = charges[o] when grad_out == NULL (finish with mipme_sr_rows_finalize: energy mode, g = gE * charges),
 *         otherwise the general weights 1/2 (g[a] q[o] + g[o] q[a]) (half list) -- force is then d L / d positions.
 *   records: device scratch of 4 * N reals (16-byte aligned); holds (x, y, z, src) per atom so that one gather per entry
 *         fetches the partner atom; records_ready != 0: already filled with (positions, src) -- e.g. by
 *         mipme_kspace_forward(out_records) -- and not repacked.
 *   partials nullable: float64[mipme_rows_partials_size(N)] per-block sums of the cell gradient; with grad_out != NULL
 *         and grad_cell != NULL they are reduced into grad_cell (3,3).
 *   dist_out (P) nullable, potential passes only (out != NULL, no pair mask, transpose == 0): the pair distances d_p,
 *         written as a by-product by the row that owns the pair's FIRST atom -- the tensor the caller's compute_distances
 *         (tests/helpers.py:278-304) would have produced, without a separate pass over the list.  Requires a pair list
 *         ordered by its first index (pairs[p][0] non-decreasing), as neighbour-list builders emit it: the role-i entries
 *         of a row are then consecutive pairs. */
int mipme_topology_pack_entries(void* stream, int dtype, int64_t n_pairs, int64_t n_atoms, const void* row_ptr,
                                const void* entries, const void* shifts, int shift_format, void* entries_shift,
                                void* flag);
int mipme_sr_rows_fused(void* stream, int dtype, int64_t n_atoms, const void* row_ptr, const void* entries_shift,
                        const void* entries, const void* pair_mask, const void* positions, const void* cell,
                        const void* charges, const void* src, const void* grad_out, int transpose, int full_list,
                        const mipme_potential_t* pot, int accumulate, int shift_format, void* records,
                        int records_ready, void* out, void* force, void* partials, void* grad_cell, void* dist_out);
/* grad_positions[a] = gE charges[a] (f force[a] + field[a]); grad_cell = f gE sum(partials); f = 1/2 for a full list,
 * gE = grad_scale[0].  force: from mipme_sr_rows_fused (nullable); field: out_field of mipme_kspace_forward (nullable). */
int mipme_sr_rows_finalize(void* stream, int dtype, int64_t n_atoms, const void* force, const void* field,
                           const void* charges, const void* grad_scale, int full_list, const void* partials,
                           void* grad_positions, void* grad_cell);

/* ---- explicit Ewald sum: EwaldCalculator._compute_kspace, calculators/ewald.py:76-142 -------
 * k-vectors (K,3) are caller supplied (lib/kvectors.py:105-166 builds them from the cell; the host layer does that with
 * differentiable tensor ops so that the cell gradient flows through them).  The (K,N) phase tables of the reference are
 * never materialised.  All arrays in `dtype`; s_* / t_* are (K,C), G / dG (K).
 *   mipme_ewald_filter     G[k] = v_LR^(|k|^2), dG[k] = dG/d(|k|^2) (nullable)         (Potential.lr_from_k_sq)
 *   mipme_ewald_structure  out_cos[k,c] = sum_i w[i,c] cos(k r_i), out_sin likewise    (w = charges, or the upstream gradient)
 *   mipme_ewald_potential  out[i,c] = sum_k G[k] (cos(k r_i) s_cos[k,c] + sin(k r_i) s_sin[k,c])   (no 1/V; OVERWRITES)
 *   mipme_ewald_backward   with S = structure(charges), T = structure(grad_out):
 *        grad_positions[i] = sum_k G[k] k B(i,k),  grad_kvectors[k] = G[k] sum_i r_i B(i,k) + 2 dG[k] k sum_c (T.S),
 *        B(i,k) = sum_c [ g_ic (cos S_sin - sin S_cos) + q_ic (cos T_sin - sin T_cos) ]   (either output nullable)
 *   (the charge gradient is mipme_ewald_potential with T in place of S) */
int mipme_ewald_filter(void* stream, int dtype, const mipme_potential_t* pot, int64_t n_k, const void* kvectors, void* G,
                       void* dG);
/* n_batch >= 1: padded batches (reference tests/calculators/test_padding.py: torch.vmap over zero-padded structures with
 * node_mask / pair_mask / batched k-vectors) in ONE launch per kernel, blockIdx.y = structure: every array gains a leading
 * batch dimension -- positions (B,N,3), weights / charges / grad_out / out (B,N,C), kvectors (B,K,3), G / dG (B,K), structure
 * factors (B,K,C) -- with n_atoms = N and n_k = K the padded sizes.  Padding atoms carry zero weight, padding k-vectors are
 * zero (G(0) = 0), so they drop out of every sum; mipme_ewald_filter takes the flat (B K, 3) k-vectors as it is. */
int mipme_ewald_structure(void* stream, int dtype, int64_t n_atoms, int n_channels, int64_t n_k, const void* positions,
                          const void* weights, const void* kvectors, void* out_cos, void* out_sin, int64_t n_batch);
int mipme_ewald_potential(void* stream, int dtype, int64_t n_atoms, int n_channels, int64_t n_k, const void* positions,
                          const void* kvectors, const void* G, const void* s_cos, const void* s_sin, void* out,
                          int64_t n_batch);
int mipme_ewald_backward(void* stream, int dtype, int64_t n_atoms, int n_channels, int64_t n_k, const void* positions,
                         const void* charges, const void* grad_out, const void* kvectors, const void* G, const void* dG,
                         const void* s_cos, const void* s_sin, const void* t_cos, const void* t_sin,
                         void* grad_positions, void* grad_kvectors, int64_t n_batch);

/* ---- device neighbour list (the reference uses third-party vesin on the host,
 * tests/helpers.py:240-275, and hands a fresh list to every call, examples/02-neighbor-lists-usage.py:97-164) -------------
 * One cell-list traversal, two products:
 *   (a) the reference's quantities: pairs (P,2) int64, integer cell shifts (P,3) as reals, distances (P), strict d < cutoff,
 *       half or full list:  mipme_nl_bin -> mipme_nl_count -> (caller: exclusive scan of counts into int64 offsets[N+1],
 *       allocate P = offsets[N]) -> mipme_nl_fill;
 *   (b) the ROW STREAM the fused pair kernels read, directly:  mipme_nl_bin -> mipme_nl_stream.  For every atom a row of fixed
 *       capacity holding the 4-byte words  other | shift code << 22  (shift_format 2) of ALL its neighbours; pass the
 *       row_ptr / words pair to mipme_sr_rows_fused / mipme_sr_job_t with shift_format = 2 | MIPME_ROWS_PADDED.  Every buffer
 *       keeps its address and nothing is read back by the host, so a captured HIP graph of these two calls refreshes the
 *       list of a captured energy + forces step in place.
 * Any cell, box size and cutoff: n_cells[d] cells along axis d (the caller picks them at least cutoff / 2 wide where the box
 * allows it), reach[d] = ceil(cutoff / cell width) cells are walked to either side, and the walk wraps with the image shift
 * it crosses (a box smaller than the cutoff meets the same cell several times with different shifts).  A non-periodic axis
 * (periodic[d] = 0: no images, shift 0) bins the fractional coordinate (f_d - frac_offset[d]) * frac_scale[d], which the
 * caller chooses so that the atoms fall in [0, 1); atoms beyond join the edge cells.
 * workspace: mipme_nl_workspace_bytes() bytes of device memory owned by the list, ZERO-INITIALISED ONCE by the caller (the
 * kernels leave their counters at zero); it carries the binned atoms from mipme_nl_bin to the walks. */
typedef struct {
  double cell[9];      /* row-major, rows = lattice vectors */
  double inv_cell[9];
  int32_t n_cells[3];
  int32_t periodic[3];
  double cutoff;
  int32_t full_list;   /* mipme_nl_count / mipme_nl_fill only; the stream always holds every neighbour of every atom */
  int32_t _pad;
  double frac_offset[3]; /* non-periodic axes only; 0 / 1 for periodic axes */
  double frac_scale[3];
  int32_t reach[3];    /* cells walked to either side along each axis */
  int32_t position_stride; /* reals per atom in `positions`: 0 or 3 = (N,3); 4 = (N,4) records x, y, z, charge (mipme_md_step) */
} mipme_nl_t;
#define MIPME_ROWS_PADDED 0x100 /* OR-ed into shift_format: row_ptr is int32[3N+1] = {begin, end, end} per atom + buffer size */
int64_t mipme_nl_workspace_bytes(const mipme_nl_t* nl, int64_t n_atoms);
int mipme_nl_bin(void* stream, int dtype, const mipme_nl_t* nl, int64_t n_atoms, const void* positions, void* workspace);
int mipme_nl_count(void* stream, int dtype, const mipme_nl_t* nl, int64_t n_atoms, void* workspace,
                   void* counts /* int32[N] */);
int mipme_nl_fill(void* stream, int dtype, const mipme_nl_t* nl, int64_t n_atoms, void* workspace,
                  const void* offsets /* int64[N+1] */, void* pairs, void* shifts, void* dist /* nullable */);
/* row_ptr int32[3N+1], words int32[N * row_capacity + 1], ZERO-FILLED ONCE by the caller (the pair kernels prefetch one step
 * past the end of a row and give what they find zero weight: it must decode to a valid atom and shift code, which zero and
 * any stale entry do).  host_status (nullable): int32[4] of PINNED host memory that
 * receives, when the call has run, {longest row, flags, refresh counter}: flags bit 0 = a row exceeded row_capacity (its
 * surplus entries were dropped: enlarge and rebuild), bit 1 = a cell shift beyond the pair kernels' table (|S| > 3: wrap the
 * positions or use (a)), bit 2 = an atom more than 400 cells outside the unit cell.  The counter (written last, release order)
 * increases by one per call. */
int mipme_nl_stream(void* stream, int dtype, const mipme_nl_t* nl, int64_t n_atoms, void* workspace, int64_t row_capacity,
                    void* row_ptr, void* words, void* host_status);

/* ---- energy + forces step of an MD-like loop on device-resident neighbour structures (no reference counterpart: the
 * reference rebuilds everything from a fresh list every call, examples/02-neighbor-lists-usage.py:97-164) ------------------
 * Between two refreshes of the neighbour list the atoms move by a fraction of a mesh spacing.  mipme_md_rebin does, once per
 * refresh, what mipme_kspace_forward does every call: it bins the atoms by mesh brick, and it writes per brick the list of
 * atoms whose stencil can reach the brick while they stay within one mesh point of where they were binned.  mipme_md_step then
 * evaluates E = sum q V and seed * dE/dpositions for the CURRENT records with five launches -- spread from the lists + the pair
 * sum over the rows of mipme_nl_stream (co-scheduled), (y,z) transforms, x stage * G, (y,z) transforms, gather + energy +
 * forces -- with every weight evaluated on the fly from the current positions; only the atom -> brick bookkeeping is reused.
 * An atom that has moved more than one mesh point since the rebin sets bit 1 of host_flags (the step's results are invalid:
 * rebin sooner) AND makes the step write NaN as its energy, so that the offending step says so in what it returns; bit 0: a
 * brick list overflowed at the rebin.  P3M / PME with 1/r or 1/r^6, one channel, meshes the brick
 * kernels cover (mipme_md_supported). */
typedef struct {
  uint32_t size, version;   /* sizeof(mipme_md_args_t), 1 */
  struct mipme_fft_plan* plan;
  void* stream;
  int32_t dtype, shift_format; /* shift_format of the rows: 2 | MIPME_ROWS_PADDED for mipme_nl_stream's */
  const mipme_mesh_t* mesh;
  const mipme_potential_t* pot;
  int64_t n_atoms;
  const void* records;      /* (N,4) reals: x, y, z, charge -- the atoms' storage, read by every kernel of the step */
  const void* cell;         /* DEVICE, 9 reals */
  const void* G;            /* mipme_kfilter_build */
  void* rho_mesh;           /* work: (nx,ny,nz) */
  void* hat_work;           /* work: (nx,ny,nz/2+1) complex */
  void* phi_mesh;           /* work: (nx,ny,nz) */
  void* dc;                 /* work: 1 real */
  void* atom_bins;          /* mipme_atom_bins_bytes(); written by mipme_md_rebin, read by the steps */
  void* live_lists;         /* 4 * mipme_md_lists_ints() bytes, ZERO-FILLED ONCE by the caller; likewise */
  const void* row_ptr;      /* neighbour rows */
  const void* words;
  void* potentials;         /* out (N) */
  void* pair_force;         /* work (N,3) */
  void* energy;             /* out: 1 real */
  void* grad_positions;     /* out (N,3): grad_seed[0] * dE/dpositions */
  const void* grad_seed;    /* device scalar, nullable (= 1) */
  void* nan_flag;           /* pinned int32, nullable (see mipme_kspace_forward) */
  void* host_flags;         /* pinned int32, nullable: bit 0 list overflow (rebin), bit 1 an atom moved beyond the margin (step) */
  /* appended in version 400 (size-checked like every field): the rest of the autograd contract, as in
   * mipme_kspace_forward_args_t -- grad_charges (N) = 2 s V; grad_cell (27 reals: mesh part, pair part, sum) with G_deriv and
   * cell_work = float64[mipme_cell_tail_work()]; all nullable */
  void* grad_charges;
  void* grad_cell;
  const void* G_deriv;
  void* cell_work;
  const void* aux_seed;     /* device scalar, nullable (= grad_seed): the factor of grad_charges / grad_cell */
  /* appended in version 408: the energy log of the frame farm as in the arguments of mipme_kspace_forward; all NULL / 0: off */
  void* energy_log;
  void* energy_log_cursor;
  int64_t energy_log_capacity;
} mipme_md_args_t;
int mipme_md_supported(const mipme_mesh_t* mesh, const mipme_potential_t* pot, int64_t n_atoms, int dtype);
int64_t mipme_md_lists_ints(const mipme_mesh_t* mesh, int64_t n_atoms);
int mipme_md_rebin(const mipme_md_args_t* args);
int mipme_md_step(const mipme_md_args_t* args);

#ifdef __cplusplus
}
#endif
#endif /* MIPME_H */
