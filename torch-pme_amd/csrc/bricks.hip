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
// (device code: bricks_device.h; this file holds the host wrappers of the single-frame path)
#define MIPME_BRICKS_MAIN_TU
#include "bricks_device.h"

namespace mipme {
// bands of rows a plane of this mesh is spread in (1: whole planes; 0: no plane spread) -- api.hip mipme_plane_spread_parts
int plane_bands(const mipme_mesh_t* m, int dtype) {
  const int rows = plane_band_rows(m, dtype);
  return rows > 0 ? m->ny / rows : 0;
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
  // whole planes, or bands of rows for planes whose tile does not fit the launch's LDS (PlaneArgs::band_rows); the banded
  // co-scheduled kernels exist for 4-byte pair entries only (what every caller of this package uses): others keep the bricks
  const int band_rows = v.idx.pcap > 0 ? plane_band_rows(m, sizeof(T) == 4 ? MIPME_F32 : MIPME_F64) : 0;
  const bool bands_ok = band_rows == m->ny || !job || (job->shift_format & kShiftFormatMask) == kShiftTable32;
  if (ph && ph->hat && ph->slot_values && !ph->keep_mesh && used_planes && clear_count && v.idx.pcap > 0 && sa.C == 1 && !sparse &&
      !sa.det && N > 0 && band_rows > 0 && bands_ok) {
    size_t need = 0;
    pa.band_rows = band_rows;
    pa.bands = m->ny / pa.band_rows;
    plane_lds_layout<T>(pa.band_rows, m->nz, pa, need, pa.bands == 1);
    pa.hat = (Cplx<T>*)ph->hat;
    pa.parts = (pa.bands == 1 && ph->parts > 1 && ph->hat_more) ? ph->parts : 1;
    ph->ycols_pending = pa.bands > 1;
    ph->parts_used = pa.parts;
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
      const unsigned n_planes = unsigned(m->nx) * unsigned(pa.bands) * unsigned(pa.parts);
      const bool compact_p = (job->shift_format & kShiftFormatMask) == kShiftTable32;
      const unsigned n_here = n_rows_blocks;
      const unsigned pgrid = pad8(n_planes) + pad8(n_here);
#define MIPME_PLANE_ROWS(PF, CO, CE) \
  MIPME_DISPATCH_STENCIL_B(m->scheme, m->order, ([&] {                                                                          \
    /* fp32: the build held to 80 scalar registers; planes spread in bands of rows: an instantiation of its own (4-byte entries) */ \
    if constexpr (sizeof(T) == 4) {                                                                                            \
      if (pa.bands > 1) {                                                                                                      \
        if constexpr (CO)                                                                                                      \
          plane_rows_capped_kernel<S, N, T, PF, CO, CE, true><<<pgrid, SPREAD_THREADS, plane_lds, st>>>(sa, pa, ra_e, n_planes, n_here); \
      } else {                                                                                                                 \
        plane_rows_capped_kernel<S, N, T, PF, CO, CE><<<pgrid, SPREAD_THREADS, plane_lds, st>>>(sa, pa, ra_e, n_planes, n_here); \
      }                                                                                                                        \
    } else {                                                                                                                   \
      if (pa.bands > 1) {                                                                                                      \
        if constexpr (CO)                                                                                                      \
          plane_rows_kernel<S, N, T, PF, CO, CE, true><<<pgrid, SPREAD_THREADS, plane_lds, st>>>(sa, pa, ra_e, n_planes, n_here); \
      } else {                                                                                                                 \
        plane_rows_kernel<S, N, T, PF, CO, CE><<<pgrid, SPREAD_THREADS, plane_lds, st>>>(sa, pa, ra_e, n_planes, n_here);       \
      }                                                                                                                        \
    }                                                                                                                          \
  }()))
      note_cosched_kernel(sizeof(T) == 4 ? "plane_rows_capped_kernel" : "plane_rows_kernel");
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
                             ([&] {
                               const unsigned pg = unsigned(m->nx) * unsigned(pa.bands) * unsigned(pa.parts);
                               if (pa.bands > 1)
                                 plane_spread_kernel<S, N, T, true><<<pg, 1024, plane_lds, st>>>(sa, pa);
                               else
                                 plane_spread_kernel<S, N, T><<<pg, 1024, plane_lds, st>>>(sa, pa);
                             }()));
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

