// Fused distance + pair-sum row kernel body (see topology.hip for the topology it walks and include/mipme.h,
// mipme_sr_rows_fused, for the contract).  A header so that the same body can run as its own kernel (topology.hip) and
// inside the horizontally fused spread + pair-sum launch (bricks.hip), where workgroups of the latency-bound spread and of
// the VALU-bound pair sum share the CUs.
#pragma once

#include <type_traits>

#include "common.h"
#include "srpot.h"

namespace mipme {

// Entry stream of the fused kernels: int2 {other atom, cell-shift code} per entry (shifts == NULL -> zero shifts).
// The shift is stored ROLE-ADJUSTED: S for a role-i entry, -S for a role-j entry, so that for both roles
//   u_e = r_other - r_a + S'_e A   has |u_e| = d_p and  d d_p / d r_a = -u_e / d_p   (no per-entry sign in the kernels).
// Code formats: kShiftPacked = 3 x int8 (little end first); kShiftTable = index (sx+3) + 7 (sy+3) + 49 (sz+3) into the
// 343-entry table of Cartesian shift vectors the kernel keeps in LDS (needs |s| <= 3).
// flag bits: 1 = some shift is not an integer in [-127,127]; 2 = some |shift| > 3 (table format unusable).
// kShiftTable32: the table code AND the partner in ONE 32-bit word per entry -- other | code << 22 (343 codes < 2^9,
// atoms < 2^22) -- which halves the entry stream of the co-scheduled pair sum (76 -> 38 MB at cfg3).
enum ShiftFormat { kShiftPacked = 0, kShiftTable = 1, kShiftTable32 = 2 };
static constexpr int kCompactAtomBits = 22;
static constexpr int64_t kCompactMaxAtoms = int64_t(1) << kCompactAtomBits;
static constexpr int kShiftTableRange = 3, kShiftTableBase = 2 * kShiftTableRange + 1;
static constexpr int kShiftTableSize = kShiftTableBase * kShiftTableBase * kShiftTableBase;
// Row layout flag, OR-ed into the shift format a caller passes: PADDED rows as the device neighbour list writes them
// (neighbors.hip, mipme_nl_stream) -- row_ptr is int32[3 N + 1] = {begin, middle, end} of every row (the last word: the size of
// the entry buffer) instead of the shared-boundary int32[2 N + 1], and a row holds EVERY neighbour of its atom once (both
// directions of each pair are in the stream, middle == end), so sums that must count a pair once -- the cell gradient -- take
// half of every entry.  Such rows always stand for a half list.
static constexpr int kRowsPadded = 0x100, kShiftFormatMask = 0xff;

// ---- owner-computes pair kernels ---------------------------------------------------------------
#ifndef MIPME_ROW_UNROLL
#define MIPME_ROW_UNROLL 4
#endif
#ifndef MIPME_ROW_UNROLL_F64
#define MIPME_ROW_UNROLL_F64 2  // entries in flight per lane of the fused fp64 body (see sr_fused_rows_body)
#endif

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// out[a,c] (+)= 1/2 sum_{entries of a} src[other,c] * v_SR(dist[p])
//   roles: forward uses role i (+ role j for a half list) with src = charges;
//          the charge gradient uses role j (+ role i for a half list) with src = upstream gradient.
// Row kernels: kRowLanes lanes own one atom's row (4 atoms per wavefront -- a wavefront per atom was limited by the
// fixed per-wave latency: row_ptr fetch, reduction, store), and every lane keeps kRowUnroll entries in flight:
// all entry loads are issued first, then the dependent gathers (dist[p], src[other]), then the arithmetic.
static constexpr int kRowUnroll = MIPME_ROW_UNROLL;
#ifndef MIPME_ROW_LANES
#define MIPME_ROW_LANES 16
#endif
static constexpr int kRowLanes = MIPME_ROW_LANES;
static constexpr int kRowsPerBlock = 256 / kRowLanes;

template <typename T>
__device__ __forceinline__ T row_sum(T v) {
#pragma unroll
  for (int off = kRowLanes / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kRowLanes);
  return v;
}

// The sums at the end of the packed / fp64 row bodies (16 lanes per row): DPP + readlane, or the ds_bpermute shuffles they
// replaced (-DMIPME_ROWS_SHUFFLE_SUMS=1: A/B builds only)
#ifndef MIPME_ROWS_SHUFFLE_SUMS
#define MIPME_ROWS_SHUFFLE_SUMS 0
#endif
template <typename T>
__device__ __forceinline__ T body_row_sum(T v) {
#if MIPME_ROWS_SHUFFLE_SUMS
  return row_sum(v);
#else
  return row16_sum_dpp(v);
#endif
}
template <typename T>
__device__ __forceinline__ T body_wave_sum(T v) {
#if MIPME_ROWS_SHUFFLE_SUMS
  return wave_sum(v);
#else
  return wave_sum_dpp(v);
#endif
}

__device__ __forceinline__ int unpack8(int word, int k) { return (word << (24 - 8 * k)) >> 24; }

enum FusedMode {
  kPot = 0,       // potentials from src
  kPotForce = 1,  // potentials from charges + speculative force sums (w_e = q[o])
  kForceQ = 2,    // force sums with w_e = q[o] (energy mode, finished by the finalize kernel)
  kForceG = 3,    // force sums with the general weights built from g and q
};

template <typename T>
struct FusedRowsArgs {
  SRPot s;
  FastRS cf;
  int64_t N;
  const int* row_ptr;
  const int2* ent_sh;
  const int2* entries;
  const uint8_t* mask;
  const T* pos;
  const AtomRecord<T>* rec;  // rec[o] = (position of o, src[o]) with src = charges except in the transposed potential pass
  const T* cell;
  const T* q;
  const T* g;
  int pot_lo, pot_hi;
  bool full, accumulate;
  T* out;
  T* force;
  double* partials;
  T* dist_out;
  // kPotForce only, nullable: per-WAVE sums {sum_a q_a out_a, sum_a q_a^2} over the rows of the wave (fp64[2] per wave, wave
  // w of workgroup b at index b * BS/64 + w) -- the pair part of the energy and the self-term sum, reduced later by the
  // gather's tail (bricks.hip)
  double* epart;
  // packed / fp64 bodies with CELL only: per-WAVE sums (fp64[9] per wave, indexed like epart) of
  //   C[i][c] = sum_rows q_a sum_e w_e v_SR'(d_e)/d_e sh_i u_c      (sh = S' A: the entry's Cartesian cell shift, u = pair vector)
  // -- the pair part of dE/dcell is  f/2 * inv(A)^T-contracted C (every pair sits in two rows; kfilter.hip
  // cell_tail_finalize_kernel)
  double* cpart;
  const int* skip;  // nullable: the stand-alone kernels return at once if *skip == 1 (mipme_set_skip_flag)
  int row_stride;  // words of row_ptr per atom: 2 (rows share their boundaries) or 3 (kRowsPadded)
  bool symmetric;  // kRowsPadded: every pair appears in both of its rows as a "role i" entry
};

template <typename T>
static inline FusedRowsArgs<T> make_fused_rows_args(const SRPot& s, const FastRS& cf, int64_t N, const void* row_ptr,
                                                    const void* ent_sh, const void* entries, const void* mask,
                                                    const void* pos, const void* records, const void* cell, const void* q,
                                                    const void* g, int pot_lo, int pot_hi, int full_list, int accumulate,
                                                    void* out, void* force, void* partials, void* dist_out,
                                                    int shift_format = 0) {
  FusedRowsArgs<T> a;
  a.s = s;
  a.cf = cf;
  a.N = N;
  a.row_ptr = (const int*)row_ptr;
  a.ent_sh = (const int2*)ent_sh;
  a.entries = (const int2*)entries;
  a.mask = (const uint8_t*)mask;
  a.pos = (const T*)pos;
  a.rec = (const AtomRecord<T>*)records;
  a.cell = (const T*)cell;
  a.q = (const T*)q;
  a.g = (const T*)g;
  a.pot_lo = pot_lo;
  a.pot_hi = pot_hi;
  a.full = full_list != 0;
  a.accumulate = accumulate != 0;
  a.out = (T*)out;
  a.force = (T*)force;
  a.partials = (double*)partials;
  a.dist_out = (T*)dist_out;
  a.epart = nullptr;
  a.cpart = nullptr;
  a.skip = nullptr;
  a.symmetric = (shift_format & kRowsPadded) != 0;
  a.row_stride = a.symmetric ? 3 : 2;
  if (a.symmetric) a.full = false;
  return a;
}

// BS = threads of the workgroup (BS / kRowLanes rows per workgroup); block = index of the workgroup among the row workgroups
// UNROLL: entries in flight per lane (0 = the measured default for the dtype, see below)
// COMPACT: the entry stream is kShiftTable32 (one int per entry; implies TABLE)
// shift_tab: LDS storage of the workgroup for the kShiftTableSize Cartesian shift vectors (TABLE formats; the caller owns it so
// that a kernel with dynamic LDS of its own -- the co-scheduled launch -- does not pay for a second, static allocation)
template <typename T, int MODE, bool CELLGRAD, int PFAST, bool MASK, bool TABLE, int BS, int UNROLL = 0, bool COMPACT = false>
__device__ __forceinline__ void sr_fused_rows_body(const FusedRowsArgs<T>& args, unsigned block,
                                                   AtomRecord<T>* __restrict__ shift_tab = nullptr) {
  static_assert(!COMPACT || (TABLE && !MASK), "compact entries carry table codes and have no pair-mask variant");
  const SRPot& s = args.s;
  const FastRS& cf = args.cf;
  const int64_t N = args.N;
  const int* __restrict__ row_ptr = args.row_ptr;
  const int2* __restrict__ ent_sh = args.ent_sh;
  const int* __restrict__ ent32 = reinterpret_cast<const int*>(args.ent_sh);
  auto load_entry = [&](int e) -> int2 {
    if constexpr (COMPACT) {
      const int w = ent32[e];
      return make_int2(w & int(kCompactMaxAtoms - 1), int(unsigned(w) >> kCompactAtomBits));
    } else {
      return ent_sh[e];
    }
  };
  const int2* __restrict__ entries = args.entries;
  const uint8_t* __restrict__ mask = args.mask;
  const T* __restrict__ pos = args.pos;
  const AtomRecord<T>* __restrict__ rec = args.rec;
  const T* __restrict__ cell = args.cell;
  const T* __restrict__ q = args.q;
  const T* __restrict__ g = args.g;
  const int pot_lo = args.pot_lo, pot_hi = args.pot_hi;
  const bool full = args.full, accumulate = args.accumulate;
  T* __restrict__ out = args.out;
  T* __restrict__ force = args.force;
  double* __restrict__ partials = args.partials;
  T* __restrict__ dist_out = args.dist_out;
  // rec[o] = (position of o, src[o]) with src = charges except in the transposed potential pass
  // dist_out (potential passes without a mask, pair list ordered by its first index): the role-i entries of a row are the
  // consecutive pairs starting at the pair of its first entry, and their distances are written as a by-product
  // entries in flight per lane: measured on MI355X -- fp32: 2 (cfg3 0.0840 ms; 0.0853 with 4, 0.0862 with 1, 0.0858 with 3);
  // fp64: 2 since the loop went from 126 to 86 VGPRs (cfg2 0.0548 against 0.0560 ms with 1; with the earlier, register-heavier
  // loop 1 was better for a single frame: 0.0708 against 0.0797); beyond that the extra registers cost more than the loads
  // they overlap
  constexpr int U = UNROLL ? UNROLL : (sizeof(T) == 8 ? MIPME_ROW_UNROLL_F64 : 2);
  constexpr bool POT = MODE == kPot || MODE == kPotForce;
  constexpr bool FORCE = MODE != kPot;
  T A[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) A[k] = cell ? cell[k] : T(0);
  const T c_inv2s2 = T(cf.inv_2s2), c1 = T(cf.c1), cpref = T(cf.pref);
  if constexpr (TABLE) {  // Cartesian shift vector of every table code
    for (int k = threadIdx.x; k < kShiftTableSize; k += BS) {
      const T sx = T(k % kShiftTableBase - kShiftTableRange), sy = T((k / kShiftTableBase) % kShiftTableBase - kShiftTableRange),
              sz = T(k / (kShiftTableBase * kShiftTableBase) - kShiftTableRange);
      shift_tab[k] = AtomRecord<T>{sx * A[0] + sy * A[3] + sz * A[6], sx * A[1] + sy * A[4] + sz * A[7],
                                   sx * A[2] + sy * A[5] + sz * A[8], T(0)};
    }
    __syncthreads();
  }
  const int sub = threadIdx.x % kRowLanes;
  unsigned a = block * (BS / kRowLanes) + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = unsigned(N - 1);
  const int* __restrict__ rp = row_ptr + int64_t(args.row_stride) * a;
  const int r0 = rp[0], mid = rp[1], r2 = rp[2];
  const int pbeg = pot_lo == 0 ? r0 : mid, pend = pot_hi == 0 ? mid : r2;  // entries that feed the potential
  // in the potential + force pass (roles i and j both visited) a full list feeds the potential from role i only
  const int pot_end = (MODE == kPotForce && full) ? mid : 0x7fffffff;
  const int beg = FORCE ? r0 : pbeg;
  const int end = valid ? (FORCE ? r2 : pend) : beg;
  // own position (and charge): from the caller's arrays, or -- pos == NULL: the records ARE the atoms' storage (live-bin step,
  // bricks.hip) -- from the atom's own record
  T ax, ay, az;
  if (pos) {
    ax = pos[3 * a];
    ay = pos[3 * a + 1];
    az = pos[3 * a + 2];
  } else {
    const AtomRecord<T> own = rec[a];
    ax = own.x;
    ay = own.y;
    az = own.z;
  }
  constexpr bool CAN_WRITE_D = POT && !MASK;
  int64_t pair_base = 0;  // pair index of entry r0 minus r0
  if constexpr (CAN_WRITE_D) {
    if (dist_out) pair_base = int64_t(entries[r0].y) - r0;
  }
  T qa = T(0), ga = T(0);
  if constexpr (FORCE) qa = q ? q[a] : rec[a].w;
  if constexpr (MODE == kForceG) ga = g[a];
  // general weights: half list 1/2 (g_a q_o + g_o q_a); full list keeps the first term for role i, the second for role j
  const T keep_i = T(0.5), keep_j_of_i = full ? T(0) : T(0.5);
  T pot = T(0), fx = T(0), fy = T(0), fz = T(0);
  double cg[9];
  if constexpr (CELLGRAD) {
#pragma unroll
    for (int k = 0; k < 9; ++k) cg[k] = 0.0;
  }
  int2 en_next[U];
  int pm_next[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = beg + u * kRowLanes + sub;
    const int ec = e < end ? e : beg;
    en_next[u] = load_entry(ec);
    if constexpr (MASK) pm_next[u] = entries[ec].y;
  }
  for (int base = beg; base < end; base += kRowLanes * U) {
    int2 en[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = base + u * kRowLanes + sub < end;
      en[u] = en_next[u];
    }
    T ox[U], oy[U], oz[U], so[U], go[U];
    uint8_t mk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = en[u].x;
      const AtomRecord<T> r = rec[o];
      ox[u] = r.x;
      oy[u] = r.y;
      oz[u] = r.z;
      so[u] = r.w;
      if constexpr (MODE == kForceG) go[u] = g[o];
      if constexpr (MASK) mk[u] = mask[pm_next[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + kRowLanes * U + u * kRowLanes + sub;
      const int ec = e < end ? e : beg;
      en_next[u] = load_entry(ec);
      if constexpr (MASK) pm_next[u] = entries[ec].y;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * kRowLanes + sub;
      const bool role_i = e < mid;
      bool use = ok[u];
      if constexpr (MASK) use = use && mk[u] != 0;
      // u = r_o - r_a + S' A with the role-adjusted shift S' (see topo_pack_entries_kernel): |u| = d, d d / d r_a = -u / d
      T shx, shy, shz;
      if constexpr (TABLE) {
        const AtomRecord<T> sh = shift_tab[en[u].y];
        shx = sh.x;
        shy = sh.y;
        shz = sh.z;
      } else {
        const T sx = T(unpack8(en[u].y, 0)), sy = T(unpack8(en[u].y, 1)), sz = T(unpack8(en[u].y, 2));
        shx = sx * A[0] + sy * A[3] + sz * A[6];
        shy = sx * A[1] + sy * A[4] + sz * A[7];
        shz = sx * A[2] + sy * A[5] + sz * A[8];
      }
      const T vx = (ox[u] - ax) + shx, vy = (oy[u] - ay) + shy, vz = (oz[u] - az) + shz;
      const T d2 = vx * vx + vy * vy + vz * vz;
      T v, dvd;  // v_SR(d) and v_SR'(d) / d
      if constexpr (PFAST > 0) {
        fast_rs_eval<PFAST, FORCE, T>(c_inv2s2, c1, cpref, d2, v, dvd, cf.cheb);
      } else {
        const T d = fsqrt(d2);
        T dv;
        sr_eval<T, FORCE>(s, d, v, dv);
        dvd = dv / d;
      }
      if constexpr (CAN_WRITE_D) {
        if (dist_out && role_i && ok[u]) {
          const T d2c = d2 < T(1e-30) ? T(1e-30) : d2;
          dist_out[pair_base + e] = PFAST > 0 ? d2c * rs_rsqrt(d2c) : fsqrt(d2);
        }
      }
      const T sv = use ? so[u] : T(0);  // masked / padding entries carry zero weight
      if constexpr (POT) {
        const bool in_pot = MODE == kPot || e < pot_end;
        pot += (in_pot ? sv : T(0)) * v;
      }
      if constexpr (FORCE) {
        T w;
        if constexpr (MODE == kForceG) {
          const T gq = role_i ? keep_i * ga * sv + keep_j_of_i * go[u] * qa : keep_j_of_i * ga * sv + keep_i * go[u] * qa;
          w = use ? gq : T(0);
        } else {
          w = sv;
        }
        const T sc = w * dvd;
        fx -= sc * vx;
        fy -= sc * vy;
        fz -= sc * vz;
        if constexpr (CELLGRAD) {
          if (role_i) {  // role i: S' = S and u is the pair vector itself
            T sx, sy, sz;
            if constexpr (TABLE) {
              const int code = en[u].y;
              sx = T(code % kShiftTableBase - kShiftTableRange);
              sy = T((code / kShiftTableBase) % kShiftTableBase - kShiftTableRange);
              sz = T(code / (kShiftTableBase * kShiftTableBase) - kShiftTableRange);
            } else {
              sx = T(unpack8(en[u].y, 0));
              sy = T(unpack8(en[u].y, 1));
              sz = T(unpack8(en[u].y, 2));
            }
            const double wq = (MODE == kForceG ? 1.0 : double(qa)) * (args.symmetric ? 0.5 : 1.0);
            const double px = wq * double(sc * vx), py = wq * double(sc * vy), pz = wq * double(sc * vz);
            cg[0] += double(sx) * px; cg[1] += double(sx) * py; cg[2] += double(sx) * pz;
            cg[3] += double(sy) * px; cg[4] += double(sy) * py; cg[5] += double(sy) * pz;
            cg[6] += double(sz) * px; cg[7] += double(sz) * py; cg[8] += double(sz) * pz;
          }
        }
      }
    }
  }
  if constexpr (POT) {
    pot = row_sum(pot);
    if (sub == 0 && valid) out[a] = (accumulate ? out[a] : T(0)) + T(0.5) * pot;
    if constexpr (MODE == kPotForce) {
      if (args.epart) {  // uniform: energy partial sums of this WAVE's rows (no barrier, no LDS: the waves retire independently)
        const bool mine = sub == 0 && valid;
        const T e1 = wave_sum(mine ? qa * (T(0.5) * pot) : T(0));
        const T e2 = wave_sum(mine ? qa * qa : T(0));
        if ((threadIdx.x & 63) == 0) {
          const int64_t w = int64_t(block) * (BS / 64) + (threadIdx.x >> 6);
          args.epart[2 * w] = double(e1);
          args.epart[2 * w + 1] = double(e2);
        }
      }
    }
  }
  if constexpr (FORCE) {
    fx = row_sum(fx);
    fy = row_sum(fy);
    fz = row_sum(fz);
    if (sub == 0 && valid) {
      force[3 * a] = fx;
      force[3 * a + 1] = fy;
      force[3 * a + 2] = fz;
    }
  }
  if constexpr (CELLGRAD) {
    __shared__ double red[BS / 64][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double v = wave_sum(cg[k]);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
      double v = 0.0;
      for (int w = 0; w < BS / 64; ++w) v += red[w][threadIdx.x];
      partials[int64_t(block) * 9 + threadIdx.x] = v;
    }
  }
}

// ---- packed fp32 body of the potential + force pass -------------------------------------------------------------------
// The row kernels are bound by VALU issue (profiles/r02_g_sq_counters.txt), so this body, used for the one case that carries
// the headline workloads -- float, kPotForce, 4-byte entries, no mask / cell gradient / distance by-product --, is written for
// the instruction count rather than for generality:
//  * the two entries a lane has in flight are evaluated as ONE 2-vector, so the whole scalar chain of fast_rs_eval (the erfc
//    polynomial, the powers of 1/d, the final products) issues as v_pk_{mul,add,fma}_f32 -- two entries per instruction --,
//    and within an entry (x, y) travel as a pair (z alone); only rsq / exp2 / rcp remain one per entry;
//  * entry words and partner records are fetched with buffer loads (32-bit offsets into a resource: no 64-bit address
//    arithmetic per entry, and reads past the end of the stream return 0 = atom 0, code 0, so the prefetch needs no clamp);
//  * invalid lanes (row tail) are neutralised by two selects (d^2 := 1, weight := 0) instead of clamped addresses.
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2v pk_rsq(f2v x) { return f2v{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
__device__ __forceinline__ f2v pk_rcp(f2v x) { return f2v{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f2v pk_exp2(f2v x) { return f2v{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

// fast_rs_eval for two distances at once (same formulas, same coefficients; see there)
template <int P>
__device__ __forceinline__ void fast_rs_eval_pk(float inv_2s2, float c1, float pref, f2v d2, f2v& v, f2v& dvd) {
  const f2v inv = pk_rsq(d2);
  const f2v inv2 = inv * inv;
  const f2v e = pk_exp2(d2 * (-1.4426950408889634f * inv_2s2));
  f2v Q, two_x_dens;
  if constexpr (P % 2 == 0) {
    const f2v x = d2 * inv_2s2;
    f2v term = f2v{1.f, 1.f}, sum = f2v{1.f, 1.f};
#pragma unroll
    for (int k = 1; k < P / 2; ++k) {
      term *= x * float(1.0 / k);
      sum += term;
    }
    Q = e * sum;
    two_x_dens = 2.f * x * e * term;
  } else {
    constexpr int m = (P - 1) / 2;
    const f2v d = d2 * inv;
    const f2v y = c1 * d;
    const f2v t = pk_rcp(1.0f + 0.4f * y);
    f2v p = 2.646481385e-02f * t + -6.557867191e-02f;
    p = p * t + -7.738398321e-02f;
    p = p * t + 2.820383187e-01f;
    p = p * t + -6.220284696e-02f;
    p = p * t + 2.579626189e-01f;
    p = p * t + 1.840778096e-01f;
    p = p * t + 2.291638826e-01f;
    p = p * t + 2.254580744e-01f;
    Q = (e * t) * p;
    f2v term = (float(2.0 * 0.56418958354775628695) * c1) * d * e;  // term_1
    if constexpr (m > 0) {
      const f2v x = d2 * inv_2s2;
#pragma unroll
      for (int k = 1; k <= m; ++k) {
        Q += term;
        term *= x * float(1.0 / (k + 0.5));
      }
    }
    two_x_dens = float(2 * m + 1) * term;
  }
  f2v invp = inv;
#pragma unroll
  for (int k = 1; k < P; ++k) invp *= inv;
  const f2v pi = pref * invp;
  v = pi * Q;
  dvd = -(pi * inv2) * (two_x_dens + float(P) * Q);
}

// Raw buffer loads through the LLVM intrinsics (the __builtin_amdgcn_raw_buffer_load_b128 of this compiler lowers to a
// one-dword load).  Resource over [base, base + bytes): stride 0, DATA_FORMAT = 32 bit -- the word composable_kernel uses for
// gfx90a / gfx94x / gfx950; out-of-range reads return 0.
typedef int i4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ f4v llvm_raw_buffer_load_f4(i4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ int llvm_raw_buffer_load_i1(i4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ f4v llvm_struct_buffer_load_f4(i4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
typedef float f3v __attribute__((ext_vector_type(3)));
__device__ f3v llvm_struct_buffer_load_f3(i4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v3f32");
__device__ float llvm_struct_buffer_load_f1(i4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.f32");
#ifndef MIPME_ROWS_FROM_POS
#define MIPME_ROWS_FROM_POS 0  // 1 (experiment builds): the packed fp32 body gathers its partners from positions (N,3) + charges instead of the (x,y,z,q) records
#endif
// structured resource: `records` elements of `stride` bytes (indexed loads: address = base + index * stride, range check on the index)
__device__ __forceinline__ i4v struct_buffer(const void* base, unsigned stride, unsigned records) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return i4v{int(unsigned(a)), int((unsigned(a >> 32) & 0xffffu) | (stride << 16)), int(records), 0x00020000};
}
__device__ __forceinline__ i4v raw_buffer(const void* base, unsigned bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return i4v{int(unsigned(a)), int(unsigned(a >> 32) & 0xffffu), int(bytes), 0x00020000};
}

// The fp64 counterpart of the packed body below (Coulomb, 4-byte entries, potential + force sums, nobody asks for the
// distances): raw buffer loads (no 64-bit address arithmetic), no distance by-product and its branches, erfc from the LDS table
// of srpot.h instead of the whole-range polynomial whose 21 scalar-register constants made the generic body spill.  ISA of the
// hot loop: 223 -> ... VALU per two entries (profiles/r03_experiments.txt).
// lds: kShiftTableSize AtomRecord<double> (the shift table) followed by kErfcxLdsDoubles doubles (the erfcx table) and kExp2Tab
// doubles (2^(j/64) for exp_neg_table2).
__device__ __forceinline__ i4v uniform_rsrc(i4v r) {
  return i4v{__builtin_amdgcn_readfirstlane(r.x), __builtin_amdgcn_readfirstlane(r.y), __builtin_amdgcn_readfirstlane(r.z),
             __builtin_amdgcn_readfirstlane(r.w)};
}
static constexpr size_t kRowsF64LdsBytes = size_t(kShiftTableSize) * sizeof(AtomRecord<double>) + sizeof(double) * (kErfcxLdsDoubles + kExp2Tab);
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ d2v llvm_raw_buffer_load_d2(i4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f64");
__device__ d2v llvm_struct_buffer_load_d2(i4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v2f64");

#ifdef MIPME_WG_TIMELINE  // measurement builds only (tools/rows_phases.py): clock stamps of the first 1024 row workgroups
static __device__ long long g_rows_phase[8 * 1024];
#define MIPME_ROWS_PHASE(k, wait)                                                                                   \
  do {                                                                                                              \
    if (wait) __builtin_amdgcn_s_waitcnt(0);                                                                        \
    if (threadIdx.x == 0 && block < 1024) g_rows_phase[block * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define MIPME_ROWS_PHASE(k, wait)
#endif

#ifndef MIPME_ENTRY_AUX
// cache policy of the entry-stream loads of the packed fp32 / fp64 bodies (aux operand of buffer_load; 2 = nt, "non-temporal").
// The stream is read once per launch next to partner records that are gathered ~300 times each, so marking it non-temporal
// looked right -- and measured 18 % SLOWER (cfg3 launch 20.9 -> 24.8 us, cfg5 174.8 -> 206.7 us, 1 029 000 atoms 1.08 -> 1.22 ms;
// profiles/r04_k_ab_nt.txt): a row's 16 lanes take 64 bytes per load, so every 128-byte line of the stream serves two loads, and
// the hint throws the line away in between.  Default policy.
#define MIPME_ENTRY_AUX 0
#endif
#ifndef MIPME_ROWS_UNMASKED
#define MIPME_ROWS_UNMASKED 1  // 0: every iteration of the packed fp32 and the fp64 bodies with its tail selects (the form before round 4's end)
#endif
// PFAST = 1: Coulomb (erfc from the LDS table); PFAST even (6: round 5): Q_p(x) = e^{-x} sum_{k < p/2} x^k / k!, no table
// exp(-x) of the fp64 pair body from a 64-entry table + 5 FMAs (srpot.h exp_neg_table2) instead of 13 FMAs; 0: the latter (A/B)
#ifndef MIPME_F64_EXP_TABLE
#define MIPME_F64_EXP_TABLE 1
#endif
// Newton steps behind v_rsq_f64 in the fp64 pair body.  ONE: the instruction is good to ~2^-26 on gfx950, a step squares that --
// energies and forces of cfg2 / cfg4 against the fp64 oracle are the same to the digit with one step as with two (3.2e-16 /
// 2.4e-15), three fp64 instructions per entry less (cfg4 0.1223 -> 0.1202 ms; profiles/r05_experiments.txt item 14).
// -DMIPME_F64_NEWTON=2: the old two (A/B builds).
#ifndef MIPME_F64_NEWTON
#define MIPME_F64_NEWTON 1
#endif
template <int BS, bool CELL = false, int PFAST = 1>
__device__ __forceinline__ void sr_rows_f64_body(const FusedRowsArgs<double>& args, unsigned block, char* __restrict__ lds) {
  static_assert(kRowLanes == 16, "two groups of 16 entries per row and iteration");
  static_assert(PFAST == 1 || PFAST % 2 == 0, "odd exponents above 1 take the generic body");
  AtomRecord<double>* __restrict__ shift_tab = reinterpret_cast<AtomRecord<double>*>(lds);
  double* __restrict__ etab = reinterpret_cast<double*>(lds + size_t(kShiftTableSize) * sizeof(AtomRecord<double>));
  const int64_t N = args.N;
  const int* __restrict__ row_ptr = args.row_ptr;
  const double* __restrict__ pos = args.pos;
  const double* __restrict__ cell = args.cell;
  double* __restrict__ out = args.out;
  double* __restrict__ force = args.force;
  const double c_inv2s2 = args.cf.inv_2s2, c1 = args.cf.c1, cpref = args.cf.pref;
  const int sub = threadIdx.x % kRowLanes;
  unsigned a = block * (BS / kRowLanes) + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = unsigned(N - 1);
  MIPME_ROWS_PHASE(0, false);
  const int* __restrict__ rp = row_ptr + int64_t(args.row_stride) * a;
  const int r0 = rp[0], mid = rp[1], r2 = rp[2];
  // (uniform by construction; said so explicitly, or the buffer descriptor built from it is treated as divergent and every
  // entry load becomes a waterfall loop)
  const int n_entries = __builtin_amdgcn_readfirstlane(row_ptr[int64_t(args.row_stride) * N]);
  double ax, ay, az, qa;
  if (pos) {
    ax = pos[3 * a];
    ay = pos[3 * a + 1];
    az = pos[3 * a + 2];
    qa = args.q[a];
  } else {
    const AtomRecord<double> own = args.rec[a];
    ax = own.x;
    ay = own.y;
    az = own.z;
    qa = own.w;
  }
  {
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = cell ? cell[k] : 0.0;
    for (int k = threadIdx.x; k < kShiftTableSize; k += BS) {
      const double sx = double(k % kShiftTableBase - kShiftTableRange),
                   sy = double((k / kShiftTableBase) % kShiftTableBase - kShiftTableRange),
                   sz = double(k / (kShiftTableBase * kShiftTableBase) - kShiftTableRange);
      shift_tab[k] = AtomRecord<double>{sx * A[0] + sy * A[3] + sz * A[6], sx * A[1] + sy * A[4] + sz * A[7],
                                        sx * A[2] + sy * A[5] + sz * A[8], 0.0};
    }
    if constexpr (PFAST == 1) erfcx_table_to_lds(etab, threadIdx.x, BS);
#if MIPME_F64_EXP_TABLE
    exp2_table_to_lds(etab + kErfcxLdsDoubles, threadIdx.x, BS);
#endif
  }
  const int pot_end = args.full ? mid : 0x7fffffff;
  const int beg = r0, end = valid ? r2 : r0;
  const i4v ent_rs = uniform_rsrc(raw_buffer(args.ent_sh, unsigned(n_entries) * 4u));
  const i4v rec_rs = uniform_rsrc(struct_buffer(args.rec, 32u, unsigned(N)));  // (indexed loads: see the packed fp32 body)
  constexpr unsigned kAtomMask = unsigned(kCompactMaxAtoms - 1);
  // (loop bookkeeping in scalar registers: see the packed fp32 body)
  const int lane_e0 = beg + sub;
  const int voff = lane_e0 * 4;
  const int remA = end - lane_e0, remP = pot_end - lane_e0, row_len = end - beg;
  int k32 = 0;
  unsigned wA = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff, 0, MIPME_ENTRY_AUX));
  unsigned wB = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff + 4 * kRowLanes, 0, MIPME_ENTRY_AUX));
  MIPME_ROWS_PHASE(1, false);
  __syncthreads();  // shift table + erfcx table
  MIPME_ROWS_PHASE(2, true);
  double pot = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
  double cg[CELL ? 9 : 1];
#pragma unroll
  for (int k = 0; k < (CELL ? 9 : 1); ++k) cg[k] = 0.0;
  // (MASKED / unmasked iterations: see the packed fp32 body below)
  auto iteration = [&](auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int iA = int(wA & kAtomMask), iB = int(wB & kAtomMask);
    const d2v pAxy = llvm_struct_buffer_load_d2(rec_rs, iA, 0, 0, 0), pAzq = llvm_struct_buffer_load_d2(rec_rs, iA, 16, 0, 0);
    const d2v pBxy = llvm_struct_buffer_load_d2(rec_rs, iB, 0, 0, 0), pBzq = llvm_struct_buffer_load_d2(rec_rs, iB, 16, 0, 0);
    unsigned codeA = wA >> kCompactAtomBits, codeB = wB >> kCompactAtomBits;
    asm("" : "+v"(codeA));
    asm("" : "+v"(codeB));
    const AtomRecord<double> sA = shift_tab[codeA], sB = shift_tab[codeB];
    const int kk = k32;  // (uniform) entries before this iteration
    k32 += 2 * kRowLanes;
    wA = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff, k32 * 4, MIPME_ENTRY_AUX));
    wB = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff + 4 * kRowLanes, k32 * 4, MIPME_ENTRY_AUX));
    const bool okA = !MASKED || kk < remA, okB = !MASKED || kk + kRowLanes < remA;
    // the two entries side by side, operation by operation: each constant of the polynomials then serves two FMAs from the
    // same scalar register pair.  No select on d2 for entries beyond the row's end: they read some valid record (the buffer
    // descriptor bounds the loads), everything stays finite in double, and their weight sv is zero.
    const double vx[2] = {(pAxy.x - ax) + sA.x, (pBxy.x - ax) + sB.x};
    const double vy[2] = {(pAxy.y - ay) + sA.y, (pBxy.y - ay) + sB.y};
    const double vz[2] = {(pAzq.x - az) + sA.z, (pBzq.x - az) + sB.z};
    const double sv[2] = {okA ? pAzq.y : 0.0, okB ? pBzq.y : 0.0};
    const double sp[2] = {(!MASKED || kk < remP) ? sv[0] : 0.0, (!MASKED || kk + kRowLanes < remP) ? sv[1] : 0.0};
    double d2[2], inv[2], x[2], e[2], y[2], Q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) d2[u] = __builtin_fmax(__builtin_fma(vz[u], vz[u], __builtin_fma(vy[u], vy[u], vx[u] * vx[u])), 1e-30);
#pragma unroll
    for (int u = 0; u < 2; ++u) inv[u] = __builtin_amdgcn_rsq(d2[u]);
#pragma unroll
    for (int k = 0; k < MIPME_F64_NEWTON; ++k) {
#pragma unroll
      for (int u = 0; u < 2; ++u) inv[u] = inv[u] * __builtin_fma(-0.5 * d2[u] * inv[u], inv[u], 1.5);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      x[u] = d2[u] * c_inv2s2;
      y[u] = c1 * (d2[u] * inv[u]);
    }
#if MIPME_F64_EXP_TABLE
    exp_neg_table2(x, e, etab + kErfcxLdsDoubles);
#else
    exp_neg_fast2(x, e);
#endif
    constexpr unsigned kCentre = unsigned(kShiftTableRange * (1 + kShiftTableBase + kShiftTableBase * kShiftTableBase));
    const bool any_cross = CELL && __builtin_amdgcn_ballot_w64((okA && codeA != kCentre) || (okB && codeB != kCentre)) != 0;
    double dens[2];  // 2 x * (density term of Q_p): term_1 for Coulomb, 2 x e x^{p/2-1} / (p/2-1)! for even p
    if constexpr (PFAST == 1) {
#pragma unroll
      for (int u = 0; u < 2; ++u) Q[u] = erfc_from_table(y[u], e[u], etab);
#pragma unroll
      for (int u = 0; u < 2; ++u) dens[u] = (2.0 * 0.56418958354775628695) * y[u] * e[u];
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        double term = 1.0, sum = 1.0;
#pragma unroll
        for (int k = 1; k < PFAST / 2; ++k) {
          term *= x[u] * (1.0 / k);
          sum += term;
        }
        Q[u] = e[u] * sum;
        dens[u] = 2.0 * x[u] * e[u] * term;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const double term = dens[u];
      double invp = inv[u];
#pragma unroll
      for (int k = 1; k < PFAST; ++k) invp *= inv[u];
      const double pi = cpref * invp;
      pot = __builtin_fma(sp[u], pi * Q[u], pot);
      const double sc = sv[u] * ((pi * (inv[u] * inv[u])) * (term + double(PFAST) * Q[u]));  // = -sv dv/dd / d
      fx = __builtin_fma(sc, vx[u], fx);
      fy = __builtin_fma(sc, vy[u], fy);
      fz = __builtin_fma(sc, vz[u], fz);
      if constexpr (CELL) {
      if (any_cross) {  // (wave-uniform: see the packed body; sc = -w v'/d here: the sign is restored below)
        const AtomRecord<double>& sh = u == 0 ? sA : sB;
        const double tx = sc * vx[u], ty = sc * vy[u], tz = sc * vz[u];
        cg[0] = __builtin_fma(sh.x, tx, cg[0]); cg[1] = __builtin_fma(sh.x, ty, cg[1]); cg[2] = __builtin_fma(sh.x, tz, cg[2]);
        cg[3] = __builtin_fma(sh.y, tx, cg[3]); cg[4] = __builtin_fma(sh.y, ty, cg[4]); cg[5] = __builtin_fma(sh.y, tz, cg[5]);
        cg[6] = __builtin_fma(sh.z, tx, cg[6]); cg[7] = __builtin_fma(sh.z, ty, cg[7]); cg[8] = __builtin_fma(sh.z, tz, cg[8]);
      }
      }
    }
#ifdef MIPME_WG_TIMELINE
    if (kk == 0) MIPME_ROWS_PHASE(3, false);
    if (kk == 2 * kRowLanes) MIPME_ROWS_PHASE(4, false);
#endif
  };
#if MIPME_ROWS_UNMASKED
  if (!args.full) {  // uniform
    while (__builtin_amdgcn_ballot_w64(k32 + kRowLanes < remA) == ~0ull) iteration(std::false_type{});
  }
#endif
  while (k32 < row_len) iteration(std::true_type{});
  MIPME_ROWS_PHASE(5, false);
  if constexpr (CELL) {
    if (args.cpart) {
      const int64_t w = int64_t(block) * (BS / 64) + (threadIdx.x >> 6);
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const double v = wave_sum_dpp(valid ? -qa * cg[k] : 0.0);
        if ((threadIdx.x & 63) == 0) args.cpart[9 * w + k] = v;
      }
    }
  }
  // (row and wave sums through DPP / readlane: the shuffle forms are ds_bpermute, six LDS operations and their index arithmetic
  // per step -- 28 of them per wave were a quarter of the vector instructions of a wave's ten loop iterations)
  pot = body_row_sum(pot);
  if (sub == 0 && valid) out[a] = (args.accumulate ? out[a] : 0.0) + 0.5 * pot;
  if (args.epart) {  // see the generic body
    const bool mine = sub == 0 && valid;
    const double e1 = body_wave_sum(mine ? qa * (0.5 * pot) : 0.0);
    const double e2 = body_wave_sum(mine ? qa * qa : 0.0);
    if ((threadIdx.x & 63) == 0) {
      const int64_t w = int64_t(block) * (BS / 64) + (threadIdx.x >> 6);
      args.epart[2 * w] = e1;
      args.epart[2 * w + 1] = e2;
    }
  }
  fx = body_row_sum(fx);
  fy = body_row_sum(fy);
  fz = body_row_sum(fz);
  if (sub == 0 && valid) {
    force[3 * a] = fx;
    force[3 * a + 1] = fy;
    force[3 * a + 2] = fz;
  }
  MIPME_ROWS_PHASE(6, true);
}

#if MIPME_ROW_LANES == 16
template <int PFAST, int BS, bool CELL = false>
__device__ __forceinline__ void sr_rows_pk_body(const FusedRowsArgs<float>& args, unsigned block,
                                                AtomRecord<float>* __restrict__ shift_tab) {
  static_assert(kRowLanes == 16, "the packed body walks 2 x 16 entries per row and iteration");
  const int64_t N = args.N;
  const int* __restrict__ row_ptr = args.row_ptr;
  const float* __restrict__ pos = args.pos;
  const float* __restrict__ cell = args.cell;
  float* __restrict__ out = args.out;
  float* __restrict__ force = args.force;
  const float c_inv2s2 = float(args.cf.inv_2s2), c1 = float(args.cf.c1), cpref = float(args.cf.pref);
  // The dependent chain of a row -- row_ptr -> entry words -> partner records -> arithmetic -- is a large part of a row
  // workgroup's lifetime (tools/wg_timeline.py: 9-14 us for 2.8 us of arithmetic), so the first loads are issued as early as
  // their addresses exist: row bounds and own position first, the shift table is built while they are in flight (moving these
  // loads behind the table's barrier costs 3 us per launch).
  const int sub = threadIdx.x % kRowLanes;
  unsigned a = block * (BS / kRowLanes) + threadIdx.x / kRowLanes;
  const bool valid = a < N;
  if (!valid) a = unsigned(N - 1);
  const int* __restrict__ rp = row_ptr + int64_t(args.row_stride) * a;
  const int r0 = rp[0], mid = rp[1], r2 = rp[2];
  // (uniform by construction; said so explicitly, or the buffer descriptor built from it lives in vector registers and every
  // entry load of the loop becomes a WATERFALL loop -- readfirstlane x 4, compares, a branch: 24 of the loop's 117 instructions,
  // which is what round 3's variable row stride had silently done to this body)
  const int n_entries = __builtin_amdgcn_readfirstlane(row_ptr[int64_t(args.row_stride) * N]);
  f2v axy;
  float az, qa;
  if (pos) {
    axy = f2v{pos[3 * a], pos[3 * a + 1]};
    az = pos[3 * a + 2];
    qa = args.q[a];
  } else {  // the records are the atoms' storage (live-bin step): own position and charge from the atom's own record
    const AtomRecord<float> own = args.rec[a];
    axy = f2v{own.x, own.y};
    az = own.z;
    qa = own.w;
  }
  {
    float A[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) A[k] = cell ? cell[k] : 0.f;
    for (int k = threadIdx.x; k < kShiftTableSize; k += BS) {
      const float sx = float(k % kShiftTableBase - kShiftTableRange),
                  sy = float((k / kShiftTableBase) % kShiftTableBase - kShiftTableRange),
                  sz = float(k / (kShiftTableBase * kShiftTableBase) - kShiftTableRange);
      shift_tab[k] = AtomRecord<float>{sx * A[0] + sy * A[3] + sz * A[6], sx * A[1] + sy * A[4] + sz * A[7],
                                       sx * A[2] + sy * A[5] + sz * A[8], 0.f};
    }
  }
  const int pot_end = args.full ? mid : 0x7fffffff;  // a full list feeds the potential from role i only
  const int beg = r0, end = valid ? r2 : r0;
  const i4v ent_rs = uniform_rsrc(raw_buffer(args.ent_sh, unsigned(n_entries) * 4u));
  // partner records through a STRUCTURED resource (stride 16, indexed loads): the atom number is the address operand as it
  // stands -- no shift per entry; an index beyond N returns zeros like a raw offset beyond the range does
  // (tools/buffer_semantics.hip)
  const i4v rec_rs = uniform_rsrc(struct_buffer(args.rec, 16u, unsigned(N)));
#if MIPME_ROWS_FROM_POS
  const i4v pos_rs = uniform_rsrc(struct_buffer(args.pos, 12u, unsigned(N)));
  const i4v q_rs = uniform_rsrc(struct_buffer(args.q, 4u, unsigned(N)));
#endif
  constexpr unsigned kAtomMask = unsigned(kCompactMaxAtoms - 1);
  // entry words one iteration ahead, partner records fetched where they are used (62 VGPRs).  Fetching the records one
  // iteration ahead as well (two register sets alternating, +10 VGPRs) changed nothing measurable: 18.3 vs 18.1 us alone,
  // 27.2 vs 26.6 us inside the co-scheduled launch (profiles/r02_experiments.txt).
  // Loop bookkeeping in SCALAR registers: every lane advances by the same 2 x 16 entries per iteration, so the lane keeps only
  // loop-invariant quantities (its byte offset into the entry stream, how many entries its row has from its first one on) and
  // the running count k32 is wave-uniform -- the entry loads take it as their scalar offset, the tail tests compare against it.
  // (Was: a per-lane offset and entry index incremented and copied every iteration, 5 of the loop's 68 vector instructions.)
  const int lane_e0 = beg + sub;
  const int voff = lane_e0 * 4;
  const int remA = end - lane_e0;          // entry eA = lane_e0 + k32 is inside the row  <=>  k32 < remA
  const int remP = pot_end - lane_e0;      // ... feeds the potential (full lists: role-i entries only)
  const int row_len = end - beg;
  int k32 = 0;
  unsigned wA = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff, 0, MIPME_ENTRY_AUX));
  unsigned wB = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff + 4 * kRowLanes, 0, MIPME_ENTRY_AUX));
  __syncthreads();  // shift table
  f2v pot2 = f2v{0.f, 0.f}, fxy = f2v{0.f, 0.f};
  float fz = 0.f;
  // CELL: C[i][(x,y)] and C[i][z] for i = x, y, z of the entry's Cartesian shift (9 sums: 3 packed + 3 scalar)
  f2v cxy[CELL ? 3 : 1];
  float cz[CELL ? 3 : 1];
#pragma unroll
  for (int k = 0; k < (CELL ? 3 : 1); ++k) {
    cxy[k] = f2v{0.f, 0.f};
    cz[k] = 0.f;
  }
  // One iteration = two entries per lane.  MASKED: the row's tail (entries beyond `end` are neutralised by two selects: d^2 := 1,
  // weight := 0) and full lists (the potential takes role-i entries only).  The unmasked form serves every iteration in which
  // ALL lanes of the wavefront have both entries inside their rows of a half list -- eight of a row's ten at cfg3 --: the
  // launch is bound by VALU issue (SQ counters: VALUBusy 81 %, profiles/r04_c_sq_counters.txt), and the selects, their compares
  // and the exec-mask loop control were 14 of the 76 vector instructions of an iteration.
  auto iteration = [&](auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
#if MIPME_ROWS_FROM_POS
    const f3v pA3 = llvm_struct_buffer_load_f3(pos_rs, int(wA & kAtomMask), 0, 0, 0);
    const f3v pB3 = llvm_struct_buffer_load_f3(pos_rs, int(wB & kAtomMask), 0, 0, 0);
    const f4v cRA = f4v{pA3.x, pA3.y, pA3.z, llvm_struct_buffer_load_f1(q_rs, int(wA & kAtomMask), 0, 0, 0)};
    const f4v cRB = f4v{pB3.x, pB3.y, pB3.z, llvm_struct_buffer_load_f1(q_rs, int(wB & kAtomMask), 0, 0, 0)};
#else
    const f4v cRA = llvm_struct_buffer_load_f4(rec_rs, int(wA & kAtomMask), 0, 0, 0);
    const f4v cRB = llvm_struct_buffer_load_f4(rec_rs, int(wB & kAtomMask), 0, 0, 0);
#endif
    // shift code -> table row: shift, then ONE shift-and-add onto the table's address (the compiler's own form is shift, mask,
    // add: the empty asm keeps it from folding the two shifts into shift + mask)
    unsigned codeA = wA >> kCompactAtomBits, codeB = wB >> kCompactAtomBits;
    asm("" : "+v"(codeA));
    asm("" : "+v"(codeB));
    const AtomRecord<float> sA = shift_tab[codeA], sB = shift_tab[codeB];
    k32 += 2 * kRowLanes;
    wA = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff, k32 * 4, MIPME_ENTRY_AUX));
    wB = unsigned(llvm_raw_buffer_load_i1(ent_rs, voff + 4 * kRowLanes, k32 * 4, MIPME_ENTRY_AUX));
    const int kk = k32 - 2 * kRowLanes;  // (uniform) entries before this iteration
    const bool okA = !MASKED || kk < remA, okB = !MASKED || kk + kRowLanes < remA;
    const f2v vA = (f2v{cRA.x, cRA.y} - axy) + f2v{sA.x, sA.y};
    const f2v vB = (f2v{cRB.x, cRB.y} - axy) + f2v{sB.x, sB.y};
    // z and the squared distances entry by entry (four scalar + five mixed instructions writing a register pair).  Left to
    // itself the SLP vectoriser pairs the two entries' z chains and x^2 + y^2 sums through six register moves (the records
    // arrive as {x, y, z, q} quads, so zA and zB are never neighbours): the empty asm statements cut its search.
    float zA = (cRA.z - az) + sA.z, zB = (cRB.z - az) + sB.z;
    asm("" : "+v"(zA));
    asm("" : "+v"(zB));
    const f2v Z = f2v{zA, zB};
    const f2v Z2 = Z * Z;
    float dA = __builtin_fmaf(vA.x, vA.x, __builtin_fmaf(vA.y, vA.y, Z2.x));
    float dB = __builtin_fmaf(vB.x, vB.x, __builtin_fmaf(vB.y, vB.y, Z2.y));
    asm("" : "+v"(dA));
    asm("" : "+v"(dB));
    const f2v d2 = MASKED ? f2v{okA ? dA : 1.f, okB ? dB : 1.f} : f2v{dA, dB};
    const f2v sv = MASKED ? f2v{okA ? cRA.w : 0.f, okB ? cRB.w : 0.f} : f2v{cRA.w, cRB.w};
    f2v v, dvd;
    fast_rs_eval_pk<PFAST>(c_inv2s2, c1, cpref, d2, v, dvd);
    if constexpr (MASKED)
      pot2 += f2v{kk < remP ? sv.x : 0.f, kk + kRowLanes < remP ? sv.y : 0.f} * v;
    else
      pot2 += sv * v;
    const f2v sc = sv * dvd;
    if constexpr (CELL) {
      const f2v tA = sc.x * vA, tB = sc.y * vB;
      const float tzA = sc.x * zA, tzB = sc.y * zB;
      fxy -= tA;
      fz -= tzA;
      fxy -= tB;
      fz -= tzB;
      // only pairs that cross a cell boundary contribute (sh = 0 otherwise), and at 9 A in a 68 A box that is no entry at all in
      // most iterations of most wavefronts: a wave-uniform branch around the twelve instructions
      constexpr unsigned kCentre = unsigned(kShiftTableRange * (1 + kShiftTableBase + kShiftTableBase * kShiftTableBase));
      if (__builtin_amdgcn_ballot_w64((okA && codeA != kCentre) || (okB && codeB != kCentre)) != 0) {
        cxy[0] += sA.x * tA; cz[0] += sA.x * tzA;
        cxy[1] += sA.y * tA; cz[1] += sA.y * tzA;
        cxy[2] += sA.z * tA; cz[2] += sA.z * tzA;
        cxy[0] += sB.x * tB; cz[0] += sB.x * tzB;
        cxy[1] += sB.y * tB; cz[1] += sB.y * tzB;
        cxy[2] += sB.z * tB; cz[2] += sB.z * tzB;
      }
    } else {
      fxy -= sc.x * vA;
      fxy -= sc.y * vB;
      fz = __builtin_fmaf(-sc.x, zA, fz);
      fz = __builtin_fmaf(-sc.y, zB, fz);
    }
  };
#if MIPME_ROWS_UNMASKED
  if (!args.full) {  // uniform
    // (a ballot over the whole wavefront: every lane is active here; lanes of rows beyond N have end == beg and fail it)
    while (__builtin_amdgcn_ballot_w64(k32 + kRowLanes < remA) == ~0ull) iteration(std::false_type{});
  }
#endif
  while (k32 < row_len) iteration(std::true_type{});
  if constexpr (CELL) {
    if (args.cpart) {  // uniform
      const int64_t w = int64_t(block) * (BS / 64) + (threadIdx.x >> 6);
      const float qv = valid ? qa : 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float vx = wave_sum_dpp(qv * cxy[i].x), vy = wave_sum_dpp(qv * cxy[i].y), vz = wave_sum_dpp(qv * cz[i]);
        if ((threadIdx.x & 63) == 0) {
          args.cpart[9 * w + 3 * i] = double(vx);
          args.cpart[9 * w + 3 * i + 1] = double(vy);
          args.cpart[9 * w + 3 * i + 2] = double(vz);
        }
      }
    }
  }
  // (DPP / readlane sums: see the fp64 body)
  const float pot = body_row_sum(pot2.x + pot2.y);
  if (sub == 0 && valid) out[a] = (args.accumulate ? out[a] : 0.f) + 0.5f * pot;
  if (args.epart) {  // see the generic body
    const bool mine = sub == 0 && valid;
    const float e1 = body_wave_sum(mine ? qa * (0.5f * pot) : 0.f);
    const float e2 = body_wave_sum(mine ? qa * qa : 0.f);
    if ((threadIdx.x & 63) == 0) {
      const int64_t w = int64_t(block) * (BS / 64) + (threadIdx.x >> 6);
      args.epart[2 * w] = double(e1);
      args.epart[2 * w + 1] = double(e2);
    }
  }
  const float fx = body_row_sum(fxy.x), fy = body_row_sum(fxy.y);
  fz = body_row_sum(fz);
  if (sub == 0 && valid) {
    force[3 * a] = fx;
    force[3 * a + 1] = fy;
    force[3 * a + 2] = fz;
  }
}

#else
// experiment builds with another row width (tools/build_variant.sh ... -DMIPME_ROW_LANES=32): no packed body, the generic one
template <int PFAST, int BS, bool CELL = false>
__device__ __forceinline__ void sr_rows_pk_body(const FusedRowsArgs<float>& args, unsigned block,
                                                AtomRecord<float>* __restrict__ shift_tab) {
  static_assert(!CELL, "the cell sums of the energy step need the packed body (MIPME_ROW_LANES == 16)");
  sr_fused_rows_body<float, kPotForce, false, PFAST, false, true, BS, 0, true>(args, block, shift_tab);
}
#endif

}  // namespace mipme
