// Internal helpers shared by the libmipme translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstdio>

#include "../../include/mipme.h"

namespace mipme {

void set_error(const char* fmt, ...);
// name (without template arguments) of the last co-scheduled spread + pair-sum kernel this thread launched or captured
// (mipme_last_cosched_kernel: what a benchmark labels its dominant launch with)
void note_cosched_kernel(const char* name);

#define MIPME_CHECK_HIP(expr)                                                                 \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      mipme::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MIPME_EHIP;                                                                      \
    }                                                                                         \
  } while (0)

#define MIPME_REQUIRE(cond, ...)     \
  do {                               \
    if (!(cond)) {                   \
      mipme::set_error(__VA_ARGS__); \
      return MIPME_EINVAL;           \
    }                                \
  } while (0)

#define MIPME_LAUNCH_CHECK() MIPME_CHECK_HIP(hipGetLastError())

// Geometry handed to kernels by value (fits the kernarg segment; no device-side cell read).
struct Geom {
  double inv[9];  // inverse cell, row-major: u_d = n_d * sum_c r_c inv[c][d]
  int nx, ny, nz;
};

inline Geom make_geom(const mipme_mesh_t* m) {
  Geom g;
  for (int i = 0; i < 9; ++i) g.inv[i] = m->inv_cell[i];
  g.nx = m->nx;
  g.ny = m->ny;
  g.nz = m->nz;
  return g;
}

inline int validate_mesh(const mipme_mesh_t* m) {
  MIPME_REQUIRE(m != nullptr, "mesh descriptor is NULL");
  MIPME_REQUIRE(m->nx > 0 && m->ny > 0 && m->nz > 0, "mesh sizes must be positive, got %d %d %d", m->nx, m->ny, m->nz);
  MIPME_REQUIRE(m->n_channels > 0, "n_channels must be positive");
  if (m->scheme == MIPME_LAGRANGE) {
    MIPME_REQUIRE(m->order >= 3 && m->order <= 7,
                  "`interpolation_nodes` is %d but only values from 3 to 7 for method 'Lagrange' are allowed", m->order);
  } else if (m->scheme == MIPME_P3M) {
    MIPME_REQUIRE(m->order >= 1 && m->order <= 5,
                  "`interpolation_nodes` is %d but only values from 1 to 5 for method 'P3M' are allowed", m->order);
  } else {
    MIPME_REQUIRE(false, "unknown interpolation scheme %d", m->scheme);
  }
  return MIPME_OK;
}

// Zero-fill with our own kernel instead of hipMemsetAsync: a memset NODE captured into a HIP graph stopped zeroing
// its buffer once an RCCL collective had run after the capture (ROCm 7.0 runtime under PyTorch; found with
// tools/dist_probe.py -- the stale brick counters then sent the binning kernels out of bounds).  A kernel node is
// immune, and the fill is as fast (the buffers are <= a few MB).
static __global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n_words) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_words; i += size_t(gridDim.x) * blockDim.x) p[i] = 0u;
}

inline hipError_t zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  // every buffer zeroed by the library is 4-byte aligned and a multiple of 4 bytes long
  const size_t n_words = bytes / 4;
  size_t blocks = (n_words + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  zero_words_kernel<<<unsigned(blocks), 256, 0, st>>>(static_cast<uint32_t*>(ptr), n_words);
  return hipGetLastError();
}

// Hardware float atomics (global_atomic_add_f32 / _f64 on gfx950; memory is coarse-grained hipMalloc).
__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

__device__ __forceinline__ int posmod(int a, int n) {
  int r = a % n;
  return r < 0 ? r + n : r;
}

// ---------------------------------------------------------------------------------------------
// 1-D interpolation weights, generated from their defining rules rather
// than from expanded polynomial tables (the reference stores those: lib/mesh_interpolator.py:156-301).
//   P3M      : w_t(x) = M_n(t - (n-1)/2 - x), centred cardinal B-spline of order n.
//              With f = x + 1/2 and N_k the cardinal B-spline on [0,k]: w_t = N_n(f + n-1-t),
//              N_k(y) = [y N_{k-1}(y) + (k-y) N_{k-1}(y-1)]/(k-1),  N_n'(y) = N_{n-1}(y) - N_{n-1}(y-1).
//   Lagrange : nodes xi_t = t - (n-1)/2,  w_t(x) = prod_{s!=t} (x - xi_s)/(xi_t - xi_s).
// x in [-1/2, 1/2].  `DERIV` also fills dw/dx.
// ---------------------------------------------------------------------------------------------
// One Cox-de Boor step a[j] = N_K(f + j) from a[j] = N_{K-1}(f + j); K is a compile-time constant so that every
// array index is static and the arrays live in VGPRs (a runtime-bounded loop nest sent them to scratch memory).
template <int K, int N, typename T>
__device__ __forceinline__ void bspline_step(T f, T (&a)[N]) {
  constexpr T inv = T(1) / T(K - 1);
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
    const T lo = (j < K - 1) ? a[j] : T(0);
    const T hi = (j >= 1) ? a[j >= 1 ? j - 1 : 0] : T(0);
    a[j] = ((f + T(j)) * lo + (T(K - j) - f) * hi) * inv;
  }
}

template <int K, int N, typename T>
__device__ __forceinline__ void bspline_raise(T f, T (&a)[N], T (&b)[N]) {
  if constexpr (K <= N) {
    if constexpr (K == N) {
#pragma unroll
      for (int j = 0; j < N; ++j) b[j] = (j < N - 1) ? a[j] : T(0);
    }
    bspline_step<K, N, T>(f, a);
    bspline_raise<K + 1, N, T>(f, a, b);
  }
}

template <int SCHEME, int N, bool DERIV, typename T>
__device__ __forceinline__ void weights_1d(T x, T (&w)[N], T (&dw)[N]) {
  if constexpr (SCHEME == MIPME_P3M) {
    if constexpr (N == 1) {
      w[0] = T(1);
      dw[0] = T(0);
    } else {
      const T f = x + T(0.5);
      T a[N];
      T b[N];
      a[0] = T(1);
#pragma unroll
      for (int j = 1; j < N; ++j) a[j] = T(0);
#pragma unroll
      for (int j = 0; j < N; ++j) b[j] = T(0);
      bspline_raise<2, N, T>(f, a, b);
#pragma unroll
      for (int t = 0; t < N; ++t) {
        w[t] = a[N - 1 - t];
        if constexpr (DERIV) dw[t] = b[N - 1 - t] - ((N - 2 - t >= 0) ? b[(N - 2 - t >= 0) ? N - 2 - t : 0] : T(0));
      }
    }
  } else {
    T e[N];
#pragma unroll
    for (int s = 0; s < N; ++s) e[s] = x - (T(s) - T(0.5) * T(N - 1));
#pragma unroll
    for (int t = 0; t < N; ++t) {
      // den = prod_{s != t} (t - s) = (-1)^(N-1-t) t! (N-1-t)!
      double den = 1.0;
#pragma unroll
      for (int s = 0; s < N; ++s)
        if (s != t) den *= double(t - s);
      T p = T(1), dp = T(0);
#pragma unroll
      for (int s = 0; s < N; ++s) {
        if (s != t) {
          if constexpr (DERIV) dp = dp * e[s] + p;
          p = p * e[s];
        }
      }
      const T rden = T(1.0 / den);
      w[t] = p * rden;
      if constexpr (DERIV) dw[t] = dp * rden;
    }
  }
}

// Register-array select with a runtime index, as a SUM of masked elements.  The obvious select chain
// (r = idx == t ? a[t] : r) is turned back by LLVM into an indexed load of a private array, which it then promotes to LDS; the
// per-thread slice index of that needs the workgroup size, and every wave reads it from the AQL dispatch packet in HOST memory
// (3-16 us for the first load of a workgroup: profiles/r03_experiments.txt item 4; tell-tale: "LDS Size" above the kernel's
// __shared__ declarations in -Rpass-analysis=kernel-resource-usage).  A sum of selects has no such lowering.
template <int N, typename T>
__device__ __forceinline__ T pick(const T (&a)[N], int idx) {
  T r = T(0);
#pragma unroll
  for (int t = 0; t < N; ++t) r += (idx == t) ? a[t] : T(0);
  return r;
}

// Fractional mesh coordinate -> base index m and offset x in [-1/2,1/2]
// (lib/mesh_interpolator.py:326-341: even n -> floor, x = u-(m+1/2); odd n -> round-half-even, x = u-m).
template <int N>
__device__ __forceinline__ void split_coordinate(double u, int& m, double& x) {
  if constexpr (N % 2 == 0) {
    const double fl = floor(u);
    m = int(fl);
    x = u - (fl + 0.5);
  } else {
    const double r = rint(u);  // round-half-even
    m = int(r);
    x = u - r;
  }
}

// first stencil offset: range(1-(n+1)//2, 1+n//2)
template <int N>
__device__ __forceinline__ constexpr int stencil_start() {
  return 1 - (N + 1) / 2;
}

template <int N>
struct StencilGroup {
  // lanes cooperating on one atom: (ty,tz) plane of the stencil, padded to a power of two
  static constexpr int PLANE = N * N;
  static constexpr int LANES = PLANE <= 1 ? 1 : PLANE <= 4 ? 4 : PLANE <= 16 ? 16 : PLANE <= 32 ? 32 : 64;
};

// dispatch on (scheme, order) to compile-time S, N (mesh.hip, jets.hip)
#define MIPME_DISPATCH_STENCIL(SCHEME_V, ORDER_V, BODY)                                   \
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

// Plane lists of the plane spread (bricks.hip) come in kPlaneSub sub-lists per x plane; their live counters follow the brick
// counters in one int32 buffer.  plan_counter_words: the words of that buffer -- bricks + their overflow counter + the plane
// sub-lists + their overflow counter -- for everyone who allocates or clears it (fft_plan_create, CounterGuard, the frames).
static constexpr int kPlaneSub = 8;
static inline size_t plan_counter_words(int nx, int ny, int nz) {
  return size_t((nx + 7) / 8) * size_t((ny + 7) / 8) * size_t((nz + 7) / 8) + 1 + size_t(kPlaneSub) * size_t(nx) + 1;
}

// Host-side description of the gather's tail (energy + force assembly in the gather launch, csrc/bricks.hip GatherTail)
struct GatherTailHost {
  const void* force;   // (N,3) pair force sums
  double force_scale;  // 1/2 for a full list
  const void* seed;    // device scalar, nullable
  void* grad_pos;      // (N,3)
  void* energy;        // 1 real
  const void* epart_k; // fp64[n_k]: per-workgroup sums of mu G |rho^|^2 written by the x stage of the convolution
  int64_t n_k;
  int sr_reduced;      // the x stage has also reduced the pair kernel's per-wave partial sums to epart_k[n_k + 2 b ...]
  // the rest of the autograd contract of E = sum q V from the same launch (all nullable; see mipme_kspace_forward_args_t)
  void* grad_q;        // (N): seed * dE/dq = 2 seed V
  double* rpart;       // fp64[9 * bricks]: per-brick sums r_a (x) (seed q_a field_a) for the cell gradient
  const void* records; // (N,4) reals x, y, z, q: the atoms' positions for rpart
  const void* aux_seed; // device scalar, nullable (= seed): the factor of grad_q and of the cell gradient
  const void* live_flags; // live-bin step: its pinned flag word (bit 1 = an atom beyond the margin -> the energy becomes NaN)
  // frame farm (mipme.h, energy_log): the energy is also appended to a float64 log, slot = cursor mod capacity (nullable)
  double* elog;
  int* elog_cursor;
  int elog_cap;
};

// The cell gradient of an energy step inside the fused convolution (kfilter.hip, convolve_xfused): the x stage stores
// w = mu |rho^|^2 per k-point, and rider workgroups of the launch behind it form the sums against the filter's derivative table
// (+ slices of the pair kernel's cell sums and of the x stage's energy sums): one row of 25 doubles per rider
struct ConvCell {
  const void* G_deriv;  // kfilter_deriv_kernel table
  const double* cwave;  // per-wave cell partial sums of the co-scheduled pair kernel (9 per wave), nullable
  int64_t n_waves;
  void* wbuf;           // (nx, ny, nz/2+1) reals
  double* rows;         // [n_riders][25]
  int n_riders;         // 0: no riders (only rho_hat_out)
  void* rho_hat_out;    // nullable: rfftn(mesh_in), natural layout, stored by the x stage (forward pass, for a later backward)
  const void* rho_hat_in;  // nullable: the general adjoint -- w = mu Re[rho^ conj psi^] instead of mu |rho^|^2
  double* kh_rows;      // nullable: 12 sums K[c][d], H[c] per rider, the rows cellgrad_finalize_kernel reads
  int* ticket;          // ... and its ticket counter, cleared by the first rider
};

// Row workgroups of the co-scheduled pair sum that ride on the persistent convolution launch instead of the spread launch:
// rows (atoms) [first_row, first_row + n_rows), first_row a multiple of 64.
// what spread_bricks (bricks.hip) needs to use the plane spread -- the charges scattered straight into the tiles of the
// convolution's forward (y,z) transform (api.hip fills it from the plan, see fft_plan_plane_forward_ok)
struct PlaneHost {
  void* hat = nullptr;    // (nx, ny, nz/2 + 1) complex: receives the transformed planes
  bool keep_mesh = true;  // the caller reads the real charge mesh afterwards: no plane spread (it does not form it)
  bool slot_values = false;  // the binning pass of this call wrote the spread's values (single-channel charges) by bin slot
  // several workgroups per plane, each with a part of the plane's atoms and a transform of its own: part 0 -> hat, part k ->
  // hat_more + (k - 1) * more_stride (complex values); the x stage of the convolution adds them up
  int parts = 1;
  void* hat_more = nullptr;
  int64_t more_stride = 0;
  // out: how the spread left `hat` -- parts_used partial transforms that add up, and (planes spread in bands of rows) only the z
  // rows transformed: the convolution runs the y columns before its x stage (fft_plan_set_forward_ycols)
  mutable int parts_used = 1;
  mutable bool ycols_pending = false;
};
struct RowRideHost {
  const mipme_sr_job_t* job;
  void* epart;  // per-wave energy partial sums (bins buffer), nullable
  int64_t first_row, n_rows;
};

// (x, y, z, w) per atom, 16-byte aligned: one gather per entry of the fused pair kernels fetches the partner's position
// and charge (or source value).  Written by pack_atom_records_kernel (topology.hip) or by the binning pass (bricks.hip).
// XCD-aware workgroup -> work-item mapping.  The dispatcher hands consecutive workgroups of a launch to the eight XCDs round
// robin, so neighbouring items (bricks of the mesh, blocks of rows of the pair list) land on eight different L2s and each L2
// ends up caching the whole working set.  With item = (wg % 8) * ceil(n / 8) + wg / 8 the workgroups ONE XCD receives cover
// a contiguous eighth of the items: halo tiles of neighbouring bricks and partner records of neighbouring rows are re-used in
// that XCD's L2.  The launch is padded to a multiple of 8 workgroups; the ones mapped beyond n exit.  (A bijection whatever
// the real dispatch order is -- only the locality depends on it.)
__host__ __device__ __forceinline__ unsigned xcd_contiguous(unsigned wg, unsigned n_items) {
  const unsigned chunk = (n_items + 7u) >> 3;
  return (wg & 7u) * chunk + (wg >> 3);
}
__host__ __device__ __forceinline__ unsigned pad8(unsigned n) { return (n + 7u) & ~7u; }

// Conditional launches without a host round trip (mipme_set_skip_flag, include/mipme.h): a device int32 that the kernels of the
// general backward path read first; 1 = return at once.  Per host thread; kernels that do not take the pointer simply run.
inline const int*& skip_flag_slot() {
  static thread_local const int* p = nullptr;
  return p;
}
#define MIPME_SKIP_IF_SET(ptr)                          \
  do {                                                  \
    if ((ptr) != nullptr && *(ptr) == 1) return;        \
  } while (0)

// boolean switch from the environment ("0" = off, anything else = on), for A/B measurements of kernel variants
inline bool env_flag(const char* name, bool dflt) {
  const char* e = getenv(name);
  return e ? (e[0] != '0') : dflt;
}

// Wave-wide sums without LDS traffic: four DPP steps inside every row of 16 lanes (quad permutes, then the half-row and row
// mirrors -- after the quad steps the lanes of a quad agree, so a mirror serves as the xor exchange), then the four row sums
// through v_readlane.  `__shfl_xor` is ds_bpermute: six LDS operations per value (twelve for a double), which for the 25-value
// rows of the cell riders was most of an 18 us launch.  Result in every lane.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float read_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <typename T>
__device__ __forceinline__ T row16_sum_dpp(T v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
  v = row16_sum_dpp(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

template <typename T>
struct alignas(4 * sizeof(T)) AtomRecord {
  T x, y, z, w;
};

}  // namespace mipme
