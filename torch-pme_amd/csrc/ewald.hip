// Explicit reciprocal-space Ewald sum.
//
// Replaces EwaldCalculator._compute_kspace (reference calculators/ewald.py:76-142): the (K,N) cos / sin tables
// (`kvectors @ positions.T`, two einsums) are never materialised -- every (k, atom) phase is evaluated in registers.
//   structure factors   S_c[k,ch] = sum_i w[i,ch] cos(k r_i),  S_s[k,ch] = sum_i w[i,ch] sin(k r_i)      32 lanes per k
//   potentials          out[i,ch] = sum_k G_k (cos(k r_i) S_c[k,ch] + sin(k r_i) S_s[k,ch])             block per atom
//   position gradient   dL/dr_i   = sum_k G_k k sum_ch [ g(-s S_c + c S_s) + q(-s T_c + c T_s) ]        block per atom
//   k-vector gradient   dL/dk     = 2 dG_k k sum_ch (T_c S_c + T_s S_s) + G_k sum_i r_i sum_ch [ .. ]   32 lanes per k
// with S the structure factors of the charges q and T those of the upstream gradient g.  The 1/V factor, self /
// background / slab terms and the cell dependence of k and V stay with the caller (host layer: autograd through the
// k-vector generation).  There is no dense contraction worth MFMA at n_channels = 1 (the einsums are matrix-vector).
#include "common.h"
#include "kpot.h"

namespace mipme {

__device__ __forceinline__ void phase(float a, float& s, float& c) { sincosf(a, &s, &c); }
__device__ __forceinline__ void phase(double a, double& s, double& c) { sincos(a, &s, &c); }

static constexpr int kEwaldTile = 256;   // atoms staged per LDS tile
static constexpr int kEwaldCMax = 4;     // channels per pass

template <typename T>
__global__ __launch_bounds__(256) void ewald_filter_kernel(KPot kp, int64_t K, const T* __restrict__ kvec,
                                                          T* __restrict__ G, T* __restrict__ dG) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const double kx = double(kvec[3 * k]), ky = double(kvec[3 * k + 1]), kz = double(kvec[3 * k + 2]);
  double v, dv;
  lr_kernel_dev(kp, kx * kx + ky * ky + kz * kz, v, dv);
  G[k] = T(v);
  if (dG) dG[k] = T(dv);
}

// kEwaldKPerBlock k-vectors per block, 32 lanes each: a lane walks every 32nd atom of the LDS tile for its k-vector and the 32
// partial sums are added with a shuffle tree (fixed order: deterministic).  A thread per k-vector -- the obvious mapping --
// gives the kernel K / 64 wavefronts: 313 for 20 000 k-vectors, one per CU, and it ran 15 x longer than the potential kernel
// below, which evaluates the same N K phases with a block per atom (8 000 atoms: 1.3 ms against 0.08).
static constexpr int kEwaldKPerBlock = 8;
template <typename T>
__global__ __launch_bounds__(256) void ewald_structure_kernel(int64_t N, int C, int64_t K, const T* __restrict__ pos,
                                                             const T* __restrict__ w, const T* __restrict__ kvec,
                                                             T* __restrict__ out_c, T* __restrict__ out_s) {
  __shared__ T sp[kEwaldTile * 3];
  __shared__ T sw[kEwaldTile * kEwaldCMax];
  {  // blockIdx.y = structure of a padded batch: (B,N,3) positions, (B,N,C) weights, (B,K,3) k-vectors, (B,K,C) outputs
    const int64_t b = blockIdx.y;
    pos += b * N * 3;
    w += b * N * C;
    kvec += b * K * 3;
    out_c += b * K * C;
    out_s += b * K * C;
  }
  constexpr int AL = 256 / kEwaldKPerBlock;  // atom lanes per k-vector (32: half a wavefront)
  static_assert(AL == 32, "the reduction below adds 32 lanes");
  const int al = threadIdx.x % AL;
  const int64_t k = int64_t(blockIdx.x) * kEwaldKPerBlock + threadIdx.x / AL;
  const bool valid = k < K;
  const T kx = valid ? kvec[3 * k] : T(0), ky = valid ? kvec[3 * k + 1] : T(0), kz = valid ? kvec[3 * k + 2] : T(0);
  for (int c0 = 0; c0 < C; c0 += kEwaldCMax) {
    const int nc = min(kEwaldCMax, C - c0);
    T ac[kEwaldCMax], as[kEwaldCMax];
#pragma unroll
    for (int c = 0; c < kEwaldCMax; ++c) ac[c] = as[c] = T(0);
    for (int64_t base = 0; base < N; base += kEwaldTile) {
      const int n = int(min<int64_t>(kEwaldTile, N - base));
      __syncthreads();
      for (int t = threadIdx.x; t < 3 * n; t += 256) sp[t] = pos[3 * base + t];
      for (int t = threadIdx.x; t < n * kEwaldCMax; t += 256) {
        const int i = t / kEwaldCMax, c = t % kEwaldCMax;
        sw[t] = c < nc ? w[(base + i) * C + c0 + c] : T(0);
      }
      __syncthreads();
      for (int i = al; i < n; i += AL) {
        T s, c;
        phase(kx * sp[3 * i] + ky * sp[3 * i + 1] + kz * sp[3 * i + 2], s, c);
#pragma unroll
        for (int ch = 0; ch < kEwaldCMax; ++ch) {
          ac[ch] += sw[i * kEwaldCMax + ch] * c;
          as[ch] += sw[i * kEwaldCMax + ch] * s;
        }
      }
    }
#pragma unroll
    for (int ch = 0; ch < kEwaldCMax; ++ch) {
#pragma unroll
      for (int off = AL / 2; off > 0; off >>= 1) {
        ac[ch] += __shfl_xor(ac[ch], off, AL);
        as[ch] += __shfl_xor(as[ch], off, AL);
      }
    }
    if (valid && al == 0)
      for (int ch = 0; ch < nc; ++ch) {
        out_c[k * C + c0 + ch] = ac[ch];
        out_s[k * C + c0 + ch] = as[ch];
      }
  }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// block per atom, threads stride over the k-vectors
template <typename T>
__global__ __launch_bounds__(256) void ewald_potential_kernel(int C, int64_t K, const T* __restrict__ pos,
                                                             const T* __restrict__ kvec, const T* __restrict__ G,
                                                             const T* __restrict__ Sc, const T* __restrict__ Ss,
                                                             T* __restrict__ out) {
  __shared__ T red[4];
  {  // blockIdx.y = structure of a padded batch (gridDim.x = atoms per structure)
    const int64_t b = blockIdx.y, Nb = gridDim.x;
    pos += b * Nb * 3;
    kvec += b * K * 3;
    G += b * K;
    Sc += b * K * C;
    Ss += b * K * C;
    out += b * Nb * C;
  }
  const int64_t i = blockIdx.x;
  const T x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  for (int c0 = 0; c0 < C; c0 += kEwaldCMax) {
    const int nc = min(kEwaldCMax, C - c0);
    T acc[kEwaldCMax];
#pragma unroll
    for (int c = 0; c < kEwaldCMax; ++c) acc[c] = T(0);
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      T s, c;
      phase(kvec[3 * k] * x + kvec[3 * k + 1] * y + kvec[3 * k + 2] * z, s, c);
      const T g = G[k];
      for (int ch = 0; ch < nc; ++ch) acc[ch] += g * (c * Sc[k * C + c0 + ch] + s * Ss[k * C + c0 + ch]);
    }
    for (int ch = 0; ch < nc; ++ch) {
      const T tot = block_sum(acc[ch], red);
      if (threadIdx.x == 0) out[i * C + c0 + ch] = tot;
    }
  }
}

// B_i,k = sum_ch [ g_i (-s S_c + c S_s) + q_i (-s T_c + c T_s) ]
template <typename T>
__device__ __forceinline__ T ewald_bracket(int C, const T* __restrict__ gi, const T* __restrict__ qi,
                                           const T* __restrict__ Sc, const T* __restrict__ Ss,
                                           const T* __restrict__ Tc, const T* __restrict__ Ts, T s, T c) {
  T b = T(0);
  for (int ch = 0; ch < C; ++ch) b += gi[ch] * (c * Ss[ch] - s * Sc[ch]) + qi[ch] * (c * Ts[ch] - s * Tc[ch]);
  return b;
}

template <typename T>
__global__ __launch_bounds__(256) void ewald_grad_positions_kernel(int C, int64_t K, const T* __restrict__ pos,
                                                                  const T* __restrict__ q, const T* __restrict__ g,
                                                                  const T* __restrict__ kvec, const T* __restrict__ G,
                                                                  const T* __restrict__ Sc, const T* __restrict__ Ss,
                                                                  const T* __restrict__ Tc, const T* __restrict__ Ts,
                                                                  T* __restrict__ grad_pos) {
  __shared__ T red[4];
  {  // blockIdx.y = structure of a padded batch (gridDim.x = atoms per structure)
    const int64_t b = blockIdx.y, Nb = gridDim.x;
    pos += b * Nb * 3;
    q += b * Nb * C;
    g += b * Nb * C;
    kvec += b * K * 3;
    G += b * K;
    Sc += b * K * C;
    Ss += b * K * C;
    Tc += b * K * C;
    Ts += b * K * C;
    grad_pos += b * Nb * 3;
  }
  const int64_t i = blockIdx.x;
  const T x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
  T ax = T(0), ay = T(0), az = T(0);
  for (int64_t k = threadIdx.x; k < K; k += 256) {
    const T kx = kvec[3 * k], ky = kvec[3 * k + 1], kz = kvec[3 * k + 2];
    T s, c;
    phase(kx * x + ky * y + kz * z, s, c);
    const T b = G[k] * ewald_bracket<T>(C, g + i * C, q + i * C, Sc + k * C, Ss + k * C, Tc + k * C, Ts + k * C, s, c);
    ax += b * kx;
    ay += b * ky;
    az += b * kz;
  }
  ax = block_sum(ax, red);
  ay = block_sum(ay, red);
  az = block_sum(az, red);
  if (threadIdx.x == 0) {
    grad_pos[3 * i] = ax;
    grad_pos[3 * i + 1] = ay;
    grad_pos[3 * i + 2] = az;
  }
}

// thread per k-vector; atoms streamed through LDS (positions, q, g; single pass over the channels -> C <= kEwaldCMax
// per LDS tile, larger C loops)
template <typename T>
__global__ __launch_bounds__(256) void ewald_grad_kvectors_kernel(int64_t N, int C, int64_t K, const T* __restrict__ pos,
                                                                 const T* __restrict__ q, const T* __restrict__ g,
                                                                 const T* __restrict__ kvec, const T* __restrict__ G,
                                                                 const T* __restrict__ dG, const T* __restrict__ Sc,
                                                                 const T* __restrict__ Ss, const T* __restrict__ Tc,
                                                                 const T* __restrict__ Ts, T* __restrict__ grad_k) {
  __shared__ T sp[kEwaldTile * 3];
  __shared__ T sq[kEwaldTile * kEwaldCMax];
  __shared__ T sg[kEwaldTile * kEwaldCMax];
  {  // blockIdx.y = structure of a padded batch
    const int64_t b = blockIdx.y;
    pos += b * N * 3;
    q += b * N * C;
    g += b * N * C;
    kvec += b * K * 3;
    G += b * K;
    dG += b * K;
    Sc += b * K * C;
    Ss += b * K * C;
    Tc += b * K * C;
    Ts += b * K * C;
    grad_k += b * K * 3;
  }
  constexpr int AL = 256 / kEwaldKPerBlock;  // as ewald_structure_kernel: 32 lanes per k-vector
  const int al = threadIdx.x % AL;
  const int64_t k = int64_t(blockIdx.x) * kEwaldKPerBlock + threadIdx.x / AL;
  const bool valid = k < K;
  const int64_t kc = valid ? k : 0;
  const T kx = kvec[3 * kc], ky = kvec[3 * kc + 1], kz = kvec[3 * kc + 2];
  T ax = T(0), ay = T(0), az = T(0), dot = T(0);
  for (int c0 = 0; c0 < C; c0 += kEwaldCMax) {
    const int nc = min(kEwaldCMax, C - c0);
    T lSc[kEwaldCMax], lSs[kEwaldCMax], lTc[kEwaldCMax], lTs[kEwaldCMax];
#pragma unroll
    for (int ch = 0; ch < kEwaldCMax; ++ch) {
      const bool on = ch < nc;
      lSc[ch] = on ? Sc[kc * C + c0 + ch] : T(0);
      lSs[ch] = on ? Ss[kc * C + c0 + ch] : T(0);
      lTc[ch] = on ? Tc[kc * C + c0 + ch] : T(0);
      lTs[ch] = on ? Ts[kc * C + c0 + ch] : T(0);
      dot += lTc[ch] * lSc[ch] + lTs[ch] * lSs[ch];
    }
    for (int64_t base = 0; base < N; base += kEwaldTile) {
      const int n = int(min<int64_t>(kEwaldTile, N - base));
      __syncthreads();
      for (int t = threadIdx.x; t < 3 * n; t += 256) sp[t] = pos[3 * base + t];
      for (int t = threadIdx.x; t < n * kEwaldCMax; t += 256) {
        const int i = t / kEwaldCMax, c = t % kEwaldCMax;
        sq[t] = c < nc ? q[(base + i) * C + c0 + c] : T(0);
        sg[t] = c < nc ? g[(base + i) * C + c0 + c] : T(0);
      }
      __syncthreads();
      for (int i = al; i < n; i += AL) {
        const T x = sp[3 * i], y = sp[3 * i + 1], z = sp[3 * i + 2];
        T s, c;
        phase(kx * x + ky * y + kz * z, s, c);
        const T b = ewald_bracket<T>(kEwaldCMax, sg + i * kEwaldCMax, sq + i * kEwaldCMax, lSc, lSs, lTc, lTs, s, c);
        ax += b * x;
        ay += b * y;
        az += b * z;
      }
    }
  }
#pragma unroll
  for (int off = AL / 2; off > 0; off >>= 1) {
    ax += __shfl_xor(ax, off, AL);
    ay += __shfl_xor(ay, off, AL);
    az += __shfl_xor(az, off, AL);
  }
  if (valid && al == 0) {
    const T gk = G[k], two_dg = T(2) * dG[k] * dot;
    grad_k[3 * k] = gk * ax + two_dg * kx;
    grad_k[3 * k + 1] = gk * ay + two_dg * ky;
    grad_k[3 * k + 2] = gk * az + two_dg * kz;
  }
}

template <typename T>
static int ewald_filter_t(hipStream_t st, const mipme_potential_t* pot, int64_t K, const void* kvec, void* G, void* dG) {
  KPot kp;
  int rc = make_kpot(pot, kp);
  if (rc) return rc;
  if (K == 0) return MIPME_OK;
  ewald_filter_kernel<T><<<unsigned((K + 255) / 256), 256, 0, st>>>(kp, K, (const T*)kvec, (T*)G, (T*)dG);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int ewald_structure_t(hipStream_t st, int64_t N, int C, int64_t K, const void* pos, const void* w,
                             const void* kvec, void* out_c, void* out_s, int64_t B) {
  if (K == 0) return MIPME_OK;
  ewald_structure_kernel<T><<<dim3(unsigned((K + kEwaldKPerBlock - 1) / kEwaldKPerBlock), unsigned(B)), 256, 0, st>>>(N, C, K, (const T*)pos, (const T*)w,
                                                                      (const T*)kvec, (T*)out_c, (T*)out_s);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int ewald_potential_t(hipStream_t st, int64_t N, int C, int64_t K, const void* pos, const void* kvec,
                             const void* G, const void* Sc, const void* Ss, void* out, int64_t B) {
  if (N == 0) return MIPME_OK;
  ewald_potential_kernel<T><<<dim3(unsigned(N), unsigned(B)), 256, 0, st>>>(C, K, (const T*)pos, (const T*)kvec, (const T*)G, (const T*)Sc,
                                                        (const T*)Ss, (T*)out);
  MIPME_LAUNCH_CHECK();
  return MIPME_OK;
}

template <typename T>
static int ewald_backward_t(hipStream_t st, int64_t N, int C, int64_t K, const void* pos, const void* q, const void* g,
                            const void* kvec, const void* G, const void* dG, const void* Sc, const void* Ss,
                            const void* Tc, const void* Ts, void* grad_pos, void* grad_k, int64_t B) {
  if (grad_pos && N > 0) {
    ewald_grad_positions_kernel<T><<<dim3(unsigned(N), unsigned(B)), 256, 0, st>>>(C, K, (const T*)pos, (const T*)q, (const T*)g,
                                                               (const T*)kvec, (const T*)G, (const T*)Sc, (const T*)Ss,
                                                               (const T*)Tc, (const T*)Ts, (T*)grad_pos);
    MIPME_LAUNCH_CHECK();
  }
  if (grad_k && K > 0) {
    ewald_grad_kvectors_kernel<T><<<dim3(unsigned((K + kEwaldKPerBlock - 1) / kEwaldKPerBlock), unsigned(B)), 256, 0, st>>>(
        N, C, K, (const T*)pos, (const T*)q, (const T*)g, (const T*)kvec, (const T*)G, (const T*)dG, (const T*)Sc,
        (const T*)Ss, (const T*)Tc, (const T*)Ts, (T*)grad_k);
    MIPME_LAUNCH_CHECK();
  }
  return MIPME_OK;
}

}  // namespace mipme

using namespace mipme;

#define EW_DT(dtype, F32CALL, F64CALL)      \
  do {                                      \
    if ((dtype) == MIPME_F32) return F32CALL; \
    if ((dtype) == MIPME_F64) return F64CALL; \
    set_error("invalid dtype %d", dtype);   \
    return MIPME_EINVAL;                    \
  } while (0)

extern "C" {

int mipme_ewald_filter(void* stream, int dtype, const mipme_potential_t* pot, int64_t n_k, const void* kvectors, void* G,
                       void* dG) {
  MIPME_REQUIRE(n_k >= 0 && (n_k == 0 || (kvectors && G)), "invalid arguments to mipme_ewald_filter");
  hipStream_t st = (hipStream_t)stream;
  EW_DT(dtype, ewald_filter_t<float>(st, pot, n_k, kvectors, G, dG), ewald_filter_t<double>(st, pot, n_k, kvectors, G, dG));
}

int mipme_ewald_structure(void* stream, int dtype, int64_t n_atoms, int n_channels, int64_t n_k, const void* positions,
                          const void* weights, const void* kvectors, void* out_cos, void* out_sin, int64_t n_batch) {
  MIPME_REQUIRE(n_atoms >= 0 && n_k >= 0 && n_channels > 0 && n_batch >= 1 && n_batch <= 65535,
                "invalid sizes passed to mipme_ewald_structure");
  MIPME_REQUIRE(n_k == 0 || (kvectors && out_cos && out_sin && (n_atoms == 0 || (positions && weights))),
                "NULL buffer passed to mipme_ewald_structure");
  hipStream_t st = (hipStream_t)stream;
  EW_DT(dtype, ewald_structure_t<float>(st, n_atoms, n_channels, n_k, positions, weights, kvectors, out_cos, out_sin, n_batch),
        ewald_structure_t<double>(st, n_atoms, n_channels, n_k, positions, weights, kvectors, out_cos, out_sin, n_batch));
}

int mipme_ewald_potential(void* stream, int dtype, int64_t n_atoms, int n_channels, int64_t n_k, const void* positions,
                          const void* kvectors, const void* G, const void* s_cos, const void* s_sin, void* out,
                          int64_t n_batch) {
  MIPME_REQUIRE(n_atoms >= 0 && n_k >= 0 && n_channels > 0 && n_batch >= 1 && n_batch <= 65535,
                "invalid sizes passed to mipme_ewald_potential");
  MIPME_REQUIRE(n_atoms == 0 || (positions && out && (n_k == 0 || (kvectors && G && s_cos && s_sin))),
                "NULL buffer passed to mipme_ewald_potential");
  hipStream_t st = (hipStream_t)stream;
  EW_DT(dtype, ewald_potential_t<float>(st, n_atoms, n_channels, n_k, positions, kvectors, G, s_cos, s_sin, out, n_batch),
        ewald_potential_t<double>(st, n_atoms, n_channels, n_k, positions, kvectors, G, s_cos, s_sin, out, n_batch));
}

int mipme_ewald_backward(void* stream, int dtype, int64_t n_atoms, int n_channels, int64_t n_k, const void* positions,
                         const void* charges, const void* grad_out, const void* kvectors, const void* G, const void* dG,
                         const void* s_cos, const void* s_sin, const void* t_cos, const void* t_sin,
                         void* grad_positions, void* grad_kvectors, int64_t n_batch) {
  MIPME_REQUIRE(n_atoms >= 0 && n_k >= 0 && n_channels > 0 && n_batch >= 1 && n_batch <= 65535,
                "invalid sizes passed to mipme_ewald_backward");
  MIPME_REQUIRE(n_atoms == 0 || n_k == 0 || (positions && charges && grad_out && kvectors && G && s_cos && s_sin && t_cos && t_sin),
                "NULL buffer passed to mipme_ewald_backward");
  MIPME_REQUIRE(!grad_kvectors || dG || n_k == 0, "grad_kvectors needs dG");
  hipStream_t st = (hipStream_t)stream;
  EW_DT(dtype,
        ewald_backward_t<float>(st, n_atoms, n_channels, n_k, positions, charges, grad_out, kvectors, G, dG, s_cos, s_sin,
                                t_cos, t_sin, grad_positions, grad_kvectors, n_batch),
        ewald_backward_t<double>(st, n_atoms, n_channels, n_k, positions, charges, grad_out, kvectors, G, dG, s_cos, s_sin,
                                 t_cos, t_sin, grad_positions, grad_kvectors, n_batch));
}

}  // extern "C"
