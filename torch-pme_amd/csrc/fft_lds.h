// LDS FFT building blocks shared by the (y,z) plane kernels of the convolution (kfilter.hip) and the plane spread of the
// co-scheduled launch (bricks.hip), which scatters the charges straight into the plane tile and transforms it in place.
#pragma once
#include "common.h"

namespace mipme {

template <typename T>
struct Cplx {
  T re, im;
};

__device__ __forceinline__ void unit_root(int j, int n, float& re, float& im) { sincospif(-2.0f * float(j) / float(n), &im, &re); }
__device__ __forceinline__ void unit_root(int j, int n, double& re, double& im) { sincospi(-2.0 * double(j) / double(n), &im, &re); }

template <typename T>
__device__ __forceinline__ Cplx<T> cmul(Cplx<T> a, Cplx<T> w) { return Cplx<T>{a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re}; }
template <typename T>
__device__ __forceinline__ Cplx<T> cmulc(Cplx<T> a, Cplx<T> w) { return Cplx<T>{a.re * w.re + a.im * w.im, a.im * w.re - a.re * w.im}; }
template <typename T>
__device__ __forceinline__ Cplx<T> cadd(Cplx<T> a, Cplx<T> b) { return Cplx<T>{a.re + b.re, a.im + b.im}; }
template <typename T>
__device__ __forceinline__ Cplx<T> csub(Cplx<T> a, Cplx<T> b) { return Cplx<T>{a.re - b.re, a.im - b.im}; }

template <typename T>
__device__ __forceinline__ Cplx<T> cconj(Cplx<T> a) { return Cplx<T>{a.re, -a.im}; }

// Transforms of `nbatch` sequences of length L = 2^logL in LDS: element i of sequence b at data[b * bstride + i * estride].
// DIT: bit-reversed in -> natural out;  DIF: natural in -> bit-reversed out.  tw[j] = exp(-2 pi i j / Ltab), j < Ltab / 2.
// Two radix-2 stages per pass (the thread that owns the four coupled points does both: half the barriers and LDS round
// trips of the textbook schedule, same data order), plus one single stage when logL is odd.
template <typename T, bool INVERSE>
__device__ __forceinline__ Cplx<T> tw_at(const Cplx<T>* tw, int idx) {
  Cplx<T> w = tw[idx];
  if constexpr (INVERSE) w.im = -w.im;
  return w;
}

// Work item t -> (sequence b, group r of its points).  BMAP = 0: the groups of one sequence go to consecutive lanes -- right
// for sequences that are contiguous in LDS (estride = 1: the z rows).  BMAP = 1 / 2: the SEQUENCES go to consecutive lanes
// (1: nbatch a power of two, 2: any nbatch) -- right for column transforms (bstride = 1), where consecutive sequences are
// consecutive addresses and the groups of one sequence sit whole rows apart, a power-of-two stride in the later passes that
// no row padding takes off the same banks.
#ifndef MIPME_FFT_BFAST
#define MIPME_FFT_BFAST 1
#endif
#ifndef MIPME_FFT_ZROWS
#define MIPME_FFT_ZROWS 1  // the z rows too: their pitch (nz / 2 + 1 elements) is odd, so rows on consecutive lanes are conflict free in every pass
#endif
template <int BMAP>
__device__ __forceinline__ void fft_item(int t, int log_groups, int nbatch, int& b, int& r) {
  if constexpr (BMAP == 0) {
    b = t >> log_groups;
    r = t & ((1 << log_groups) - 1);
  } else if constexpr (BMAP == 1) {
    r = t >> (31 - __clz(nbatch));
    b = t & (nbatch - 1);
  } else {
    r = t / nbatch;
    b = t - r * nbatch;
  }
}

template <typename T, bool DIT, bool INVERSE, int BMAP = 0>
__device__ __forceinline__ void lds_fft_single(Cplx<T>* data, int logL, int s, int nbatch, int bstride, int estride,
                                               const Cplx<T>* tw, int Ltab, int tid, int nthr) {
  const int half_total = nbatch << (logL - 1);
  const int hm = 1 << (s - 1), f = Ltab >> s;
  for (int t = tid; t < half_total; t += nthr) {
    int b, r;
    fft_item<BMAP>(t, logL - 1, nbatch, b, r);
    const int j = r & (hm - 1), i = ((r >> (s - 1)) << s) + j;
    Cplx<T>* p = data + b * bstride + i * estride;
    const Cplx<T> w = tw_at<T, INVERSE>(tw, j * f);
    const Cplx<T> u = p[0], v = p[hm * estride];
    if constexpr (DIT) {
      const Cplx<T> tv = cmul(v, w);
      p[0] = cadd(u, tv);
      p[hm * estride] = csub(u, tv);
    } else {
      p[0] = cadd(u, v);
      p[hm * estride] = cmul(csub(u, v), w);
    }
  }
  __syncthreads();
}

// multiply by exp(-+ 2 pi i k / 8) (upper sign: forward), k = 0..3 a compile-time constant
template <typename T, bool INVERSE, int K>
__device__ __forceinline__ Cplx<T> rot8(Cplx<T> a) {
  constexpr T r = T(0.70710678118654752440);
  if constexpr (K == 0) return a;
  if constexpr (K == 2) return INVERSE ? Cplx<T>{-a.im, a.re} : Cplx<T>{a.im, -a.re};
  if constexpr (K == 1) return INVERSE ? Cplx<T>{r * (a.re - a.im), r * (a.re + a.im)} : Cplx<T>{r * (a.re + a.im), r * (a.im - a.re)};
  return INVERSE ? Cplx<T>{-r * (a.re + a.im), r * (a.re - a.im)} : Cplx<T>{r * (a.im - a.re), -r * (a.re + a.im)};  // K == 3
}

// THREE radix-2 stages in one pass, by the thread that owns the eight points they couple (same data order as the radix-2
// schedule, a third of its barriers and LDS round trips).  Three twiddle loads per eight points: the others differ from them by
// eighth roots of unity.
// DIT, stages with half-lengths h = 2^(s-1), 2h, 4h: points p[k * st], k < 8, st = h * estride.
template <typename T, bool INVERSE>
__device__ __forceinline__ void r8_dit(Cplx<T>* p, int st, Cplx<T> w1, Cplx<T> w2, Cplx<T> w3) {
  Cplx<T> x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = p[k * st];
#pragma unroll
  for (int k = 0; k < 8; k += 2) {  // length 2h: (k, k+1), twiddle w^j
    const Cplx<T> t = cmul(x[k + 1], w1);
    x[k + 1] = csub(x[k], t);
    x[k] = cadd(x[k], t);
  }
#pragma unroll
  for (int g = 0; g < 8; g += 4) {  // length 4h: (k, k+2); the pair that starts h further carries a quarter turn
    const Cplx<T> t0 = cmul(x[g + 2], w2), t1 = rot8<T, INVERSE, 2>(cmul(x[g + 3], w2));
    x[g + 2] = csub(x[g], t0);
    x[g] = cadd(x[g], t0);
    x[g + 3] = csub(x[g + 1], t1);
    x[g + 1] = cadd(x[g + 1], t1);
  }
  {  // length 8h: (k, k+4), twiddle w^j times the k-th eighth root
    const Cplx<T> t0 = cmul(x[4], w3), t1 = rot8<T, INVERSE, 1>(cmul(x[5], w3)), t2 = rot8<T, INVERSE, 2>(cmul(x[6], w3)),
                  t3 = rot8<T, INVERSE, 3>(cmul(x[7], w3));
    p[0] = cadd(x[0], t0);
    p[4 * st] = csub(x[0], t0);
    p[st] = cadd(x[1], t1);
    p[5 * st] = csub(x[1], t1);
    p[2 * st] = cadd(x[2], t2);
    p[6 * st] = csub(x[2], t2);
    p[3 * st] = cadd(x[3], t3);
    p[7 * st] = csub(x[3], t3);
  }
}
// DIF, stages of length 8q, 4q, 2q (q = 2^(s-3)): points p[k * st], st = q * estride.
template <typename T, bool INVERSE>
__device__ __forceinline__ void r8_dif(Cplx<T>* p, int st, Cplx<T> w8, Cplx<T> w4, Cplx<T> w2) {
  Cplx<T> x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = p[k * st];
  {  // length 8q: (k, k+4)
    const Cplx<T> d0 = csub(x[0], x[4]), d1 = csub(x[1], x[5]), d2 = csub(x[2], x[6]), d3 = csub(x[3], x[7]);
    x[0] = cadd(x[0], x[4]);
    x[1] = cadd(x[1], x[5]);
    x[2] = cadd(x[2], x[6]);
    x[3] = cadd(x[3], x[7]);
    x[4] = cmul(d0, w8);
    x[5] = cmul(rot8<T, INVERSE, 1>(d1), w8);
    x[6] = cmul(rot8<T, INVERSE, 2>(d2), w8);
    x[7] = cmul(rot8<T, INVERSE, 3>(d3), w8);
  }
#pragma unroll
  for (int g = 0; g < 8; g += 4) {  // length 4q: (k, k+2)
    const Cplx<T> d0 = csub(x[g], x[g + 2]), d1 = csub(x[g + 1], x[g + 3]);
    x[g] = cadd(x[g], x[g + 2]);
    x[g + 1] = cadd(x[g + 1], x[g + 3]);
    x[g + 2] = cmul(d0, w4);
    x[g + 3] = cmul(rot8<T, INVERSE, 2>(d1), w4);
  }
#pragma unroll
  for (int k = 0; k < 8; k += 2) {  // length 2q: (k, k+1)
    const Cplx<T> d = csub(x[k], x[k + 1]);
    p[k * st] = cadd(x[k], x[k + 1]);
    p[(k + 1) * st] = cmul(d, w2);
  }
}

// radix of the passes: logL = 3a (+2: one radix-4 pass; +1: two radix-4 passes, or a single radix-2 stage for logL = 1)
#ifndef MIPME_FFT_RADIX8
#define MIPME_FFT_RADIX8 1
#endif

template <typename T, bool DIT, bool INVERSE, int BMAP = 0>
__device__ __forceinline__ void lds_fft_radix2(Cplx<T>* data, int logL, int nbatch, int bstride, int estride,
                                               const Cplx<T>* tw, int Ltab, int tid = int(threadIdx.x),
                                               int nthr = int(blockDim.x)) {
  if (logL == 1) {
    lds_fft_single<T, DIT, INVERSE, BMAP>(data, logL, 1, nbatch, bstride, estride, tw, Ltab, tid, nthr);
    return;
  }
  const int quarter_total = nbatch << (logL - 2);
  // radix-4 passes this transform takes besides its radix-8 ones
  int n4 = MIPME_FFT_RADIX8 ? ((logL % 3 == 0) ? 0 : (logL % 3 == 2 ? 1 : 2)) : (logL >> 1);
  const bool odd = !MIPME_FFT_RADIX8 && (logL & 1);
  const int eighth_total = logL >= 3 ? nbatch << (logL - 3) : 0;
  if constexpr (DIT) {
    int s = 1;  // next stage has sub-transform length 2^s
    if (odd) {
      lds_fft_single<T, true, INVERSE, BMAP>(data, logL, 1, nbatch, bstride, estride, tw, Ltab, tid, nthr);
      s = 2;
    }
    while (s <= logL) {
      if (n4 > 0) {
        --n4;
        const int h = 1 << (s - 1);          // stages of length 2h then 4h
        const int f2 = Ltab >> s, f4 = Ltab >> (s + 1);
        for (int t = tid; t < quarter_total; t += nthr) {
          int b, r;
          fft_item<BMAP>(t, logL - 2, nbatch, b, r);
          const int j = r & (h - 1), i = ((r >> (s - 1)) << (s + 1)) + j;
          Cplx<T>* p = data + b * bstride + i * estride;
          const int st = h * estride;
          const Cplx<T> x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
          const Cplx<T> w1 = tw_at<T, INVERSE>(tw, j * f2);
          const Cplx<T> b1 = cmul(x1, w1), b3 = cmul(x3, w1);
          const Cplx<T> a0 = cadd(x0, b1), a1 = csub(x0, b1), a2 = cadd(x2, b3), a3 = csub(x2, b3);
          const Cplx<T> c2 = cmul(a2, tw_at<T, INVERSE>(tw, j * f4)), c3 = cmul(a3, tw_at<T, INVERSE>(tw, (j + h) * f4));
          p[0] = cadd(a0, c2);
          p[2 * st] = csub(a0, c2);
          p[st] = cadd(a1, c3);
          p[3 * st] = csub(a1, c3);
        }
        s += 2;
      } else {
        const int h = 1 << (s - 1);          // stages of length 2h, 4h, 8h
        const int f2 = Ltab >> s, f4 = Ltab >> (s + 1), f8 = Ltab >> (s + 2);
        for (int t = tid; t < eighth_total; t += nthr) {
          int b, r;
          fft_item<BMAP>(t, logL - 3, nbatch, b, r);
          const int j = r & (h - 1), i = ((r >> (s - 1)) << (s + 2)) + j;
          r8_dit<T, INVERSE>(data + b * bstride + i * estride, h * estride, tw_at<T, INVERSE>(tw, j * f2),
                             tw_at<T, INVERSE>(tw, j * f4), tw_at<T, INVERSE>(tw, j * f8));
        }
        s += 3;
      }
      __syncthreads();
    }
  } else {
    int s = logL;  // current sub-transform length 2^s
    // the mirror image of the DIT schedule: radix-8 passes first, the radix-4 ones last
    while (s >= 2) {
      const int left8 = (s - 2 * n4) / 3;  // radix-8 passes still to come
      if (MIPME_FFT_RADIX8 && left8 > 0) {
        const int q = 1 << (s - 3);          // stages of length 8q, 4q, 2q
        const int f8 = Ltab >> s, f4 = Ltab >> (s - 1), f2 = Ltab >> (s - 2);
        for (int t = tid; t < eighth_total; t += nthr) {
          int b, r;
          fft_item<BMAP>(t, logL - 3, nbatch, b, r);
          const int j = r & (q - 1), i = ((r >> (s - 3)) << s) + j;
          r8_dif<T, INVERSE>(data + b * bstride + i * estride, q * estride, tw_at<T, INVERSE>(tw, j * f8),
                             tw_at<T, INVERSE>(tw, j * f4), tw_at<T, INVERSE>(tw, j * f2));
        }
        s -= 3;
      } else {
        const int q = 1 << (s - 2);          // stages of length 4q then 2q
        const int f4 = Ltab >> s, f2 = Ltab >> (s - 1);
        for (int t = tid; t < quarter_total; t += nthr) {
          int b, r;
          fft_item<BMAP>(t, logL - 2, nbatch, b, r);
          const int j = r & (q - 1), i = ((r >> (s - 2)) << s) + j;
          Cplx<T>* p = data + b * bstride + i * estride;
          const int st = q * estride;
          const Cplx<T> x0 = p[0], x1 = p[st], x2 = p[2 * st], x3 = p[3 * st];
          const Cplx<T> u0 = cadd(x0, x2), u2 = cmul(csub(x0, x2), tw_at<T, INVERSE>(tw, j * f4));
          const Cplx<T> u1 = cadd(x1, x3), u3 = cmul(csub(x1, x3), tw_at<T, INVERSE>(tw, (j + q) * f4));
          const Cplx<T> w2 = tw_at<T, INVERSE>(tw, j * f2);
          p[0] = cadd(u0, u1);
          p[st] = cmul(csub(u0, u1), w2);
          p[2 * st] = cadd(u2, u3);
          p[3 * st] = cmul(csub(u2, u3), w2);
        }
        s -= 2;
        if (n4 > 0) --n4;
      }
      __syncthreads();
    }
    if (s == 1) lds_fft_single<T, false, INVERSE, BMAP>(data, logL, 1, nbatch, bstride, estride, tw, Ltab, tid, nthr);
  }
}

// ---- (y, z) plane transform of one workgroup (see kfilter.hip for the conventions) ---------------------------------------
// LDS layout of a plane: tile[ny][RZ] complex (RZ = nz/2 + 1), then the twiddles tw[Ltab/2] and twr[nz/2 + 1].
template <typename T>
struct YzTile {
  Cplx<T>* tile;
  Cplx<T>* tw;
  Cplx<T>* twr;
  int Ltab, Lz, RZ;
};
template <typename T, bool YSTAGE = true>
__device__ __forceinline__ YzTile<T> yz_tile_setup(int ny, int nz, char* smem_yz, char* smem_tw = nullptr) {
  YzTile<T> t;
  t.Lz = nz >> 1;
  t.RZ = t.Lz + 1;
  t.tile = reinterpret_cast<Cplx<T>*>(smem_yz);      // [ny][RZ]
  // exp(-2 pi i j / Ltab), j < Ltab / 2: behind the tile, or where the caller says
  t.tw = smem_tw ? reinterpret_cast<Cplx<T>*>(smem_tw) : t.tile + size_t(ny) * t.RZ;
  t.Ltab = (YSTAGE && ny > t.Lz) ? ny : t.Lz;
  t.twr = t.tw + (t.Ltab >> 1);                      // exp(-2 pi i k / nz), k <= nz / 2 (split / merge steps)
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int j = tid; j < (t.Ltab >> 1); j += nthr) unit_root(j, t.Ltab, t.tw[j].re, t.tw[j].im);
  for (int k = tid; k <= t.Lz; k += nthr) unit_root(k, nz, t.twr[k].re, t.twr[k].im);
  return t;
}
// where the real sample (y, z) of the plane sits in the tile for the forward transform, counted in reals: rows as complex
// sequences c_j = a_2j + i a_2j+1, stored bit-reversed for the DIT z transform
__device__ __forceinline__ int yz_real_slot(int y, int z, int RZ, int loglz) {
  const int j = z >> 1;
  const int jr = loglz ? int(__brev(unsigned(j)) >> (32 - loglz)) : 0;
  return ((y * RZ + jr) << 1) | (z & 1);
}
// forward transform of a tile that holds the real plane (yz_real_slot layout; barrier done by the caller) and store to `dst`
// (the plane's ny x RZ block of the half-complex mesh)
template <typename T, bool YSTAGE = true>
__device__ __forceinline__ void yz_forward_finish(const YzTile<T>& t, int ny, int nz, int logny, int loglz, Cplx<T>* __restrict__ dst) {
  Cplx<T>* tile = t.tile;
  const Cplx<T>*tw = t.tw, *twr = t.twr;
  const int Lz = t.Lz, RZ = t.RZ, Ltab = t.Ltab;
  const int tid = threadIdx.x, nthr = blockDim.x;
  lds_fft_radix2<T, true, false, MIPME_FFT_ZROWS>(tile, loglz, ny, RZ, 1, tw, Ltab);
  // split step, pairs (k, Lz - k):  A_k = E_k + e^{-2 pi i k / nz} O_k,  E = (C_k + conj C_{Lz-k}) / 2,  O = -i (C_k - conj C_{Lz-k}) / 2
  for (int idx = tid; idx < ny * (Lz / 2 + 1); idx += nthr) {
    const int y = idx / (Lz / 2 + 1), k = idx - y * (Lz / 2 + 1);
    Cplx<T>* row = tile + y * RZ;
    const int k2 = Lz - k;
    const Cplx<T> ck = row[k == Lz ? 0 : k], cm = row[k2 == Lz ? 0 : k2];
    const Cplx<T> wk = twr[k], wm = twr[k2];
    auto split = [](Cplx<T> a, Cplx<T> b, Cplx<T> w) {  // a = C_k, b = C_{Lz-k}
      const Cplx<T> e{T(0.5) * (a.re + b.re), T(0.5) * (a.im - b.im)};
      const Cplx<T> d{T(0.5) * (a.re - b.re), T(0.5) * (a.im + b.im)};  // (C_k - conj C_{Lz-k}) / 2
      const Cplx<T> o{d.im, -d.re};                                      // -i d
      return cadd(e, cmul(o, w));
    };
    const Cplx<T> ak = split(ck, cm, wk), am = split(cm, ck, wm);
    row[k] = ak;
    row[k2] = am;
  }
  __syncthreads();
  // columns: DIF along y (natural in, bit-reversed out); the store undoes the bit reversal
  if constexpr (YSTAGE) lds_fft_radix2<T, false, false, MIPME_FFT_BFAST ? 2 : 0>(tile, logny, RZ, 1, RZ, tw, Ltab);
  for (int idx = tid; idx < ny * RZ; idx += nthr) {
    const int y = idx / RZ, k = idx - y * RZ;
    const int yr = YSTAGE ? int(__brev(unsigned(y)) >> (32 - logny)) : y;
    dst[int64_t(yr) * RZ + k] = tile[idx];
  }
}

// YSTAGE = false: only the z rows are transformed (`ny` is then just the number of rows this workgroup takes, y stays in natural
// order, blockIdx.x = row group): the first / last of the two launches for planes that do not fit LDS (split_yz).
template <typename T, bool INVERSE, bool YSTAGE = true>
__device__ __forceinline__ void yz_plane_body(int ny, int nz, int logny, int loglz, const T* __restrict__ real_in,
                                              Cplx<T>* __restrict__ hat, T* __restrict__ real_out, int64_t plane,
                                              char* smem_yz) {
  const YzTile<T> yt = yz_tile_setup<T, YSTAGE>(ny, nz, smem_yz);
  const int Lz = yt.Lz, RZ = yt.RZ, Ltab = yt.Ltab;
  Cplx<T>* tile = yt.tile;
  const Cplx<T>*tw = yt.tw, *twr = yt.twr;
  const int tid = threadIdx.x, nthr = blockDim.x;
  if constexpr (!INVERSE) {
    // rows: c_j = a_2j + i a_2j+1, stored bit-reversed for the DIT z transform
    const T* src = real_in + plane * int64_t(ny) * nz;
    for (int idx = tid; idx < ny * Lz; idx += nthr) {
      const int y = idx / Lz, j = idx - y * Lz;
      const int jr = loglz ? int(__brev(unsigned(j)) >> (32 - loglz)) : 0;
      tile[y * RZ + jr] = reinterpret_cast<const Cplx<T>*>(src)[idx];  // (a_2j, a_2j+1): the plane as ny x Lz pairs
    }
    __syncthreads();
    yz_forward_finish<T, YSTAGE>(yt, ny, nz, logny, loglz, hat + plane * int64_t(ny) * RZ);
  } else {
    const Cplx<T>* src = hat + plane * int64_t(ny) * RZ;
    for (int idx = tid; idx < ny * RZ; idx += nthr) {
      const int y = idx / RZ, k = idx - y * RZ;
      const int yr = YSTAGE ? int(__brev(unsigned(y)) >> (32 - logny)) : y;
      tile[yr * RZ + k] = src[idx];
    }
    __syncthreads();
    if constexpr (YSTAGE) lds_fft_radix2<T, true, true, MIPME_FFT_BFAST ? 2 : 0>(tile, logny, RZ, 1, RZ, tw, Ltab);
    // merge step (un-normalised: twice the textbook one):  C_k = (A_k + conj A_{Lz-k}) + i e^{+2 pi i k / nz} (A_k - conj A_{Lz-k})
    for (int idx = tid; idx < ny * (Lz / 2 + 1); idx += nthr) {
      const int y = idx / (Lz / 2 + 1), k = idx - y * (Lz / 2 + 1);
      Cplx<T>* row = tile + y * RZ;
      const int k2 = Lz - k;
      Cplx<T> ak = row[k], am = row[k2];
      if (k == 0) {
        // a complex-to-real transform ignores the imaginary parts of the k_z = 0 and Nyquist entries (they vanish for a
        // Hermitian input; G on the Nyquist plane of a triclinic cell is not exactly symmetric, so they do not here)
        ak.im = T(0);
        am.im = T(0);
      }
      const Cplx<T> wk = twr[k], wm = twr[k2];
      auto merge = [](Cplx<T> a, Cplx<T> b, Cplx<T> w) {  // a = A_k, b = A_{Lz-k}, w = e^{-2 pi i k / nz}
        const Cplx<T> e{a.re + b.re, a.im - b.im};
        const Cplx<T> d{a.re - b.re, a.im + b.im};      // A_k - conj A_{Lz-k}
        const Cplx<T> dw = cmulc(d, w);                  // times e^{+2 pi i k / nz}
        return Cplx<T>{e.re - dw.im, e.im + dw.re};      // e + i dw
      };
      const Cplx<T> ck = merge(ak, am, wk), cm = merge(am, ak, wm);
      if (k < Lz) row[k] = ck;  // C has Lz entries: index Lz is the alias of 0
      if (k2 < Lz && k2 != k) row[k2] = cm;
    }
    __syncthreads();
    // rows: DIF along z (natural in, bit-reversed out), read back through the bit reversal
    lds_fft_radix2<T, false, true, MIPME_FFT_ZROWS>(tile, loglz, ny, RZ, 1, tw, Ltab);
    T* dst = real_out + plane * int64_t(ny) * nz;
    for (int idx = tid; idx < ny * Lz; idx += nthr) {
      const int y = idx / Lz, j = idx - y * Lz;
      const int jr = loglz ? int(__brev(unsigned(j)) >> (32 - loglz)) : 0;
      reinterpret_cast<Cplx<T>*>(dst)[idx] = tile[y * RZ + jr];
    }
  }
}

}  // namespace mipme
