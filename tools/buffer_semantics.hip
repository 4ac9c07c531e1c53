// What gfx950 buffer loads return outside their resource (the pair bodies of rows_body.h lean on it):
//   1. raw load, voffset in range, voffset + soffset past num_records            -> 0 expected (the prefetch of the entry stream)
//   2. raw load, voffset + soffset in range                                      -> the element
//   3. structured load (stride 16, idxen), index < num_records                   -> the element
//   4. structured load, index >= num_records                                     -> 0 expected
// Build: hipcc --offload-arch=gfx950 -O2 tools/buffer_semantics.hip -o tools/buffer_semantics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int i4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ int raw_load_i1(i4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ f4v struct_load_f4(i4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");

__device__ i4v uniform(i4v r) {
  return i4v{__builtin_amdgcn_readfirstlane(r.x), __builtin_amdgcn_readfirstlane(r.y), __builtin_amdgcn_readfirstlane(r.z),
             __builtin_amdgcn_readfirstlane(r.w)};
}

__global__ void probe(const int* words, int n_words, const float* recs, int n_recs, int soff, int* out_i, float* out_f) {
  const uint64_t a = reinterpret_cast<uint64_t>(words), b = reinterpret_cast<uint64_t>(recs);
  const i4v raw = uniform(i4v{int(unsigned(a)), int(unsigned(a >> 32) & 0xffffu), n_words * 4, 0x00020000});
  const i4v st = uniform(i4v{int(unsigned(b)), int((unsigned(b >> 32) & 0xffffu) | (16u << 16)), n_recs, 0x00020000});
  const int t = threadIdx.x;
  out_i[t] = raw_load_i1(raw, t * 4, __builtin_amdgcn_readfirstlane(soff), 0);
  const f4v r = struct_load_f4(st, t, 0, 0, 0);
  out_f[4 * t] = r.x; out_f[4 * t + 1] = r.y; out_f[4 * t + 2] = r.z; out_f[4 * t + 3] = r.w;
}

int main() {
  const int NW = 96, NR = 40, T = 64;  // allocations are larger than the resources: an unchecked read returns the sentinel, not a fault
  std::vector<int> w(4096);
  for (int i = 0; i < 4096; ++i) w[i] = 1000 + i;
  std::vector<float> r(4096);
  for (int i = 0; i < 4096; ++i) r[i] = 0.5f + i;
  int *dw, *oi; float *dr, *of;
  hipMalloc(&dw, 4096 * 4); hipMalloc(&dr, 4096 * 4); hipMalloc(&oi, T * 4); hipMalloc(&of, T * 16);
  hipMemcpy(dw, w.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dr, r.data(), 4096 * 4, hipMemcpyHostToDevice);
  int bad = 0;
  for (int soff : {0, 128, 256, 512}) {
    probe<<<1, T>>>(dw, NW, dr, NR, soff, oi, of);
    std::vector<int> hi(T); std::vector<float> hf(T * 4);
    hipMemcpy(hi.data(), oi, T * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hf.data(), of, T * 16, hipMemcpyDeviceToHost);
    int raw_bad = 0, st_bad = 0;
    for (int t = 0; t < T; ++t) {
      const int e = t + soff / 4, want = e < NW ? 1000 + e : 0;
      if (hi[t] != want) { if (!raw_bad) printf("  raw: lane %d soffset %d got %d want %d\n", t, soff, hi[t], want); ++raw_bad; }
      for (int c = 0; c < 4; ++c) {
        const float wantf = t < NR ? 0.5f + 4 * t + c : 0.f;
        if (hf[4 * t + c] != wantf) { if (!st_bad) printf("  struct: lane %d got %g want %g\n", t, hf[4 * t + c], wantf); ++st_bad; }
      }
    }
    printf("soffset %4d: raw mismatches %d, struct mismatches %d\n", soff, raw_bad, st_bad);
    bad += raw_bad + st_bad;
  }
  printf(bad ? "BUFFER SEMANTICS: UNEXPECTED\n" : "BUFFER SEMANTICS: as assumed (out of range -> 0 with scalar offsets and with idxen)\n");
  return bad != 0;
}
