// Round-5 microbenchmark: LDS float atomics in the access pattern of a plane spread (one lane = one atom, 5 x 5 stencil points
// of a 64 x 64 real plane stored as 64 x 33 complex, z pairs bit-reversed), against plain LDS stores of the same pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_bench lds_atomic_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <typename T, int MODE>  // MODE 0: atomic add, 1: plain store (lower bound of the pattern), 2: atomics, lanes of a wave = atoms of one 8x8 cell
__global__ __launch_bounds__(512) void scatter(const int2* __restrict__ atoms, int n_per_wg, T* out, int iters) {
  extern __shared__ char smem[];
  T* tile = (T*)smem;
  const int NY = 64, RZ = 33, LOGLZ = 5;
  for (int i = threadIdx.x; i < 2 * NY * RZ; i += blockDim.x) tile[i] = T(0);
  __syncthreads();
  const int2* mine = atoms + (size_t)blockIdx.x * n_per_wg;
  for (int it = 0; it < iters; ++it) {
    for (int a = threadIdx.x; a < n_per_wg; a += blockDim.x) {
      const int2 m = mine[a];
      int rowa[5], zo[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        int y = m.x - 2 + j; y += y < 0 ? NY : 0; y -= y >= NY ? NY : 0;
        rowa[j] = y * RZ * 2;
        int z = m.y - 2 + j; z += z < 0 ? 64 : 0; z -= z >= 64 ? 64 : 0;
        zo[j] = (int(__brev(unsigned(z >> 1)) >> (32 - LOGLZ)) << 1) | (z & 1);
      }
      const T v = T(1 + (a & 3));
      T keep = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          if (MODE == 1) tile[rowa[j] + zo[k]] = v;
          else if (MODE == 3) {
            if constexpr (sizeof(T) == 4) atomicAdd(reinterpret_cast<unsigned*>(&tile[rowa[j] + zo[k]]), unsigned(int(v * T(j + 1) * T(65536))));
            else atomicAdd(reinterpret_cast<unsigned long long*>(&tile[rowa[j] + zo[k]]), (unsigned long long)((long long)(v * T(j + 1) * T(4294967296.0))));
          } else if (MODE == 4) keep += atomicAdd(&tile[rowa[j] + zo[k]], v * T(j + 1));
          else atomicAdd(&tile[rowa[j] + zo[k]], v * T(j + 1));
        }
      if (keep == T(-3)) out[1] = keep;
    }
  }
  __syncthreads();
  T s = 0;
  for (int i = threadIdx.x; i < 2 * NY * RZ; i += blockDim.x) s += tile[i];
  if (s == T(-1)) out[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = tile[17];
}

template <typename T, int MODE>
void run(const char* name, int wgs, int n_per_wg, const std::vector<int2>& h, int iters) {
  int2* d; T* o;
  CK(hipMalloc(&d, h.size() * sizeof(int2))); CK(hipMalloc(&o, 4096 * sizeof(T)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t lds = 2 * 64 * 33 * sizeof(T);
  scatter<T, MODE><<<wgs, 512, lds>>>(d, n_per_wg, o, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 10; ++r) scatter<T, MODE><<<wgs, 512, lds>>>(d, n_per_wg, o, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
  const double ops = double(wgs) * n_per_wg * 25.0 * iters;
  printf("%-44s wgs %4d atoms/wg %5d iters %2d : %8.2f us  %7.1f G lane-ops/s  (%.1f us per 2500-atom plane pass)\n", name, wgs, n_per_wg, iters,
         ms * 1e3, ops / ms * 1e-6, ms * 1e3 / iters * 2500.0 / n_per_wg);
  CK(hipFree(d)); CK(hipFree(o));
}

int main() {
  const int n_per_wg = 2560;
  for (int wgs : {64, 256}) {
    std::vector<int2> rnd((size_t)wgs * n_per_wg), cell((size_t)wgs * n_per_wg);
    srand(3);
    for (size_t i = 0; i < rnd.size(); ++i) rnd[i] = int2{rand() % 64, rand() % 64};
    // brick-ordered: consecutive groups of 40 atoms share an 8 x 8 (y,z) cell (what the bins deliver)
    for (size_t i = 0; i < cell.size(); ++i) { const int c = int((i / 40) % 64); cell[i] = int2{(c / 8) * 8 + rand() % 8, (c % 8) * 8 + rand() % 8}; }
    for (int iters : {4}) {
      run<float, 0>("f32 ds_add, random atoms", wgs, n_per_wg, rnd, iters);
      run<float, 0>("f32 ds_add, brick-ordered atoms", wgs, n_per_wg, cell, iters);
      run<float, 1>("f32 plain store, brick-ordered", wgs, n_per_wg, cell, iters);
      run<double, 0>("f64 ds_add, brick-ordered atoms", wgs, n_per_wg, cell, iters);
      run<float, 3>("u32 ds_add (fixed point), brick-ordered", wgs, n_per_wg, cell, iters);
      run<double, 3>("u64 ds_add (fixed point), brick-ordered", wgs, n_per_wg, cell, iters);
      run<float, 4>("f32 ds_add_rtn, brick-ordered", wgs, n_per_wg, cell, iters);
    }
  }
  return 0;
}
