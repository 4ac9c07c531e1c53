// Microbenchmark: scattered float atomic adds at agent vs workgroup scope, and random 4-byte gathers.
// Build: hipcc --offload-arch=gfx950 -O3 -o atomic_bench atomic_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SCOPE>
__global__ void scatter_add(const int* __restrict__ idx, long n, float* out, long stride_copy) {
  unsigned xcc = 0;
  if (stride_copy) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
  }
  float* o = out + xcc * stride_copy;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
    int j = idx[p];
    if (SCOPE == 0) unsafeAtomicAdd(o + j, 1.0f);
    else if (SCOPE == 1) __hip_atomic_fetch_add(o + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(o + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void gather4(const int* __restrict__ idx, long n, const float* __restrict__ src, float* out) {
  float acc = 0.f;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) acc += src[idx[p]];
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const long n = 28500000;
  for (long tbl : {32768L, 98304L, 4750000L, 14250000L}) {
    std::vector<int> h(n);
    srand(1);
    for (long i = 0; i < n; ++i) h[i] = (int)(((long)rand() * 32768L + rand()) % tbl);
    int* d_idx; float* d_out; float* d_src;
    CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&d_out, tbl * 4 * 8)); CK(hipMalloc(&d_src, tbl * 4));
    CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0, tbl * 4 * 8)); CK(hipMemset(d_src, 0, tbl * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      printf("table %8ld  %-34s %8.3f ms  %7.2f G ops/s\n", tbl, name, ms, n / ms * 1e-6);
    };
    run("atomic agent (unsafeAtomicAdd)", [&] { scatter_add<0><<<4096, 256>>>(d_idx, n, d_out, 0); });
    run("atomic agent (__hip_atomic)", [&] { scatter_add<2><<<4096, 256>>>(d_idx, n, d_out, 0); });
    run("atomic workgroup scope, shared", [&] { scatter_add<1><<<4096, 256>>>(d_idx, n, d_out, 0); });
    run("atomic workgroup scope, per-XCD copy", [&] { scatter_add<1><<<4096, 256>>>(d_idx, n, d_out, tbl); });
    run("random 4B gather", [&] { gather4<<<4096, 256>>>(d_idx, n, d_src, d_out); });
    // verify the per-XCD privatised sum
    CK(hipMemset(d_out, 0, tbl * 4 * 8));
    scatter_add<1><<<4096, 256>>>(d_idx, n, d_out, tbl);
    CK(hipDeviceSynchronize());
    std::vector<float> o(tbl * 8);
    CK(hipMemcpy(o.data(), d_out, tbl * 4 * 8, hipMemcpyDeviceToHost));
    double tot = 0; for (float v : o) tot += v;
    printf("table %8ld  per-XCD privatised total = %.0f (expected %ld)\n", tbl, tot, n);
    CK(hipFree(d_idx)); CK(hipFree(d_out)); CK(hipFree(d_src));
  }
  return 0;
}
