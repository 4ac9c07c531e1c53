// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the pair kernel (round-1 verdict
// item 5c): known byte counts, streamed at 4 / 8 / 16 bytes per lane, random 16-byte gathers from a small (L2 / MALL
// resident) and a large table, and 4-byte streaming stores.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib
// tools/fetch_calib.hip ; run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and again with WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <typename V>
__global__ void stream_read(const V* __restrict__ in, int64_t n, float* __restrict__ sink) {
  float acc = 0.f;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const V v = in[i];
    acc += reinterpret_cast<const float*>(&v)[0];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

__global__ void gather16(const int* __restrict__ idx, int64_t n, const float4* __restrict__ table, float* __restrict__ sink) {
  float acc = 0.f;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) acc += table[idx[i]].x;
  if (acc == 12345.678f) sink[0] = acc;
}

__global__ void stream_write4(float* __restrict__ out, int64_t n) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) out[i] = float(i);
}

int main() {
  const int64_t bytes = int64_t(1) << 30;  // 1 GiB streams: far beyond the 256 MiB Infinity Cache
  void* buf;
  float* sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 4);
  hipMemset(buf, 0, bytes);
  const int grid = 256 * 8, block = 256;
  for (int rep = 0; rep < 3; ++rep) {
    stream_read<float><<<grid, block>>>((const float*)buf, bytes / 4, sink);
    stream_read<float2><<<grid, block>>>((const float2*)buf, bytes / 8, sink);
    stream_read<float4><<<grid, block>>>((const float4*)buf, bytes / 16, sink);
    stream_write4<<<grid, block>>>((float*)buf, bytes / 4);
  }
  // random 16-byte gathers: 64 M indices (256 MiB index stream, 4 B / lane) into a 512 KiB table (the atom records of cfg3)
  // and into a 768 MiB table
  const int64_t n_idx = int64_t(1) << 26;
  for (int64_t table_bytes : {int64_t(512) << 10, int64_t(768) << 20}) {
    const int64_t n_tab = table_bytes / 16;
    std::vector<int> h(n_idx);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      v = int(s % uint64_t(n_tab));
    }
    int* idx;
    hipMalloc(&idx, n_idx * 4);
    hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) gather16<<<grid, block>>>(idx, n_idx, (const float4*)buf, sink);
    hipDeviceSynchronize();
    hipFree(idx);
  }
  hipDeviceSynchronize();
  printf("expected bytes: stream_read<float|float2|float4> %lld each; stream_write4 %lld; gather16: index stream %lld + "
         "%lld gathered (small table: cache resident; large table: %lld lines of 128 B if no line is shared)\n",
         (long long)bytes, (long long)bytes, (long long)(n_idx * 4), (long long)(n_idx * 16), (long long)n_idx);
  return 0;
}
