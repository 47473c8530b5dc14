// LDS read throughput per CU by instruction: ds_read_b128 against ds_read_b64_tr_b16 (the transposing read both operands of the TN GEMM use) and
// plain ds_read_b64 -- 8 waves per CU, every wave streams reads over a 32-KB image with independent destinations.
//   hipcc --offload-arch=gfx950 -O3 tools/native/lds_read_bench.hip -o tools/native/lds_read_bench.bin && tools/native/lds_read_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int MODE>   // 0: ds_read_b128, 1: ds_read_b64_tr_b16, 2: ds_read_b64
__global__ __launch_bounds__(512) void lds_read(unsigned* out, int iters) {
  __shared__ __attribute__((aligned(16))) char sm[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 16; i += 512) ((u32x4*)sm)[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  unsigned acc = 0;
  // conflict-free addressing: b128: lane * 16 within a 1-KB row block; b64: lane * 8
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm + (tid >> 6) * 8192 + lane * (MODE == 0 ? 16 : 8);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(base), "n"(u * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u].x;
    } else {
      s16x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (MODE == 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[u]) : "v"(base), "n"(u * 512));
        else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[u]) : "v"(base), "n"(u * 512));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
      for (int u = 0; u < 16; ++u) acc ^= (unsigned)v[u].x;
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}

template <int MODE>
static void run(const char* name, unsigned* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((lds_read<MODE>), dim3(256), dim3(512), 0, 0, out, 100);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((lds_read<MODE>), dim3(256), dim3(512), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes_per_cu = 8.0 * 8192.0 * iters;        // 8 waves x 8 KB per iteration
  printf("%-22s %7.1f GB/s per CU   (%.1f B/clk at 2.0 GHz)\n", name, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu / (ms * 1e-3) / 2.0e9);
}

int main() {
  unsigned* out;
  CK(hipMalloc(&out, 4));
  for (int r = 0; r < 2; ++r) {
    run<0>("ds_read_b128", out);
    run<1>("ds_read_b64_tr_b16", out);
    run<2>("ds_read_b64", out);
  }
  return 0;
}
