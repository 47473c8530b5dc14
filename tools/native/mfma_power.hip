// Sustained matrix-core throughput of the box under its power cap, by MFMA shape: v_mfma_f32_16x16x32_bf16 against
// v_mfma_f32_32x32x16_bf16 streams of equal FLOPs on random operands held in registers (no memory traffic), optionally with the
// LDS fragment reads of a 256 x 256 GEMM tile beside them (12 ds_read_b128 per 32 K-deep step of a 128 x 64 wave tile).
// The kernels of this repository run power-limited (1.74 - 1.9 GHz under the GEMMs against 2.4 nominal): what a shape
// sustains is TFLOP/s = rate x clock-under-that-load, which this measures directly.
//   hipcc --offload-arch=gfx950 -O3 tools/native/mfma_power.hip -o tools/native/mfma_power.bin && tools/native/mfma_power.bin [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SHAPE 16 / 32; LDS = 1: every 32-deep K step of the wave's 128 x 64 tile re-reads its 12 fragments from LDS (random data staged once)
template <int SHAPE, int LDS>
__global__ __launch_bounds__(256, 2) void mfma_loop(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) char sm[LDS ? 12 * 1024 * 4 : 16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = in[(i * 256 + tid) & 4095];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = in[((8 + i) * 256 + tid) & 4095];
  if (LDS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *(bf16x8*)(sm + (wid * 12 + i) * 1024 + lane * 16) = a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *(bf16x8*)(sm + (wid * 12 + 8 + i) * 1024 + lane * 16) = b[i];
    __syncthreads();
  }
  float r = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(sm + (wid * 12 + i) * 1024 + ((lane * 16 + it * 64) & 1023));
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(sm + (wid * 12 + 8 + i) * 1024 + ((lane * 16 + it * 64) & 1023));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    // the same 128 x 64 x 32 step as 4 x 2 tiles of 32 x 32, two 16-deep halves: fragment i of the 16-shape = rows 16 i .., K 32;
    // here a[2 m + kh] stands for (rows 32 m .., K half kh), b[2 n + kh] likewise: 8 A + 4 B fragments of 16 B per lane, as above
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(sm + (wid * 12 + i) * 1024 + ((lane * 16 + it * 64) & 1023));
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(sm + (wid * 12 + 8 + i) * 1024 + ((lane * 16 + it * 64) & 1023));
      }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * i + kh], b[2 * j + kh], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[i][j][e];
  }
  if (r == 1.2345f) out[0] = r;
}

template <int SHAPE, int LDS>
static double run(const bf16x8* in, float* out, int iters, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(512), dim3(256), 0, 0, in, out, iters);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(512), dim3(256), 0, 0, in, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 128 * 64 * 32 * (double)iters * 2048 * reps;     // per wave and iteration: 128 x 64 x 32 MACs; 2048 waves
  return flop / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.0;
  std::vector<unsigned short> h(4096 * 8);
  unsigned s = 12345u;
  for (auto& v : h) {                         // bf16 normal-ish values: random sign, exponent 120..130, random mantissa
    s = s * 1664525u + 1013904223u;
    v = (unsigned short)(((s >> 31) << 15) | ((120 + ((s >> 8) % 11)) << 7) | ((s >> 16) & 0x7F));
  }
  bf16x8* in; float* out;
  CK(hipMalloc(&in, h.size() * 2)); CK(hipMalloc(&out, 4));
  CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int iters = 20000;                    // ~ 0.67 TFLOP per launch: a few hundred microseconds at 1.5 PFLOP/s ... reps make the seconds
  const int reps = (int)(secs * 1500.0 / (2.0 * 128 * 64 * 32 * iters * 2048 / 1e12)) + 1;
  for (int round = 0; round < 3; ++round) {
    printf("16x16x32 regs   %7.1f TFLOP/s\n", run<16, 0>(in, out, iters, reps));
    printf("32x32x16 regs   %7.1f TFLOP/s\n", run<32, 0>(in, out, iters, reps));
    printf("16x16x32 + LDS  %7.1f TFLOP/s\n", run<16, 1>(in, out, iters, reps));
    printf("32x32x16 + LDS  %7.1f TFLOP/s\n", run<32, 1>(in, out, iters, reps));
  }
  return 0;
}
