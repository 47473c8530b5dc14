// HBM ceilings of the box this runs on, for the streaming kernels' roofline (LayerNorm, loss, EMA ...): read-only, write-only and
// copy rates of 16-B-per-lane kernels over buffers far larger than the Infinity Cache, by grid size / unroll / cache policy.
//   hipcc --offload-arch=gfx950 -O3 tools/native/copy_bench.hip -o tools/native/copy_bench.bin && tools/native/copy_bench.bin [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, int NT>
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long j = i + (long long)u * 256;
      if (j < n) { if (NT & 1) { const v4u t_ = __builtin_nontemporal_load((const v4u*)(src + j)); v[u] = make_uint4(t_.x, t_.y, t_.z, t_.w); } else v[u] = src[j]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long j = i + (long long)u * 256;
      if (j < n) { if (NT & 2) { v4u t_ = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t_, (v4u*)(dst + j)); } else dst[j] = v[u]; }
    }
  }
}
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ src, unsigned* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * 256 * U;
  unsigned acc = 0;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; v[u] = j < n ? src[j] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write_kernel(uint4* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * U;
  const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride)
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) dst[j] = v; }
}
// the LayerNorm access shape: a wave owns whole 2-KB rows (two 16-B chunks per lane), R rows per batch, persistent
template <int R>
__global__ __launch_bounds__(256) void rowcopy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long rows) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long nb = (rows + R - 1) / R, nw = (long long)gridDim.x * 4;
  for (long long b = (long long)blockIdx.x * 4 + wid; b < nb; b += nw) {
    uint4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = b * R + r < rows ? b * R + r : rows - 1;
      v[r][0] = src[row * 128 + lane]; v[r][1] = src[row * 128 + 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = b * R + r;
      if (row < rows) { dst[row * 128 + lane] = v[r][0]; dst[row * 128 + 64 + lane] = v[r][1]; }
    }
  }
}


// non-persistent: a workgroup copies ONE contiguous chunk of 256 x U x 16 B and retires
template <int U>
__global__ __launch_bounds__(256) void copy_once_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * 256 * U + threadIdx.x;
  uint4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) v[u] = src[j]; }
#pragma unroll
  for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) dst[j] = v[u]; }
}
template <int U>
__global__ __launch_bounds__(256) void write_once_kernel(uint4* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * 256 * U + threadIdx.x;
  const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
#pragma unroll
  for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) dst[j] = v; }
}
template <int U>
__global__ __launch_bounds__(256) void read_once_kernel(const uint4* __restrict__ src, unsigned* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 * U + threadIdx.x;
  unsigned acc = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) { const uint4 v = src[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; } }
  if (acc == 0x12345678u) out[0] = acc;
}
// non-persistent LayerNorm shape: a wave copies R whole 2-KB rows, 4 waves per workgroup
template <int R>
__global__ __launch_bounds__(256) void rowcopy_once_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long rows) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long b = (long long)blockIdx.x * 4 + wid;
  uint4 v[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long long row = b * R + r < rows ? b * R + r : rows - 1;
    v[r][0] = src[row * 128 + lane]; v[r][1] = src[row * 128 + 64 + lane];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long long row = b * R + r;
    if (row < rows) { dst[row * 128 + lane] = v[r][0]; dst[row * 128 + 64 + lane] = v[r][1]; }
  }
}
// persistent, but the workgroups walk the buffer as ONE front: chunk = iteration * grid + block (same as copy_kernel), a
// DYNAMIC variant hands chunks out by an atomic counter so that the front stays tight whatever the workgroups' speeds
template <int U>
__global__ __launch_bounds__(256) void copy_dyn_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n, unsigned* ctr) {
  __shared__ unsigned s_chunk;
  const long long nchunk = (n + 256 * U - 1) / (256 * U);
  for (;;) {
    if (threadIdx.x == 0) s_chunk = atomicAdd(ctr, 1u);
    __syncthreads();
    const long long c = s_chunk;
    __syncthreads();
    if (c >= nchunk) break;
    const long long i = c * 256 * U + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) v[u] = src[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long j = i + (long long)u * 256; if (j < n) dst[j] = v[u]; }
  }
}

// persistent LayerNorm-shaped copy with hand-counted waits: the loads of batch i+1 are issued before batch i is stored, and the wait
// in front of batch i's stores is vmcnt(4R) = "all but the 2R stores of batch i-1 and the 2R loads of batch i+1": store
// acknowledgements are never waited for (hipcc's own wait for a load is vmcnt(0) whenever a store is pending).
#define ASM_LOAD(dst_, ptr_, off_) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off_ : "=v"(dst_) : "v"(ptr_) : "memory")
#define ASM_STORE(ptr_, src_, off_) asm volatile("global_store_dwordx4 %0, %1, off offset:" #off_ "\n\ts_nop 1" : : "v"(ptr_), "v"(src_) : "memory")
typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
template <int R>
__global__ __launch_bounds__(256) void rowcopy_asm_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long rows) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long nb = rows / R, nw = (long long)gridDim.x * 4;      // (rows % R == 0 here)
  long long b = (long long)blockIdx.x * 4 + wid;
  if (b >= nb) return;
  v4u_ a[R][2], c[R][2];
#define LOADB(buf_, b_)                                                       \
  _Pragma("unroll") for (int r = 0; r < R; ++r) {                             \
    const uint4* p_ = src + ((b_) * R + r) * 128 + lane;                      \
    ASM_LOAD(buf_[r][0], p_, 0); ASM_LOAD(buf_[r][1], p_, 1024);              \
  }
#define STOREB(buf_, b_)                                                      \
  _Pragma("unroll") for (int r = 0; r < R; ++r) {                             \
    uint4* p_ = dst + ((b_) * R + r) * 128 + lane;                            \
    ASM_STORE(p_, buf_[r][0], 0); ASM_STORE(p_, buf_[r][1], 1024);            \
  }
#define WAITB(buf_, n_)                                                                                   \
  if (R == 1) asm volatile("s_waitcnt vmcnt(" #n_ ")" : "+v"(buf_[0][0]), "+v"(buf_[0][1]) : : "memory"); \
  else asm volatile("s_waitcnt vmcnt(" #n_ ")" : "+v"(buf_[0][0]), "+v"(buf_[0][1]), "+v"(buf_[R - 1][0]), "+v"(buf_[R - 1][1]) : : "memory");
  LOADB(a, b)
  long long b1 = b + nw;
  if (b1 < nb) { LOADB(c, b1) } else { LOADB(c, b) }
  if (R == 1) { WAITB(a, 2) } else { WAITB(a, 4) }        // (first batch: no stores in flight yet)
  STOREB(a, b)
  while (b1 < nb) {
    const long long b2 = b1 + nw;
    if (b2 < nb) { LOADB(a, b2) } else { LOADB(a, b1) }
    if (R == 1) { WAITB(c, 4) } else { WAITB(c, 8) }
    STOREB(c, b1)
    if (b2 >= nb) break;
    b = b2;
    b1 = b + nw;
    if (b1 < nb) { LOADB(c, b1) } else { LOADB(c, b) }
    if (R == 1) { WAITB(a, 4) } else { WAITB(a, 8) }
    STOREB(a, b)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename F> static double timeit(F f, int iters = 8) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int g = 0; g < 5; ++g) {
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms / iters);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2] * 1e-3;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 3.0;
  const long long bytes = (long long)(gib * (1LL << 30)), n = bytes / 16;
  uint4 *src, *dst; unsigned* out;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 2, bytes));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("# %s, %d CUs, buffers %.1f GiB each\n", p.name, cus, gib);
  double t = timeit([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); });
  printf("hipMemcpy D2D                          %7.1f us  %5.2f TB/s (read + write)\n", t * 1e6, 2.0 * bytes / t / 1e12);
#define RUN_COPY(U, NT, WPC)                                                                                          \
  { const int grid = cus * WPC; t = timeit([&] { hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n); }); \
    printf("copy  unroll %d nt %d  %2d wg/CU           %7.1f us  %5.2f TB/s\n", U, NT, WPC, t * 1e6, 2.0 * bytes / t / 1e12); }
  RUN_COPY(1, 0, 8) RUN_COPY(2, 0, 8) RUN_COPY(4, 0, 8) RUN_COPY(8, 0, 8)
  RUN_COPY(4, 0, 2) RUN_COPY(4, 0, 4) RUN_COPY(4, 0, 16) RUN_COPY(4, 0, 32)
  RUN_COPY(4, 1, 8) RUN_COPY(4, 2, 8) RUN_COPY(4, 3, 8) RUN_COPY(8, 3, 8) RUN_COPY(2, 3, 16)
  { const long long need = (n + 255) / 256; t = timeit([&] { hipLaunchKernelGGL((copy_kernel<1, 0>), dim3((unsigned)need), dim3(256), 0, 0, src, dst, n); });
    printf("copy  one 16-B element per thread      %7.1f us  %5.2f TB/s\n", t * 1e6, 2.0 * bytes / t / 1e12); }
#define RUN_ROW(R, WPC)                                                                                               \
  { const int grid = cus * WPC; const long long rows = n / 128; t = timeit([&] { hipLaunchKernelGGL((rowcopy_kernel<R>), dim3(grid), dim3(256), 0, 0, src, dst, rows); }); \
    printf("rowcopy (2-KB rows) R %d  %2d wg/CU       %7.1f us  %5.2f TB/s\n", R, WPC, t * 1e6, 2.0 * bytes / t / 1e12); }
  RUN_ROW(1, 8) RUN_ROW(2, 8) RUN_ROW(4, 8) RUN_ROW(4, 4) RUN_ROW(8, 4)
#define RUN_READ(U, WPC)                                                                                              \
  { const int grid = cus * WPC; t = timeit([&] { hipLaunchKernelGGL((read_kernel<U>), dim3(grid), dim3(256), 0, 0, src, out, n); }); \
    printf("read  unroll %d  %2d wg/CU                %7.1f us  %5.2f TB/s\n", U, WPC, t * 1e6, 1.0 * bytes / t / 1e12); }
  RUN_READ(4, 8) RUN_READ(8, 8) RUN_READ(4, 16)
#define RUN_WRITE(U, WPC)                                                                                             \
  { const int grid = cus * WPC; t = timeit([&] { hipLaunchKernelGGL((write_kernel<U>), dim3(grid), dim3(256), 0, 0, dst, n); }); \
    printf("write unroll %d  %2d wg/CU                %7.1f us  %5.2f TB/s\n", U, WPC, t * 1e6, 1.0 * bytes / t / 1e12); }
  RUN_WRITE(4, 8) RUN_WRITE(1, 16)

#define RUN_ONCE(U)                                                                                                   \
  { const long long need = (n + 256 * U - 1) / (256 * U); t = timeit([&] { hipLaunchKernelGGL((copy_once_kernel<U>), dim3((unsigned)need), dim3(256), 0, 0, src, dst, n); }); \
    printf("copy  once, %d x 16 B per thread         %7.1f us  %5.2f TB/s\n", U, t * 1e6, 2.0 * bytes / t / 1e12); }
  RUN_ONCE(1) RUN_ONCE(2) RUN_ONCE(4) RUN_ONCE(8)
#define RUN_ROWONCE(R)                                                                                                \
  { const long long rows = n / 128, need = ((rows + R - 1) / R + 3) / 4; t = timeit([&] { hipLaunchKernelGGL((rowcopy_once_kernel<R>), dim3((unsigned)need), dim3(256), 0, 0, src, dst, rows); }); \
    printf("rowcopy once (2-KB rows) R %d            %7.1f us  %5.2f TB/s\n", R, t * 1e6, 2.0 * bytes / t / 1e12); }
  RUN_ROWONCE(1) RUN_ROWONCE(2) RUN_ROWONCE(4)
  { unsigned* ctr; CK(hipMalloc(&ctr, 4));
#define RUN_DYN(U, WPC)                                                                                               \
  { const int grid = cus * WPC; t = timeit([&] { CK(hipMemsetAsync(ctr, 0, 4, 0)); hipLaunchKernelGGL((copy_dyn_kernel<U>), dim3(grid), dim3(256), 0, 0, src, dst, n, ctr); }); \
    printf("copy  dynamic chunks %d x 16 B %2d wg/CU   %7.1f us  %5.2f TB/s\n", U, WPC, t * 1e6, 2.0 * bytes / t / 1e12); }
    RUN_DYN(1, 8) RUN_DYN(2, 8) RUN_DYN(4, 8) RUN_DYN(4, 4) }
  { const int grid = cus * 8 - 24; t = timeit([&] { hipLaunchKernelGGL((copy_kernel<4, 0>), dim3(grid), dim3(256), 0, 0, src, dst, n); });
    printf("copy  unroll 4, grid %d (not 2^k)      %7.1f us  %5.2f TB/s\n", grid, t * 1e6, 2.0 * bytes / t / 1e12); }
  { const long long need = (n + 255) / 256; t = timeit([&] { hipLaunchKernelGGL((read_once_kernel<1>), dim3((unsigned)need), dim3(256), 0, 0, src, out, n); });
    printf("read  once, 16 B per thread             %7.1f us  %5.2f TB/s\n", t * 1e6, 1.0 * bytes / t / 1e12); }
  { const long long need = (n + 1023) / 1024; t = timeit([&] { hipLaunchKernelGGL((read_once_kernel<4>), dim3((unsigned)need), dim3(256), 0, 0, src, out, n); });
    printf("read  once, 4 x 16 B per thread         %7.1f us  %5.2f TB/s\n", t * 1e6, 1.0 * bytes / t / 1e12); }
  { const long long need = (n + 255) / 256; t = timeit([&] { hipLaunchKernelGGL((write_once_kernel<1>), dim3((unsigned)need), dim3(256), 0, 0, dst, n); });
    printf("write once, 16 B per thread             %7.1f us  %5.2f TB/s\n", t * 1e6, 1.0 * bytes / t / 1e12); }
  { const long long need = (n + 1023) / 1024; t = timeit([&] { hipLaunchKernelGGL((write_once_kernel<4>), dim3((unsigned)need), dim3(256), 0, 0, dst, n); });
    printf("write once, 4 x 16 B per thread         %7.1f us  %5.2f TB/s\n", t * 1e6, 1.0 * bytes / t / 1e12); }

#define RUN_ROWASM(R, WPC)                                                                                            \
  { const int grid = cus * WPC; const long long rows = n / 128; t = timeit([&] { hipLaunchKernelGGL((rowcopy_asm_kernel<R>), dim3(grid), dim3(256), 0, 0, src, dst, rows); }); \
    printf("rowcopy persistent, counted waits R %d %2d wg/CU %7.1f us  %5.2f TB/s\n", R, WPC, t * 1e6, 2.0 * bytes / t / 1e12); }
  RUN_ROWASM(1, 8) RUN_ROWASM(1, 4) RUN_ROWASM(2, 8) RUN_ROWASM(2, 4) RUN_ROWASM(2, 2)
  return 0;
}
