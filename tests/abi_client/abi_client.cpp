// A torch-free client of the drop-in boundary: plain HIP allocations, the C ABI of include/a2v.h, CPU checks.
// Built by __graft_entry__.build() (hipcc, links liba2v_hip.so) and run by tests/test_abi_client_gpu.py.
//   1. a2v_abi_version
//   2. a2v_layernorm_fwd (fp32) against a double-precision CPU LayerNorm
//   3. a2v_conv_gemm as a Linear with bias + GELU (fp32 parity mode) against a CPU reference
//   4. a2v_attention_fwd (fp32, ALiBi from explicit positions) against a CPU softmax attention
//   5. caller-owned scratch (ABI 10): a2v_query_workspace / a2v_bind_workspace on SIX streams at once (the library's own static
//      sets stop at four), a bf16 Linear with column sums on the persistent kernel on each, against a CPU reference
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "a2v.h"

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) { std::printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } \
  } while (0)
#define A2V(x)                                                                   \
  do {                                                                           \
    int r_ = (x);                                                                \
    if (r_ != 0) { std::printf("a2v error %d at %s:%d\n", r_, __FILE__, __LINE__); return 3; } \
  } while (0)

static unsigned lcg = 12345u;
static float rnd() {
  lcg = lcg * 1664525u + 1013904223u;
  return ((lcg >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
template <typename T> static T* dev(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}
static double relerr(const std::vector<float>& a, const std::vector<double>& b) {
  double num = 0, den = 0;
  for (size_t i = 0; i < a.size(); ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
  return std::sqrt(num / (den + 1e-30));
}

int main() {
  if (a2v_abi_version() < 5) { std::printf("abi version %d\n", a2v_abi_version()); return 1; }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // ---- LayerNorm
  {
    const int M = 37, C = 96;
    std::vector<float> x(M * C), g(C), b(C), y(M * C);
    for (auto& v : x) v = rnd() * 3.f;
    for (auto& v : g) v = 1.f + 0.2f * rnd();
    for (auto& v : b) v = 0.3f * rnd();
    float *dx = dev(x), *dg = dev(g), *db = dev(b), *dy = dev(y);
    A2V(a2v_layernorm_fwd(A2V_F32, dx, nullptr, dg, db, dy, nullptr, nullptr, nullptr, nullptr, M, C, C, C, 1e-5f, 0, 0.f, 0u, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref(M * C);
    for (int m = 0; m < M; ++m) {
      double mu = 0, var = 0;
      for (int c = 0; c < C; ++c) mu += x[m * C + c];
      mu /= C;
      for (int c = 0; c < C; ++c) var += (x[m * C + c] - mu) * (x[m * C + c] - mu);
      const double rs = 1.0 / std::sqrt(var / C + 1e-5);
      for (int c = 0; c < C; ++c) ref[m * C + c] = (x[m * C + c] - mu) * rs * g[c] + b[c];
    }
    const double e = relerr(y, ref);
    std::printf("layernorm_fwd rel err %.2e\n", e);
    if (!(e < 1e-5)) return 10;
  }
  // ---- Linear + bias + GELU through a2v_conv_gemm
  {
    const int M = 130, N = 72, K = 64;
    std::vector<float> x(M * K), w(N * K), bias(N), y(M * N);
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = rnd() * 0.125f;
    for (auto& v : bias) v = rnd() * 0.1f;
    float *dx = dev(x), *dw = dev(w), *dbias = dev(bias), *dy = dev(y);
    A2V(a2v_conv_gemm(A2V_F32, dx, K, 0, 0, M, M, 1, 0, 1, K, K, nullptr, 1, dw, K, 0, dy, N, 0, nullptr, nullptr, 0, dbias, nullptr,
                      M, N, K, 1, A2V_EPI_BIAS | A2V_EPI_GELU, 0.f, 0u, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref(M * N);
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = bias[n];
        for (int k = 0; k < K; ++k) s += (double)x[m * K + k] * w[n * K + k];
        ref[m * N + n] = 0.5 * s * (1.0 + std::erf(s / std::sqrt(2.0)));
      }
    const double e = relerr(y, ref);
    std::printf("conv_gemm linear+bias+gelu rel err %.2e\n", e);
    if (!(e < 1e-5)) return 11;
  }
  // ---- attention forward with ALiBi
  {
    const int Nb = 2, S = 50, H = 2, HD = 16;
    std::vector<float> qkv((size_t)Nb * S * 3 * H * HD), out((size_t)Nb * S * H * HD), slopes = {0.5f, 0.25f}, asc = {1.3f, 0.7f};
    std::vector<int> pos(Nb * S);
    for (auto& v : qkv) v = rnd();
    for (int n = 0; n < Nb; ++n)
      for (int s = 0; s < S; ++s) pos[n * S + s] = 3 * s + n;
    float *dq = dev(qkv), *dout = dev(out), *dsl = dev(slopes), *dasc = dev(asc);
    int* dpos = dev(pos);
    float* dlse = nullptr;
    CK(hipMalloc(&dlse, (size_t)Nb * H * S * 4));
    A2V(a2v_attention_fwd(A2V_F32, dq, dout, dlse, nullptr, nullptr, nullptr, dpos, dsl, dasc, Nb, S, H, HD, 1.0f / std::sqrt((float)HD), 0.f, 0u, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref(out.size());
    const int W = 3 * H * HD;
    for (int n = 0; n < Nb; ++n)
      for (int h = 0; h < H; ++h)
        for (int i = 0; i < S; ++i) {
          std::vector<double> sc(S);
          double mx = -1e300;
          for (int j = 0; j < S; ++j) {
            double s = 0;
            for (int d = 0; d < HD; ++d)
              s += (double)qkv[((size_t)n * S + i) * W + h * HD + d] * qkv[((size_t)n * S + j) * W + H * HD + h * HD + d];
            s = s / std::sqrt((double)HD) - slopes[h] * asc[h] * std::abs(pos[n * S + i] - pos[n * S + j]);
            sc[j] = s;
            mx = std::max(mx, s);
          }
          double den = 0;
          for (int j = 0; j < S; ++j) { sc[j] = std::exp(sc[j] - mx); den += sc[j]; }
          for (int d = 0; d < HD; ++d) {
            double o = 0;
            for (int j = 0; j < S; ++j) o += sc[j] / den * qkv[((size_t)n * S + j) * W + 2 * H * HD + h * HD + d];
            ref[((size_t)n * S + i) * H * HD + h * HD + d] = o;
          }
        }
    const double e = relerr(out, ref);
    std::printf("attention_fwd rel err %.2e\n", e);
    if (!(e < 1e-5)) return 12;
  }
  // ---- caller-owned workspaces, more streams than the library has static sets
  {
    if (a2v_abi_version() < 10) { std::printf("abi version %d has no a2v_bind_workspace\n", a2v_abi_version()); return 13; }
    const size_t wsb = a2v_query_workspace();
    if (wsb == 0 || wsb > (size_t)1 << 30) { std::printf("a2v_query_workspace: %zu\n", wsb); return 13; }
    const int NS = 6, M = 2048, N = 256, K = 256;
    std::vector<unsigned short> x((size_t)M * K), w((size_t)N * K);
    auto tobf = [](float f) { unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); };
    auto frombf = [](unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; };
    for (auto& v : x) v = tobf(rnd());
    for (auto& v : w) v = tobf(rnd() * 0.0625f);
    std::vector<double> ref((size_t)M * N), refcs(N, 0.0);
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s_ = 0;
        for (int k = 0; k < K; ++k) s_ += (double)frombf(x[(size_t)m * K + k]) * frombf(w[(size_t)n * K + k]);
        ref[(size_t)m * N + n] = s_;
      }
    unsigned short *dx = dev(x), *dw = dev(w);
    hipStream_t ss[NS];
    void* wsp[NS];
    unsigned short* dy[NS];
    float* dcs[NS];
    if (a2v_bind_workspace(st, nullptr, 0) == 0) { std::printf("unbinding a stream that was never bound must fail\n"); return 13; }
    for (int i = 0; i < NS; ++i) {
      CK(hipStreamCreate(&ss[i]));
      CK(hipMalloc(&wsp[i], wsb));
      CK(hipMemset(wsp[i], 0, wsb));
      CK(hipMalloc((void**)&dy[i], (size_t)M * N * 2));
      CK(hipMalloc((void**)&dcs[i], N * 4));
      CK(hipMemset(dcs[i], 0, N * 4));
      if (a2v_bind_workspace(ss[i], wsp[i], wsb - 1) == 0) { std::printf("a short workspace must be refused\n"); return 13; }
      A2V(a2v_bind_workspace(ss[i], wsp[i], wsb));
    }
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep)          // twice: the workspace must come back self-consistent (counters reset, copies cleared)
      for (int i = 0; i < NS; ++i)
        A2V(a2v_conv_gemm(A2V_BF16, dx, K, 0, 0, M, M, 1, 0, 1, K, K, nullptr, 1, dw, K, 0, dy[i], N, 0, nullptr, nullptr, 0, nullptr, dcs[i],
                          M, N, K, 1, A2V_EPI_COLSUM, 0.f, 0u, ss[i]));
    CK(hipDeviceSynchronize());
    for (int i = 0; i < NS; ++i) {
      std::vector<unsigned short> y((size_t)M * N);
      std::vector<float> cs(N), yf((size_t)M * N);
      CK(hipMemcpy(y.data(), dy[i], y.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(cs.data(), dcs[i], N * 4, hipMemcpyDeviceToHost));
      for (size_t j = 0; j < y.size(); ++j) yf[j] = frombf(y[j]);
      if (i == 0) {                              // column sums of the bf16-ROUNDED outputs, summed over the two launches
        for (int n = 0; n < N; ++n) { refcs[n] = 0; for (int m = 0; m < M; ++m) refcs[n] += 2.0 * yf[(size_t)m * N + n]; }
      }
      const double e = relerr(yf, ref), ec = relerr(cs, refcs);
      std::printf("workspace stream %d: linear rel err %.2e, column sums rel err %.2e\n", i, e, ec);
      if (!(e < 5e-3) || !(ec < 1e-3)) return 14;
      A2V(a2v_bind_workspace(ss[i], nullptr, 0));                      // unbind
      CK(hipFree(wsp[i]));
    }
  }
  std::printf("ABI_CLIENT_OK\n");
  return 0;
}
