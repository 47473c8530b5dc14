// Vector-instruction throughput on gfx950 by opcode and by waves per SIMD (round 6): the student attention kernels turned out to be
// bound by their VALU instruction COUNT (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.2 cycles per instruction at 76-87 % VALU-busy), so
// what each opcode costs decides which arithmetic to write.  Every wave runs REP x 64 copies of one instruction on 16 independent
// register sets (no dependent chain shorter than 16 instructions) between two s_memtime stamps; W waves per SIMD are made co-resident
// by the launch (256-thread workgroups, W per CU, every CU busy).  Reported: shader cycles per instruction per SIMD =
// (stamp delta of a wave) / (instructions of ONE wave) / W ... i.e. the SIMD's throughput when W waves compete.
//   hipcc --offload-arch=gfx950 -O3 tools/native/valu_bench.hip -o tools/native/valu_bench.bin && tools/native/valu_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP 64

#define OP16(STR)                                                                                      \
  asm volatile(STR(0) STR(1) STR(2) STR(3) STR(4) STR(5) STR(6) STR(7) STR(8) STR(9) STR(10) STR(11) STR(12) STR(13) STR(14) STR(15) \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),                \
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])          \
               : "v"(x), "v"(y))

// one 32-bit register per copy: %N is copy N, %16 and %17 are two loop-invariant sources
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %16\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %16\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %16\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %16\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %16, %17\n"
#define I_CNDM(n) "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n"
#define I_CVTPK(n) "v_cvt_pk_bf16_f32 %" #n ", %" #n ", %16\n"
#define I_BFE(n) "v_bfe_i32 %" #n ", %" #n ", %16, 1\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %16\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %16\n"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 15, %" #n "\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %16, %17\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %16, 31\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %16\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %16, %17\n"
#define I_XAD(n) "v_xad_u32 %" #n ", %" #n ", %16, %17\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 2, %16\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %16, %17\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %16, %17\n"
#define I_PKSUBU16(n) "v_pk_sub_u16 %" #n ", %" #n ", %16 clamp\n"
#define I_PKMINU16(n) "v_pk_min_u16 %" #n ", %" #n ", %16\n"
#define I_PKMULBF(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %16\n"
#define I_SUBABS(n) "v_sub_f32 %" #n ", %" #n ", %16\n"
#define I_FMAABS(n) "v_fma_f32 %" #n ", |%" #n "|, %16, -%17\n"
#define I_CMP(n) "v_cmp_gt_i32 vcc, %" #n ", %16\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %16\n"
#define I_DPP(n) "v_add_f32_dpp %" #n ", %" #n ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_BITOP(n) "v_bitop3_b32 %" #n ", %" #n ", %16, %17 bitop3:0x96\n"
#define I_CNDS(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %16, s[6:7]\n"
#define I_SUBABS2(n) "v_sub_f32_e64 %" #n ", |%" #n "|, %16\n"
#define I_XORSDWA(n) "v_xor_b32_sdwa %" #n ", %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I_CMPS(n) "v_cmp_gt_i32_e64 s[8:9], %" #n ", %16\n"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %16\n"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define I_OR(n) "v_or_b32 %" #n ", %" #n ", %16\n"
#define I_SUBU(n) "v_sub_u32 %" #n ", %16, %" #n "\n"
#define I_MULNEG(n) "v_mul_f32_e64 %" #n ", -%" #n ", %16\n"
#define I_CVTU(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_EXPFMA(n) "v_exp_f32 %" #n ", %" #n "\nv_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_EXPADD3(n) "v_exp_f32 %" #n ", %" #n "\nv_add_f32 %" #n ", %" #n ", %16\nv_mul_f32 %" #n ", %" #n ", %16\nv_and_b32 %" #n ", %" #n ", %17\n"
#define I_MULLOADD3(n) "v_mul_lo_u32 %" #n ", %" #n ", %16\nv_add_f32 %" #n ", %" #n ", %16\nv_mul_f32 %" #n ", %" #n ", %16\nv_and_b32 %" #n ", %" #n ", %17\n"
#define I_MED3(n) "v_med3_f32 %" #n ", %" #n ", %16, %17\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %16, %17\n"
#define I_FMAMK(n) "v_fmamk_f32 %" #n ", %" #n ", 0x3fb8aa3b, %16\n"

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(unsigned long long* __restrict__ stamps, unsigned* __restrict__ sink, int loops) {
  unsigned r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + 1u;
  unsigned x = 0x3f800123u + threadIdx.x, y = 0x3e000001u;
  asm volatile("v_cmp_gt_u32 vcc, %0, %1\ns_mov_b64 s[6:7], vcc" :: "v"(x), "v"(y) : "vcc", "s6", "s7", "s8", "s9");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int l = 0; l < loops; ++l) {
#pragma unroll
    for (int k = 0; k < REP / 16; ++k) {
      if constexpr (OP == 0) OP16(I_FMA);
      if constexpr (OP == 1) OP16(I_ADD);
      if constexpr (OP == 2) OP16(I_MUL);
      if constexpr (OP == 3) OP16(I_EXP);
      if constexpr (OP == 4) OP16(I_RCP);
      if constexpr (OP == 5) OP16(I_MULLO);
      if constexpr (OP == 6) OP16(I_MUL24);
      if constexpr (OP == 7) OP16(I_MAD24);
      if constexpr (OP == 8) OP16(I_CNDM);
      if constexpr (OP == 9) OP16(I_CVTPK);
      if constexpr (OP == 10) OP16(I_BFE);
      if constexpr (OP == 11) OP16(I_AND);
      if constexpr (OP == 12) OP16(I_XOR);
      if constexpr (OP == 13) OP16(I_LSHR);
      if constexpr (OP == 14) OP16(I_MAX3);
      if constexpr (OP == 15) OP16(I_ALIGN);
      if constexpr (OP == 16) OP16(I_ADDU);
      if constexpr (OP == 17) OP16(I_ADD3);
      if constexpr (OP == 18) OP16(I_XAD);
      if constexpr (OP == 19) OP16(I_LSHLOR);
      if constexpr (OP == 20) OP16(I_ANDOR);
      if constexpr (OP == 21) OP16(I_PERM);
      if constexpr (OP == 22) OP16(I_PKSUBU16);
      if constexpr (OP == 23) OP16(I_PKMINU16);
      if constexpr (OP == 24) OP16(I_PKMULBF);
      if constexpr (OP == 25) OP16(I_FMAABS);
      if constexpr (OP == 26) OP16(I_CMP);
      if constexpr (OP == 27) OP16(I_MOV);
      if constexpr (OP == 28) OP16(I_DPP);
      if constexpr (OP == 29) OP16(I_BITOP);
      if constexpr (OP == 30) OP16(I_CNDS);
      if constexpr (OP == 31) OP16(I_SUBABS2);
      if constexpr (OP == 32) OP16(I_XORSDWA);
      if constexpr (OP == 33) OP16(I_CMPS);
      if constexpr (OP == 34) OP16(I_MAX);
      if constexpr (OP == 35) OP16(I_ASHR);
      if constexpr (OP == 36) OP16(I_OR);
      if constexpr (OP == 37) OP16(I_SUBU);
      if constexpr (OP == 38) OP16(I_MULNEG);
      if constexpr (OP == 39) OP16(I_CVTU);
      if constexpr (OP == 40) OP16(I_EXPFMA);
      if constexpr (OP == 41) OP16(I_EXPADD3);
      if constexpr (OP == 42) OP16(I_MULLOADD3);
      if constexpr (OP == 43) OP16(I_MED3);
      if constexpr (OP == 44) OP16(I_LSHL);
      if constexpr (OP == 45) OP16(I_FMAC);
      if constexpr (OP == 46) OP16(I_FMAMK);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc ^= r[i];
  if (acc == 0x9e3779b9u) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// 64-bit register pairs: packed fp32
#define OP8P(STR)                                                                                                              \
  asm volatile(STR(0) STR(1) STR(2) STR(3) STR(4) STR(5) STR(6) STR(7)                                                         \
               : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])                \
               : "v"(px), "v"(py))
#define P_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define P_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define P_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void pk_kernel(unsigned long long* __restrict__ stamps, unsigned* __restrict__ sink, int loops) {
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = (f2){1.0f + threadIdx.x * 1e-6f, 0.5f + i};
  f2 px = {1.0000001f, 0.9999999f}, py = {1e-7f, -1e-7f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int l = 0; l < loops; ++l) {
#pragma unroll
    for (int k = 0; k < REP / 8; ++k) {
      if constexpr (OP == 0) OP8P(P_FMA);
      if constexpr (OP == 1) OP8P(P_MUL);
      if constexpr (OP == 2) OP8P(P_ADD);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += p[i][0] + p[i][1];
  if (acc == 12345.678f) sink[0] = 1;
  if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static unsigned long long* d_st; static unsigned* d_sink;
template <typename K>
static double run(K kern, int W, int loops) {
  const int blocks = 256 * W;
  kern<<<blocks, 256, 0, 0>>>(d_st, d_sink, loops);
  CK(hipDeviceSynchronize());
  kern<<<blocks, 256, 0, 0>>>(d_st, d_sink, loops);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)blocks * 4);
  CK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[h.size() / 2] / ((double)loops * REP) / W;        // cycles per instruction per SIMD
}

int main() {
  CK(hipMalloc(&d_st, 256 * 8 * 4 * 8)); CK(hipMalloc(&d_sink, 64));
  const int loops = 400;
  printf("# gfx950 vector-instruction throughput: shader cycles per wave64 instruction per SIMD, by waves per SIMD (every CU busy, %d x %d instructions per wave)\n", loops, REP);
  printf("%-28s %8s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=3", "W=4");
#define ROW(NAME, KERN) { printf("%-28s", NAME); for (int W = 1; W <= 4; ++W) printf(" %8.2f", run(KERN, W, loops)); printf("\n"); }
  ROW("v_fma_f32", valu_kernel<0>) ROW("v_add_f32", valu_kernel<1>) ROW("v_mul_f32", valu_kernel<2>)
  ROW("v_exp_f32", valu_kernel<3>) ROW("v_rcp_f32", valu_kernel<4>)
  ROW("v_mul_lo_u32", valu_kernel<5>) ROW("v_mul_u32_u24", valu_kernel<6>) ROW("v_mad_u32_u24", valu_kernel<7>)
  ROW("v_cndmask_b32 (vcc)", valu_kernel<8>) ROW("v_cvt_pk_bf16_f32", valu_kernel<9>) ROW("v_bfe_i32", valu_kernel<10>)
  ROW("v_and_b32", valu_kernel<11>) ROW("v_xor_b32", valu_kernel<12>) ROW("v_lshrrev_b32", valu_kernel<13>)
  ROW("v_max3_f32", valu_kernel<14>) ROW("v_alignbit_b32", valu_kernel<15>) ROW("v_add_u32", valu_kernel<16>)
  ROW("v_add3_u32", valu_kernel<17>) ROW("v_xad_u32", valu_kernel<18>) ROW("v_lshl_or_b32", valu_kernel<19>)
  ROW("v_and_or_b32", valu_kernel<20>) ROW("v_perm_b32", valu_kernel<21>) ROW("v_pk_sub_u16 clamp", valu_kernel<22>)
  ROW("v_pk_min_u16", valu_kernel<23>) ROW("v_pk_mul_lo_u16", valu_kernel<24>) ROW("v_fma_f32 |a|, b, -c", valu_kernel<25>)
  ROW("v_cmp_gt_i32 vcc", valu_kernel<26>) ROW("v_mov_b32", valu_kernel<27>) ROW("v_add_f32_dpp quad_perm", valu_kernel<28>)
  ROW("v_bitop3_b32", valu_kernel<29>)
  ROW("v_cndmask_b32_e64 (sgpr pair)", valu_kernel<30>) ROW("v_sub_f32_e64 |a|", valu_kernel<31>) ROW("v_xor_b32_sdwa WORD_1", valu_kernel<32>)
  ROW("v_cmp_gt_i32_e64 -> sgpr", valu_kernel<33>) ROW("v_max_f32", valu_kernel<34>) ROW("v_ashrrev_i32", valu_kernel<35>)
  ROW("v_or_b32", valu_kernel<36>) ROW("v_sub_u32", valu_kernel<37>) ROW("v_mul_f32_e64 -a", valu_kernel<38>) ROW("v_cvt_f32_u32", valu_kernel<39>)
  ROW("v_med3_f32", valu_kernel<43>) ROW("v_lshlrev_b32", valu_kernel<44>) ROW("v_fmac_f32", valu_kernel<45>) ROW("v_fmamk_f32", valu_kernel<46>)
  printf("# mixed streams, per INSTRUCTION (2 resp. 4 instructions per copy)\n");
  { printf("%-28s", "exp + fma (x 1/2)"); for (int W = 1; W <= 4; ++W) printf(" %8.2f", run(valu_kernel<40>, W, loops) / 2); printf("\n"); }
  { printf("%-28s", "exp + add + mul + and (x 1/4)"); for (int W = 1; W <= 4; ++W) printf(" %8.2f", run(valu_kernel<41>, W, loops) / 4); printf("\n"); }
  { printf("%-28s", "mul_lo + add + mul + and (1/4)"); for (int W = 1; W <= 4; ++W) printf(" %8.2f", run(valu_kernel<42>, W, loops) / 4); printf("\n"); }
  ROW("v_pk_fma_f32 (2 per lane)", pk_kernel<0>) ROW("v_pk_mul_f32 (2 per lane)", pk_kernel<1>) ROW("v_pk_add_f32 (2 per lane)", pk_kernel<2>)
  return 0;
}
