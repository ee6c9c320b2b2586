// VALU issue-rate microbenchmark for gfx950 (MI355X).
// Measures sustained wave-instruction throughput of the vector ops the renderer's
// inner loops are built from, so DESIGN.md can state the *measured* vector-ALU peak.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 32768;
constexpr int CHAINS = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters, unsigned long long* clk) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x[CHAINS];
  double xd[CHAINS];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 xp[CHAINS];
  unsigned xu[CHAINS];
  for (int i = 0; i < CHAINS; ++i) {
    x[i] = threadIdx.x * 1e-3f + i; xd[i] = x[i]; xp[i] = f2{x[i], x[i] + 1.f}; xu[i] = threadIdx.x * 2654435761u + i;
  }
  f2 ap{a, a}, bp{b, b};
  double ad = a, bd = b;
  unsigned ua = __float_as_uint(a) | 1u;
#pragma unroll 4
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[i]) : "v"(ap), "v"(bp));
      if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(xd[i]) : "v"(ad), "v"(bd));
      if (OP == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
      if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
      if (OP == 5) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
      if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(xu[i]) : "v"(ua));
      if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(xu[i]) : "v"(ua));
      if (OP == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(xu[i]) : "v"(ua));
      if (OP == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
      if (OP == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (OP == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(xd[i]) : "v"(ad));
      if (OP == 12) asm volatile("v_add_f64 %0, %0, %1" : "+v"(xd[i]) : "v"(ad));
      if (OP == 13) asm volatile("v_sqrt_f64 %0, %0" : "+v"(xd[i]));
      if (OP == 14) asm volatile("v_rcp_f64 %0, %0" : "+v"(xd[i]));
      if (OP == 15) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(xu[i]) : "v"(ua));
      if (OP == 16) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(xu[i]));
      if (OP == 17) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(xp[i]) : "v"(ap));
      if (OP == 18) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[i]) : "v"(ap));
      if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (OP == 20) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(xu[i]) : "v"(ua));
      if (OP == 21) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
      if (OP == 22) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      if (OP == 23) asm volatile("v_log_f32 %0, %0" : "+v"(x[i]));
      if (OP == 24) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(b));
      if (OP == 25) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(xu[i]) : "v"(ua));
      if (OP == 26) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x[i]), "v"(a) : "vcc");
      if (OP == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
  float s = 0;
  for (int i = 0; i < CHAINS; ++i) s += x[i] + (float)xd[i] + xp[i].x + xp[i].y + (float)xu[i];
  if (s == 123.456f) out[0] = s;
}

template <int OP>
void run(const char* name, double flop_per_lane_op, float* d_out, int wps = 8) {
  const int blocks = 256 * wps, threads = 256;  // wps waves / SIMD
  unsigned long long* d_clk; CHECK(hipMalloc(&d_clk, 16)); unsigned long long h_clk[2];
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001f, 1e-9f, 64, d_clk);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001f, 1e-9f, ITERS, d_clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost)); CHECK(hipFree(d_clk));
  double ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1;  // s_memrealtime ticks at 100 MHz
  double wave_instr = (double)blocks * (threads / 64) * ITERS * CHAINS;
  double rate = wave_instr / (best * 1e-3);                 // wave-instr / s, whole chip
  double per_simd_cyc = (1024.0 * ghz * 1e9) / rate;       // cycles per wave-instr per SIMD at the measured clock
  printf("%-22s %8.3f ms  %7.3f T wave-instr/s  %6.2f cyc/instr/SIMD  clk %.2f GHz  wps %d  %8.2f TFLOP/s\n",
         name, best, rate * 1e-12, per_simd_cyc, ghz, wps, rate * 64 * flop_per_lane_op * 1e-12);
}

int main() {
  float* d_out; CHECK(hipMalloc(&d_out, 4));
  run<0>("v_fma_f32", 2, d_out);
  run<0>("v_fma_f32", 2, d_out, 4);
  run<0>("v_fma_f32", 2, d_out, 2);
  run<0>("v_fma_f32", 2, d_out, 1);
  run<1>("v_pk_fma_f32", 4, d_out, 4);
  run<2>("v_fma_f64", 2, d_out, 4);
  run<27>("v_fmac_f32", 2, d_out);
  run<24>("v_fma_f32 (sgpr src)", 2, d_out);
  run<1>("v_pk_fma_f32", 4, d_out);
  run<17>("v_pk_mul_f32", 2, d_out);
  run<18>("v_pk_add_f32", 2, d_out);
  run<10>("v_add_f32", 1, d_out);
  run<19>("v_max_f32", 1, d_out);
  run<2>("v_fma_f64", 2, d_out);
  run<11>("v_mul_f64", 1, d_out);
  run<12>("v_add_f64", 1, d_out);
  run<3>("v_sqrt_f32", 1, d_out);
  run<4>("v_rcp_f32", 1, d_out);
  run<5>("v_rsq_f32", 1, d_out);
  run<22>("v_exp_f32", 1, d_out);
  run<23>("v_log_f32", 1, d_out);
  run<13>("v_sqrt_f64", 1, d_out);
  run<14>("v_rcp_f64", 1, d_out);
  run<6>("v_mul_lo_u32", 0, d_out);
  run<20>("v_mul_hi_u32", 0, d_out);
  run<7>("v_mul_u32_u24", 0, d_out);
  run<8>("v_mad_u32_u24", 0, d_out);
  run<15>("v_xor_b32", 0, d_out);
  run<16>("v_lshrrev_b32", 0, d_out);
  run<25>("v_xad_u32", 0, d_out);
  run<21>("v_cvt_f32_u32", 0, d_out);
  run<26>("v_cmp_gt_f32", 0, d_out);
  run<9>("v_cmp+v_cndmask (2)", 0, d_out);
  return 0;
}
