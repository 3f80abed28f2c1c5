// Instruction issue-rate microbenchmark for gfx950: cycles per wave-instruction of the VALU ops the
// tracer is made of (one wave per SIMD, independent chains).  Build: hipcc --offload-arch=gfx950 -O3 inst_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 4096
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t u0 = threadIdx.x + 7, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p0 = {f0, f1}, p1 = {f1, f2}, p2 = {f2, f3}, p3 = {f3, f0};
  const f2v pc = {1.0000001f, 0.9999999f}, pe = {0.5f, 0.25f};
  const double c = 1.0000001, e = 0.9999999;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_IT; ++i) {
    if (OP == 0) { a0 = fma(a0, c, e); a1 = fma(a1, c, e); a2 = fma(a2, c, e); a3 = fma(a3, c, e); a4 = fma(a4, c, e); a5 = fma(a5, c, e); a6 = fma(a6, c, e); a7 = fma(a7, c, e); }
    if (OP == 1) { a0 = a0 * c; a1 = a1 * c; a2 = a2 * c; a3 = a3 * c; a4 = a4 * c; a5 = a5 * c; a6 = a6 * c; a7 = a7 * c; }
    if (OP == 2) { a0 = a0 + e; a1 = a1 + e; a2 = a2 + e; a3 = a3 + e; a4 = a4 + e; a5 = a5 + e; a6 = a6 + e; a7 = a7 + e; }
    if (OP == 3) { uint64_t p0 = (uint64_t)u0 * 0xD2511F53u, p1 = (uint64_t)u1 * 0xCD9E8D57u, p2 = (uint64_t)u2 * 0xD2511F53u, p3 = (uint64_t)u3 * 0xCD9E8D57u;
                   u0 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; u1 = (uint32_t)(p1 >> 32) ^ (uint32_t)p2; u2 = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; u3 = (uint32_t)(p3 >> 32) ^ (uint32_t)p0; }
    if (OP == 4) { a0 = sqrt(a0) + 2.0; a1 = sqrt(a1) + 2.0; a2 = sqrt(a2) + 2.0; a3 = sqrt(a3) + 2.0; }
    if (OP == 5) { a0 = 3.0 / a0 + 2.0; a1 = 3.0 / a1 + 2.0; a2 = 3.0 / a2 + 2.0; a3 = 3.0 / a3 + 2.0; }
    if (OP == 6) { f0 = fmaf(f0, 1.0000001f, 0.5f); f1 = fmaf(f1, 1.0000001f, 0.5f); f2 = fmaf(f2, 1.0000001f, 0.5f); f3 = fmaf(f3, 1.0000001f, 0.5f); a0 = a0; }
    if (OP == 7) { a0 = __builtin_amdgcn_rcp(a0) + 2.0; a1 = __builtin_amdgcn_rcp(a1) + 2.0; a2 = __builtin_amdgcn_rcp(a2) + 2.0; a3 = __builtin_amdgcn_rcp(a3) + 2.0; }
    if (OP == 8) { u0 = u0 * 0xD2511F53u + 1; u1 = u1 * 0xCD9E8D57u + 1; u2 = u2 * 0xD2511F53u + 1; u3 = u3 * 0xCD9E8D57u + 1; }
    if (OP == 10) { p0 = __builtin_elementwise_fma(p0, pc, pe); p1 = __builtin_elementwise_fma(p1, pc, pe); p2 = __builtin_elementwise_fma(p2, pc, pe); p3 = __builtin_elementwise_fma(p3, pc, pe); }
    if (OP == 11) { p0 = p0 * pc; p1 = p1 * pc; p2 = p2 * pc; p3 = p3 * pc; }
    if (OP == 12) { u0 = (f0 < f1) ? u1 : u0 + 1; u1 = (f1 < f2) ? u2 : u1 + 1; u2 = (f2 < f3) ? u3 : u2 + 1; u3 = (f3 < f0) ? u0 : u3 + 1; }
    if (OP == 9) { u0 = (u0 ^ u1) + 0x9E3779B9u; u1 = (u1 ^ u2) + 0x9E3779B9u; u2 = (u2 ^ u3) + 0x9E3779B9u; u3 = (u3 ^ u0) + 0x9E3779B9u; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int per_iter, int waves_per_simd) {
  double* out; unsigned long long* cyc;
  const int blocks = 256, threads = 256 * waves_per_simd;   // 4 SIMDs per CU
  hipMalloc(&out, blocks * threads * 8); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(out, cyc, 1.5);
  hipEventRecord(e0);
  for (int rep = 0; rep < 20; ++rep) k<OP><<<blocks, threads>>>(out, cyc, 1.5);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  // wave-level source ops executed per SIMD per second
  const double wave_ops = 20.0 * blocks * (threads / 64) * (double)N_IT * per_iter;
  const double ns_per_op_per_simd = ms * 1e6 / (wave_ops / 1024.0);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i]; s /= blocks;
  // s_memtime ticks at 100 MHz?  report raw ticks per (iteration*ops) and per wave-instruction assuming waves_per_simd share a SIMD
  printf("%-26s waves/SIMD %d: %7.3f memtime ticks/op/wave | wall: %6.3f ns per wave-op per SIMD = %5.2f cycles @2.4GHz\n", name, waves_per_simd,
         s / N_IT / per_iter, ns_per_op_per_simd, ns_per_op_per_simd * 2.4);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 4}) {
    run<0>("v_fma_f64", 8, w); run<1>("v_mul_f64", 8, w); run<2>("v_add_f64", 8, w);
    run<3>("u64=u32*u32 (+xor)", 4, w); run<8>("u32 mul+add", 4, w); run<9>("u32 xor+add", 4, w);
    run<4>("sqrt f64 (IEEE) + add", 4, w); run<5>("div f64 (IEEE) + add", 4, w); run<7>("v_rcp_f64 + add", 4, w); run<6>("v_fma_f32", 4, w);
    run<10>("v_pk_fma_f32", 4, w); run<11>("v_pk_mul_f32", 4, w); run<12>("v_cmp_f32+cndmask+add (3)", 4, w);
  }
  return 0;
}
