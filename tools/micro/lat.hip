// single-lane latency micro-benchmarks on gfx950: dependent LDS reads, dependent VALU adds, dependent taken branches
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_lds_chase(int *out, int n, int lanes) {
  __shared__ int s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 97 + 13) & 4095;
  __syncthreads();
  if ((int)threadIdx.x < lanes) {
    int p = threadIdx.x;
    for (int i = 0; i < n; i++) p = s[p];
    out[threadIdx.x] = p;
  }
}
__global__ void k_valu_chain(int *out, int n, int lanes) {
  if ((int)threadIdx.x < lanes) {
    int p = threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
      asm volatile("v_add_u32 %0, %0, 3\n v_xor_b32 %0, %0, 5\n v_add_u32 %0, %0, 7\n v_xor_b32 %0, %0, 9\n v_add_u32 %0, %0, 3\n v_xor_b32 %0, %0, 5\n v_add_u32 %0, %0, 7\n v_xor_b32 %0, %0, 9" : "+v"(p));
    }
    out[threadIdx.x] = p;
  }
}
__global__ void k_branchy(int *out, int n, int lanes) {
  if ((int)threadIdx.x < lanes) {
    int p = threadIdx.x, q = 0;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
      if (p & 1) { p = p * 3 + 1; q++; } else { p >>= 1; }
      if (p & 2) { p ^= 0x55; } else { p += 11; q += 2; }
      if (p & 4) { p += q; } else { p -= 3; }
      if (p & 8) { p ^= q; } else { p += 5; }
    }
    out[threadIdx.x] = p + q;
  }
}
template <class F> double run(F f) {
  f();
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  f();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
  int *out;
  hipMalloc(&out, 4096);
  const int n = 200000;
  for (int lanes : {1, 16, 64}) {
    double a = run([&] { hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, out, n, lanes); });
    double b = run([&] { hipLaunchKernelGGL(k_valu_chain, dim3(1), dim3(64), 0, 0, out, n, lanes); });
    double c = run([&] { hipLaunchKernelGGL(k_branchy, dim3(1), dim3(64), 0, 0, out, n, lanes); });
    printf("lanes %2d: dependent LDS read %.1f ns, dependent VALU op %.2f ns, divergent if/else pair %.1f ns\n", lanes, a * 1e3 / n, b * 1e3 / (8.0 * n), c * 1e3 / (4.0 * n));
  }
  // many waves on one CU: 8 waves (2 per SIMD), each one lane chasing
  double d = run([&] { hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(512), 0, 0, out, n, 512); });
  printf("8 full waves in one workgroup: dependent LDS read %.1f ns per wave-step\n", d * 1e3 / n);
  return 0;
}
