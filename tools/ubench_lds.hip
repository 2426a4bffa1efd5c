// ubench_lds.hip -- what the LDS pipe of a gfx950 CU moves per cycle for the access shapes of the GEMM kernels:
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_lds tools/ubench_lds.hip
// One 256-thread workgroup per CU (4 waves, one per SIMD) or two; every wave issues N conflict-free ds_read / ds_write of
// 4 / 8 / 16 bytes per lane back to back; shader cycles (s_memtime) per CU per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

enum { R128, R64, R32, W128, W64, RW128, R128_PAD };
__device__ unsigned long long ticks[2];

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16384; i += blockDim.x) smem[i] = i;
  __syncthreads();
  // lane-contiguous accesses inside a per-wave 4 KB window (conflict-free by construction); R128_PAD: the fp32 GEMM's pattern --
  // lane (c, q) reads 16 bytes at row c (stride 144 bytes) + 32 q
  const int c = lane & 15, q = lane >> 4;
  const unsigned base = (KIND == R128_PAD) ? (unsigned)(w * 4096 + c * 144 + q * 32) : (unsigned)(w * 4096 + lane * 16);
  u4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == R128 || KIND == R128_PAD) {
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(base));
    } else if (KIND == R64) {
      u2v b0, b1, b2, b3;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n ds_read_b64 %3, %4 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"((unsigned)(w * 4096 + lane * 8)));
      a0.x ^= b0.x ^ b1.x ^ b2.x ^ b3.x;
    } else if (KIND == R32) {
      unsigned b0, b1, b2, b3;
      asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"((unsigned)(w * 4096 + lane * 4)));
      a0.x ^= b0 ^ b1 ^ b2 ^ b3;
    } else if (KIND == W128) {
      asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" :: "v"(base), "v"(a1) : "memory");
    } else if (KIND == W64) {
      u2v b = {1, 2};
      asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" :: "v"((unsigned)(w * 4096 + lane * 8)), "v"(b) : "memory");
    } else if (KIND == RW128) {
      asm volatile("ds_read_b128 %0, %2\n ds_write_b128 %2, %3 offset:1024\n ds_read_b128 %1, %2 offset:2048\n ds_write_b128 %2, %3 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1) : "v"(base), "v"(a3) : "memory");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) { atomicAdd(&ticks[0], t1 - t0); atomicAdd(&ticks[1], 1ull); }
  if (a0.x + a1.y + a2.z + a3.w == 0x12345678u) sink[0] = a0.x;
}

template <int KIND>
static void run(const char* name, int threads, int bytes_per_lane) {
  unsigned* sink;
  hipMalloc(&sink, 4);
  unsigned long long z[2] = {0, 0}, t[2];
  hipMemcpyToSymbol(HIP_SYMBOL(ticks), z, sizeof(z));
  const int iters = 2000;
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 65536, 0, iters, sink);
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(t, HIP_SYMBOL(ticks), sizeof(t));
  const double cyc = (double)t[0] / (double)t[1] / iters;                 // per group of 4 instructions per wave
  const int waves = threads / 64;
  printf("%-44s %d waves/CU: %.1f cycles per 4 instr of a wave -> %.1f cycles per wave-instruction on the CU's pipe = %.1f bytes / cycle / CU\n",
         name, waves, cyc, cyc / (4.0 * waves), 4.0 * waves * 64 * bytes_per_lane / cyc);
  hipFree(sink);
}

int main() {
  for (int threads : {256, 512}) {
    run<R32>("ds_read_b32  lane-contiguous", threads, 4);
    run<R64>("ds_read_b64  lane-contiguous", threads, 8);
    run<R128>("ds_read_b128 lane-contiguous", threads, 16);
    run<R128_PAD>("ds_read_b128 rows of 144 B (fp32 GEMM tile)", threads, 16);
    run<W64>("ds_write_b64 lane-contiguous", threads, 8);
    run<W128>("ds_write_b128 lane-contiguous", threads, 16);
    run<RW128>("ds_read_b128 + ds_write_b128 alternating", threads, 16);
  }
  return 0;
}
