// LDS read rate per CU on gfx950 for the three fragment-read forms the GEMM kernels use: ds_read_b128 (forward A/B
// fragments), ds_read_b64 and ds_read_b64_tr_b16 (weight-gradient / WT data-gradient fragments).  One 1024-thread
// workgroup (16 waves) per CU, conflict-free addresses, results folded into a value that is stored so nothing is optimised away.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip ; run: ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef s16x4 __attribute__((address_space(3))) * lds_p;
typedef const u32x4 __attribute__((address_space(3))) * lds_p128;
typedef const u32x2 __attribute__((address_space(3))) * lds_p64;

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[64 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 1024) ((unsigned*)smem)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0, accs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned char* base = smem + (wave & 3) * 16384;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int off = ((it + u) & 7) * 1024;
      asm volatile("" : "+v"(off));
      if (MODE == 0) {          // b128: 16 B per lane, consecutive lanes consecutive addresses
        u32x4 v = *(lds_p128)(base + off + lane * 16);
        accs[u] ^= v[0] ^ v[1] ^ v[2] ^ v[3];
      } else if (MODE == 1) {   // b64
        u32x2 v = *(lds_p64)(base + off + lane * 8);
        accs[u] ^= v[0] ^ v[1];
      } else {                  // b64 transposing read, the access shape of the weight-gradient kernel (128-byte rows)
        const int fr = lane & 15, fq = lane >> 4;
        const int row = 8 * fq + (fr >> 2), ch = (fr & 3) >> 1, o = 8 * (fr & 1);
        const int swz = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;          // tr_swz<64> of gemm_bf16.hip
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off / 2 + row * 128 + ((ch ^ swz) << 4) + o));
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        accs[u] ^= w[0] ^ w[1];
      }
    }
  }
  for (int u = 0; u < 8; u++) acc ^= accs[u];
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, int bytes_per_lane, unsigned* out) {
  const int iters = 10000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 1024 * iters * 8 * bytes_per_lane;
  printf("%-22s %8.3f ms  %8.1f GB/s per CU  (%5.1f B/clk at 2.4 GHz)\n", name, ms, bytes / ms / 1e6 / blocks,
         bytes / ms / 1e6 / blocks / 2.4);
}

int main() {
  unsigned* out;
  hipMalloc(&out, 256 * 1024 * 4);
  run<0>("ds_read_b128", 16, out);
  run<1>("ds_read_b64", 8, out);
  run<2>("ds_read_b64_tr_b16", 8, out);
  run<0>("ds_read_b128", 16, out);
  hipFree(out);
  return 0;
}
