// Inner-loop probe (no global traffic): how fast can a CU turn LDS-resident 256 x 256 x 64 K-tiles into MFMAs,
//   variant A: 8 waves, wave tile 128 x 64, v_mfma_f32_16x16x32_bf16 (the product kernel's shape: 24 ds_read_b128 per 64 MFMA per wave)
//   variant B: 4 waves, wave tile 128 x 128, v_mfma_f32_32x32x16_bf16, accumulators 256 registers (32 ds_read_b128 per 64 MFMA per wave;
//              128 KB of LDS reads per K-tile per CU instead of 192 KB)
// Both alternate between two LDS images of the K-tile (so the reads are not loop-invariant) and, with DMA=1, every wave also issues its
// share of 64 KB of direct-to-LDS loads per K-tile into a third image (a 1 MB global buffer that stays in L2), to load the LDS write port
// as the real staging does.  One workgroup per CU (128 KB+ of LDS), 256 workgroups.  Prints ns per K-tile and the implied TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_loop_probe.hip -o /tmp/mfma_loop_probe && /tmp/mfma_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define IMG (256 * 128)            // one operand image: 256 rows x 64 bf16 = 32 KB
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ int off(int row, int ch) { return row * 128 + ((ch ^ (row & 7)) << 4); }

template <int DMA>
__device__ __forceinline__ void dma_issue(const char* g, char* lds_dst, int w, int nw, int lane, int n_instr) {
  if (!DMA) return;
  // n_instr wave-instructions of 1 KB each (16 B per lane) into consecutive 1-KB blocks
  for (int i = 0; i < n_instr; ++i) {
    const char* src = g + ((size_t)(i * nw + w) * 1024 + lane * 16);
    const unsigned dst = (unsigned)(uintptr_t)LDS_PTR(void, lds_dst) + (unsigned)__builtin_amdgcn_readfirstlane((i * nw + w) * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
  }
}

template <int DMA>
__global__ __launch_bounds__(512) void probe_a(const char* g, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 x (A img, B img)] + DMA image
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
  const int lr = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 4 * IMG / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const char* A = lds + (it & 1) * 2 * IMG;
    const char* B = A + IMG;
    dma_issue<DMA>(g + (size_t)(it & 7) * 65536, lds + 4 * IMG, w, 8, lane, 8);   // 8 waves x 8 x 1 KB = 64 KB
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xa[8], wb[4];
#pragma unroll
      for (int m = 0; m < 8; ++m) xa[m] = *reinterpret_cast<const bf16x8*>(A + off(wr * 128 + m * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int n = 0; n < 4; ++n) wb[n] = *reinterpret_cast<const bf16x8*>(B + off(wc * 64 + n * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[n], xa[m], acc[m][n], 0, 0, 0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
}

// variant A2: variant A with the fragment reads of the NEXT k-substep issued before the MFMAs of the current one (two register sets),
// so that a wave's own LDS latency hides under its own MFMAs instead of relying on the SIMD's other wave
__global__ __launch_bounds__(512) void probe_a2(const char* g, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
  const int lr = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 4 * IMG / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  bf16x8 xa[2][8], wb[2][4];
  auto load = [&](int buf, const char* A, const char* B, int ks) {
#pragma unroll
    for (int m = 0; m < 8; ++m) xa[buf][m] = *reinterpret_cast<const bf16x8*>(A + off(wr * 128 + m * 16 + lr, ks * 4 + lg));
#pragma unroll
    for (int n = 0; n < 4; ++n) wb[buf][n] = *reinterpret_cast<const bf16x8*>(B + off(wc * 64 + n * 16 + lr, ks * 4 + lg));
  };
  load(0, lds, lds + IMG, 0);
  for (int it = 0; it < iters; ++it) {
    const char* A = lds + (it & 1) * 2 * IMG;
    const char* An = lds + ((it + 1) & 1) * 2 * IMG;
    load(1, A, A + IMG, 1);
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0][n], xa[0][m], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_s_barrier();   // (the image pair of the next iteration is "ready": where the real kernel's barrier sits)
    load(0, An, An + IMG, 0);
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1][n], xa[1][m], acc[m][n], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s + (float)xa[0][0][0];
}

template <int DMA>
__global__ __launch_bounds__(256) void probe_b(const char* g, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  const int r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 4 * IMG / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const char* A = lds + (it & 1) * 2 * IMG;
    const char* B = A + IMG;
    dma_issue<DMA>(g + (size_t)(it & 7) * 65536, lds + 4 * IMG, w, 4, lane, 16);  // 4 waves x 16 x 1 KB = 64 KB
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {   // 16-deep k-substeps: lane (r, h) holds k = 8h .. 8h+7 -> 16-B chunk ks*2 + h of the row
      bf16x8 xa[4], wb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) xa[m] = *reinterpret_cast<const bf16x8*>(A + off(wr * 128 + m * 32 + r, ks * 2 + h));
#pragma unroll
      for (int n = 0; n < 4; ++n) wb[n] = *reinterpret_cast<const bf16x8*>(B + off(wc * 128 + n * 32 + r, ks * 2 + h));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[n], xa[m], acc[m][n], 0, 0, 0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
  out[blockIdx.x * 256 + tid] = s;
}

// variant C: the matrix pipe alone -- the same 64 MFMAs per wave and K-tile on register operands that never change (no LDS, no barrier):
// the sustained dense bf16 rate of the chip at whatever clock it holds under that load
__global__ __launch_bounds__(512) void probe_c(float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 xa[8], wb[4];
  for (int m = 0; m < 8; ++m) for (int j = 0; j < 8; ++j) xa[m][j] = (__bf16)(0.01f * ((tid + m + j) & 15));
  for (int n = 0; n < 4; ++n) for (int j = 0; j < 8; ++j) wb[n][j] = (__bf16)(0.02f * ((tid + n + j) & 7));
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[n], xa[m], acc[m][n], 0, 0, 0);
    asm volatile("" : "+v"(xa[0]), "+v"(wb[0]));
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(s);
  launch();
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  return ms;
}

int main() {
  char* g; float* out;
  hipMalloc(&g, 8 * 65536); hipMemset(g, 0, 8 * 65536);
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000, grid = 256;
  const size_t smem = 5 * IMG;   // 160 KB: two (A, B) image pairs + one DMA image
  hipFuncSetAttribute((const void*)probe_a<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)probe_a<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)probe_b<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)probe_b<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const double flop = 2.0 * 256 * 256 * 64;
  hipFuncSetAttribute((const void*)probe_a2, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  struct { const char* name; float ms; } r[7];
  r[6] = {"A2 variant A, next k-substep's fragments read before the current MFMAs", time_ms([&] { hipLaunchKernelGGL(probe_a2, dim3(grid), dim3(512), smem, 0, g, out, iters); })};
  r[4] = {"C  8 waves, MFMAs on register operands only         ", time_ms([&] { hipLaunchKernelGGL(probe_c, dim3(grid), dim3(512), 0, 0, out, iters); })};
  r[5] = {"C  ... on 32 CUs only (one workgroup per 8 CUs)      ", time_ms([&] { hipLaunchKernelGGL(probe_c, dim3(32), dim3(512), 0, 0, out, iters); })};
  r[0] = {"A  8 waves, 128 x 64 per wave, 16x16x32, LDS only ", time_ms([&] { hipLaunchKernelGGL(probe_a<0>, dim3(grid), dim3(512), smem, 0, g, out, iters); })};
  r[1] = {"A  ... + 64 KB of direct-to-LDS loads per K-tile    ", time_ms([&] { hipLaunchKernelGGL(probe_a<1>, dim3(grid), dim3(512), smem, 0, g, out, iters); })};
  r[2] = {"B  4 waves, 128 x 128 per wave, 32x32x16, LDS only  ", time_ms([&] { hipLaunchKernelGGL(probe_b<0>, dim3(grid), dim3(256), smem, 0, g, out, iters); })};
  r[3] = {"B  ... + 64 KB of direct-to-LDS loads per K-tile    ", time_ms([&] { hipLaunchKernelGGL(probe_b<1>, dim3(grid), dim3(256), smem, 0, g, out, iters); })};
  for (auto& x : r) {
    const double ns = x.ms * 1e6 / iters;
    printf("%s: %7.1f ns per K-tile  = %6.0f TFLOP/s if 256 CUs ran at this rate\n", x.name, ns, flop * 256 / (ns * 1e-9) / 1e12);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(err));
  return 0;
}
