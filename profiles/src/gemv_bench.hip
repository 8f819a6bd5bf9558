// micro-benchmark: weight-streaming GEMV variants at the 7B gate/up shape (N=2*18944, K=3584, M=8)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// V0: as shipped: 4 waves split K, fragment-shaped loads, pragma unroll U
template <int U>
__global__ __launch_bounds__(256) void v0(const bf16* __restrict__ x, const bf16* __restrict__ W, float* __restrict__ out, int N, int K) {
  __shared__ float red[4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int steps = K / 32 / 4, ks0 = wave * steps;
  const bf16* wp = W + (long)(n0 + r) * K + q * 8;
  const bool xv = r < 8;
  const bf16* xp = x + (long)(xv ? r : 0) * K + q * 8;
  f32x4 acc = {0, 0, 0, 0};
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll U
  for (int ks = ks0; ks < ks0 + steps; ++ks) {
    bf16x8 a = *(const bf16x8*)(wp + ks * 32);
    bf16x8 b = xv ? *(const bf16x8*)(xp + ks * 32) : z;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) red[wave][lane][i] = acc[i];
  __syncthreads();
  if (wave == 0 && xv) {
    float4 o;
    o.x = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
    o.y = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
    o.z = red[0][lane][2] + red[1][lane][2] + red[2][lane][2] + red[3][lane][2];
    o.w = red[0][lane][3] + red[1][lane][3] + red[2][lane][3] + red[3][lane][3];
    *(float4*)(out + (long)r * N + n0 + 4 * q) = o;
  }
}

// V1: explicit batch of D loads in flight (registers), then D MFMAs; 4 waves split K
template <int D>
__global__ __launch_bounds__(256) void v1(const bf16* __restrict__ x, const bf16* __restrict__ W, float* __restrict__ out, int N, int K) {
  __shared__ float red[4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int steps = K / 32 / 4, ks0 = wave * steps;
  const bf16* wp = W + (long)(n0 + r) * K + q * 8 + (long)ks0 * 32;
  const bool xv = r < 8;
  const bf16* xp = x + (long)(xv ? r : 0) * K + q * 8 + (long)ks0 * 32;
  f32x4 acc = {0, 0, 0, 0};
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < steps; s += D) {
    bf16x8 a[D], b[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = __builtin_nontemporal_load((const bf16x8*)(wp + (s + i) * 32));
#pragma unroll
    for (int i = 0; i < D; ++i) b[i] = xv ? *(const bf16x8*)(xp + (s + i) * 32) : z;
#pragma unroll
    for (int i = 0; i < D; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[i], acc, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) red[wave][lane][i] = acc[i];
  __syncthreads();
  if (wave == 0 && xv) {
    float4 o;
    o.x = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
    o.y = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
    o.z = red[0][lane][2] + red[1][lane][2] + red[2][lane][2] + red[3][lane][2];
    o.w = red[0][lane][3] + red[1][lane][3] + red[2][lane][3] + red[3][lane][3];
    *(float4*)(out + (long)r * N + n0 + 4 * q) = o;
  }
}

// V2: PACKED weights: tile (nt, kt) is 1 KiB contiguous, lane-linear (lane l at +16*l) -> each load = 1 KiB contiguous
template <int D>
__global__ __launch_bounds__(256) void v2(const bf16* __restrict__ x, const bf16* __restrict__ Wp, float* __restrict__ out, int N, int K) {
  __shared__ float red[4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x;
  const int ksteps = K / 32, steps = ksteps / 4, ks0 = wave * steps;
  const bf16* wp = Wp + ((long)nt * ksteps + ks0) * 512 + lane * 8;
  const bool xv = r < 8;
  const bf16* xp = x + (long)(xv ? r : 0) * K + q * 8 + (long)ks0 * 32;
  f32x4 acc = {0, 0, 0, 0};
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < steps; s += D) {
    bf16x8 a[D], b[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = __builtin_nontemporal_load((const bf16x8*)(wp + (long)(s + i) * 512));
#pragma unroll
    for (int i = 0; i < D; ++i) b[i] = xv ? *(const bf16x8*)(xp + (s + i) * 32) : z;
#pragma unroll
    for (int i = 0; i < D; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[i], acc, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) red[wave][lane][i] = acc[i];
  __syncthreads();
  if (wave == 0 && xv) {
    float4 o;
    o.x = red[0][lane][0] + red[1][lane][0] + red[2][lane][0] + red[3][lane][0];
    o.y = red[0][lane][1] + red[1][lane][1] + red[2][lane][1] + red[3][lane][1];
    o.z = red[0][lane][2] + red[1][lane][2] + red[2][lane][2] + red[3][lane][2];
    o.w = red[0][lane][3] + red[1][lane][3] + red[2][lane][3] + red[3][lane][3];
    *(float4*)(out + (long)r * N + nt * 16 + 4 * q) = o;
  }
}

// V3: packed, one wave = one n-tile over the full K (no LDS reduce), 4 independent waves per WG
template <int D>
__global__ __launch_bounds__(256) void v3(const bf16* __restrict__ x, const bf16* __restrict__ Wp, float* __restrict__ out, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x * 4 + wave;
  const int ksteps = K / 32;
  const bf16* wp = Wp + ((long)nt * ksteps) * 512 + lane * 8;
  const bool xv = r < 8;
  const bf16* xp = x + (long)(xv ? r : 0) * K + q * 8;
  f32x4 acc = {0, 0, 0, 0};
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < ksteps; s += D) {
    bf16x8 a[D], b[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = __builtin_nontemporal_load((const bf16x8*)(wp + (long)(s + i) * 512));
#pragma unroll
    for (int i = 0; i < D; ++i) b[i] = xv ? *(const bf16x8*)(xp + (s + i) * 32) : z;
#pragma unroll
    for (int i = 0; i < D; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[i], acc, 0, 0, 0);
  }
  if (xv) *(float4*)(out + (long)r * N + nt * 16 + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}


// V4: packed, NW waves split K for one tile pair (T tiles per wave per k-step share the B fragment), depth D
template <int NW, int T, int D>
__global__ __launch_bounds__(NW * 64) void v4(const bf16* __restrict__ x, const bf16* __restrict__ Wp, float* __restrict__ out, int N, int K, int tile_stride) {
  __shared__ float red[T][NW][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x;
  const int ksteps = K / 32, steps = ksteps / NW, ks0 = wave * steps;
  const bf16* wp = Wp + ((long)nt * ksteps + ks0) * 512 + lane * 8;
  const bool xv = r < 8;
  const bf16* xp = x + (long)(xv ? r : 0) * K + q * 8 + (long)ks0 * 32;
  f32x4 acc[T];
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < steps; s += D) {
    bf16x8 a[T][D], b[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) a[t][i] = __builtin_nontemporal_load((const bf16x8*)(wp + (long)t * tile_stride * ksteps * 512 + (long)(s + i) * 512));
#pragma unroll
    for (int i = 0; i < D; ++i) b[i] = xv ? *(const bf16x8*)(xp + (s + i) * 32) : z;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][i], b[i], acc[t], 0, 0, 0);
  }
  if (NW > 1) {
    for (int t = 0; t < T; ++t) for (int i = 0; i < 4; ++i) red[t][wave][lane][i] = acc[t][i];
    __syncthreads();
    if (wave == 0 && xv) {
      for (int t = 0; t < T; ++t) {
        float o[4];
        for (int i = 0; i < 4; ++i) { o[i] = 0; for (int w = 0; w < NW; ++w) o[i] += red[t][w][lane][i]; }
        *(float4*)(out + (long)r * N + (nt + t * tile_stride) * 16 + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  } else if (xv) {
    for (int t = 0; t < T; ++t) *(float4*)(out + (long)r * N + (nt + t * tile_stride) * 16 + 4 * q) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
}

// V5: like V4 but x slice staged in LDS once per workgroup (B fragments from LDS)
template <int NW, int T, int D>
__global__ __launch_bounds__(NW * 64) void v5(const bf16* __restrict__ x, const bf16* __restrict__ Wp, float* __restrict__ out, int N, int K, int tile_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* xs = (bf16*)smem;                         // [8][K]
  float (*red)[NW][64][4] = (float (*)[NW][64][4])(smem + 8 * K * 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  for (int i = threadIdx.x; i < 8 * K / 8; i += NW * 64) ((bf16x8*)xs)[i] = ((const bf16x8*)x)[i];
  __syncthreads();
  const int nt = blockIdx.x;
  const int ksteps = K / 32, steps = ksteps / NW, ks0 = wave * steps;
  const bf16* wp = Wp + ((long)nt * ksteps + ks0) * 512 + lane * 8;
  const bool xv = r < 8;
  const bf16* xp = xs + (long)(xv ? r : 0) * K + q * 8 + (long)ks0 * 32;
  f32x4 acc[T];
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < steps; s += D) {
    bf16x8 a[T][D], b[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) a[t][i] = __builtin_nontemporal_load((const bf16x8*)(wp + (long)t * tile_stride * ksteps * 512 + (long)(s + i) * 512));
#pragma unroll
    for (int i = 0; i < D; ++i) b[i] = xv ? *(const bf16x8*)(xp + (s + i) * 32) : z;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][i], b[i], acc[t], 0, 0, 0);
  }
  for (int t = 0; t < T; ++t) for (int i = 0; i < 4; ++i) red[t][wave][lane][i] = acc[t][i];
  __syncthreads();
  if (wave == 0 && xv) {
    for (int t = 0; t < T; ++t) {
      float o[4];
      for (int i = 0; i < 4; ++i) { o[i] = 0; for (int w = 0; w < NW; ++w) o[i] += red[t][w][lane][i]; }
      *(float4*)(out + (long)r * N + (nt + t * tile_stride) * 16 + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// copy kernel: float4 streaming read (+ tiny write) as the achievable-HBM yardstick
__global__ __launch_bounds__(256) void stream_read(const f32x4* __restrict__ p, long n, float* out) {
  float4 s = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    f32x4 v = __builtin_nontemporal_load(p + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1;
}

int main() {
  const int N = 2 * 18944, K = 3584, NB = 6;   // 6 distinct 271 MB weight buffers > 256 MiB infinity cache
  const size_t wbytes = (size_t)N * K * 2;
  std::vector<bf16*> W(NB);
  for (auto& p : W) { CK(hipMalloc(&p, wbytes)); CK(hipMemset(p, 0x3c, wbytes)); }
  bf16* x; CK(hipMalloc(&x, 8 * K * 2)); CK(hipMemset(x, 0x3c, 8 * K * 2));
  float* out; CK(hipMalloc(&out, 16 * N * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < NB; ++i) launch(W[i]);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) for (int i = 0; i < NB; ++i) launch(W[i]);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * NB);
    printf("%-28s %8.2f us  %7.1f GB/s\n", name, us, wbytes / us / 1e3);
  };
  run("stream_read 2048 blk", [&](bf16* w) { hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const f32x4*)w, (long)(wbytes / 16), out); });
  run("stream_read 8192 blk", [&](bf16* w) { hipLaunchKernelGGL(stream_read, dim3(8192), dim3(256), 0, 0, (const f32x4*)w, (long)(wbytes / 16), out); });
  run("v0 unroll4", [&](bf16* w) { hipLaunchKernelGGL(v0<4>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v0 unroll7", [&](bf16* w) { hipLaunchKernelGGL(v0<7>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v1 depth4", [&](bf16* w) { hipLaunchKernelGGL(v1<4>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v1 depth7", [&](bf16* w) { hipLaunchKernelGGL(v1<7>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v1 depth14", [&](bf16* w) { hipLaunchKernelGGL(v1<14>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v2 packed depth7", [&](bf16* w) { hipLaunchKernelGGL(v2<7>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v2 packed depth14", [&](bf16* w) { hipLaunchKernelGGL(v2<14>, dim3(N / 16), dim3(256), 0, 0, x, w, out, N, K); });
  run("v3 packed fullK depth8", [&](bf16* w) { hipLaunchKernelGGL(v3<8>, dim3(N / 64), dim3(256), 0, 0, x, w, out, N, K); });
  run("v3 packed fullK depth16", [&](bf16* w) { hipLaunchKernelGGL(v3<16>, dim3(N / 64), dim3(256), 0, 0, x, w, out, N, K); });

  const int NT = N / 16;
  run("v4 NW2 T1 D7", [&](bf16* w) { hipLaunchKernelGGL((v4<2,1,7>), dim3(NT), dim3(128), 0, 0, x, w, out, N, K, 0); });
  run("v4 NW2 T1 D14", [&](bf16* w) { hipLaunchKernelGGL((v4<2,1,14>), dim3(NT), dim3(128), 0, 0, x, w, out, N, K, 0); });
  run("v4 NW4 T1 D7", [&](bf16* w) { hipLaunchKernelGGL((v4<4,1,7>), dim3(NT), dim3(256), 0, 0, x, w, out, N, K, 0); });
  run("v4 NW4 T2 D7 (gate/up)", [&](bf16* w) { hipLaunchKernelGGL((v4<4,2,7>), dim3(NT/2), dim3(256), 0, 0, x, w, out, N, K, NT/2); });
  run("v4 NW4 T2 D4 (gate/up)", [&](bf16* w) { hipLaunchKernelGGL((v4<4,2,4>), dim3(NT/2), dim3(256), 0, 0, x, w, out, N, K, NT/2); });
  run("v4 NW2 T2 D7 (gate/up)", [&](bf16* w) { hipLaunchKernelGGL((v4<2,2,7>), dim3(NT/2), dim3(128), 0, 0, x, w, out, N, K, NT/2); });
  run("v4 NW8 T2 D7 (gate/up)", [&](bf16* w) { hipLaunchKernelGGL((v4<8,2,7>), dim3(NT/2), dim3(512), 0, 0, x, w, out, N, K, NT/2); });
  run("v4 NW8 T1 D7", [&](bf16* w) { hipLaunchKernelGGL((v4<8,1,7>), dim3(NT), dim3(512), 0, 0, x, w, out, N, K, 0); });
  run("v4 NW1 T2 D8", [&](bf16* w) { hipLaunchKernelGGL((v4<1,2,8>), dim3(NT/2), dim3(64), 0, 0, x, w, out, N, K, NT/2); });
  run("v5 lds-x NW4 T2 D7", [&](bf16* w) { hipLaunchKernelGGL((v5<4,2,7>), dim3(NT/2), dim3(256), 8*K*2 + 2*4*64*4*4, 0, x, w, out, N, K, NT/2); });
  run("v5 lds-x NW8 T2 D7", [&](bf16* w) { hipLaunchKernelGGL((v5<8,2,7>), dim3(NT/2), dim3(512), 8*K*2 + 2*8*64*4*4, 0, x, w, out, N, K, NT/2); });
  return 0;
}
