// micro-benchmark of NT GEMM variants (activations row-major bf16 x tile-packed bf16 weights -> bf16), LM / ViT shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// ---------------- V0: shipped kernel (register staged, 128x128x64, 1 barrier / k-tile)
__global__ __launch_bounds__(256) void v0(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128, fr = lane & 15, fq = lane >> 4;
  const bf16* ga[4]; const bf16* gw[4]; int lo[4], lw[4];
  const int ksteps = K >> 5;
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i, row = id >> 3, c = id & 7;
    int am = m0 + row; if (am > M - 1) am = M - 1;
    ga[i] = A + (long)am * lda + c * 8; lo[i] = swz(row, c);
    const int st = id >> 6, sub = st >> 1, kk = st & 1;
    int nt = (n0 >> 4) + sub; if (nt > n_tiles - 1) nt = n_tiles - 1;
    gw[i] = Wp + ((long)nt * ksteps + kk) * 512 + (id & 63) * 8; lw[i] = st * 1024 + (id & 63) * 16;
  }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int nt = K / 64;
  bf16x8 ra[4], rw[4];
  for (int i = 0; i < 4; ++i) { ra[i] = *(const bf16x8*)ga[i]; rw[i] = *(const bf16x8*)gw[i]; }
  for (int i = 0; i < 4; ++i) { *(bf16x8*)(smem + lo[i]) = ra[i]; *(bf16x8*)(smem + 16384 + lw[i]) = rw[i]; }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 32768; char* nxt = smem + ((t + 1) & 1) * 32768;
    const bool more = t + 1 < nt;
    if (more) for (int i = 0; i < 4; ++i) { ra[i] = *(const bf16x8*)(ga[i] + (long)(t + 1) * 64); rw[i] = *(const bf16x8*)(gw[i] + (long)(t + 1) * 1024); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fw[4];
      for (int i = 0; i < 4; ++i) { fw[i] = *(const bf16x8*)(cur + 16384 + ((wn * 4 + i) * 2 + kk) * 1024 + lane * 16); fa[i] = *(const bf16x8*)(cur + swz(wm * 64 + i * 16 + fr, kk * 4 + fq)); }
      for (int in = 0; in < 4; ++in) for (int im = 0; im < 4; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[in], fa[im], acc[in][im], 0, 0, 0);
    }
    if (more) for (int i = 0; i < 4; ++i) { *(bf16x8*)(nxt + lo[i]) = ra[i]; *(bf16x8*)(nxt + 16384 + lw[i]) = rw[i]; }
    __syncthreads();
  }
  for (int in = 0; in < 4; ++in) { const int n = n0 + wn * 64 + in * 16 + fq * 4; if (n >= N) continue;
    for (int im = 0; im < 4; ++im) { const int m = m0 + wm * 64 + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}

// ---------------- V1: direct-to-LDS staging (global_load_lds dwordx4), 128x128x64, 2 LDS buffers, stage-before-compute
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int XCD>
__global__ __launch_bounds__(256) void v1(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {  // XCD-aware remap: blocks sharing an XCD (id % 8) get a contiguous chunk of the (n-major) tile space
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nb / 8, r = nb % 8, x = id % 8;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
    by = wg / gridDim.x; bx = wg % gridDim.x;
  }
  const int m0 = by * 128, n0 = bx * 128, fr = lane & 15, fq = lane >> 4;
  const int ksteps = K >> 5;
  // per wave: 4 A pieces (8 rows x 128 B each) + 4 W pieces (1 packed tile each)
  const bf16* ga[4]; const bf16* gw[4]; int la[4], lw[4];
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;            // 0..15
    const int row = piece * 8 + (lane >> 3), c = lane & 7;
    int am = m0 + row; if (am > M - 1) am = M - 1;
    ga[i] = A + (long)am * lda + ((c ^ (row & 7)) << 3);
    la[i] = piece * 1024;                      // wave-uniform LDS base of the piece
    const int sub = piece >> 1, kk = piece & 1;
    int nt = (n0 >> 4) + sub; if (nt > n_tiles - 1) nt = n_tiles - 1;
    gw[i] = Wp + ((long)nt * ksteps + kk) * 512 + lane * 8;
    lw[i] = 16384 + piece * 1024;
  }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int nt = K / 64;
  for (int i = 0; i < 4; ++i) { glds16(ga[i], smem + la[i]); glds16(gw[i], smem + lw[i]); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 32768; char* nxt = smem + ((t + 1) & 1) * 32768;
    if (t + 1 < nt) for (int i = 0; i < 4; ++i) { glds16(ga[i] + (long)(t + 1) * 64, nxt + la[i]); glds16(gw[i] + (long)(t + 1) * 1024, nxt + lw[i]); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fw[4];
      for (int i = 0; i < 4; ++i) { fw[i] = *(const bf16x8*)(cur + 16384 + ((wn * 4 + i) * 2 + kk) * 1024 + lane * 16); fa[i] = *(const bf16x8*)(cur + swz(wm * 64 + i * 16 + fr, kk * 4 + fq)); }
      __builtin_amdgcn_s_setprio(1);
      for (int in = 0; in < 4; ++in) for (int im = 0; im < 4; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[in], fa[im], acc[in][im], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int in = 0; in < 4; ++in) { const int n = n0 + wn * 64 + in * 16 + fq * 4; if (n >= N) continue;
    for (int im = 0; im < 4; ++im) { const int m = m0 + wm * 64 + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}


// ---------------- V2: glds, BM = 32*MT, NBUF LDS buffers, counted vmcnt + raw barrier (prefetch depth NBUF-1)
template <int MT, int NBUF>
__global__ __launch_bounds__(256) void v2(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM_ = 32 * MT, ABYTES = BM_ * 128, BUF = ABYTES + 16384, NA = BM_ / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM_, n0 = blockIdx.x * 128, fr = lane & 15, fq = lane >> 4;
  const int ksteps = K >> 5;
  const bf16* ga[NA]; const bf16* gw[4]; int la[NA], lw[4];
  for (int i = 0; i < NA; ++i) {
    const int piece = wave * NA + i, row = piece * 8 + (lane >> 3), c = lane & 7;
    int am = m0 + row; if (am > M - 1) am = M - 1;
    ga[i] = A + (long)am * lda + ((c ^ (row & 7)) << 3); la[i] = piece * 1024;
  }
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i, sub = piece >> 1, kk = piece & 1;
    int nt = (n0 >> 4) + sub; if (nt > n_tiles - 1) nt = n_tiles - 1;
    gw[i] = Wp + ((long)nt * ksteps + kk) * 512 + lane * 8; lw[i] = ABYTES + piece * 1024;
  }
  f32x4 acc[4][MT];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int nt = K / 64;
  auto stage = [&](int t) {
    char* b = smem + (t % NBUF) * BUF;
    for (int i = 0; i < NA; ++i) glds16(ga[i] + (long)t * 64, b + la[i]);
    for (int i = 0; i < 4; ++i) glds16(gw[i] + (long)t * 1024, b + lw[i]);
  };
  for (int t = 0; t < NBUF - 1 && t < nt; ++t) stage(t);
  // wait for tile 0 only: NBUF-2 tiles may stay in flight
  if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NA + 4) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t % NBUF) * BUF;
    if (t + NBUF - 1 < nt) stage(t + NBUF - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MT], fw[4];
      for (int i = 0; i < 4; ++i) fw[i] = *(const bf16x8*)(cur + ABYTES + ((wn * 4 + i) * 2 + kk) * 1024 + lane * 16);
      for (int i = 0; i < MT; ++i) fa[i] = *(const bf16x8*)(cur + swz(wm * 16 * MT + i * 16 + fr, kk * 4 + fq));
      __builtin_amdgcn_s_setprio(1);
      for (int in = 0; in < 4; ++in) for (int im = 0; im < MT; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[in], fa[im], acc[in][im], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    // tile t+1 must have landed; tile t+2 (just issued, NA+4 instructions) may stay in flight
    if (NBUF == 3 && t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NA + 4) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int in = 0; in < 4; ++in) { const int n = n0 + wn * 64 + in * 16 + fq * 4; if (n >= N) continue;
    for (int im = 0; im < MT; ++im) { const int m = m0 + wm * 16 * MT + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}


// ---------------- V3: 256x256x64 tile, 8 waves (2 m x 4 n), wave tile 128x64, glds, 2 LDS buffers (128 KiB), 1 WG/CU
__global__ __launch_bounds__(512) void v3(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ABYTES = 256 * 128, BUF = ABYTES + 32768;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 256, fr = lane & 15, fq = lane >> 4;
  const int ksteps = K >> 5;
  const bf16* ga[4]; const bf16* gw[4]; int la[4], lw[4];
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i, row = piece * 8 + (lane >> 3), c = lane & 7;   // 32 A pieces
    int am = m0 + row; if (am > M - 1) am = M - 1;
    ga[i] = A + (long)am * lda + ((c ^ (row & 7)) << 3); la[i] = piece * 1024;
    const int sub = piece >> 1, kk = piece & 1;                                        // 32 W pieces
    int nt = (n0 >> 4) + sub; if (nt > n_tiles - 1) nt = n_tiles - 1;
    gw[i] = Wp + ((long)nt * ksteps + kk) * 512 + lane * 8; lw[i] = ABYTES + piece * 1024;
  }
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const int nt = K / 64;
  for (int i = 0; i < 4; ++i) { glds16(ga[i], smem + la[i]); glds16(gw[i], smem + lw[i]); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF; char* nxt = smem + ((t + 1) & 1) * BUF;
    if (t + 1 < nt) for (int i = 0; i < 4; ++i) { glds16(ga[i] + (long)(t + 1) * 64, nxt + la[i]); glds16(gw[i] + (long)(t + 1) * 1024, nxt + lw[i]); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[8], fw[4];
      for (int i = 0; i < 4; ++i) fw[i] = *(const bf16x8*)(cur + ABYTES + ((wn * 4 + i) * 2 + kk) * 1024 + lane * 16);
      for (int i = 0; i < 8; ++i) fa[i] = *(const bf16x8*)(cur + swz(wm * 128 + i * 16 + fr, kk * 4 + fq));
      __builtin_amdgcn_s_setprio(1);
      for (int in = 0; in < 4; ++in) for (int im = 0; im < 8; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[in], fa[im], acc[in][im], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int in = 0; in < 4; ++in) { const int n = n0 + wn * 64 + in * 16 + fq * 4; if (n >= N) continue;
    for (int im = 0; im < 8; ++im) { const int m = m0 + wm * 128 + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}


// ---------------- V4: 256x256x64, 8 waves (2 M x 4 N), 4 phases per k-tile, counted vmcnt, staggered wave groups
// LDS: 2 buffers (k-tile parity) x [X-lo 16K | X-hi 16K | W-lo 16K | W-hi 16K].  Per k-tile T the four quadrants of a wave's
// 128 x 64 output are multiplied in phases p0..p3; the LDS reads are W (all 4 n-frags) + X(mq0) in p0 and X(mq1) in p2.
// Half-tile staging, one per phase: p0 -> X-lo(T+1), p1 -> X-hi(T+1), p2 -> W-lo(T+2), p3 -> W-hi(T+2); the wait for tile T+1
// is vmcnt(4) at p3 (two younger half-tiles stay in flight).  Waves 4-7 run one barrier slot behind waves 0-3, so on every
// SIMD one wave multiplies while the other reads LDS / issues loads.
template <int XCD>
__global__ __launch_bounds__(512) void v4(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nb / 8, r = nb % 8, x = id % 8;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
    // column-major walk over groups of 8 M-tiles keeps a W panel + 8 X panels hot in one XCD's L2
    by = wg % gridDim.y; bx = wg / gridDim.y;
  }
  const int m0 = by * 256, n0 = bx * 256, fr = lane & 15, fq = lane >> 4;
  const int ksteps = K >> 5, NT = K >> 6;
  // staging sources: 2 pieces per wave per half-tile
  const bf16* gx[2][2]; const bf16* gw[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i, row = piece * 8 + (lane >> 3), c = lane & 7;
      int am = m0 + h * 128 + row; if (am > M - 1) am = M - 1;
      gx[h][i] = A + (long)am * lda + ((c ^ (row & 7)) << 3);
      int nt = (n0 >> 4) + h * 8 + (piece >> 1); if (nt > n_tiles - 1) nt = n_tiles - 1;
      gw[h][i] = Wp + ((long)nt * ksteps + (piece & 1)) * 512 + lane * 8;
    }
  auto stage_x = [&](int h, int T) {
    if (T >= NT) return;
    char* dst = smem + (T & 1) * 65536 + h * 16384 + wave * 2048;
    glds16(gx[h][0] + (long)T * 64, dst); glds16(gx[h][1] + (long)T * 64, dst + 1024);
  };
  auto stage_w = [&](int h, int T) {
    if (T >= NT) return;
    char* dst = smem + (T & 1) * 65536 + 32768 + h * 16384 + wave * 2048;
    glds16(gw[h][0] + (long)T * 1024, dst); glds16(gw[h][1] + (long)T * 1024, dst + 1024);
  };
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  // prologue: half-tiles 0..5 (tile 0 complete + W of tile 1)
  stage_w(0, 0); stage_w(1, 0); stage_x(0, 0); stage_x(1, 0); stage_w(0, 1); stage_w(1, 1);
  if (NT > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // stagger: waves 4-7 run one slot behind
  bf16x8 Wf[4][2], Xs[4][2];
  const int xoff = wm * 16384, woff = 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192 + lane * 16;
  for (int T = 0; T < NT; ++T) {
    const char* cur = smem + (T & 1) * 65536;
    // ---- p0
#pragma unroll
    for (int in = 0; in < 4; ++in)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Wf[in][kk] = *(const bf16x8*)(cur + woff + (in * 2 + kk) * 1024);
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Xs[im][kk] = *(const bf16x8*)(cur + xoff + swz(im * 16 + fr, kk * 4 + fq));
    stage_x(0, T + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][im], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- p1
    stage_x(1, T + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 2; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][im], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- p2
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Xs[im][kk] = *(const bf16x8*)(cur + xoff + swz(64 + im * 16 + fr, kk * 4 + fq));
    stage_w(0, T + 2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 2; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) acc[in][4 + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][4 + im], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- p3
    stage_w(1, T + 2);
    if (T + 2 < NT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) acc[in][4 + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][4 + im], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();       // match the stagger barrier of waves 4-7
  for (int in = 0; in < 4; ++in) { const int n = n0 + wn * 64 + in * 16 + fq * 4; if (n >= N) continue;
    for (int im = 0; im < 8; ++im) { const int m = m0 + wm * 128 + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}

template <int XCD, int ABL>
__global__ __launch_bounds__(512) void v5(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nb / 8, r = nb % 8, x = id % 8;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
    // column-major walk over groups of 8 M-tiles keeps a W panel + 8 X panels hot in one XCD's L2
    by = wg % gridDim.y; bx = wg / gridDim.y;
  }
  const int m0 = by * 256, n0 = bx * 256, fr = lane & 15, fq = lane >> 4;
  const int ksteps = K >> 5, NT = K >> 6;
  // staging sources: 2 pieces per wave per half-tile
  const bf16* gx[2][2]; const bf16* gw[2][2];
  unsigned ox[2][2], ow[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i, row = piece * 8 + (lane >> 3), c = lane & 7;
      int am = m0 + h * 128 + row; if (am > M - 1) am = M - 1;
      gx[h][i] = A + (long)am * lda + ((c ^ (row & 7)) << 3);
      int nt = (n0 >> 4) + h * 8 + (piece >> 1); if (nt > n_tiles - 1) nt = n_tiles - 1;
      gw[h][i] = Wp + ((long)nt * ksteps + (piece & 1)) * 512 + lane * 8;
      ox[h][i] = (unsigned)(((long)am * lda + ((c ^ (row & 7)) << 3)) * 2);
      ow[h][i] = (unsigned)((((long)nt * ksteps + (piece & 1)) * 512 + lane * 8) * 2);
    }
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((long)M * lda * 2 > 0x7fffffffL ? 0x7fffffff : (long)M * lda * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)((long)n_tiles * 16 * K * 2 > 0x7fffffffL ? 0x7fffffff : (long)n_tiles * 16 * K * 2), 0x00020000);
  auto stage_x = [&](int h, int T) {
    if (T >= NT || (ABL & 4) || (ABL & 8)) return;
    char* dst = smem + (T & 1) * 65536 + h * 16384 + wave * 2048;
    if (ABL & 32) {   // timing only: the same bytes as pieces of 4 rows x 256 B (two k-tiles of 64 rows) instead of 8 rows x 128 B
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i, row = (T & 1) * 64 + piece * 4 + (lane >> 4);
        int am = m0 + h * 128 + row; if (am > M - 1) am = M - 1;
        glds16(A + (long)am * lda + (long)(T & ~1) * 64 + (lane & 15) * 8, dst + i * 1024);
      }
      return;
    }
    if (ABL & 64) {   // buffer_load ... lds: 32-bit per-lane offset + scalar k-tile offset
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)dst, 16, ox[h][0], T * 128, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + 1024), 16, ox[h][1], T * 128, 0, 0);
      return;
    }
    glds16(gx[h][0] + (long)T * 64, dst); glds16(gx[h][1] + (long)T * 64, dst + 1024);
  };
  auto stage_w = [&](int h, int T) {
    if (T >= NT || (ABL & 4) || (ABL & 16)) return;
    char* dst = smem + (T & 1) * 65536 + 32768 + h * 16384 + wave * 2048;
    if (ABL & 64) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)dst, 16, ow[h][0], T * 2048, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(dst + 1024), 16, ow[h][1], T * 2048, 0, 0);
      return;
    }
    glds16(gw[h][0] + (long)T * 1024, dst); glds16(gw[h][1] + (long)T * 1024, dst + 1024);
  };
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  // prologue: half-tiles 0..5 (tile 0 complete + W of tile 1)
  stage_w(0, 0); stage_w(1, 0); stage_x(0, 0); stage_x(1, 0); stage_w(0, 1); stage_w(1, 1);
  if (NT > 1) { if ((ABL & 8) || (ABL & 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // stagger: waves 4-7 run one slot behind
  bf16x8 Wf[4][2], Xs[4][2];
  if (ABL & 2) { for (int i = 0; i < 4; ++i) for (int k = 0; k < 2; ++k) { Wf[i][k] = bf16x8{1,1,1,1,1,1,1,1}; Xs[i][k] = bf16x8{1,1,1,1,1,1,1,1}; } }
  const int xoff = wm * 16384, woff = 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192 + lane * 16;
  for (int T = 0; T < NT; ++T) {
    const char* cur = smem + (T & 1) * 65536;
    // ---- p0
#pragma unroll
    for (int in = 0; in < 4; ++in)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { if (!(ABL & 2)) Wf[in][kk] = *(const bf16x8*)(cur + woff + (in * 2 + kk) * 1024); }
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { if (!(ABL & 2)) Xs[im][kk] = *(const bf16x8*)(cur + xoff + swz(im * 16 + fr, kk * 4 + fq)); }
    stage_x(0, T + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) { if (ABL & 1) asm volatile("" :: "v"(Wf[in][kk]), "v"(Xs[im][kk])); else acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][im], 0, 0, 0); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- p1
    stage_x(1, T + 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 2; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) { if (ABL & 1) asm volatile("" :: "v"(Wf[in][kk]), "v"(Xs[im][kk])); else acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][im], 0, 0, 0); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- p2
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { if (!(ABL & 2)) Xs[im][kk] = *(const bf16x8*)(cur + xoff + swz(64 + im * 16 + fr, kk * 4 + fq)); }
    stage_w(0, T + 2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 2; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) { if (ABL & 1) asm volatile("" :: "v"(Wf[in][kk]), "v"(Xs[im][kk])); else acc[in][4 + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][4 + im], 0, 0, 0); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- p3
    stage_w(1, T + 2);
    if (T + 2 < NT) { if ((ABL & 8) || (ABL & 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im) { if (ABL & 1) asm volatile("" :: "v"(Wf[in][kk]), "v"(Xs[im][kk])); else acc[in][4 + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][4 + im], 0, 0, 0); }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();       // match the stagger barrier of waves 4-7
  for (int in = 0; in < 4; ++in) { const int n = n0 + wn * 64 + in * 16 + fq * 4; if (n >= N) continue;
    for (int im = 0; im < 8; ++im) { const int m = m0 + wm * 128 + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}

// ---------------- V7 (round 4): W fragments straight from the tile-packed weights into REGISTERS (the layout already is the MFMA A
// operand), 3 k-tiles ahead in a 4-set register ring; X through a 4-slot LDS ring filled 3 k-tiles ahead by direct-to-LDS loads.
// 256 x 256 x 64 tile, 8 waves, wave w owns ALL 256 rows x columns 32 w .. 32 w + 31 (its two weight tiles are nobody else's: no W
// byte is loaded twice, LDS carries only the activation half).  Why: the shipped kernel's staging is latency-bound, not byte-bound
// (half the bytes take 70-86 % of the time, r02_training_gemms.md): LDS holds two 64-KiB stages, ~1.5 k-tiles in flight; here the
// bytes in flight are 3 x 32 KiB of X in LDS + 3 x 32 KiB of W in registers per CU, and there is ONE barrier per k-tile.
// Every global access and wait is inline asm (a global_load_lds in the stream makes hipcc turn later waits into vmcnt(0)); the W
// waits carry the register set as "+v" operands so that no MFMA is scheduled above them.
typedef __attribute__((ext_vector_type(4))) float wreg;
__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}
template <int XCD>
__global__ __launch_bounds__(512) void v7(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp, int n_tiles, bf16* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];          // X ring: 4 slots x [256 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx = blockIdx.x, by = blockIdx.y;
  if (XCD) {
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nb / 8, r = nb % 8, x = id % 8;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
    const int per_group = 8 * gridDim.x, grp = wg / per_group, in_grp = wg % per_group;
    const int rows = ((int)gridDim.y - grp * 8) < 8 ? ((int)gridDim.y - grp * 8) : 8;
    by = grp * 8 + in_grp % rows; bx = in_grp / rows;
  }
  const int m0 = by * 256, n0 = bx * 256, fr = lane & 15, fq = lane >> 4;
  const int ksteps = K >> 5, NT = K >> 6;
  // X staging: 32 pieces of 8 rows x 128 B per k-tile, 4 per wave
  const bf16* gx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i, row = piece * 8 + (lane >> 3), c = lane & 7;
    int am = m0 + row; if (am > M - 1) am = M - 1;
    gx[i] = A + (long)am * lda + ((c ^ (row & 7)) << 3);
  }
  const unsigned ring = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  const unsigned xdst = ring + wave * 4096;                           // + slot * 32768 + i * 1024
  // W: the wave's two 16-column weight tiles
  int nt0 = (n0 >> 4) + wave * 2, nt1 = nt0 + 1;
  if (nt0 > n_tiles - 1) nt0 = n_tiles - 1;
  if (nt1 > n_tiles - 1) nt1 = n_tiles - 1;
  const bf16* wp0 = Wp + (long)nt0 * ksteps * 512 + lane * 8;
  const bf16* wp1 = Wp + (long)nt1 * ksteps * 512 + lane * 8;
  wreg w[4][4];                                                       // [set][tile 0 k-step 0, tile 0 k-step 1, tile 1 k-step 0, tile 1 k-step 1]
#define V7_ISSUE(S_, T_)                                                                                              \
  do {                                                                                                                \
    const int tt = (T_) < NT ? (T_) : NT - 1;                         /* phantom issues past the end re-read the last k-tile */ \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) dma16(gx[i] + (long)tt * 64, xdst + (S_) * 32768 + i * 1024);        \
    const bf16* p0 = wp0 + (long)tt * 1024; const bf16* p1 = wp1 + (long)tt * 1024;                                   \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[S_][0]) : "v"(p0) : "memory");                            \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[S_][1]) : "v"(p0) : "memory");                \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[S_][2]) : "v"(p1) : "memory");                            \
    asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(w[S_][3]) : "v"(p1) : "memory");                \
  } while (0)
  f32x4 acc[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  V7_ISSUE(0, 0); V7_ISSUE(1, 1); V7_ISSUE(2, 2);
  const int xrd = fr * 128;                                           // + rb * 2048, chunk swizzled with row & 7 = fr & 7
  const int c0 = ((0 + fq) ^ (fr & 7)) << 4, c1 = ((4 + fq) ^ (fr & 7)) << 4;
#define V7_TILE(S_, T_)                                                                                               \
  do {                                                                                                                \
    /* k-tile T: its 8 memory operations (issued three tiles ago) have landed; the 16 younger ones stay in flight */   \
    asm volatile("s_waitcnt vmcnt(16)" : "+v"(w[S_][0]), "+v"(w[S_][1]), "+v"(w[S_][2]), "+v"(w[S_][3]) :: "memory");  \
    asm volatile("s_barrier" ::: "memory");                                                                           \
    V7_ISSUE(((S_) + 3) & 3, (T_) + 3);                               /* slot / set (T - 1) & 3: every wave is past tile T - 1 */ \
    const char* xs = smem + (S_) * 32768 + xrd;                                                                       \
    _Pragma("unroll") for (int ph = 0; ph < 4; ++ph) {                                                                \
      bf16x8 X0[4], X1[4];                                                                                            \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                 \
        X0[r] = *(const bf16x8*)(xs + (ph * 4 + r) * 2048 + c0);                                                      \
        X1[r] = *(const bf16x8*)(xs + (ph * 4 + r) * 2048 + c1);                                                      \
      }                                                                                                               \
      __builtin_amdgcn_s_setprio(1);                                                                                  \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                 \
        acc[0][ph * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[S_][0]), X0[r], acc[0][ph * 4 + r], 0, 0, 0); \
        acc[1][ph * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[S_][2]), X0[r], acc[1][ph * 4 + r], 0, 0, 0); \
      }                                                                                                               \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                 \
        acc[0][ph * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[S_][1]), X1[r], acc[0][ph * 4 + r], 0, 0, 0); \
        acc[1][ph * 4 + r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[S_][3]), X1[r], acc[1][ph * 4 + r], 0, 0, 0); \
      }                                                                                                               \
      __builtin_amdgcn_s_setprio(0);                                                                                  \
    }                                                                                                                 \
  } while (0)
  for (int T = 0; T < NT; T += 4) {                                   // NT % 4 == 0 (host)
    V7_TILE(0, T); V7_TILE(1, T + 1); V7_TILE(2, T + 2); V7_TILE(3, T + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the phantom issues of the last three tiles
#undef V7_TILE
#undef V7_ISSUE
#pragma unroll
  for (int in = 0; in < 2; ++in) {
    const int n = n0 + wave * 32 + in * 16 + fq * 4; if (n >= N) continue;
#pragma unroll
    for (int im = 0; im < 16; ++im) { const int m = m0 + im * 16 + fr; if (m >= M) continue;
      bf16x4 o = {(bf16)acc[in][im][0], (bf16)acc[in][im][1], (bf16)acc[in][im][2], (bf16)acc[in][im][3]};
      *(bf16x4*)(C + (long)m * ldc + n) = o; } }
}

int main() {
  struct Shape { int M, N, K; const char* name; } shapes[] = {
    {9280, 4608, 3584, "lm qkv"}, {9280, 3584, 18944, "lm down"}, {9280, 37888, 3584, "lm gate/up"}, {9280, 3584, 3584, "lm o"},
    {4096, 3840, 1280, "vit qkv"}, {4096, 6848, 1280, "vit gate/up"}, {4096, 1280, 3456, "vit down"}, {4096, 1280, 1280, "vit proj"},
    {1088, 4608, 3584, "prefill qkv"}, {1088, 3584, 3584, "prefill o"}, {1088, 37888, 3584, "prefill gate/up"}, {1088, 3584, 18944, "prefill down"}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& s : shapes) {
    if (getenv("VR2_BENCH_SHAPE") && !strstr(s.name, getenv("VR2_BENCH_SHAPE"))) continue;
    bf16 *A, *W, *C;
    const int nt16 = (s.N + 15) / 16;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&W, (size_t)nt16 * 16 * s.K * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
    // random-ish bf16 bit patterns in [-2,2): exponent 0x3f/0xbf..; fill via small kernel-free trick: memset different bytes
    {
      size_t na = (size_t)s.M * s.K, nw = (size_t)nt16 * 16 * s.K;
      unsigned short* h = (unsigned short*)malloc((na > nw ? na : nw) * 2);
      unsigned x = 12345;
      const bool zero = getenv("VR2_BENCH_ZERO") != nullptr;    // all-zero operands: same instruction stream, no data-dependent power
      for (size_t i = 0; i < na; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0 : (unsigned short)(((x >> 16) & 0x807f) | 0x3f00 | ((x >> 9) & 0x80)); }
      CK(hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice));
      for (size_t i = 0; i < nw; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0 : (unsigned short)(((x >> 16) & 0x807f) | 0x3c00 | ((x >> 9) & 0x80)); }
      CK(hipMemcpy(W, h, nw * 2, hipMemcpyHostToDevice));
      free(h);
    }
    dim3 grid((s.N + 127) / 128, (s.M + 127) / 128);
    auto run = [&](const char* name, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      const int reps = 10;
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("%-16s M=%5d N=%5d K=%5d  %-22s %8.1f us  %7.1f TF/s\n", s.name, s.M, s.N, s.K, name, us, 2.0 * s.M * s.N * s.K / us / 1e6);
    };
    dim3 grid64((s.N + 127) / 128, (s.M + 63) / 64);
    static bool once = false;
    if (!once) { once = true;
      CK(hipFuncSetAttribute((const void*)v2<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768));
      CK(hipFuncSetAttribute((const void*)v2<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576)); }
    run("v1 glds 128 2buf", [&] { hipLaunchKernelGGL(v1<0>, grid, dim3(256), 65536, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v2 128 2buf", [&] { hipLaunchKernelGGL((v2<4, 2>), grid, dim3(256), 2 * 32768, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o3 = false; if (!o3) { o3 = true; CK(hipFuncSetAttribute((const void*)v3, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536)); } }
    run("v3 256x256 2buf", [&] { hipLaunchKernelGGL(v3, dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 2 * 65536, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o4 = false; if (!o4) { o4 = true; CK(hipFuncSetAttribute((const void*)v4<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); CK(hipFuncSetAttribute((const void*)v4<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    {  // correctness of v4 against v1
      bf16* C2; CK(hipMalloc(&C2, (size_t)s.M * s.N * 2));
      hipLaunchKernelGGL(v1<0>, grid, dim3(256), 65536, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K);
      hipLaunchKernelGGL(v4<0>, dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C2, (long)s.N, s.M, s.N, s.K);
      CK(hipDeviceSynchronize());
      size_t n = (size_t)s.M * s.N; unsigned short* h1 = (unsigned short*)malloc(n * 2); unsigned short* h2 = (unsigned short*)malloc(n * 2);
      CK(hipMemcpy(h1, C, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, C2, n * 2, hipMemcpyDeviceToHost));
      double maxd = 0, maxv = 0; size_t bad = 0;
      for (size_t i = 0; i < n; ++i) { unsigned a = (unsigned)h1[i] << 16, b = (unsigned)h2[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        double d = fabs((double)fa - fb); if (d > maxd) maxd = d; if (fabs(fa) > maxv) maxv = fabs(fa); if (d > 0.02 * fabs(fa) + 0.05) ++bad; }
      printf("   v4 check: max|d| %.4f (max|v| %.1f) mismatches %zu\n", maxd, maxv, bad);
      free(h1); free(h2); CK(hipFree(C2));
    }
    { static bool o7 = false; if (!o7) { o7 = true; CK(hipFuncSetAttribute((const void*)v7<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); CK(hipFuncSetAttribute((const void*)v5<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    if ((s.K / 64) % 4 == 0) {  // correctness of v7 against v1
      bf16* C2; CK(hipMalloc(&C2, (size_t)s.M * s.N * 2));
      hipLaunchKernelGGL(v1<0>, grid, dim3(256), 65536, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K);
      hipLaunchKernelGGL(v7<1>, dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C2, (long)s.N, s.M, s.N, s.K);
      CK(hipDeviceSynchronize());
      size_t n = (size_t)s.M * s.N; unsigned short* h1 = (unsigned short*)malloc(n * 2); unsigned short* h2 = (unsigned short*)malloc(n * 2);
      CK(hipMemcpy(h1, C, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, C2, n * 2, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < n; ++i) if (h1[i] != h2[i]) ++bad;
      printf("   v7 check: %zu of %zu outputs differ from v1 (bit compare)\n", bad, n);
      free(h1); free(h2); CK(hipFree(C2));
      for (int rep = 0; rep < 2; ++rep) {
        run("v5 shipped structure", [&] { hipLaunchKernelGGL((v5<1, 0>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
        run("v7 W->regs, X ring", [&] { hipLaunchKernelGGL((v7<1>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
      }
    }
    if (getenv("VR2_BENCH_V7")) { CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); continue; }
    run("v4 256 8phase", [&] { hipLaunchKernelGGL(v4<0>, dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v4 256 8phase xcd", [&] { hipLaunchKernelGGL(v4<1>, dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 xcd no mfma", [&] { hipLaunchKernelGGL((v5<1, 1>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 xcd no ds_read", [&] { hipLaunchKernelGGL((v5<1, 2>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 xcd no stage", [&] { hipLaunchKernelGGL((v5<1, 4>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 xcd no mfma+read", [&] { hipLaunchKernelGGL((v5<1, 3>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 xcd no read+stage", [&] { hipLaunchKernelGGL((v5<1, 6>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 xcd no mfma+stage", [&] { hipLaunchKernelGGL((v5<1, 5>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); CK(hipFuncSetAttribute((const void*)v5<1, 19>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 stage W only", [&] { hipLaunchKernelGGL((v5<1, 11>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v5 stage X only", [&] { hipLaunchKernelGGL((v5<1, 19>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 51>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); CK(hipFuncSetAttribute((const void*)v5<1, 35>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 stage X only, 4x256B", [&] { hipLaunchKernelGGL((v5<1, 51>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v5 stage both, X 4x256B", [&] { hipLaunchKernelGGL((v5<1, 35>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    { static bool o = false; if (!o) { o = true; CK(hipFuncSetAttribute((const void*)v5<1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); CK(hipFuncSetAttribute((const void*)v5<1, 67>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); } }
    run("v5 full, buffer_load lds", [&] { hipLaunchKernelGGL((v5<1, 64>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v5 stage only, buffer_load lds", [&] { hipLaunchKernelGGL((v5<1, 67>), dim3((s.N + 255) / 256, (s.M + 255) / 256), dim3(512), 131072, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    if (getenv("VR2_BENCH_SHORT")) { CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); continue; }
    run("v2 128 3buf", [&] { hipLaunchKernelGGL((v2<4, 3>), grid, dim3(256), 3 * 32768, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v2 64 2buf", [&] { hipLaunchKernelGGL((v2<2, 2>), grid64, dim3(256), 2 * 24576, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    run("v2 64 3buf", [&] { hipLaunchKernelGGL((v2<2, 3>), grid64, dim3(256), 3 * 24576, 0, A, (long)s.K, W, nt16, C, (long)s.N, s.M, s.N, s.K); });
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C));
  }
  return 0;
}
