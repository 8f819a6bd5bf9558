// Weight-gradient GEMM ("TN"): C[m][n] (+)= sum_k A[k][m] * B[k][n],  A: [K][lda] bf16, B: [K][ldb] bf16 (k = token rows),
// C fp32 row-major.  dW[out][in] += dY[t][out]^T X[t][in] for every trainable nn.Linear.
//
// Both operands are k-major in memory (a token row holds all its features), the opposite of what an MFMA fragment wants
// (8 consecutive k per lane).  The tiles are staged HBM -> LDS as they are (coalesced 16-byte rows) and the fragments
// are read with ds_read_b64_tr_b16 (hardware transpose: a 16-lane group reads a 4(k) x 16(feature) block and each lane
// receives one feature's 4 k values), two reads per v_mfma_f32_16x16x32_bf16 operand.  The k order inside a 32-step is
// permuted (lane quarter q, element j: k = 4q + j for j < 4, 16 + 4q + (j-4) otherwise) identically for both operands,
// which makes every transposed read conflict-free with a 288-byte row stride.  The B tile is the MFMA A operand so a
// lane owns 4 consecutive n of one m: the fp32 (+)= epilogue is one 16-byte access.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int ROWB = 288;                     // LDS bytes per k row: 128 features x 2 B + 32 B pad
constexpr int TILEB = TK * ROWB;              // 18432

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int kk, int col0, int lane) {
  // 16-lane group q reads rows 4q + a (first read) and 16 + 4q + a (second), cols col0 + 4p .. +3  (a = (lane&15)>>2, p = lane&3)
  typedef __attribute__((ext_vector_type(4))) __bf16 vb4;
  const int q = lane >> 4, a = (lane & 15) >> 2, p = lane & 3;
  const char* p0 = tile + (kk * 32 + 4 * q + a) * ROWB + (col0 + 4 * p) * 2;
  const vb4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(p0));
  const vb4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(p0 + 16 * ROWB));
  bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return r;
}

template <bool ACC>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                      float* __restrict__ C, long ldc, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  // staging: 64 rows x 16 chunks per tile, 4 chunks of A and 4 of B per thread
  int srow[4], scol[4];
  bool va[4], vb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    srow[i] = id >> 4;
    scol[i] = (id & 15) * 8;
    va[i] = (m0 + scol[i]) < M;   // M, N multiples of 8 (host)
    vb[i] = (n0 + scol[i]) < N;
  }
  const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = (K + TK - 1) / TK;
  bf16x8 ra[4], rb[4];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long k = (long)t * TK + srow[i];
      const bool kv = k < K;
      ra[i] = (kv && va[i]) ? *reinterpret_cast<const bf16x8*>(A + k * lda + m0 + scol[i]) : z8;
      rb[i] = (kv && vb[i]) ? *reinterpret_cast<const bf16x8*>(B + k * ldb + n0 + scol[i]) : z8;
    }
  };
  auto lstore = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<bf16x8*>(buf + srow[i] * ROWB + scol[i] * 2) = ra[i];
      *reinterpret_cast<bf16x8*>(buf + TILEB + srow[i] * ROWB + scol[i] * 2) = rb[i];
    }
  };
  gload(0);
  lstore(smem);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 2 * TILEB;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILEB;
    const bool more = (t + 1) < nt;
    if (more) gload(t + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fm[4], fn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fn[i] = tr_frag(cur + TILEB, kk, wn * 64 + i * 16, lane);
        fm[i] = tr_frag(cur, kk, wm * 64 + i * 16, lane);
      }
#pragma unroll
      for (int in = 0; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < 4; ++im)
          acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[in], fm[im], acc[in][im], 0, 0, 0);
    }
    if (more) lstore(nxt);
    __syncthreads();
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int in = 0; in < 4; ++in) {
    const int n = n0 + wn * 64 + in * 16 + fq * 4;
    if (n >= N) continue;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
      const int m = m0 + wm * 64 + im * 16 + fr;
      if (m >= M) continue;
      float* cp = C + (long)m * ldc + n;
      float4 v = make_float4(acc[in][im][0], acc[in][im][1], acc[in][im][2], acc[in][im][3]);
      if (ACC) {
        const float4 o = *reinterpret_cast<const float4*>(cp);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(cp) = v;
    }
  }
}

// ---- 256 x 256 x 64 tile, 8 waves, the pipeline of gemm.hip's gemm256_kernel with transposing fragment reads ------------------
// Both operands are k-major, so a half-tile (64 token rows x 128 features) is staged by direct-to-LDS loads as 16 pieces of
// 4 rows x 256 B (one wave instruction each, lane-linear) and the MFMA fragments are read with ds_read_b64_tr_b16.  A 16-lane
// group of such a read touches the 4 rows of ONE piece (same banks at the natural 256-B row stride) and the two groups of a
// half-wave touch two pieces 1 KiB apart, so the 16-column fragment slot is XOR-swizzled with (row & 3) | (piece & 1) << 2 --
// applied to the per-lane SOURCE address of the load and again to the read address: every half-wave read covers all 64 banks once.
// Phases / staging / counted vmcnt / wave stagger exactly as in gemm256_kernel (dY halves take the X role, x halves the W role).
// K must be a multiple of 64 here (the host sends a k tail to the 128-tile kernel above).
// A launch covers the tiles [tile0, tile0 + gridDim.x / kparts) of the grouped walk, each cut into kparts ranges of k-tiles
// (kparts > 1 only with MODE 2).  MODE 0: C = ..., 1: C += ... (read-modify-write), 2: C += ... by fp32 atomics -- the last,
// partly filled round of a tiling (1036 tiles on 256 CUs = 4.05 rounds) runs as tiles x k-parts that fill the chip instead.
template <int MODE>
__global__ __launch_bounds__(512) void gemm_tn256_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ B, long ldb,
                                                         float* __restrict__ C, long ldc, int M, int N, int K, int tiles_m,
                                                         int tiles_n, int tile0, int kparts) {
  constexpr bool ACC = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) __bf16 vb4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  int bm, bn;
  int kpart = 0;
  {
    int nb = gridDim.x, id = blockIdx.x;
    if (MODE == 2) { nb /= kparts; kpart = id % kparts; id /= kparts; }
    const int q = nb / 8, r = nb % 8, x = id % 8;
    const int wg = tile0 + (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;   // bijective XCD remap
    const int per_group = 8 * tiles_n, grp = wg / per_group, in_grp = wg % per_group;
    const int rows = (tiles_m - grp * 8) < 8 ? (tiles_m - grp * 8) : 8;
    bm = grp * 8 + in_grp % rows;
    bn = in_grp / rows;
  }
  const int m0 = bm * 256, n0 = bn * 256;
  int NT = K >> 6;
  if (MODE == 2) {   // this workgroup's k-tiles
    const int t0 = (int)((long)kpart * NT / kparts), t1 = (int)((long)(kpart + 1) * NT / kparts);
    A += (long)t0 * 64 * lda;
    B += (long)t0 * 64 * ldb;
    NT = t1 - t0;
  }
  const bf16* ga[2][2];
  const bf16* gb[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i, ar = lane >> 4, j = (lane & 15) ^ ((ar << 1) | ((piece & 1) << 3));
      int ca = m0 + h * 128 + j * 8; if (ca > M - 8) ca = M - 8;                // columns past the edge: clamp, never stored
      int cb = n0 + h * 128 + j * 8; if (cb > N - 8) cb = N - 8;
      ga[h][i] = A + (long)(4 * piece + ar) * lda + ca;
      gb[h][i] = B + (long)(4 * piece + ar) * ldb + cb;
    }
  auto glds = [](const bf16* g, char* l) { vr2_dma16(g, l); };   // asm form: no compiler vmcnt(0) in front of the transposed reads
  auto stage_a = [&](int h, int T) {
    if (T >= NT) return;
    char* dst = smem + (T & 1) * 65536 + h * 16384 + wave * 2048;
    glds(ga[h][0] + (long)T * 64 * lda, dst);
    glds(ga[h][1] + (long)T * 64 * lda, dst + 1024);
  };
  auto stage_b = [&](int h, int T) {
    if (T >= NT) return;
    char* dst = smem + (T & 1) * 65536 + 32768 + h * 16384 + wave * 2048;
    glds(gb[h][0] + (long)T * 64 * ldb, dst);
    glds(gb[h][1] + (long)T * 64 * ldb, dst + 1024);
  };
  f32x4 acc[4][8];
  // (+)=: the accumulators START from the old dW values -- 32 reads in flight beside the prologue's direct-to-LDS loads, one
  // exposed latency per tile -- so the epilogue only stores (a read-add-store epilogue paid four read latencies per tile)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ACC) {
        const int n = n0 + wn * 64 + i * 16 + fq * 4, m = m0 + wm * 128 + j * 16 + fr;
        if (n < N && m < M) acc[i][j] = *reinterpret_cast<const f32x4*>(C + (long)m * ldc + n);
      }
    }
  stage_b(0, 0); stage_b(1, 0); stage_a(0, 0); stage_a(1, 0); stage_b(0, 1); stage_b(1, 1);
  if (ACC) {   // wait for them here: inside the loop the compiler's (uncounted-for) vmcnt would drain the staging loads each k-tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
  }
  if (NT > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();        // stagger: waves 4-7 run one slot behind
  // fragment read: lane (q = lane>>4 -> piece, a = row in piece, p -> 4 columns); slot fi of the half is swizzled with sw
  const int q4 = lane >> 4, ar = (lane & 15) >> 2, pc = lane & 3, sw = ar | ((q4 & 1) << 2);
  const int lane_off = q4 * 1024 + ar * 256 + pc * 8;
  auto frag = [&](const char* half, int kk, int fi) -> bf16x8 {
    const char* p0 = half + kk * 8192 + lane_off + ((fi ^ sw) << 5);
    const vb4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(p0));
    const vb4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(p0 + 4096));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
  };
  bf16x8 Wf[4][2], Xs[4][2];
  const int aoff = wm * 16384, boff = 32768 + (wn >> 1) * 16384;
#define VR2_MMA(N0, M0)                                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                                              \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                            \
    _Pragma("unroll") for (int in = N0; in < N0 + 2; ++in)                                                                    \
      _Pragma("unroll") for (int im = 0; im < 4; ++im)                                                                        \
        acc[in][M0 + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][M0 + im], 0, 0, 0);       \
  __builtin_amdgcn_s_setprio(0);                                                                                              \
  __builtin_amdgcn_s_barrier()
  for (int T = 0; T < NT; ++T) {
    const char* cur = smem + (T & 1) * 65536;
#pragma unroll
    for (int in = 0; in < 4; ++in)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Wf[in][kk] = frag(cur + boff, kk, (wn & 1) * 4 + in);
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Xs[im][kk] = frag(cur + aoff, kk, im);
    stage_a(0, T + 1);
    VR2_MMA(0, 0);
    stage_a(1, T + 1);
    VR2_MMA(2, 0);
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Xs[im][kk] = frag(cur + aoff, kk, 4 + im);
    stage_b(0, T + 2);
    VR2_MMA(2, 4);
    stage_b(1, T + 2);
    if (T + 2 < NT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VR2_MMA(0, 4);
  }
#undef VR2_MMA
  if (wm == 0) __builtin_amdgcn_s_barrier();
  if (MODE == 2) {
    // atomic (+)=: the memory-side float atomics run at full rate only on whole 256-byte row segments per wave instruction, so
    // the wave's 128 x 64 block goes through its own LDS region 32 rows at a time (272-byte row stride: conflict-free writes)
    float* stg = reinterpret_cast<float*>(smem + wave * 8704);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int in = 0; in < 4; ++in)
          *reinterpret_cast<f32x4*>(stg + (j * 16 + fr) * 68 + in * 16 + fq * 4) = acc[in][p * 2 + j];
      const int n = n0 + wn * 64 + lane;
      for (int rr = 0; rr < 32; ++rr) {
        const int m = m0 + wm * 128 + p * 32 + rr;
        if (m < M && n < N) unsafeAtomicAdd(C + (long)m * ldc + n, stg[rr * 68 + lane]);
      }
    }
    return;
  }
  // epilogue: stores only (MODE 1 started from the old values)
#pragma unroll
  for (int in = 0; in < 4; ++in) {
    const int n = n0 + wn * 64 + in * 16 + fq * 4;
    if (n >= N) continue;
#pragma unroll
    for (int im = 0; im < 8; ++im) {
      const int m = m0 + wm * 128 + im * 16 + fr;
      if (m < M) *reinterpret_cast<f32x4*>(C + (long)m * ldc + n) = acc[in][im];
    }
  }
}

}  // namespace

extern "C" int vr2_gemm_tn_bf16(const void* A_bf16, int64_t lda, const void* B_bf16, int64_t ldb, void* C_f32, int64_t ldc,
                                int64_t M, int64_t N, int64_t K, int accumulate, void* stream) {
  VR2_REQUIRE(M >= 8 && N >= 8 && K >= 1, "vr2_gemm_tn_bf16: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  VR2_REQUIRE(M % 8 == 0 && N % 8 == 0, "vr2_gemm_tn_bf16: M=%ld and N=%ld must be multiples of 8", (long)M, (long)N);
  VR2_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= M && ldb >= N && ldc >= N, "vr2_gemm_tn_bf16: bad strides");
  hipStream_t s = (hipStream_t)stream;
  // 256-row tiles (one workgroup per CU) when their last round over the 256 CUs is mostly full; the k tail (< 64 token rows)
  // then goes to the 128-tile kernel as an accumulating second launch
  const long tm = (M + 255) / 256, tn = (N + 255) / 256, t256 = tm * tn, rounds = (t256 + 255) / 256;
  static const bool no256 = getenv("VR2_GEMM_NO256") != nullptr;
  if (!no256 && K >= 256 && M >= 128 && N >= 128 && t256 * 4 >= 3 * rounds * 256) {
    static bool attr256 = false;
    if (!attr256) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      attr256 = true;
    }
    const int64_t Kmain = K / 64 * 64;
    // (+)= with more than one round whose last round is under 60 % full: that round's tiles run as tiles x k-parts with atomic
    // adds (>= 8 k-tiles per part), the whole rounds before it as they are
    long tail_tiles = t256 % 256, parts = 0;
    if (accumulate && t256 > 256 && tail_tiles > 0 && tail_tiles * 10 < 256 * 6) {
      parts = 256 / tail_tiles;
      if (parts > Kmain / 64 / 8) parts = Kmain / 64 / 8;
    }
    const long main_tiles = parts >= 2 ? t256 - tail_tiles : t256;
    if (accumulate)
      hipLaunchKernelGGL(gemm_tn256_kernel<1>, dim3((unsigned)main_tiles), dim3(512), 131072, s, (const bf16*)A_bf16, (long)lda,
                         (const bf16*)B_bf16, (long)ldb, (float*)C_f32, (long)ldc, (int)M, (int)N, (int)Kmain, (int)tm, (int)tn, 0, 1);
    else
      hipLaunchKernelGGL(gemm_tn256_kernel<0>, dim3((unsigned)main_tiles), dim3(512), 131072, s, (const bf16*)A_bf16, (long)lda,
                         (const bf16*)B_bf16, (long)ldb, (float*)C_f32, (long)ldc, (int)M, (int)N, (int)Kmain, (int)tm, (int)tn, 0, 1);
    if (parts >= 2)
      hipLaunchKernelGGL(gemm_tn256_kernel<2>, dim3((unsigned)(tail_tiles * parts)), dim3(512), 131072, s, (const bf16*)A_bf16,
                         (long)lda, (const bf16*)B_bf16, (long)ldb, (float*)C_f32, (long)ldc, (int)M, (int)N, (int)Kmain, (int)tm,
                         (int)tn, (int)main_tiles, (int)parts);
    if (vr2_check_launch("vr2_gemm_tn_bf16(256)")) return -1;
    if (Kmain == K) return 0;
    A_bf16 = (const bf16*)A_bf16 + Kmain * lda;
    B_bf16 = (const bf16*)B_bf16 + Kmain * ldb;
    K -= Kmain;
    accumulate = 1;
  }
  dim3 grid((unsigned)((N + TN - 1) / TN), (unsigned)((M + TM - 1) / TM));
  const size_t lds = 4 * TILEB;  // 73,728 B
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (accumulate)
    hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), lds, s, (const bf16*)A_bf16, (long)lda, (const bf16*)B_bf16,
                       (long)ldb, (float*)C_f32, (long)ldc, (int)M, (int)N, (int)K);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), lds, s, (const bf16*)A_bf16, (long)lda, (const bf16*)B_bf16,
                       (long)ldb, (float*)C_f32, (long)ldc, (int)M, (int)N, (int)K);
  return vr2_check_launch("vr2_gemm_tn_bf16");
}
