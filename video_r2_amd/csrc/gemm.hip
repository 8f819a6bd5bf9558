// Prefill / teacher-forced bf16 GEMM on the CDNA4 matrix cores:
//
//   C[m][n] (op)= sum_k A[m][k] * W[n][k] + bias[n]      A: [M][K] bf16 activations row-major, W: TILE-PACKED bf16
//   (16-row x 32-k tiles of 1 KiB in MFMA A-fragment order, see gemv.hip / weights.py pack_tiles)
//
// MFMA-bound (ViT K1/K3/K7, LM K11, lm_head K15).  128x128x64 workgroup tile, 4 waves (2x2), each wave a
// 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  The WEIGHT tile is the MFMA A operand and
// the activation tile the B operand, so an accumulator lane holds 4 consecutive n for one m: the epilogue
// (bias / GELU / fp32 residual add) is a single 8- or 16-byte row-major access per accumulator.
// The activation LDS tile is [128 rows][64 k] bf16 with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) so the
// ds_read_b128 fragment reads of 16 different rows at one k-chunk spread over the banks.  The weight LDS tile is a
// verbatim copy of 16 packed 1-KiB tiles: every fragment read is base + lane*16 (conflict-free).  Both tiles are
// staged with direct-to-LDS loads (global_load_lds_dwordx4), double-buffered, the next tile issued before the MFMA
// block, one vmcnt(0) + barrier per k-tile.
#include "common.h"

//
// F8 instantiations (BASELINE config 5, the no-grad passes: rollout prefill, reference prefill + scoring): the SAME kernels on
// e4m3 operands.  Activations = the bytes of vr2_quant_rows_fp8 / vr2_add_rmsnorm_q8 (per-row scale), weights = the 1-KiB
// 16-row x 64-k fp8 tiles of the decode GEMVs (per-channel scale): a 128-byte row segment is now 128 k instead of 64, a pair of
// fp8 weight tiles takes the place of a pair of bf16 ones, so every staging address, LDS image and fragment read below is
// byte-for-byte the bf16 kernel's with K/2 "bf16 columns".  What changes is the multiply: the two 16-byte fragments a lane
// used for two v_mfma_f32_16x16x32_bf16 (k = 64) are the 32-byte operand of ONE v_mfma_scale_f32_16x16x128_f8f6f4 (k = 128,
// unit block scales) -- the block-scaled form runs fp8 at twice the bf16 rate (MI355X_MICROARCH.md, Matrix cores), the
// non-scaled fp8 MFMA only at the bf16 rate -- and the epilogue multiplies the fp32 accumulator by scale_w[n] * scale_x[m].
// Both operands order the k of a 128-block the same way (slot s of each 64-block: k = 8s.., 32 + 8s..), which is all a dot
// product needs.
namespace {

constexpr int BM = 128, BN = 128, BK = 64;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// one block-scaled fp8 MFMA over the 128 k held by the fragment pairs (w0|w1) x (x0|x1); all block scales = 2^0 (e8m0 127)
__device__ __forceinline__ f32x4 mma_f8(const bf16x8& w0, const bf16x8& w1, const bf16x8& x0, const bf16x8& x1, const f32x4& c) {
  const i32x8 a = __builtin_shufflevector(__builtin_bit_cast(i32x4, w0), __builtin_bit_cast(i32x4, w1), 0, 1, 2, 3, 4, 5, 6, 7);
  const i32x8 b = __builtin_shufflevector(__builtin_bit_cast(i32x4, x0), __builtin_bit_cast(i32x4, x1), 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_ADD_F32 = 2, EPI_GELU_BF16 = 3, EPI_SWIGLU_BF16 = 4 };

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ void glds16(const void* g, void* l) {   // 16 B per lane, LDS dst = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// epilogue shared by the tile variants: lane holds D[row n = 4*fq + r][col m = fr] of each 16x16 accumulator; mw / nw are the
// wave's first output row / column
// PRELOADED: the accumulators of an EPI_ADD_F32 launch started from the old output values, nothing is read back here
// F8: acc holds sums of e4m3 products; value = acc * wscale[n] * xscale[m] (applied here, before bias / activation)
template <int EPI, int MT, bool UNUSED = false, bool PRELOADED = false, bool F8 = false>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[4][MT], int mw, int nw, int n0, int wn, int fr, int fq,
                                         const float* __restrict__ bias, void* __restrict__ Cv, long ldc, int M, int N,
                                         bf16* __restrict__ aux, long ldaux, const float* __restrict__ wscale = nullptr,
                                         const float* __restrict__ xscale = nullptr) {
  if constexpr (F8) {
    float xs[MT];
    float4 ws[4];
#pragma unroll
    for (int im = 0; im < MT; ++im) { const int m = mw + im * 16 + fr; xs[im] = m < M ? xscale[m] : 0.f; }
#pragma unroll
    for (int in = 0; in < 4; ++in) {
      const int n = nw + in * 16 + fq * 4;     // (SwiGLU: interleaved-space row, the scales are stored in that order)
      ws[in] = n < N ? *reinterpret_cast<const float4*>(wscale + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int in = 0; in < 4; ++in) asm volatile("" ::"v"(ws[in].x), "v"(ws[in].y), "v"(ws[in].z), "v"(ws[in].w));
#pragma unroll
    for (int in = 0; in < 4; ++in)
#pragma unroll
      for (int im = 0; im < MT; ++im) {
        acc[in][im][0] *= ws[in].x * xs[im]; acc[in][im][1] *= ws[in].y * xs[im];
        acc[in][im][2] *= ws[in].z * xs[im]; acc[in][im][3] *= ws[in].w * xs[im];
      }
  }
  if (EPI == EPI_SWIGLU_BF16) {
    // weight tiles alternate gate / up (weights.py interleave_gate_up): acc[2j] = gate, acc[2j+1] = up of the SAME 16
    // output columns, held by the same lane -> act = silu(gate + bg) * (up + bu), one rounding, no gate/up round trip
    float4 bgs[2], bus[2];   // bias reads first and waited for once (see the generic path below)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ng = nw + (2 * j) * 16 + fq * 4;
      const bool ld = bias && ng < N;
      bgs[j] = ld ? *reinterpret_cast<const float4*>(bias + ng) : make_float4(0.f, 0.f, 0.f, 0.f);
      bus[j] = ld ? *reinterpret_cast<const float4*>(bias + ng + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      asm volatile("" ::"v"(bgs[j].x), "v"(bgs[j].y), "v"(bgs[j].z), "v"(bgs[j].w), "v"(bus[j].x), "v"(bus[j].y), "v"(bus[j].z), "v"(bus[j].w));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ng = nw + (2 * j) * 16 + fq * 4;      // interleaved-space row of the gate values
      if (ng >= N) continue;
      const float4 bg = bgs[j], bu = bus[j];
      const int no = (nw >> 5) * 16 + j * 16 + fq * 4;  // output column
#pragma unroll
      for (int im = 0; im < MT; ++im) {
        const int m = mw + im * 16 + fr;
        if (m >= M) continue;
        const f32x4 g = acc[2 * j][im], u = acc[2 * j + 1][im];
        const float g0 = g[0] + bg.x, g1 = g[1] + bg.y, g2 = g[2] + bg.z, g3 = g[3] + bg.w;
        const float u0 = u[0] + bu.x, u1 = u[1] + bu.y, u2 = u[2] + bu.z, u3 = u[3] + bu.w;
        bf16x4 o = {f2bf(g0 / (1.0f + expf(-g0)) * u0), f2bf(g1 / (1.0f + expf(-g1)) * u1),
                    f2bf(g2 / (1.0f + expf(-g2)) * u2), f2bf(g3 / (1.0f + expf(-g3)) * u3)};
        *reinterpret_cast<bf16x4*>((bf16*)Cv + (long)m * ldc + no) = o;
        if (aux) {   // training forward: keep the pre-activations (interleaved layout) for the backward
          bf16x4 sg = {f2bf(g0), f2bf(g1), f2bf(g2), f2bf(g3)}, su = {f2bf(u0), f2bf(u1), f2bf(u2), f2bf(u3)};
          *reinterpret_cast<bf16x4*>(aux + (long)m * ldaux + ng) = sg;
          *reinterpret_cast<bf16x4*>(aux + (long)m * ldaux + ng + 16) = su;
        }
      }
    }
    return;
  }
  float4 bvs[4];   // all bias reads before the first store: a read between store groups makes the compiler wait for the stores too
#pragma unroll
  for (int in = 0; in < 4; ++in) {
    const int n = nw + in * 16 + fq * 4;
    bvs[in] = (bias && n < N) ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ... and waited for HERE: behind the per-row branches below the wait-count pass no longer knows the reads have landed and
  // puts s_waitcnt vmcnt(0) in front of every store, i.e. one memory round trip per 16 x 4 block, 32 in a row per thread
#pragma unroll
  for (int in = 0; in < 4; ++in) asm volatile("" ::"v"(bvs[in].x), "v"(bvs[in].y), "v"(bvs[in].z), "v"(bvs[in].w));
#pragma unroll
  for (int in = 0; in < 4; ++in) {
    const int n = nw + in * 16 + fq * 4;
    if (n >= N) continue;
    const float4 bv = bvs[in];
#pragma unroll
    for (int im = 0; im < MT; ++im) {
      const int m = mw + im * 16 + fr;
      if (m >= M) continue;
      float v0 = acc[in][im][0] + bv.x, v1 = acc[in][im][1] + bv.y, v2 = acc[in][im][2] + bv.z,
            v3 = acc[in][im][3] + bv.w;
      if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16) {
        if (EPI == EPI_GELU_BF16) { v0 = gelu_erf_f(v0); v1 = gelu_erf_f(v1); v2 = gelu_erf_f(v2); v3 = gelu_erf_f(v3); }
        bf16x4 o = {f2bf(v0), f2bf(v1), f2bf(v2), f2bf(v3)};
        *reinterpret_cast<bf16x4*>((bf16*)Cv + (long)m * ldc + n) = o;
      } else {
        float* cp = (float*)Cv + (long)m * ldc + n;
        if (EPI == EPI_ADD_F32 && !PRELOADED) {
          const float4 old = *reinterpret_cast<const float4*>(cp);
          v0 += old.x; v1 += old.y; v2 += old.z; v3 += old.w;
        }
        *reinterpret_cast<float4*>(cp) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

// MT = 16-row activation fragments per wave: workgroup tile BM x 128 with BM = 32*MT (128 for big grids; 64 when the 128-row
// grid would leave CUs with fewer than ~2 workgroups -- prefill at M = S ~ 1k rows, ViT projections with N = 1280)
// SPLIT: gridDim.z k-parts of a launch with too few output tiles to fill the chip (the < 256 tail rows of a 256-row tiling at
// N = 3584 are 28 tiles, each walking the whole K: 0.47 ms at K = 37888 on 28 CUs); fp32 outputs only.  Every part stores its
// partial tile into a slab of its own (Cv = the slab workspace: [part][tile][64][128] fp32) and splitk_reduce_kernel adds the
// parts in part order: the result does not depend on which part finishes first (rounds 1-3 added the parts with fp32 atomics:
// the tail rows of a forward pass then differed in their last bits from run to run, and with them the log-probs of two passes
// over the same weights)
// F8: A / Wp are e4m3 bytes addressed as pairs (lda, K in 2-byte units = half the fp8 counts), see the file header
template <int EPI, int MT, bool SPLIT = false, bool F8 = false>
__global__ __launch_bounds__(256) void gemm_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp,
                                                   int n_tiles, const float* __restrict__ bias, void* __restrict__ Cv,
                                                   long ldc, int M, int N, int K, bf16* __restrict__ aux, long ldaux,
                                                   const float* __restrict__ wscale = nullptr,
                                                   const float* __restrict__ xscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A (BM x 128 B) | W 16K]
  constexpr int BM_ = 32 * MT;
  constexpr int ABYTES = BM_ * 128;           // activation tile bytes per buffer
  constexpr int BUF = ABYTES + 16384;
  constexpr int APIECES = BM_ / 8;            // 1-KiB pieces of the activation tile (8 rows each)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM_, n0 = blockIdx.x * BN;
  const int fr = lane & 15, fq = lane >> 4;

  // staging: DIRECT-TO-LDS loads (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  One wave instruction moves
  // 64 lanes x 16 B = 1 KiB into a lane-linear KiB of LDS.  Per k-tile each wave issues 4 activation pieces (8 rows x
  // 128 B; the XOR swizzle is applied to the per-lane SOURCE address, the LDS image stays linear) and 4 weight pieces
  // (one packed 1-KiB tile each, already in fragment order).  Measured vs register staging on MI355X
  // (profiles/src/gemm_bench.hip): 0.69-0.75 -> 0.87-1.06 PFLOP/s on the LM shapes.
  constexpr int NA = APIECES / 4;                      // activation pieces per wave (4 or 2)
  const bf16* ga[NA];
  const bf16* gw[4];
  int la[NA], lw[4];
  const int ksteps = K >> 5;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int piece = wave * NA + i;
    const int row = piece * 8 + (lane >> 3), c = lane & 7;
    int am = m0 + row; if (am > M - 1) am = M - 1;     // tail rows: clamp the load, never store
    ga[i] = A + (long)am * lda + ((c ^ (row & 7)) << 3);
    la[i] = piece * 1024;                              // wave-uniform LDS base of the piece
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;                    // 0..15
    const int sub = piece >> 1, kk = piece & 1;        // LDS sub-tile (sub, kk) = packed tile (n0/16 + sub, 2t + kk)
    int nt = (n0 >> 4) + sub; if (nt > n_tiles - 1) nt = n_tiles - 1;
    gw[i] = Wp + ((long)nt * ksteps + kk) * 512 + lane * 8;
    lw[i] = ABYTES + piece * 1024;
  }

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int t_first = 0, nt = K / BK;
  if (SPLIT) {
    const int all = K / BK;
    t_first = (int)((long)blockIdx.z * all / gridDim.z);
    nt = (int)((long)(blockIdx.z + 1) * all / gridDim.z);
    if (blockIdx.z != 0) bias = nullptr;
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) glds16(ga[i] + (long)t_first * BK, smem + (t_first & 1) * BUF + la[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) glds16(gw[i] + (long)t_first * 1024, smem + (t_first & 1) * BUF + lw[i]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = t_first; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* nxt = smem + ((t + 1) & 1) * BUF;
    if (t + 1 < nt) {   // stage tile t+1 BEFORE computing tile t: its HBM/L2 latency hides under the MFMA block
#pragma unroll
      for (int i = 0; i < NA; ++i) glds16(ga[i] + (long)(t + 1) * BK, nxt + la[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(gw[i] + (long)(t + 1) * 1024, nxt + lw[i]);   // two packed k-steps per k-tile
    }
    if constexpr (F8) {
      bf16x8 fa[MT][2], fw[4][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[i][kk] = *reinterpret_cast<const bf16x8*>(cur + ABYTES + ((wn * 4 + i) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i][kk] = *reinterpret_cast<const bf16x8*>(cur + swz(wm * (16 * MT) + i * 16 + fr, kk * 4 + fq));
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int in = 0; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < MT; ++im) acc[in][im] = mma_f8(fw[in][0], fw[in][1], fa[im][0], fa[im][1], acc[in][im]);
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MT], fw[4];
      const int chunk = kk * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(cur + ABYTES + ((wn * 4 + i) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(cur + swz(wm * (16 * MT) + i * 16 + fr, chunk));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int in = 0; in < 4; ++in)
#pragma unroll
        for (int im = 0; im < MT; ++im)
          acc[in][im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[in], fa[im], acc[in][im], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of tile t+1 has landed ...
    __syncthreads();                                   // ... for every wave, and tile t is fully consumed
  }

  if constexpr (SPLIT) {
    // slab tile of this (part, tile): rows m0.., columns n0.. of a [BM_][128] fp32 block; the plain-store epilogue addresses it
    // through a base shifted by the tile's origin
    const long tile = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* slab = (float*)Cv + tile * (BM_ * BN) - ((long)m0 * BN + n0);
    epilogue<EPI_F32, MT, false, false, F8>(acc, m0 + wm * (16 * MT), n0 + wn * 64, n0, wn, fr, fq, bias, slab, BN, M, N, nullptr, 0, wscale, xscale);
  } else {
    epilogue<EPI, MT, false, false, F8>(acc, m0 + wm * (16 * MT), n0 + wn * 64, n0, wn, fr, fq, bias, Cv, ldc, M, N, aux, ldaux, wscale, xscale);
  }
}

// C (+)= sum over the k-parts of a SPLIT launch, in part order.  grid (n tiles, m tiles, 8): a workgroup sums one eighth (8 rows) of a
// 64 x 128 output tile, one float4 per thread, the parts' loads issued eight at a time (a one-load-one-add loop over the parts is a
// chain of memory round trips: 23 us per launch at 9 parts, rocprofv3 round 4)
template <bool ADD>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int parts, float* __restrict__ C, long ldc, int M, int N) {
  const long tiles = (long)gridDim.x * gridDim.y, tile = (long)blockIdx.y * gridDim.x + blockIdx.x;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BN;
  const int e = ((int)blockIdx.z * 256 + (int)threadIdx.x) * 4, r = e >> 7, c = e & 127;
  const int m = m0 + r, n = n0 + c;
  if (m >= M || n >= N) return;
  float* cp = C + (long)m * ldc + n;
  float4 v = ADD ? *reinterpret_cast<const float4*>(cp) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* sp = slabs + tile * (64 * BN) + e;
  const long pstride = tiles * (64 * BN);
  for (int p0 = 0; p0 < parts; p0 += 8) {
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (p0 + j < parts) ? *reinterpret_cast<const float4*>(sp + (p0 + j) * pstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }   // part order (+ 0 past the last part)
  }
  *reinterpret_cast<float4*>(cp) = v;
}

// slab workspace of the SPLIT launches, owned by the caller (vr2_set_gemm_workspace): a launch needs parts x tiles x 32 KiB
// <= 256 x 32 KiB = 8 MiB; without one the tail rows run as full-K tiles (slower, still deterministic)
float* g_split_ws = nullptr;
int64_t g_split_ws_bytes = 0;

// ---- 256 x 256 x 64 tile, 8 waves (2 M x 4 N, 128 x 64 outputs each), for grids of >= 2 such tiles per CU ----------------
// LDS (128 KiB): 2 buffers (k-tile parity) x [X-lo 16K | X-hi 16K | W-lo 16K | W-hi 16K]; X halves are 128 activation rows x 64 k
// (XOR-swizzled through the source address), W halves 8 packed n-tiles x 2 k-steps (verbatim).  Per k-tile T the four
// quadrants of a wave's output are multiplied in phases p0..p3 (16 MFMAs each); the fragments are read from LDS in p0 (all of
// W + the X rows of quadrants 0/1) and p2 (the other X rows).  One half-tile is staged per phase by direct-to-LDS loads:
//   p0 -> X-lo(T+1), p1 -> X-hi(T+1), p2 -> W-lo(T+2), p3 -> W-hi(T+2)
// (a half-tile is restaged two phases after its last read), and the only wait is a counted vmcnt(4) in p3: tile T+1 has landed,
// the two younger half-tiles stay in flight across the barriers -- the k-loop never drains the load queue.  Waves 4-7 run one
// barrier slot behind waves 0-3, so on every SIMD one wave multiplies while the other reads LDS and issues loads.
// Workgroup -> tile: ids that share an XCD (id % 8) walk groups of 8 M-tiles column by column, so the ~32 tiles an XCD has in
// flight share 8 X panels and ~4 W panels in its L2 (group sizes 2 / 4 / 8 / 16 / all measured: 1195 / 1216 / 1235 / 1218 /
// 1199 TFLOP/s on the LM gate/up shape).
// Measured (profiles/src/gemm_bench.hip, MI355X, bf16, random operands): LM shapes at T = 9280 rows 0.97-1.23 PFLOP/s vs 0.82-0.91
// for the 128-row tile; ViT qkv / gate-up (M = 4096) 0.95 / 0.88 vs 0.81 / 0.78.
template <int EPI, bool F8 = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ Wp,
                                                      int n_tiles, const float* __restrict__ bias, void* __restrict__ Cv,
                                                      long ldc, int M, int N, int K, bf16* __restrict__ aux, long ldaux,
                                                      int tiles_m, int tiles_n, const float* __restrict__ wscale = nullptr,
                                                      const float* __restrict__ xscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  int bm, bn;
  {
    const int nb = tiles_m * tiles_n, id = blockIdx.x;
    const int q = nb / 8, r = nb % 8, x = id % 8;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;   // bijective XCD remap
    const int per_group = 8 * tiles_n, grp = wg / per_group, in_grp = wg % per_group;
    const int rows = (tiles_m - grp * 8) < 8 ? (tiles_m - grp * 8) : 8;          // the last group may be shorter
    bm = grp * 8 + in_grp % rows;
    bn = in_grp / rows;
  }
  const int m0 = bm * 256, n0 = bn * 256;
  const int ksteps = K >> 5, NT = K >> 6;
  const bf16* gx[2][2];
  const bf16* gw[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i, row = piece * 8 + (lane >> 3), c = lane & 7;
      int am = m0 + h * 128 + row; if (am > M - 1) am = M - 1;                   // tail rows: clamp the load, never store
      gx[h][i] = A + (long)am * lda + ((c ^ (row & 7)) << 3);
      int nt = (n0 >> 4) + h * 8 + (piece >> 1); if (nt > n_tiles - 1) nt = n_tiles - 1;
      gw[h][i] = Wp + ((long)nt * ksteps + (piece & 1)) * 512 + lane * 8;
    }
  auto stage_x = [&](int h, int T) {
    if (T >= NT) return;
    char* dst = smem + (T & 1) * 65536 + h * 16384 + wave * 2048;
    glds16(gx[h][0] + (long)T * 64, dst);
    glds16(gx[h][1] + (long)T * 64, dst + 1024);
  };
  auto stage_w = [&](int h, int T) {
    if (T >= NT) return;
    char* dst = smem + (T & 1) * 65536 + 32768 + h * 16384 + wave * 2048;
    glds16(gw[h][0] + (long)T * 1024, dst);
    glds16(gw[h][1] + (long)T * 1024, dst + 1024);
  };
  f32x4 acc[4][8];
  // residual-add epilogue: the accumulators START from the old output (32 reads in flight beside the prologue's loads, waited for
  // once, before the loop) -- a read-add-store epilogue costs a memory round trip per 16 x 4 block, 32 per thread, after the k-loop
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_ADD_F32 && !F8) {     // (F8: the accumulator is scaled in the epilogue, the old output is added there)
        const int n = n0 + wn * 64 + i * 16 + fq * 4, m = m0 + wm * 128 + j * 16 + fr;
        if (n < N && m < M) acc[i][j] = *reinterpret_cast<const f32x4*>((const float*)Cv + (long)m * ldc + n);
      }
    }
  // prologue: tile 0 and the W halves of tile 1
  stage_w(0, 0); stage_w(1, 0); stage_x(0, 0); stage_x(1, 0); stage_w(0, 1); stage_w(1, 1);
  if (EPI == EPI_ADD_F32 && !F8) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
  }
  if (NT > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();        // stagger: waves 4-7 run one slot behind
  bf16x8 Wf[4][2], Xs[4][2];
  const int xoff = wm * 16384, woff = 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192 + lane * 16;
#define VR2_MMA(N0, M0)                                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                               \
  __builtin_amdgcn_s_setprio(1);                                                                                              \
  if constexpr (F8) {          /* hipcc sinks the scaled MFMAs of all four phases below the last barrier of the k-tile   */   \
                               /* unless their accumulators are pinned on both sides (empty asm volatile, no instruction) */   \
    _Pragma("unroll") for (int in = N0; in < N0 + 2; ++in)                                                                    \
      _Pragma("unroll") for (int im = 0; im < 4; ++im) asm volatile("" : "+v"(acc[in][M0 + im]));                             \
    _Pragma("unroll") for (int in = N0; in < N0 + 2; ++in)                                                                    \
      _Pragma("unroll") for (int im = 0; im < 4; ++im)                                                                        \
        acc[in][M0 + im] = mma_f8(Wf[in][0], Wf[in][1], Xs[im][0], Xs[im][1], acc[in][M0 + im]);                             \
    _Pragma("unroll") for (int in = N0; in < N0 + 2; ++in)                                                                    \
      _Pragma("unroll") for (int im = 0; im < 4; ++im) asm volatile("" : "+v"(acc[in][M0 + im]));                             \
  } else {                                                                                                                    \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                            \
    _Pragma("unroll") for (int in = N0; in < N0 + 2; ++in)                                                                    \
      _Pragma("unroll") for (int im = 0; im < 4; ++im)                                                                        \
        acc[in][M0 + im] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[in][kk], Xs[im][kk], acc[in][M0 + im], 0, 0, 0);       \
  }                                                                                                                           \
  __builtin_amdgcn_s_setprio(0);                                                                                              \
  __builtin_amdgcn_s_barrier()
  for (int T = 0; T < NT; ++T) {
    const char* cur = smem + (T & 1) * 65536;
    // p0
#pragma unroll
    for (int in = 0; in < 4; ++in)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Wf[in][kk] = *reinterpret_cast<const bf16x8*>(cur + woff + (in * 2 + kk) * 1024);
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Xs[im][kk] = *reinterpret_cast<const bf16x8*>(cur + xoff + swz(im * 16 + fr, kk * 4 + fq));
    stage_x(0, T + 1);
    VR2_MMA(0, 0);
    // p1
    stage_x(1, T + 1);
    VR2_MMA(2, 0);
    // p2
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Xs[im][kk] = *reinterpret_cast<const bf16x8*>(cur + xoff + swz(64 + im * 16 + fr, kk * 4 + fq));
    stage_w(0, T + 2);
    VR2_MMA(2, 4);
    // p3: tile T+1 must have landed before the next p0 reads it; the two half-tiles of T+2 issued since stay in flight
    stage_w(1, T + 2);
    if (T + 2 < NT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VR2_MMA(0, 4);
  }
#undef VR2_MMA
  if (wm == 0) __builtin_amdgcn_s_barrier();        // pairs with the stagger barrier of waves 4-7
  epilogue<EPI, 8, false, !F8, F8>(acc, m0 + wm * 128, n0 + wn * 64, n0, wn, fr, fq, bias, Cv, ldc, M, N, aux, ldaux, wscale, xscale);
}

// F8 launches take A / W as byte pointers re-addressed in 2-byte units: lda, K = HALF the fp8 counts (see the file header)
template <int EPI, bool F8 = false>
void launch128(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, int64_t M, int64_t N,
               int64_t K, void* stream, void* aux, int64_t ldaux, const float* ws = nullptr, const float* xs = nullptr) {
  const long nb = (N + BN - 1) / BN;
  const long big = ((M + 127) / 128) * nb;
  if (big >= 512) {          // >= 2 workgroups per CU with 128-row tiles
    dim3 grid((unsigned)nb, (unsigned)((M + 127) / 128));
    hipLaunchKernelGGL((gemm_kernel<EPI, 4, false, F8>), grid, dim3(256), 65536, (hipStream_t)stream, (const bf16*)A, (long)lda,
                       (const bf16*)W, (int)((N + 15) / 16), (const float*)bias, C, (long)ldc, (int)M, (int)N, (int)K,
                       (bf16*)aux, (long)ldaux, ws, xs);
  } else {                   // small grid: 64-row tiles double the workgroup count (48 KiB LDS -> 3 per CU)
    dim3 grid((unsigned)nb, (unsigned)((M + 63) / 64));
    hipLaunchKernelGGL((gemm_kernel<EPI, 2, false, F8>), grid, dim3(256), 49152, (hipStream_t)stream, (const bf16*)A, (long)lda,
                       (const bf16*)W, (int)((N + 15) / 16), (const float*)bias, C, (long)ldc, (int)M, (int)N, (int)K,
                       (bf16*)aux, (long)ldaux, ws, xs);
  }
}

// few rows (a tail of <= 128), fp32 output, long K: 64-row tiles with the k-tiles dealt to enough parts to fill the CUs, each
// part into its own slab, summed in part order by splitk_reduce_kernel
template <int EPI, bool F8 = false>
bool launch_tail_splitk(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, int64_t M, int64_t N,
                        int64_t K, void* stream, const float* ws = nullptr, const float* xs = nullptr) {
  if constexpr (EPI == EPI_F32 || EPI == EPI_ADD_F32) {
    const long nb = (N + BN - 1) / BN, mb = (M + 63) / 64, kt = K / BK;
    long parts = 256 / (nb * mb);
    if (parts > kt / 4) parts = kt / 4;
    if (parts > 32) parts = 32;
    if (parts < 2) return false;
    if (g_split_ws == nullptr || g_split_ws_bytes < (int64_t)parts * nb * mb * 64 * BN * 4) return false;
    dim3 grid((unsigned)nb, (unsigned)mb, (unsigned)parts);
    hipLaunchKernelGGL((gemm_kernel<EPI, 2, true, F8>), grid, dim3(256), 49152, (hipStream_t)stream, (const bf16*)A, (long)lda,
                       (const bf16*)W, (int)((N + 15) / 16), (const float*)bias, (void*)g_split_ws, (long)ldc, (int)M, (int)N, (int)K,
                       (bf16*)nullptr, 0L, ws, xs);
    hipLaunchKernelGGL((splitk_reduce_kernel<EPI == EPI_ADD_F32>), dim3((unsigned)nb, (unsigned)mb, 8), dim3(256), 0, (hipStream_t)stream,
                       (const float*)g_split_ws, (int)parts, (float*)C, (long)ldc, (int)M, (int)N);
    return true;
  } else {
    return false;
  }
}

template <int EPI, bool F8 = false>
void launch256(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, int64_t M, int64_t N,
               int64_t K, void* stream, void* aux, int64_t ldaux, const float* ws = nullptr, const float* xs = nullptr) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<EPI, F8>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_set = true;
  }
  const long tm = (M + 255) / 256, tn = (N + 255) / 256;
  hipLaunchKernelGGL((gemm256_kernel<EPI, F8>), dim3((unsigned)(tm * tn)), dim3(512), 131072, (hipStream_t)stream, (const bf16*)A,
                     (long)lda, (const bf16*)W, (int)((N + 15) / 16), (const float*)bias, C, (long)ldc, (int)M, (int)N, (int)K,
                     (bf16*)aux, (long)ldaux, (int)tm, (int)tn, ws, xs);
}

// efficiency of a 256-row tiling of M x N, in 1/1024: tiles / (rounds * 256) * M / (tm * 256)
inline long eff256(int64_t M, int64_t N) {
  const long tm = (M + 255) / 256, tn = (N + 255) / 256, t = tm * tn, rounds = (t + 255) / 256;
  return t * M * 1024 / (rounds * 256 * tm * 256);
}

template <int EPI, bool F8 = false>
int launch(const void* A, int64_t lda, const void* W, const void* bias, void* C, int64_t ldc, int64_t M, int64_t N,
           int64_t K, void* stream, void* aux = nullptr, int64_t ldaux = 0, const float* ws = nullptr, const float* xs = nullptr) {
  // 256-row tiles (one workgroup per CU) when their last round over the 256 CUs and their last row of tiles are mostly full
  // (efficiency >= 0.75).  Measured crossovers (profiles/src/gemm_bench.hip): LM q/k/v and gate/up at T ~ 9k rows, lm_head chunks,
  // ViT q/k/v and gate/up qualify; N = 1280 and the M ~ 1k prefill projections stay on the 128-row tiles.  When the full M
  // does not qualify but its whole 256-row tiles do at >= 0.9 (N = 3584 at T = 9280: 37 x 14 tiles = 2.02 rounds, 36 x 14 =
  // 1.97), the first M - M % 256 rows take the 256-row kernel and the few tail rows the 128-row one.
  static const bool no256 = getenv("VR2_GEMM_NO256") != nullptr;
  const int64_t tail = M % 256, main = M - tail;
  const size_t esz = (EPI == EPI_F32 || EPI == EPI_ADD_F32) ? 4 : 2;
  if (!no256 && eff256(M, N) >= 768) {
    launch256<EPI, F8>(A, lda, W, bias, C, ldc, M, N, K, stream, aux, ldaux, ws, xs);
  } else if (!no256 && main >= 2048 && tail > 0 && tail <= 128 && eff256(main, N) >= 922) {
    launch256<EPI, F8>(A, lda, W, bias, C, ldc, main, N, K, stream, aux, ldaux, ws, xs);
    const float* xt = xs ? xs + main : nullptr;
    if (!launch_tail_splitk<EPI, F8>((const bf16*)A + main * lda, lda, W, bias, (char*)C + main * ldc * esz, ldc, tail, N, K, stream, ws, xt))
      launch128<EPI, F8>((const bf16*)A + main * lda, lda, W, bias, (char*)C + main * ldc * esz, ldc, tail, N, K, stream,
                         aux ? (bf16*)aux + main * ldaux : nullptr, ldaux, ws, xt);
  } else {
    launch128<EPI, F8>(A, lda, W, bias, C, ldc, M, N, K, stream, aux, ldaux, ws, xs);
  }
  return vr2_check_launch(F8 ? "vr2_gemm_a8w8" : "vr2_gemm_bf16");
}

}  // namespace

// fp32 scratch for the k-parts of split launches (8 MiB covers every shape; see g_split_ws).  The caller owns the memory, keeps it
// alive while GEMMs run and uses ONE stream per workspace.  ptr = NULL unregisters it.
extern "C" int vr2_set_gemm_workspace(void* ptr, int64_t bytes) {
  VR2_REQUIRE((ptr == nullptr) == (bytes == 0) && bytes >= 0 && ((uintptr_t)ptr & 15) == 0, "vr2_set_gemm_workspace: bad workspace (%p, %ld bytes)", ptr, (long)bytes);
  g_split_ws = (float*)ptr;
  g_split_ws_bytes = bytes;
  return 0;
}

extern "C" int vr2_gemm_bf16(const void* A, int64_t lda, const void* Wpacked, const void* bias_f32, void* C, int64_t ldc,
                             int64_t M, int64_t N, int64_t K, int epilogue, void* stream) {
  const void* W = Wpacked;
  VR2_REQUIRE(M >= 1 && N >= 4 && K >= 64, "vr2_gemm_bf16: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  VR2_REQUIRE(K % 64 == 0, "vr2_gemm_bf16: K=%ld must be a multiple of 64 (pad the operand buffers)", (long)K);
  VR2_REQUIRE(N % 4 == 0, "vr2_gemm_bf16: N=%ld must be a multiple of 4", (long)N);
  VR2_REQUIRE(lda % 8 == 0 && ldc % 4 == 0, "vr2_gemm_bf16: lda %% 8 and ldc %% 4 must be 0");
  VR2_REQUIRE(lda >= K && (ldc >= N || epilogue == EPI_SWIGLU_BF16), "vr2_gemm_bf16: leading dimensions smaller than the row length");
  switch (epilogue) {
    case EPI_BF16: return launch<EPI_BF16>(A, lda, W, bias_f32, C, ldc, M, N, K, stream);
    case EPI_F32: return launch<EPI_F32>(A, lda, W, bias_f32, C, ldc, M, N, K, stream);
    case EPI_ADD_F32: return launch<EPI_ADD_F32>(A, lda, W, bias_f32, C, ldc, M, N, K, stream);
    case EPI_GELU_BF16: return launch<EPI_GELU_BF16>(A, lda, W, bias_f32, C, ldc, M, N, K, stream);
    case EPI_SWIGLU_BF16:
      VR2_REQUIRE(N % 32 == 0 && ldc >= N / 2, "vr2_gemm_bf16: SwiGLU epilogue needs N %% 32 == 0 (interleaved gate/up tiles)");
      return launch<EPI_SWIGLU_BF16>(A, lda, W, bias_f32, C, ldc, M, N, K, stream);
  }
  vr2_set_error("vr2_gemm_bf16: unknown epilogue %d", epilogue);
  return -1;
}

// SwiGLU GEMM that also stores the bf16 gate/up pre-activations (interleaved [M][N]) for the backward pass
extern "C" int vr2_gemm_swiglu_save(const void* A, int64_t lda, const void* Wpacked, const void* bias_f32, void* act_bf16,
                                    int64_t ldact, void* gu_bf16, int64_t ldgu, int64_t M, int64_t N, int64_t K, void* stream) {
  VR2_REQUIRE(M >= 1 && K >= 64 && K % 64 == 0 && N % 32 == 0, "vr2_gemm_swiglu_save: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  VR2_REQUIRE(lda % 8 == 0 && lda >= K && ldact % 4 == 0 && ldact >= N / 2 && ldgu % 4 == 0 && ldgu >= N && gu_bf16 != nullptr,
              "vr2_gemm_swiglu_save: bad strides");
  return launch<EPI_SWIGLU_BF16>(A, lda, Wpacked, bias_f32, act_bf16, ldact, M, N, K, stream, gu_bf16, ldgu);
}

// BASELINE config 5, the no-grad GEMMs on the fp8 matrix path: C (op)= (q_x q_w^T) * scale_x[m] * scale_w[n] + bias.  x = the
// e4m3 bytes + per-row scales of vr2_quant_rows_fp8 / vr2_add_rmsnorm_q8, W = the per-channel e4m3 tiles of vr2_gemv_w8 (one
// fp8 copy of a weight serves the decode GEMVs and these GEMMs).  Same tilings, staging and epilogues as vr2_gemm_bf16; the
// multiply is v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the bf16 MFMA rate on half the staged bytes).
extern "C" int vr2_gemm_a8w8(const void* x_fp8, int64_t ldx8, const void* xscale_f32, const void* Wpacked_fp8, const void* wscale_f32,
                             const void* bias_f32, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue, void* stream) {
  VR2_REQUIRE(M >= 1 && N >= 4 && K >= 128 && K % 128 == 0, "vr2_gemm_a8w8: bad shape M=%ld N=%ld K=%ld (K %% 128)", (long)M, (long)N, (long)K);
  VR2_REQUIRE(N % 4 == 0 && ldx8 % 16 == 0 && ldx8 >= K && ldc % 4 == 0, "vr2_gemm_a8w8: N %% 4, ldx8 %% 16, ldc %% 4");
  VR2_REQUIRE(ldc >= N || epilogue == EPI_SWIGLU_BF16, "vr2_gemm_a8w8: ldc smaller than the row length");
  VR2_REQUIRE(xscale_f32 && wscale_f32, "vr2_gemm_a8w8: scales missing");
  const float* ws = (const float*)wscale_f32;
  const float* xs = (const float*)xscale_f32;
  const int64_t lda = ldx8 / 2, K2 = K / 2;     // the kernels address the bytes in pairs
  switch (epilogue) {
    case EPI_BF16: return launch<EPI_BF16, true>(x_fp8, lda, Wpacked_fp8, bias_f32, C, ldc, M, N, K2, stream, nullptr, 0, ws, xs);
    case EPI_F32: return launch<EPI_F32, true>(x_fp8, lda, Wpacked_fp8, bias_f32, C, ldc, M, N, K2, stream, nullptr, 0, ws, xs);
    case EPI_ADD_F32: return launch<EPI_ADD_F32, true>(x_fp8, lda, Wpacked_fp8, bias_f32, C, ldc, M, N, K2, stream, nullptr, 0, ws, xs);
    case EPI_SWIGLU_BF16:
      VR2_REQUIRE(N % 32 == 0 && ldc >= N / 2, "vr2_gemm_a8w8: SwiGLU epilogue needs N %% 32 == 0 (interleaved gate/up tiles)");
      return launch<EPI_SWIGLU_BF16, true>(x_fp8, lda, Wpacked_fp8, bias_f32, C, ldc, M, N, K2, stream, nullptr, 0, ws, xs);
  }
  vr2_set_error("vr2_gemm_a8w8: epilogue %d not built for fp8 operands", epilogue);
  return -1;
}
