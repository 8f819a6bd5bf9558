// Decode GEMVs with per-channel fp8 (e4m3, OCP) WEIGHTS and bf16 activations (BASELINE config 5): the weights stream from HBM
// at half the bytes and are expanded to bf16 in registers (v_cvt_pk_f32_fp8 + v_perm: an e4m3 value is exact in bf16), the MFMA
// stays v_mfma_f32_16x16x32_bf16 and the per-channel scale multiplies the fp32 accumulator in the epilogue.
//
// HBM layout (weights.py pack_tiles_fp8): tile (nt, kt) = 16 rows x 64 k = 1 KiB; lane l = 16*q + r owns 16 bytes: the 8 k's
// of MFMA k-step 2kt (k = 64kt + 8q ..) then the 8 k's of k-step 2kt+1 -- one 16-byte load feeds two MFMAs.
// Two kernels mirror gemv.hip: the direct one for M <= 16 rows (4 waves split K, LDS reduce) and the LDS-staged one for
// 16 < M <= 128 (4 waves own different tile pairs over the same k range and share the x chunk).  Up to 64 rows the staged
// kernel walks 256-k chunks (the bf16 one: 128) so a wave keeps the same 8 KiB of weight loads in flight per round trip;
// above 64 rows the activations' L2 -> LDS traffic dominates and the 128-k variant with a second k group wins
// (128 rows: 6.8 vs 5.2 ms/token; 64 rows: both 3.8).
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void dq4(unsigned d, unsigned& o0, unsigned& o1) {   // 4 fp8 -> 4 bf16 in 2 dwords
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)d, false);
  const f32x2 b = __builtin_amdgcn_cvt_pk_f32_fp8((int)d, true);
  o0 = __builtin_amdgcn_perm(__float_as_uint(a[1]), __float_as_uint(a[0]), 0x07060302u);   // bf16 = upper half of the fp32
  o1 = __builtin_amdgcn_perm(__float_as_uint(b[1]), __float_as_uint(b[0]), 0x07060302u);
}

__device__ __forceinline__ void dq16(const u32x4 w, bf16x8& lo, bf16x8& hi) {
  u32x4 a, b;
  unsigned t0, t1;
  dq4(w[0], t0, t1); a[0] = t0; a[1] = t1;
  dq4(w[1], t0, t1); a[2] = t0; a[3] = t1;
  dq4(w[2], t0, t1); b[0] = t0; b[1] = t1;
  dq4(w[3], t0, t1); b[2] = t0; b[3] = t1;
  lo = __builtin_bit_cast(bf16x8, a);
  hi = __builtin_bit_cast(bf16x8, b);
}

// ---- M <= 16: MODE 0 fp32 partial store of tiles (2p, 2p+1); MODE 1 gate tile 2p / up tile 2p+1 -> silu(g)*u bf16
template <int MODE>
__global__ __launch_bounds__(256) void gemv16_w8_kernel(const bf16* __restrict__ x, int ldx, int M, const unsigned char* __restrict__ Wp,
                                                        const float* __restrict__ scale, int n_tiles, int N, int K,
                                                        float* __restrict__ out_f32, long split_stride, int ldo,
                                                        bf16* __restrict__ out_bf16) {
  __shared__ float red[2][4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int dsteps = K >> 6;                        // 64-k double steps
  const int parts = gridDim.y * 4, part = blockIdx.y * 4 + wave;
  const int base = dsteps / parts, rem = dsteps % parts;
  const int d0 = part * base + (part < rem ? part : rem);
  const int d1 = d0 + base + (part < rem ? 1 : 0);
  const int t0 = 2 * (int)blockIdx.x;
  int t1 = t0 + 1;
  const bool t1_valid = t1 < n_tiles;
  if (!t1_valid) t1 = t0;
  const unsigned char* w0 = Wp + ((long)t0 * dsteps) * 1024 + lane * 16;
  const unsigned char* w1 = Wp + ((long)t1 * dsteps) * 1024 + lane * 16;
  const bool xvalid = r < M;
  const bf16* xp = x + (long)(xvalid ? r : 0) * ldx + q * 8;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  int d = d0;
  for (; d + 4 <= d1; d += 4) {                     // 4 double steps = 8 KiB of weights in flight per wave
    u32x4 a0[4], a1[4];
    bf16x8 b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w0 + (long)(d + i) * 1024));
      a1[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w1 + (long)(d + i) * 1024));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = xvalid ? *reinterpret_cast<const bf16x8*>(xp + (long)(d * 2 + i) * 32) : zero8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 lo, hi;
      dq16(a0[i], lo, hi);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b[2 * i], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b[2 * i + 1], acc0, 0, 0, 0);
      dq16(a1[i], lo, hi);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b[2 * i], acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b[2 * i + 1], acc1, 0, 0, 0);
    }
  }
  for (; d < d1; ++d) {
    const u32x4 a0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w0 + (long)d * 1024));
    const u32x4 a1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w1 + (long)d * 1024));
    const bf16x8 b0 = xvalid ? *reinterpret_cast<const bf16x8*>(xp + (long)(d * 2) * 32) : zero8;
    const bf16x8 b1 = xvalid ? *reinterpret_cast<const bf16x8*>(xp + (long)(d * 2 + 1) * 32) : zero8;
    bf16x8 lo, hi;
    dq16(a0, lo, hi);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b0, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b1, acc0, 0, 0, 0);
    dq16(a1, lo, hi);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b0, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, b1, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][wave][lane][i] = acc0[i]; red[1][wave][lane][i] = acc1[i]; }
  __syncthreads();
  if (wave == 0 && xvalid) {
    float g[4], u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g[i] = red[0][0][lane][i] + red[0][1][lane][i] + red[0][2][lane][i] + red[0][3][lane][i];
      u[i] = red[1][0][lane][i] + red[1][1][lane][i] + red[1][2][lane][i] + red[1][3][lane][i];
    }
    const float4 sa = *reinterpret_cast<const float4*>(scale + t0 * 16 + 4 * q);
    const float4 sb = *reinterpret_cast<const float4*>(scale + t1 * 16 + 4 * q);
    g[0] *= sa.x; g[1] *= sa.y; g[2] *= sa.z; g[3] *= sa.w;
    u[0] *= sb.x; u[1] *= sb.y; u[2] *= sb.z; u[3] *= sb.w;
    const long row = r;
    if (MODE == 0) {
      float* o = out_f32 + (long)blockIdx.y * split_stride + row * ldo;
      const int na = t0 * 16 + 4 * q, nb = t1 * 16 + 4 * q;
      if (na < N) *reinterpret_cast<float4*>(o + na) = make_float4(g[0], g[1], g[2], g[3]);
      if (t1_valid && nb < N) *reinterpret_cast<float4*>(o + nb) = make_float4(u[0], u[1], u[2], u[3]);
    } else {
      const int n = (t0 >> 1) * 16 + 4 * q;
      if (n < N) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2bf((g[i] / (1.0f + expf(-g[i]))) * u[i]);
        *reinterpret_cast<bf16x4*>(out_bf16 + row * ldo + n) = o;
      }
    }
  }
}

// ---- 16 < M <= 128: x chunk (16*MB rows x 256 k) double-buffered in LDS, 4 waves x 2 tiles, one barrier per chunk
template <int MODE, int MB>
__global__ __launch_bounds__(256) void gemv_lds_w8_kernel(const bf16* __restrict__ x, int ldx, int M, const unsigned char* __restrict__ Wp,
                                                          const float* __restrict__ scale, int n_tiles, int N, int K,
                                                          float* __restrict__ out_f32, long split_stride, int ldo,
                                                          bf16* __restrict__ out_bf16) {
  constexpr int KC = 8;                              // k-steps per chunk (4 weight double steps)
  constexpr int ROWB = KC * 64;                      // 512 bytes of x per row and chunk
  constexpr int CHUNKB = MB * 16 * ROWB;
  constexpr int XP = CHUNKB / 1024 / 4;              // 1-KiB pieces (2 rows) staged by each wave
  extern __shared__ __attribute__((aligned(16))) char xs[];   // 2 * CHUNKB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K >> 5;
  const int nchunks = (ksteps + KC - 1) / KC;
  const int parts = gridDim.y, part = blockIdx.y;
  const int pbase = nchunks / parts, prem = nchunks % parts;
  const int c0 = part * pbase + (part < prem ? part : prem);
  const int nch = pbase + (part < prem ? 1 : 0);
  const int ks0 = c0 * KC;
  const int nfull = (ks0 + KC * nch <= ksteps) ? nch : nch - 1;
  const int dsteps = K >> 6;

  int t0 = (int)blockIdx.x * 8 + wave * 2, t1 = t0 + 1;
  const bool t0_valid = t0 < n_tiles, t1_valid = t1 < n_tiles;
  if (!t0_valid) t0 = n_tiles - 1;
  if (!t1_valid) t1 = t0;
  const unsigned char* w0 = Wp + ((long)t0 * dsteps + (ks0 >> 1)) * 1024 + lane * 16;
  const unsigned char* w1 = Wp + ((long)t1 * dsteps + (ks0 >> 1)) * 1024 + lane * 16;

  const bf16* xsrc[XP];
  int xslot[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = wave * XP + i;
    const int row = 2 * p + (lane >> 5), c = (lane & 31) ^ (row & 15);   // 32 slots of 16 B per row, XOR on the low 4 bits
    int m = row; if (m > M - 1) m = M - 1;
    xsrc[i] = x + (long)m * ldx;
    xslot[i] = c * 8;
  }
  auto stage = [&](int c, char* buf) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int k = (ks0 + KC * c) * 32 + xslot[i];
      if (k > K - 8) k = K - 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + k),
                                       (__attribute__((address_space(3))) void*)(buf + (wave * XP + i) * 1024), 16, 0, 0);
    }
  };
  f32x4 acc0[MB], acc1[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc0[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto mma_step = [&](const char* cur, int i, const bf16x8& wa, const bf16x8& wb) {
    bf16x8 b[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      b[mb] = *reinterpret_cast<const bf16x8*>(cur + (mb * 16 + r) * ROWB + (((i * 4 + q) ^ r) << 4));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc0[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, b[mb], acc0[mb], 0, 0, 0);
      acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, b[mb], acc1[mb], 0, 0, 0);
    }
  };
  if (nch > 0) stage(0, xs);
  for (int c = 0; c < nfull; ++c) {
    u32x4 a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w0 + (long)(4 * c + i) * 1024));
      a1[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w1 + (long)(4 * c + i) * 1024));
    }
    // outstanding VMEM in issue order: [x chunk c: XP LDS-DMA loads][8 weight loads just issued] -> vmcnt(8) retires x
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    if (c + 1 < nch) stage(c + 1, xs + ((c + 1) & 1) * CHUNKB);
    const char* cur = xs + (c & 1) * CHUNKB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x8 l0, h0, l1, h1;
      dq16(a0[i], l0, h0);
      dq16(a1[i], l1, h1);
      mma_step(cur, 2 * i, l0, l1);
      mma_step(cur, 2 * i + 1, h0, h1);
    }
  }
  if (nfull < nch) {   // K % 256 != 0: the last chunk has 2, 4 or 6 k-steps (K is a multiple of 64)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const char* cur = xs + (nfull & 1) * CHUNKB;
    for (int i = 0; ks0 + KC * nfull + 2 * i < ksteps; ++i) {
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(w0 + (long)(4 * nfull + i) * 1024);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(w1 + (long)(4 * nfull + i) * 1024);
      bf16x8 l0, h0, l1, h1;
      dq16(a0, l0, h0);
      dq16(a1, l1, h1);
      mma_step(cur, 2 * i, l0, l1);
      mma_step(cur, 2 * i + 1, h0, h1);
    }
  }
  const float4 sa = *reinterpret_cast<const float4*>(scale + t0 * 16 + 4 * q);
  const float4 sb = *reinterpret_cast<const float4*>(scale + t1 * 16 + 4 * q);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long row = mb * 16 + r;
    if (row >= M) continue;
    const float g0 = acc0[mb][0] * sa.x, g1 = acc0[mb][1] * sa.y, g2 = acc0[mb][2] * sa.z, g3 = acc0[mb][3] * sa.w;
    const float u0 = acc1[mb][0] * sb.x, u1 = acc1[mb][1] * sb.y, u2 = acc1[mb][2] * sb.z, u3 = acc1[mb][3] * sb.w;
    if (MODE == 0) {
      float* o = out_f32 + (long)blockIdx.y * split_stride + row * ldo;
      const int na = t0 * 16 + 4 * q, nb = t1 * 16 + 4 * q;
      if (t0_valid && na < N) *reinterpret_cast<float4*>(o + na) = make_float4(g0, g1, g2, g3);
      if (t1_valid && nb < N) *reinterpret_cast<float4*>(o + nb) = make_float4(u0, u1, u2, u3);
    } else {
      const int n = (t0 >> 1) * 16 + 4 * q;
      if (t1_valid && n < N) {
        bf16x4 o = {f2bf((g0 / (1.0f + expf(-g0))) * u0), f2bf((g1 / (1.0f + expf(-g1))) * u1),
                    f2bf((g2 / (1.0f + expf(-g2))) * u2), f2bf((g3 / (1.0f + expf(-g3))) * u3)};
        *reinterpret_cast<bf16x4*>(out_bf16 + row * ldo + n) = o;
      }
    }
  }
}

// 128-k chunks and an optional second k group per workgroup, as gemv.hip's gemv_lds_kernel: used above 64 rows, where the
// activations' L2 -> LDS traffic dominates and short chunks with more waves beat the 256-k variant above.
template <int MODE, int MB, int KS>
__global__ __launch_bounds__(256 * KS) void gemv_lds_w8s_kernel(const bf16* __restrict__ x, int ldx, int M,
                                                               const unsigned char* __restrict__ Wp, const float* __restrict__ scale, int n_tiles, int N, int K,
                                                           float* __restrict__ out_f32, long split_stride, int ldo,
                                                           bf16* __restrict__ out_bf16) {
  constexpr int KC = 4;                              // k-steps per chunk
  constexpr int CHUNKB = MB * 4096;                  // bytes of one x chunk: 16*MB rows x 128 k bf16
  constexpr int XP = MB;                             // 1-KiB pieces (4 rows) staged by each of the 4 waves of a k group
  extern __shared__ __attribute__((aligned(16))) char xs_all[];   // KS * 2 * CHUNKB bytes (dynamic: 128 KiB at MB = 8, KS = 2)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pw = wave & 3, kh = wave >> 2;           // tile pair inside the workgroup, k group
  char* xs = xs_all + kh * 2 * CHUNKB;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K >> 5;
  // grid.y splits K in units of 128-k chunks, the k groups split the part again; only the very last chunk of K may be partial
  const int nchunks = (ksteps + KC - 1) / KC;
  const int parts = gridDim.y, part = blockIdx.y;
  const int pbase = nchunks / parts, prem = nchunks % parts;
  const int pc0 = part * pbase + (part < prem ? part : prem);
  const int pn = pbase + (part < prem ? 1 : 0);
  const int hbase = pn / KS, hrem = pn % KS;
  const int c0 = pc0 + kh * hbase + (kh < hrem ? kh : hrem);
  const int nch = hbase + (kh < hrem ? 1 : 0);       // chunks of this k group
  const int nmax = hbase + (hrem ? 1 : 0);           // barrier count: every wave of the workgroup runs this many rounds
  const int ks0 = c0 * KC;
  const int nfull = (ks0 + KC * nch <= ksteps) ? nch : nch - 1;   // chunks with all 4 k-steps

  int t0 = (int)blockIdx.x * 8 + pw * 2, t1 = t0 + 1;
  const bool t0_valid = t0 < n_tiles, t1_valid = t1 < n_tiles;
  if (!t0_valid) t0 = n_tiles - 1;                   // past the last tile: stream a valid tile, store nothing
  if (!t1_valid) t1 = t0;
  const int dsteps = K >> 6;                         // chunk starts are even k-steps: 2 fp8 double steps per chunk
  const unsigned char* w0 = Wp + ((long)t0 * dsteps + (ks0 >> 1)) * 1024 + lane * 16;
  const unsigned char* w1 = Wp + ((long)t1 * dsteps + (ks0 >> 1)) * 1024 + lane * 16;

  // staging: wave instruction i of this wave fills rows 4p .. 4p+3 of the chunk (p = pw*XP + i); lane -> (row, 16-B slot)
  const bf16* xsrc[XP];
  int xslot[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = pw * XP + i;
    const int row = 4 * p + (lane >> 4), c = (lane & 15) ^ (row & 15);
    int m = row; if (m > M - 1) m = M - 1;           // rows past M: a finite copy of the last row, never stored
    xsrc[i] = x + (long)m * ldx;
    xslot[i] = c * 8;                                // element offset of this lane's 16 bytes inside the 128-k chunk
  }
  auto stage = [&](int c, char* buf) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int k = (ks0 + KC * c) * 32 + xslot[i];
      if (k > K - 8) k = K - 8;                      // partial last chunk: stay inside the row (those k-steps are never multiplied)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + k),
                                       (__attribute__((address_space(3))) void*)(buf + (pw * XP + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc0[MB], acc1[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc0[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  auto mma_step = [&](const char* cur, int i, const bf16x8& wa, const bf16x8& wb) {
    bf16x8 b[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
      b[mb] = *reinterpret_cast<const bf16x8*>(cur + (mb * 16 + r) * 256 + (((i * 4 + q) ^ r) << 4));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc0[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, b[mb], acc0[mb], 0, 0, 0);
      acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, b[mb], acc1[mb], 0, 0, 0);
    }
  };
  if (nch > 0) stage(0, xs);
  for (int c = 0; c < nmax; ++c) {
    const bool active = c < nfull;                   // wave-uniform; a k group with one chunk less idles through the last round
    u32x4 a0[2], a1[2];
    if (active) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a0[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w0 + (long)(2 * c + i) * 1024));
        a1[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w1 + (long)(2 * c + i) * 1024));
      }
    }
    // outstanding VMEM in issue order: [x chunk c: XP LDS-DMA loads][4 weight loads just issued] -> vmcnt(4) retires x
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    if (c + 1 < nch) stage(c + 1, xs + ((c + 1) & 1) * CHUNKB);   // buffer last read for chunk c-1: every wave is past it
    if (active) {
      const char* cur = xs + (c & 1) * CHUNKB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 l0, h0, l1, h1;
        dq16(a0[i], l0, h0);
        dq16(a1[i], l1, h1);
        mma_step(cur, 2 * i, l0, l1);
        mma_step(cur, 2 * i + 1, h0, h1);
      }
    }
  }
  if (K & 127) {       // the last chunk of K has 1..3 k-steps: its owner multiplies it with plain loads after a full drain
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nfull < nch) {
      const char* cur = xs + (nfull & 1) * CHUNKB;
      for (int i = 0; ks0 + KC * nfull + 2 * i < ksteps; ++i) {   // K % 128 == 64: one double step left
        const u32x4 wa = *reinterpret_cast<const u32x4*>(w0 + (long)(2 * nfull + i) * 1024);
        const u32x4 wb = *reinterpret_cast<const u32x4*>(w1 + (long)(2 * nfull + i) * 1024);
        bf16x8 l0, h0, l1, h1;
        dq16(wa, l0, h0);
        dq16(wb, l1, h1);
        mma_step(cur, 2 * i, l0, l1);
        mma_step(cur, 2 * i + 1, h0, h1);
      }
    }
  }
  if constexpr (KS == 2) {   // k group 1 hands its partial sums to k group 0 through the (now idle) x buffers
    float* red = reinterpret_cast<float*>(xs_all);
    __syncthreads();
    if (kh == 1) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          red[((pw * MB * 8) + mb * 8 + i) * 64 + lane] = acc0[mb][i];
          red[((pw * MB * 8) + mb * 8 + 4 + i) * 64 + lane] = acc1[mb][i];
        }
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0[mb][i] += red[((pw * MB * 8) + mb * 8 + i) * 64 + lane];
        acc1[mb][i] += red[((pw * MB * 8) + mb * 8 + 4 + i) * 64 + lane];
      }
  }
  {   // per-channel scales of the two tiles' rows (after the k-group sum: the scale is linear)
    const float4 sa = *reinterpret_cast<const float4*>(scale + t0 * 16 + 4 * q);
    const float4 sb = *reinterpret_cast<const float4*>(scale + t1 * 16 + 4 * q);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc0[mb][0] *= sa.x; acc0[mb][1] *= sa.y; acc0[mb][2] *= sa.z; acc0[mb][3] *= sa.w;
      acc1[mb][0] *= sb.x; acc1[mb][1] *= sb.y; acc1[mb][2] *= sb.z; acc1[mb][3] *= sb.w;
    }
  }
  // D layout (16x16): lane holds col = m = lane&15, rows n = 4*(lane>>4)+i of the tile
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long row = mb * 16 + r;
    if (row >= M) continue;
    if (MODE == 0) {
      float* o = out_f32 + (long)blockIdx.y * split_stride + row * ldo;
      const int na = t0 * 16 + 4 * q, nb = t1 * 16 + 4 * q;
      if (t0_valid && na < N) *reinterpret_cast<float4*>(o + na) = make_float4(acc0[mb][0], acc0[mb][1], acc0[mb][2], acc0[mb][3]);
      if (t1_valid && nb < N) *reinterpret_cast<float4*>(o + nb) = make_float4(acc1[mb][0], acc1[mb][1], acc1[mb][2], acc1[mb][3]);
    } else {
      const int n = (t0 >> 1) * 16 + 4 * q;          // gate tile t0 (even), up tile t0 + 1 -> output column block t0/2
      if (t1_valid && n < N) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float g = acc0[mb][i]; o[i] = f2bf((g / (1.0f + expf(-g))) * acc1[mb][i]); }
        *reinterpret_cast<bf16x4*>(out_bf16 + row * ldo + n) = o;
      }
    }
  }
}

template <int MODE>
void launch_w8(dim3 grid, hipStream_t st, const bf16* x, int ldx, int M, const unsigned char* Wp, const float* scale, int n_tiles,
               int N, int K, float* of, long split_stride, int ldo, bf16* ob) {
  if (M <= 16) {
    hipLaunchKernelGGL((gemv16_w8_kernel<MODE>), grid, dim3(256), 0, st, x, ldx, M, Wp, scale, n_tiles, N, K, of, split_stride, ldo, ob);
    return;
  }
  const dim3 g4((unsigned)((n_tiles + 7) / 8), grid.y);
#define VR2_W8_LAUNCH(MB_)                                                                                                     \
  {                                                                                                                            \
    static bool attr = false;                                                                                                  \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_lds_w8_kernel<MODE, MB_>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MB_ * 8192); attr = true; }         \
    hipLaunchKernelGGL((gemv_lds_w8_kernel<MODE, MB_>), g4, dim3(256), 2 * MB_ * 8192, st, x, ldx, M, Wp, scale, n_tiles, N, K, \
                       of, split_stride, ldo, ob);                                                                             \
  }
  if (M <= 32) VR2_W8_LAUNCH(2)
  else if (M <= 64) VR2_W8_LAUNCH(4)
  else {   // > 64 rows: 128-k chunks, second k group for the split-K projections (as the bf16 kernel)
    const bool two = MODE == 0 && (K / 128) >= 4 * (int)grid.y;
    if (two) {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_lds_w8s_kernel<MODE, 8, 2>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 8 * 4096); attr = true; }
      hipLaunchKernelGGL((gemv_lds_w8s_kernel<MODE, 8, 2>), g4, dim3(512), 2 * 2 * 8 * 4096, st, x, ldx, M, Wp, scale, n_tiles, N, K,
                         of, split_stride, ldo, ob);
    } else {
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_lds_w8s_kernel<MODE, 8, 1>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4096); attr = true; }
      hipLaunchKernelGGL((gemv_lds_w8s_kernel<MODE, 8, 1>), g4, dim3(256), 2 * 8 * 4096, st, x, ldx, M, Wp, scale, n_tiles, N, K,
                         of, split_stride, ldo, ob);
    }
  }
#undef VR2_W8_LAUNCH
}

// ---- fp8 activations x fp8 weights on the fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8), 16 < M <= 256 rows, K % 128 == 0 --------
// BASELINE config 5 ("fp8 weights on CDNA4 fp8 MFMA").  Same one-workgroup-per-CU ring structure as gemv.hip's
// gemv_ring_kernel (read its header comment first); what differs:
//   * the weight tiles are the 16-row x 64-k fp8 tiles above: one 16-B load per lane = the A operands of TWO MFMA k-steps, so a
//     128-k chunk is 4 loads per wave (two tiles x two 64-k steps), the same count as the bf16 kernel's 64-k chunk;
//   * the activations arrive ALREADY quantised (vr2_quant_rows_fp8: e4m3 bytes, per-row scale, and inside every 64-k block
//     the 16-byte slot s holds k = 8s..8s+7 followed by k = 32+8s..32+8s+7, the same two-k-step order as the weight tiles), so
//     one ds_read_b128 per lane is the B operand of the same two k-steps; the LDS image of a chunk is [2 halves of 64 k][16*MB
//     rows][64 B], slot XOR-swizzled by (-(row >> 2)) & 3 through the source address: conflict-free for ds_read_b128's four
//     16-lane groups;
//   * half the activation bytes per k (the L2 -> LDS traffic that is half of a CU's fetch budget at 128 rows) and half the
//     weight bytes; the MFMA runs at the bf16 rate (twice as many per byte), scales (weight channel x activation row) multiply
//     the fp32 accumulator in the epilogue.
typedef __attribute__((ext_vector_type(4))) float wreg8;
typedef __attribute__((ext_vector_type(2))) long l64x2;

template <int MODE, int MB, int D>   // MB > 8 (129-256 rows): one workgroup per CU with the whole register file, as gemv.hip's ring kernel
__global__ __launch_bounds__(512, (MB > 8 ? 1 : 2)) void gemv_ring8_kernel(const unsigned char* __restrict__ x8, int ldx8,
                                                            const float* __restrict__ xscale, int M,
                                                            const unsigned char* __restrict__ Wp, const float* __restrict__ wscale,
                                                            int n_tiles, int N, int K, float* __restrict__ out_f32,
                                                            long split_stride, int ldo, bf16* __restrict__ out_bf16) {
  constexpr int NB = D + 1;
  static_assert(MB >= 2 && MB <= 16 && (NB & (NB - 1)) == 0, "ring shape (lgkmcnt counts <= 15 fragment reads)");
  constexpr int CHUNKB = MB * 2048;                  // 16*MB rows x 128 k fp8
  constexpr int PIECES = MB * 2;                     // 1-KiB pieces (16 rows x 64 B) of an x chunk: [half][16-row block]
  constexpr int XP = (PIECES + 7) / 8;               // (MB = 2, 10, 14: some waves repeat a piece)
  extern __shared__ __attribute__((aligned(16))) char xring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int dsteps = K >> 6, nchunks = K >> 7;
  const int parts = gridDim.y, part = blockIdx.y;
  const int pbase = nchunks / parts, prem = nchunks % parts;
  const int pc0 = part * pbase + (part < prem ? part : prem);
  const int n = pbase + (part < prem ? 1 : 0);
  const int pairs = (n_tiles + 1) >> 1, groups = gridDim.x, g = blockIdx.x;
  const int gbase = pairs / groups, grem = pairs % groups;
  const int first = g * gbase + (g < grem ? g : grem), count = gbase + (g < grem ? 1 : 0);
  const bool active = wave < count;
  const int t0 = 2 * (first + (active ? wave : 0)), t1 = (t0 + 1 < n_tiles) ? t0 + 1 : t0;
  const bool t1_valid = t0 + 1 < n_tiles;
  const char* w0 = (const char*)(Wp + ((long)t0 * dsteps + 2 * pc0) * 1024);   // + 2 KiB per 128-k chunk
  const char* w1 = (const char*)(Wp + ((long)t1 * dsteps + 2 * pc0) * 1024);
  const unsigned wlane = lane * 16;
  const char* xb = (const char*)x8 + (long)pc0 * 128;                         // + 128 B per chunk
  unsigned xoff[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = (wave * XP + i) % PIECES, half = p / MB, blk = p % MB;
    const int row = blk * 16 + (lane >> 2), sl = lane & 3;
    const int m = row < M ? row : M - 1;
    xoff[i] = (unsigned)(m * ldx8 + half * 64 + ((sl ^ ((-(row >> 2)) & 3)) << 4));
  }
  int staged = 0, loaded = 0;
  auto stage = [&]() {
    char* dst = xring + (staged & (NB - 1)) * CHUNKB;
#pragma unroll
    for (int i = 0; i < XP; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + xoff[i]),
                                       (__attribute__((address_space(3))) void*)(dst + ((wave * XP + i) % PIECES) * 1024), 16, 0, 0);
    ++staged;
    xb += staged < n ? 128 : 0;
  };
  const unsigned ring_u32 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)xring;
  const unsigned frag = ring_u32 + r * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4);

  wreg8 acc0[MB], acc1[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc0[mb] = wreg8{0.f, 0.f, 0.f, 0.f}; acc1[mb] = wreg8{0.f, 0.f, 0.f, 0.f}; }
  l64x2 wa[NB][2], wb[NB][2];
#define VR2_LOADW8(S_)                                                                                                  \
  do {                                                                                                                  \
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(wa[S_][0]) : "v"(wlane), "s"(w0) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(wb[S_][0]) : "v"(wlane), "s"(w1) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=v"(wa[S_][1]) : "v"(wlane), "s"(w0) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=v"(wb[S_][1]) : "v"(wlane), "s"(w1) : "memory");   \
    ++loaded;                                                                                                           \
    { const int adv = loaded < n ? 2048 : 0; w0 += adv; w1 += adv; }                                                    \
  } while (0)

  auto kloop = [&](auto act) {
    constexpr bool ACTIVE = decltype(act)::value;
    constexpr int WL = ACTIVE ? 4 : 0;
    constexpr int X_YOUNGER = D * WL + (D - 1) * XP;
    constexpr int W_YOUNGER = D * WL + (D + 1) * XP;
    static_assert(W_YOUNGER <= 63, "vmcnt is 6 bits");
    auto& A0 = acc0; auto& A1 = acc1; auto& WA = wa; auto& WB = wb;
    auto chunk = [&](int cc, auto jc, auto steady) {
      constexpr int j = decltype(jc)::value;
      constexpr bool STEADY = decltype(steady)::value;
      auto& C0 = A0; auto& C1 = A1; auto& CA = WA; auto& CB = WB;
      if constexpr (STEADY && ACTIVE) VR2_LOADW8((j + D) % NB);
      if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(X_YOUNGER) : "memory");
      else asm volatile("s_barrier" ::: "memory");
      if constexpr (STEADY) stage();
      if constexpr (ACTIVE) {
        const unsigned a = frag + (unsigned)(cc & (NB - 1)) * CHUNKB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                // the two 64-k halves of the chunk
          l64x2 b[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[mb]) : "v"(a), "n"((i * MB + mb) * 1024) : "memory");
          const long a0lo = CA[j][i][0], a0hi = CA[j][i][1], a1lo = CB[j][i][0], a1hi = CB[j][i][1];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            if (i == 0 && mb == 0 && STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_YOUNGER) : "memory");
            const long blo = b[mb][0], bhi = b[mb][1];
            asm volatile("s_waitcnt lgkmcnt(%8)\n\t"
                         "v_mfma_f32_16x16x32_fp8_fp8 %0, %2, %6, %0\n\tv_mfma_f32_16x16x32_fp8_fp8 %1, %4, %6, %1\n\t"
                         "v_mfma_f32_16x16x32_fp8_fp8 %0, %3, %7, %0\n\tv_mfma_f32_16x16x32_fp8_fp8 %1, %5, %7, %1"
                         : "+v"(C0[mb]), "+v"(C1[mb])
                         : "v"(a0lo), "v"(a0hi), "v"(a1lo), "v"(a1hi), "v"(blo), "v"(bhi), "n"(MB - 1 - mb));
          }
        }
      }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if constexpr (ACTIVE) VR2_LOADW8(j);
      stage();
    }
    int cb = 0;
    for (; cb < n - D; cb += NB) {
      [&]<int... J>(std::integer_sequence<int, J...>) {
        (chunk(cb + J, std::integral_constant<int, J>{}, std::true_type{}), ...);
      }(std::make_integer_sequence<int, NB>{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ((cb + J < n ? (chunk(cb + J, std::integral_constant<int, J>{}, std::false_type{}),
                      (void)({ asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); 0; }))
                   : (void)0), ...);
    }(std::make_integer_sequence<int, D>{});
  };
  if (active) kloop(std::true_type{});
  else { kloop(std::false_type{}); return; }
#undef VR2_LOADW8
  // D layout (16x16): lane holds col = m = lane&15, rows n = 4*(lane>>4)+i of the tile; scale = weight channel x activation row
  const float4 sa = *reinterpret_cast<const float4*>(wscale + t0 * 16 + 4 * q);
  const float4 sb = *reinterpret_cast<const float4*>(wscale + t1 * 16 + 4 * q);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long row = mb * 16 + r;
    if (row >= M) continue;
    const float xs = xscale[row];
    const float g0 = acc0[mb][0] * sa.x * xs, g1 = acc0[mb][1] * sa.y * xs, g2 = acc0[mb][2] * sa.z * xs, g3 = acc0[mb][3] * sa.w * xs;
    const float u0 = acc1[mb][0] * sb.x * xs, u1 = acc1[mb][1] * sb.y * xs, u2 = acc1[mb][2] * sb.z * xs, u3 = acc1[mb][3] * sb.w * xs;
    if (MODE == 0) {
      float* o = out_f32 + (long)blockIdx.y * split_stride + row * ldo;
      const int na = t0 * 16 + 4 * q, nb = t1 * 16 + 4 * q;
      if (na < N) *reinterpret_cast<float4*>(o + na) = make_float4(g0, g1, g2, g3);
      if (t1_valid && nb < N) *reinterpret_cast<float4*>(o + nb) = make_float4(u0, u1, u2, u3);
    } else {
      const int nn = (t0 >> 1) * 16 + 4 * q;
      if (t1_valid && nn < N) {
        bf16x4 o = {f2bf(g0 / (1.0f + expf(-g0)) * u0), f2bf(g1 / (1.0f + expf(-g1)) * u1), f2bf(g2 / (1.0f + expf(-g2)) * u2),
                    f2bf(g3 / (1.0f + expf(-g3)) * u3)};
        *reinterpret_cast<bf16x4*>(out_bf16 + row * ldo + nn) = o;
      }
    }
  }
}

// x bf16 [M][K] -> e4m3 bytes [M][Kpad] (Kpad = K rounded up to 64, zero padded) + scale[m] = amax / 448 (1 for an all-zero row):
// x ~= float(q) * scale.  Inside every 64-k block the 16-byte slot s holds k = 8s..8s+7 then k = 32+8s..32+8s+7 (see above).
template <int NS>   // NS = 16-byte output slots per thread kept in registers (K <= 256 * 16 * NS); the row is read ONCE
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16* __restrict__ x, long ldx, int K, unsigned char* __restrict__ q8,
                                                             long ldq, float* __restrict__ scale) {
  __shared__ float scratch[4];
  const int row = blockIdx.x;
  const bf16* xr = x + (long)row * ldx;
  const int slots = ((K + 63) >> 6) * 4;            // 16-byte output slots of the row
  bf16x8 v[NS][2];
  float am = 0.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int s = threadIdx.x + j * 256;
    const int k0 = (s >> 2) * 64 + (s & 3) * 8;     // slot s: k0 .. k0+7 and k0+32 .. k0+39
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + 32 * h;
      if (s < slots && k + 8 <= K) v[j][h] = *reinterpret_cast<const bf16x8*>(xr + k);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[j][h][i] = (s < slots && k + i < K) ? xr[k + i] : f2bf(0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) am = fmaxf(am, fabsf(bf2f(v[j][h][i])));
    }
  }
  am = block_max_256(am, scratch);
  const float inv = am > 0.f ? 448.0f / am : 1.0f;
  if (threadIdx.x == 0) scale[row] = am > 0.f ? am / 448.0f : 1.0f;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int s = threadIdx.x + j * 256;
    if (s >= slots) continue;
    unsigned o[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int lo = 0, hi = 0;                            // v_cvt_pk_fp8_f32: round to nearest even, = torch's float8_e4m3fn cast
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[j][h][0]) * inv, bf2f(v[j][h][1]) * inv, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[j][h][2]) * inv, bf2f(v[j][h][3]) * inv, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[j][h][4]) * inv, bf2f(v[j][h][5]) * inv, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[j][h][6]) * inv, bf2f(v[j][h][7]) * inv, hi, true);
      o[2 * h] = (unsigned)lo; o[2 * h + 1] = (unsigned)hi;
    }
    *reinterpret_cast<uint4*>(q8 + (long)row * ldq + (long)s * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

int ring8_groups(int pairs, int parts) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    ncu = prop.multiProcessorCount;
  }
  const long units = (long)pairs * parts;
  const int rounds = (int)((units + 8L * ncu - 1) / (8L * ncu));
  int groups = (ncu * rounds) / parts;
  if (groups < 1) groups = 1;
  if (groups > pairs) groups = pairs;
  while ((pairs + groups - 1) / groups > 8) ++groups;
  return groups;
}

template <int MODE>
int launch_a8w8(const void* x8, int64_t ldx8, const void* xscale, int64_t M, const void* W8, const void* wscale, int n_tiles, int N,
                int64_t K, float* of, long split_stride, int ldo, bf16* ob, int nsplit, void* stream) {
  const int groups = ring8_groups((n_tiles + 1) / 2, nsplit);
  if (groups <= 0) { vr2_set_error("vr2_gemv_a8w8: no device"); return -2; }
  const dim3 gr((unsigned)groups, (unsigned)nsplit);
#define VR2_RING8(MB_)                                                                                                          \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_ring8_kernel<MODE, MB_, 1>),                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MB_ * 2048); attr = true; }          \
    hipLaunchKernelGGL((gemv_ring8_kernel<MODE, MB_, 1>), gr, dim3(512), 2 * MB_ * 2048, (hipStream_t)stream,                   \
                       (const unsigned char*)x8, (int)ldx8, (const float*)xscale, (int)M, (const unsigned char*)W8,             \
                       (const float*)wscale, n_tiles, N, (int)K, of, split_stride, ldo, ob);                                    \
  } while (0)
  if (M <= 32) VR2_RING8(2); else if (M <= 64) VR2_RING8(4); else if (M <= 128) VR2_RING8(8);
  else if (M <= 160) VR2_RING8(10); else if (M <= 192) VR2_RING8(12); else if (M <= 224) VR2_RING8(14); else VR2_RING8(16);
#undef VR2_RING8
  return 0;
}

}  // namespace

extern "C" int vr2_quant_rows_fp8(const void* x_bf16, int64_t ldx, int64_t M, int64_t K, void* q_fp8, int64_t ldq, void* scale_f32,
                                  void* stream) {
  VR2_REQUIRE(M >= 1 && K >= 1 && ldx % 8 == 0 && ldq % 16 == 0 && ldq >= (K + 63) / 64 * 64, "vr2_quant_rows_fp8: bad shape");
  const long slots = (K + 63) / 64 * 4;
  VR2_REQUIRE(slots <= 256 * 8, "vr2_quant_rows_fp8: K=%ld above 32768", (long)K);
#define VR2_QUANT(NS_)                                                                                                           \
  hipLaunchKernelGGL(quant_rows_fp8_kernel<NS_>, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, (const bf16*)x_bf16, (long)ldx, \
                     (int)K, (unsigned char*)q_fp8, (long)ldq, (float*)scale_f32)
  if (slots <= 256) VR2_QUANT(1); else if (slots <= 512) VR2_QUANT(2); else if (slots <= 1280) VR2_QUANT(5); else VR2_QUANT(8);
#undef VR2_QUANT
  return vr2_check_launch("vr2_quant_rows_fp8");
}

extern "C" int vr2_gemv_a8w8(const void* x_fp8, int64_t ldx8, const void* xscale_f32, int64_t M, const void* Wpacked_fp8,
                             const void* wscale_f32, int64_t N, int64_t K, void* out_f32, int64_t split_stride, int64_t ldo,
                             int nsplit, void* stream) {
  VR2_REQUIRE(M > 16 && M <= 256, "vr2_gemv_a8w8: M=%ld must be in (16,256]", (long)M);
  VR2_REQUIRE(K % 128 == 0 && N % 4 == 0 && ldx8 % 16 == 0 && ldx8 >= K && ldo % 4 == 0, "vr2_gemv_a8w8: K=%ld %% 128, N=%ld %% 4, strides", (long)K, (long)N);
  VR2_REQUIRE(nsplit >= 1 && (K / 128) >= (int64_t)nsplit, "vr2_gemv_a8w8: nsplit=%d too large for K=%ld", nsplit, (long)K);
  const int n_tiles = (int)((N + 15) / 16);
  if (launch_a8w8<0>(x_fp8, ldx8, xscale_f32, M, Wpacked_fp8, wscale_f32, n_tiles, (int)N, K, (float*)out_f32, (long)split_stride,
                     (int)ldo, (bf16*)nullptr, nsplit, stream)) return -2;
  return vr2_check_launch("vr2_gemv_a8w8");
}

extern "C" int vr2_gemv_gateup_silu_a8w8(const void* x_fp8, int64_t ldx8, const void* xscale_f32, int64_t M, const void* Wpacked_fp8,
                                         const void* wscale_f32, int64_t I, int64_t K, void* act_bf16, int64_t ldo, void* stream) {
  VR2_REQUIRE(M > 16 && M <= 256, "vr2_gemv_gateup_silu_a8w8: M=%ld must be in (16,256]", (long)M);
  VR2_REQUIRE(K % 128 == 0 && I % 16 == 0 && ldx8 % 16 == 0 && ldx8 >= K && ldo % 4 == 0, "vr2_gemv_gateup_silu_a8w8: bad shape");
  const int it = (int)(I / 16);
  if (launch_a8w8<1>(x_fp8, ldx8, xscale_f32, M, Wpacked_fp8, wscale_f32, 2 * it, (int)I, K, (float*)nullptr, 0L, (int)ldo,
                     (bf16*)act_bf16, 1, stream)) return -2;
  return vr2_check_launch("vr2_gemv_gateup_silu_a8w8");
}

extern "C" int vr2_gemv_w8(const void* x, int64_t ldx, int64_t M, const void* Wpacked_fp8, const void* scale_f32, int64_t N,
                           int64_t K, void* out_f32, int64_t split_stride, int64_t ldo, int nsplit, void* stream) {
  VR2_REQUIRE(M >= 1 && M <= 128, "vr2_gemv_w8: M=%ld must be in [1,128]", (long)M);
  VR2_REQUIRE(K % 64 == 0 && N % 4 == 0, "vr2_gemv_w8: K=%ld %% 64 and N=%ld %% 4 must be 0", (long)K, (long)N);
  VR2_REQUIRE(ldx % 8 == 0 && ldo % 4 == 0, "vr2_gemv_w8: strides must keep 16-byte alignment");
  VR2_REQUIRE(nsplit >= 1 && (K / 64) >= (int64_t)nsplit * 4, "vr2_gemv_w8: nsplit=%d too large for K=%ld", nsplit, (long)K);
  const int n_tiles = (int)((N + 15) / 16);
  dim3 grid((unsigned)((n_tiles + 1) / 2), (unsigned)nsplit);
  launch_w8<0>(grid, (hipStream_t)stream, (const bf16*)x, (int)ldx, (int)M, (const unsigned char*)Wpacked_fp8, (const float*)scale_f32,
               n_tiles, (int)N, (int)K, (float*)out_f32, (long)split_stride, (int)ldo, (bf16*)nullptr);
  return vr2_check_launch("vr2_gemv_w8");
}

extern "C" int vr2_gemv_gateup_silu_w8(const void* x, int64_t ldx, int64_t M, const void* Wpacked_fp8, const void* scale_f32,
                                       int64_t I, int64_t K, void* act_bf16, int64_t ldo, void* stream) {
  VR2_REQUIRE(M >= 1 && M <= 128, "vr2_gemv_gateup_silu_w8: M=%ld must be in [1,128]", (long)M);
  VR2_REQUIRE(K % 64 == 0 && K >= 256 && I % 16 == 0, "vr2_gemv_gateup_silu_w8: K=%ld %% 64 (>=256) and I=%ld %% 16 must be 0", (long)K, (long)I);
  VR2_REQUIRE(ldx % 8 == 0 && ldo % 4 == 0, "vr2_gemv_gateup_silu_w8: strides must keep alignment");
  const int it = (int)(I / 16);
  dim3 grid((unsigned)it, 1);
  launch_w8<1>(grid, (hipStream_t)stream, (const bf16*)x, (int)ldx, (int)M, (const unsigned char*)Wpacked_fp8, (const float*)scale_f32,
               2 * it, (int)I, (int)K, (float*)nullptr, 0L, (int)ldo, (bf16*)act_bf16);
  return vr2_check_launch("vr2_gemv_gateup_silu_w8");
}
