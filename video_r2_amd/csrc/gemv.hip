// Decode-time weight-streaming "GEMV" for M <= 128 activation rows (the G completions of up to 16 prompt groups): a direct
// kernel for M <= 16 (below) and an LDS-staged one for 16 < M <= 128 (second half of this file).
//
//   out[m][n] = sum_k x[m][k] * W[n][k]        x: [M][K] bf16 row-major;  W: TILE-PACKED bf16 (weights.py: pack_tiles)
//
// HBM-bound: every weight byte is read exactly once per decode step (~14.1 GB/step @7B).
// Weight layout in HBM: 16-row x 32-k tiles of 1 KiB, tile (nt, kt) at ((nt * K/32) + kt) * 1 KiB, and inside a tile
// lane l = 16*(k%32/8) + n%16 owns the 16 bytes W[n][8*(k/8) .. +7] -- exactly the v_mfma_f32_16x16x32_bf16 A
// fragment.  One wave-level global_load_dwordx4 therefore fetches ONE CONTIGUOUS KiB that is already the MFMA operand:
// no LDS round trip, no cross-lane reduction, almost no VALU.  Measured on MI355X at the 7B gate/up shape
// (profiles/src/gemv_bench.hip): row-major fragment-shaped loads (16 rows x 64 B per instruction) 4.0 TB/s, packed tiles
// 5.6 TB/s, packed + two weight tiles per activation fragment 6.3 TB/s (plain streaming read: 6.9 TB/s).
// A wave streams TWO weight tiles per k-step against one activation (B) fragment; the 4 waves of a workgroup split K
// and reduce through LDS; an optional grid-level split-K writes separate fp32 partial slabs which the NEXT kernel
// (add_rmsnorm / rope_append) sums at its launch boundary -- no atomics, bitwise reproducible.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

// DEPTH = k-steps whose loads are issued before the first MFMA of the batch (4 for MB <= 2; 2 for MB = 4 to stay under
// 128 VGPRs -- the activation fragments cost MB * DEPTH * 4 registers)

// MODE 0: fp32 partial store of tiles (2p, 2p+1); MODE 1: gate tile 2p, up tile 2p+1 -> silu(g)*u bf16.
// MB = number of 16-row activation blocks (M <= 16*MB): every weight fragment is used for MB MFMAs, so 32 sequences cost the
// same HBM traffic as 8.
template <int MODE, int MB, int DEPTH>
__global__ __launch_bounds__(256) void gemv16_kernel(const bf16* __restrict__ x, int ldx, int M,
                                                     const bf16* __restrict__ Wp, int n_tiles, int N, int K,
                                                     float* __restrict__ out_f32, long split_stride, int ldo,
                                                     bf16* __restrict__ out_bf16) {
  __shared__ float red[MB][2][4][64][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15;  // B column = activation row within its 16-row block ; D column
  const int q = lane >> 4;
  const int ksteps = K >> 5;
  const int parts = gridDim.y * 4;
  const int part = blockIdx.y * 4 + wave;
  const int base = ksteps / parts, rem = ksteps % parts;
  const int ks0 = part * base + (part < rem ? part : rem);
  const int ks1 = ks0 + base + (part < rem ? 1 : 0);

  const int t0 = 2 * (int)blockIdx.x;
  int t1 = t0 + 1;
  const bool t1_valid = t1 < n_tiles;
  if (!t1_valid) t1 = t0;  // odd tile count: stream the first tile twice, store once
  const bf16* w0 = Wp + ((long)t0 * ksteps) * 512 + lane * 8;
  const bf16* w1 = Wp + ((long)t1 * ksteps) * 512 + lane * 8;
  bool xvalid[MB];
  const bf16* xp[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    xvalid[mb] = (mb * 16 + r) < M;
    xp[mb] = x + (long)(xvalid[mb] ? mb * 16 + r : 0) * ldx + q * 8;
  }

  f32x4 acc0[MB], acc1[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc0[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  int ks = ks0;
  for (; ks + DEPTH <= ks1; ks += DEPTH) {
    bf16x8 a0[DEPTH], a1[DEPTH], b[MB][DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      a0[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w0 + (long)(ks + i) * 512));
      a1[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + (long)(ks + i) * 512));
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) b[mb][i] = xvalid[mb] ? *reinterpret_cast<const bf16x8*>(xp[mb] + (ks + i) * 32) : zero8;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc0[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b[mb][i], acc0[mb], 0, 0, 0);
        acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b[mb][i], acc1[mb], 0, 0, 0);
      }
  }
  for (; ks < ks1; ++ks) {
    const bf16x8 a0 = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w0 + (long)ks * 512));
    const bf16x8 a1 = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + (long)ks * 512));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const bf16x8 b = xvalid[mb] ? *reinterpret_cast<const bf16x8*>(xp[mb] + ks * 32) : zero8;
      acc0[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0[mb], 0, 0, 0);
      acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1[mb], 0, 0, 0);
    }
  }
  // cross-wave (K-slice) reduction through LDS
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[mb][0][wave][lane][i] = acc0[mb][i];
      red[mb][1][wave][lane][i] = acc1[mb][i];
    }
  __syncthreads();
  if (wave < MB && xvalid[wave]) {   // wave mb finishes activation block mb
    const int mb = wave;
    // D layout (16x16): lane holds col = m = lane&15, rows n = 4*(lane>>4)+i of the tile
    float g[4], u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g[i] = red[mb][0][0][lane][i] + red[mb][0][1][lane][i] + red[mb][0][2][lane][i] + red[mb][0][3][lane][i];
      u[i] = red[mb][1][0][lane][i] + red[mb][1][1][lane][i] + red[mb][1][2][lane][i] + red[mb][1][3][lane][i];
    }
    const long row = mb * 16 + r;
    if (MODE == 0) {
      float* o = out_f32 + (long)blockIdx.y * split_stride + row * ldo;
      const int na = t0 * 16 + 4 * q, nb = t1 * 16 + 4 * q;
      if (na < N) *reinterpret_cast<float4*>(o + na) = make_float4(g[0], g[1], g[2], g[3]);  // N % 4 == 0 (host)
      if (t1_valid && nb < N) *reinterpret_cast<float4*>(o + nb) = make_float4(u[0], u[1], u[2], u[3]);
    } else {
      const int n = (t0 >> 1) * 16 + 4 * q;   // output column: tile pair p = t0/2
      if (n < N) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2bf((g[i] / (1.0f + expf(-g[i]))) * u[i]);
        *reinterpret_cast<bf16x4*>(out_bf16 + row * ldo + n) = o;
      }
    }
  }
}

// ---- M > 16: activation rows staged through LDS ------------------------------------------------------------------------
// With more than one 16-row activation block the direct kernel above re-reads MB KiB of x from L2 for every 2 KiB of weights
// and turns L2-bound (measured at the 7B gate/up shape: 43 us @ M=8 -> 87 us @ M=64).  Here 4 waves of a workgroup own
// DIFFERENT weight tiles (2 each: 128 output rows per workgroup) over the SAME k range, so one copy of the x chunk
// (16*MB rows x 128 k) in LDS feeds all of them: L2 traffic for x drops to M/128 of the weight bytes, and the weights still
// go HBM -> VGPR as ready-made MFMA operands.  The x chunk is double-buffered with direct-to-LDS loads (XOR swizzle applied
// to the per-lane source address so the B-fragment ds_read_b128 of 16 rows at one k slot is conflict-free); one s_barrier
// per chunk with a counted vmcnt for the x chunk, the 8 weight loads of a chunk issued before that barrier, consumed after.
// A wave is latency-bound (8 KiB in flight per ~2 us round trip) and 128 rows per workgroup leave only ~1 workgroup per CU
// at the gate/up shape, so with KS = 2 a second group of 4 waves takes the other half of the workgroup's k range (own x
// chunks) and the halves are summed through LDS at the end: twice the waves in flight, same L2 traffic, same output.
// (Tried and dropped: 256-k chunks -- slower; 8 waves x 2 tiles = 256 output rows per workgroup (half the x traffic, half the
// workgroups) -- slower, 66 vs 60.6 us at 64 rows; x staged two chunks ahead -- no change; the 128-row GEMM tile with its
// direct-to-LDS weight prefetch on the same shape -- 64-75 us, no better; a two-register-set software prefetch of the weights -- it needs inline-asm loads
// because hipcc drains vmcnt(0) for register loads it sees next to an LDS-DMA, and hipcc then copies the still-in-flight asm
// destinations at control-flow joins.)
template <int MODE, int MB, int KS>
__global__ __launch_bounds__(256 * KS) void gemv_lds_kernel(const bf16* __restrict__ x, int ldx, int M,
                                                           const bf16* __restrict__ Wp, int n_tiles, int N, int K,
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
  const bf16* w0 = Wp + ((long)t0 * ksteps + ks0) * 512 + lane * 8;
  const bf16* w1 = Wp + ((long)t1 * ksteps + ks0) * 512 + lane * 8;

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
    bf16x8 a0[KC], a1[KC];
    if (active) {
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        a0[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w0 + (long)(KC * c + i) * 512));
        a1[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + (long)(KC * c + i) * 512));
      }
    }
    // outstanding VMEM in issue order: [x chunk c: XP LDS-DMA loads][8 weight loads just issued] -> vmcnt(8) retires x
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    if (c + 1 < nch) stage(c + 1, xs + ((c + 1) & 1) * CHUNKB);   // buffer last read for chunk c-1: every wave is past it
    if (active) {
      const char* cur = xs + (c & 1) * CHUNKB;
#pragma unroll
      for (int i = 0; i < KC; ++i) mma_step(cur, i, a0[i], a1[i]);
    }
  }
  if (K & 127) {       // the last chunk of K has 1..3 k-steps: its owner multiplies it with plain loads after a full drain
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nfull < nch) {
      const char* cur = xs + (nfull & 1) * CHUNKB;
      for (int i = 0; ks0 + KC * nfull + i < ksteps; ++i) {
        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w0 + (long)(KC * nfull + i) * 512);
        const bf16x8 wb = *reinterpret_cast<const bf16x8*>(w1 + (long)(KC * nfull + i) * 512);
        mma_step(cur, i, wa, wb);
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

// ---- 16 < M <= 128, K % 64 == 0: one workgroup per CU, weights ring-buffered in REGISTERS, activations in an LDS ring ------
// What round 1's staged kernel (above) leaves on the table at 64-128 rows, measured on MI355X at the 7B gate/up shape
// (profiles/r02_gemv_ring.md):
//   * it issues a chunk's 8 weight loads, waits for them, multiplies, and only then issues the next 8 -- one HBM round trip
//     per 128 k.  It cannot do better from C++: a global_load_lds in the instruction stream makes hipcc's wait-count pass
//     turn EVERY later vmcnt / lgkmcnt wait into vmcnt(0) lgkmcnt(0) (the instruction carries an LDS memory operand on a FLAT
//     encoding, filed under "may return out of order"), so a register prefetch is drained before the first MFMA;
//   * a CU fetches at most ~40 GB/s in this kernel (weights from HBM + the activation block from L2, half each at 128 rows),
//     and its 296 workgroups of 128 output columns land one per CU on 216 CUs and TWO on the other 40: those take twice as
//     long and set the kernel time (256 workgroups: 46 us; 257: 72 us).
// Here
//   * the grid is a whole number of workgroups per CU: the weight-tile PAIRS of a launch (gate + up tile of 16 columns, or two
//     neighbouring tiles) are dealt to G = #CUs x rounds / parts groups, 4-5 pairs each at the 7B gate/up shape; a workgroup
//     has 8 waves, wave w < count streams pair w (two 1-KiB tiles per 32 k), the others only help staging;
//   * a wave keeps D chunks (64 k = 4 KiB of weights) in flight in D + 1 register sets that rotate through a loop unrolled
//     D + 1 times; the activation chunk (16*MB rows x 64 k) lives in a D + 1 slot LDS ring filled D chunks ahead by
//     direct-to-LDS loads (XOR swizzle through the source address, conflict-free ds_read_b128 fragments);
//   * every memory instruction and wait of the main loop is inline asm: ONE counted vmcnt for the x chunk (+ s_barrier) and
//     ONE for the weight set per chunk (vmcnt counts loads and LDS-DMA together, in issue order), counted lgkmcnt per fragment.
//     The MFMAs are asm too (hipcc hoists bare wait statements over MFMA builtins); gfx950 does not interlock an MFMA result
//     against a following VALU read, hence the s_nops before compiler-generated code touches the accumulators;
//   * the main loop is branch-free (hipcc may copy registers at control-flow joins; a copy of an in-flight load destination
//     reads stale data), so up to D of its load batches name chunks past the end of the part: they re-read the last chunk.
typedef __attribute__((ext_vector_type(4))) float wreg;    // 16 B of bf16x8 weight / activation fragment in 4 VGPRs

// MB > 8 (129-256 rows, round 4): 2 x MB accumulator sets + MB activation fragments per wave do not fit the 128 registers of
// two workgroups per CU; the grid is one workgroup per CU anyway (ring_groups), so those instantiations ask for one.
template <int MODE, int MB, int D, int ABL = 0>   // ABL: ablations for profiles/ (1 no x staging, 2 no LDS reads / MFMA, 3 no weight stream, 4 = 1 + 3)
__global__ __launch_bounds__(512, (MB > 8 ? 1 : 2)) void gemv_ring_kernel(const bf16* __restrict__ x, int ldx, int xadv, int M,
                                                           const bf16* __restrict__ Wp, int n_tiles, int N, int K,
                                                           float* __restrict__ out_f32, long split_stride, int ldo,
                                                           bf16* __restrict__ out_bf16) {
  constexpr int NB = D + 1;                          // LDS ring slots == weight register sets
  static_assert(MB >= 2 && MB <= 16 && (NB & (NB - 1)) == 0, "ring shape (lgkmcnt counts <= 15 fragment reads)");
  constexpr int CHUNKB = MB * 2048;                  // 16*MB rows x 64 k bf16
  constexpr int PIECES = MB * 2;                     // 1-KiB pieces (8 rows x 128 B) of an x chunk
  constexpr int XP = (PIECES + 7) / 8;               // pieces staged per wave and chunk (MB = 2, 10, 14: some waves repeat a piece)
  extern __shared__ __attribute__((aligned(16))) char xring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K >> 5, nchunks = K >> 6;
  const int parts = gridDim.y, part = blockIdx.y;
  const int pbase = nchunks / parts, prem = nchunks % parts;
  const int pc0 = part * pbase + (part < prem ? part : prem);
  const int n = pbase + (part < prem ? 1 : 0);       // chunks of this part (>= D: host), the same for every wave
  // this workgroup's weight-tile pairs; wave w streams pair first + w
  const int pairs = (n_tiles + 1) >> 1, groups = gridDim.x, g = blockIdx.x;
  const int gbase = pairs / groups, grem = pairs % groups;
  const int first = g * gbase + (g < grem ? g : grem), count = gbase + (g < grem ? 1 : 0);   // count <= 8 (host)
  const bool active = wave < count;
  const int t0 = 2 * (first + (active ? wave : 0)), t1 = (t0 + 1 < n_tiles) ? t0 + 1 : t0;   // odd tile count: last pair streams its tile twice
  const bool t1_valid = t0 + 1 < n_tiles;
  // addresses = wave-uniform 64-bit base (SGPR pair, advanced per chunk) + a per-lane 32-bit offset that never changes
  const char* w0 = (const char*)(Wp + ((long)t0 * ksteps + 2 * pc0) * 512);   // + 2 KiB per chunk
  const char* w1 = (const char*)(Wp + ((long)t1 * ksteps + 2 * pc0) * 512);
  const unsigned wlane = lane * 16;
  const char* xb = (const char*)x + (long)pc0 * xadv;                           // + xadv bytes per chunk (128 row-major)
  unsigned xoff[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = ((wave * XP + i) % PIECES) * 8 + (lane >> 3), c = lane & 7;
    const int m = row < M ? row : M - 1;             // rows past M: a finite copy of the last row, never stored
    xoff[i] = (unsigned)(m * ldx + ((c ^ (row & 7)) << 3)) * 2u;               // XOR swizzle through the source address
  }
  int staged = 0, loaded = 0;                        // chunks handed to the LDS ring / to the register sets so far
  auto stage = [&]() {
    char* dst = xring + (staged & (NB - 1)) * CHUNKB;
#pragma unroll
    for (int i = 0; i < XP; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + ((ABL == 1 || ABL == 4) ? 0u : xoff[i])),
                                       (__attribute__((address_space(3))) void*)(dst + ((wave * XP + i) % PIECES) * 1024), 16, 0, 0);
    ++staged;
    xb += staged < n ? xadv : 0;                     // phantom stages re-read the last chunk
  };
  const unsigned ring_u32 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)xring;
  const unsigned frag0 = ring_u32 + r * 128 + ((q ^ (r & 7)) << 4);            // k-step 0 of a chunk: 16-B slot q
  const unsigned frag1 = ring_u32 + r * 128 + (((4 + q) ^ (r & 7)) << 4);      // k-step 1: slot 4 + q

  wreg acc0[MB], acc1[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc0[mb] = wreg{0.f, 0.f, 0.f, 0.f}; acc1[mb] = wreg{0.f, 0.f, 0.f, 0.f}; }
  wreg wa[NB][2], wb[NB][2];
#define VR2_LOADW(S_)                                                                                                   \
  do {                                                                                                                  \
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(wa[S_][0]) : "v"(wlane), "s"(w0) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(wb[S_][0]) : "v"(wlane), "s"(w1) : "memory");               \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=v"(wa[S_][1]) : "v"(wlane), "s"(w0) : "memory");   \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=v"(wb[S_][1]) : "v"(wlane), "s"(w1) : "memory");   \
    ++loaded;                                                                                                           \
    { const int adv = (loaded < n && ABL != 3 && ABL != 4) ? 2048 : 0; w0 += adv; w1 += adv; }                                      \
  } while (0)

  // The whole k loop for one kind of wave.  ACTIVE waves stream a tile pair: 4 weight loads + XP stage ops per chunk; helper
  // waves (wave >= count) only stage and keep the barriers: XP ops per chunk.  The counted waits follow from the issue order
  // [weight loads of chunk c + D][wait x chunk c + barrier][stage chunk c + D][multiply chunk c].
  auto kloop = [&](auto act) {
    constexpr bool ACTIVE = decltype(act)::value;
    constexpr int WL = ACTIVE ? 4 : 0;
    constexpr int X_YOUNGER = D * WL + (D - 1) * XP;   // VMEM ops issued after the LDS-DMA of chunk c before its wait
    constexpr int W_YOUNGER = D * WL + (D + 1) * XP;   // ... after the weight loads of chunk c before theirs
    static_assert(W_YOUNGER <= 63, "vmcnt is 6 bits");
    auto& A0 = acc0; auto& A1 = acc1; auto& WA = wa; auto& WB = wb;   // (asm operands alone do not capture in a generic lambda)
    // one chunk.  STEADY: the load batches it names are issued (real or phantom) and the queue history is the steady one, so
    // the waits are counted; otherwise (the last <= D chunks of the part) nothing is in flight any more.
    auto chunk = [&](int cc, auto jc, auto steady) {
      constexpr int j = decltype(jc)::value;
      constexpr bool STEADY = decltype(steady)::value;
      auto& C0 = A0; auto& C1 = A1; auto& CA = WA; auto& CB = WB;       // (same capture quirk, one lambda level down)
      if constexpr (STEADY && ACTIVE) VR2_LOADW((j + D) % NB);   // chunk cc + D into the set chunk cc - 1 was multiplied from
      // the barrier publishes every wave's pieces of x chunk cc and retires ring slot cc - 1, refilled with chunk cc + D
      if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(X_YOUNGER) : "memory");
      else asm volatile("s_barrier" ::: "memory");
      if constexpr (STEADY) stage();
      if constexpr (ACTIVE) {
        const unsigned boff = (unsigned)(cc & (NB - 1)) * CHUNKB;
#pragma unroll
        for (int i = 0; i < (ABL == 2 ? 0 : 2); ++i) {
          const unsigned a = (i == 0 ? frag0 : frag1) + boff;
          wreg b[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[mb]) : "v"(a), "n"(mb * 2048) : "memory");
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            if (i == 0 && mb == 0 && STEADY)           // first use of this weight set
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_YOUNGER) : "memory");
            asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1"
                         : "+v"(C0[mb]), "+v"(C1[mb])
                         : "v"(CA[j][i]), "v"(CB[j][i]), "v"(b[mb]), "n"(MB - 1 - mb));
          }
        }
      }
    };
    // prologue = the load halves of D steady iterations, in steady order, so the counted waits hold from chunk 0 on
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if constexpr (ACTIVE) VR2_LOADW(j);
      stage();
    }
    // main loop: whole groups of NB steady chunks; covers chunks 0 .. S' - 1, S' = (n - D) rounded up to a multiple of NB,
    // every one of them a real chunk (S' <= n because NB = D + 1)
    int cb = 0;
    for (; cb < n - D; cb += NB) {
      [&]<int... J>(std::integer_sequence<int, J...>) {
        (chunk(cb + J, std::integral_constant<int, J>{}, std::true_type{}), ...);
      }(std::make_integer_sequence<int, NB>{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // nothing in flight past this point
    // the last n - S' <= D chunks: their weights and activations were loaded by the main loop
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ((cb + J < n ? (chunk(cb + J, std::integral_constant<int, J>{}, std::false_type{}),
                      (void)({ asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); 0; }))
                   : (void)0), ...);
    }(std::make_integer_sequence<int, D>{});
  };
  if (active) kloop(std::true_type{});
  else { kloop(std::false_type{}); return; }
#undef VR2_LOADW
  // D layout (16x16): lane holds col = m = lane&15, rows n = 4*(lane>>4)+i of the tile
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long row = mb * 16 + r;
    if (row >= M) continue;
    if (MODE == 0) {
      float* o = out_f32 + (long)blockIdx.y * split_stride + row * ldo;
      const int na = t0 * 16 + 4 * q, nb = t1 * 16 + 4 * q;
      if (na < N) *reinterpret_cast<float4*>(o + na) = make_float4(acc0[mb][0], acc0[mb][1], acc0[mb][2], acc0[mb][3]);
      if (t1_valid && nb < N) *reinterpret_cast<float4*>(o + nb) = make_float4(acc1[mb][0], acc1[mb][1], acc1[mb][2], acc1[mb][3]);
    } else {
      const int nn = (t0 >> 1) * 16 + 4 * q;         // gate tile t0 (even), up tile t0 + 1 -> output column block t0/2
      if (t1_valid && nn < N) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float gt = acc0[mb][i]; o[i] = f2bf((gt / (1.0f + expf(-gt))) * acc1[mb][i]); }
        *reinterpret_cast<bf16x4*>(out_bf16 + row * ldo + nn) = o;
      }
    }
  }
}

// groups per k part for the ring kernel: a whole number of workgroups per CU over all parts, <= 8 tile pairs per workgroup
inline int ring_groups(int pairs, int parts) {
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
void launch_gemv(dim3 grid, hipStream_t st, const bf16* x, int ldx, int M, const bf16* Wp, int n_tiles, int N, int K, float* of,
                 long split_stride, int ldo, bf16* ob) {
  if (M <= 16) {
    hipLaunchKernelGGL((gemv16_kernel<MODE, 1, 4>), grid, dim3(256), 0, st, x, ldx, M, Wp, n_tiles, N, K, of, split_stride, ldo, ob);
    return;
  }
  const int variant = vr2_tune_get(VR2_TUNE_GEMV_VARIANT);
  const int pairs = (n_tiles + 1) / 2;
  const int groups = (variant > 0 && K % 64 == 0 && (K / 64) / (int)grid.y >= 4) ? ring_groups(pairs, (int)grid.y) : 0;
  if (groups > 0) {
    // register-ring kernel: one workgroup per CU and round, <= 8 tile pairs each, grid.y parts of whole 64-k chunks
    const dim3 gr((unsigned)groups, grid.y);
    // experiment knob: x given chunk-major ([K/64][M][64]: every 64-k chunk of all rows contiguous) instead of row-major
    const bool chunk_major = vr2_tune_get(VR2_TUNE_GEMV_XCHUNK) != 0;
    const int rldx = chunk_major ? 64 : ldx, xadv = chunk_major ? M * 128 : 128;
#define VR2_RING_LAUNCH(MB_, D_, ...)                                                                                           \
  do {                                                                                                                          \
    static bool attr = false;                                                                                                   \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_ring_kernel<MODE, MB_, D_ __VA_OPT__(,) __VA_ARGS__>),  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (D_ + 1) * MB_ * 2048); attr = true; }   \
    hipLaunchKernelGGL((gemv_ring_kernel<MODE, MB_, D_ __VA_OPT__(,) __VA_ARGS__>), gr, dim3(512), (D_ + 1) * MB_ * 2048, st, x, \
                       rldx, xadv, M, Wp, n_tiles, N, K, of, split_stride, ldo, ob);                                            \
  } while (0)
    const int abl = vr2_tune_get(VR2_TUNE_GEMV_ABLATE);
    if (abl > 0 && MODE == 1 && M > 64) {            // profiles/: which resource bounds the kernel (results are wrong)
      if (abl == 1) VR2_RING_LAUNCH(8, 1, 1); else if (abl == 2) VR2_RING_LAUNCH(8, 1, 2); else if (abl == 3) VR2_RING_LAUNCH(8, 1, 3); else VR2_RING_LAUNCH(8, 1, 4);
    } else if (M > 128) {                            // 129-256 rows: 10 / 12 / 14 / 16 row blocks per wave, one chunk in flight
      if (M <= 160) VR2_RING_LAUNCH(10, 1); else if (M <= 192) VR2_RING_LAUNCH(12, 1); else if (M <= 224) VR2_RING_LAUNCH(14, 1); else VR2_RING_LAUNCH(16, 1);
    } else if (variant == 2) {                       // shallow ring (2 slots, one chunk in flight): A/B reference point
      if (M <= 32) VR2_RING_LAUNCH(2, 1); else if (M <= 64) VR2_RING_LAUNCH(4, 1); else VR2_RING_LAUNCH(8, 1);
    } else {
      if (M <= 32) VR2_RING_LAUNCH(2, 3); else if (M <= 64) VR2_RING_LAUNCH(4, 3); else VR2_RING_LAUNCH(8, 3);
    }
#undef VR2_RING_LAUNCH
    return;
  }
  if (M > 128) { vr2_set_error("decode GEMV: %d rows need the ring kernel (K %% 64 == 0, >= 4 chunks per k part)", M); return; }
  // LDS-staged variant, 8 weight tiles (128 output rows) per workgroup; two k groups per workgroup when the part is long
  const dim3 g4((unsigned)((n_tiles + 7) / 8), grid.y);
  // (measured at M = 64, 7B shapes: the second k group pays off for the split-K projections -- down 37 -> 31.5 us at 8 slabs,
  // qkv 14.5 -> 12.4 us at 4 slabs, i.e. the same time with half the partial slabs for the consumer to sum -- but not for
  // gate/up: 59.8 vs 62.6 us at 64 rows, 72 vs 87 us at 128; ablation there: x staging costs 11 us, the MFMA/LDS-read side nothing)
  const bool two = MODE == 0 && (K / 128) >= 4 * (int)grid.y;
#define VR2_GEMV_LAUNCH(MB_)                                                                                                   \
  if (two) {                                                                                                                   \
    static bool attr = false;                                                                                                  \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_lds_kernel<MODE, MB_, 2>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * MB_ * 4096); attr = true; }     \
    hipLaunchKernelGGL((gemv_lds_kernel<MODE, MB_, 2>), g4, dim3(512), 2 * 2 * MB_ * 4096, st, x, ldx, M, Wp, n_tiles, N, K, of, \
                       split_stride, ldo, ob);                                                                                 \
  } else {                                                                                                                     \
    static bool attr = false;                                                                                                  \
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_lds_kernel<MODE, MB_, 1>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MB_ * 4096); attr = true; }         \
    hipLaunchKernelGGL((gemv_lds_kernel<MODE, MB_, 1>), g4, dim3(256), 2 * MB_ * 4096, st, x, ldx, M, Wp, n_tiles, N, K, of,    \
                       split_stride, ldo, ob);                                                                                 \
  }
  if (M <= 32) { VR2_GEMV_LAUNCH(2); }
  else if (M <= 64) { VR2_GEMV_LAUNCH(4); }
  else { VR2_GEMV_LAUNCH(8); }
#undef VR2_GEMV_LAUNCH
}

}  // namespace

extern "C" int vr2_gemv_bf16(const void* x, int64_t ldx, int64_t M, const void* Wpacked, int64_t N, int64_t K,
                             void* out_f32, int64_t split_stride, int64_t ldo, int nsplit, void* stream) {
  VR2_REQUIRE(M >= 1 && M <= 256, "vr2_gemv_bf16: M=%ld must be in [1,256]", (long)M);
  VR2_REQUIRE(M <= 128 || (K % 64 == 0 && (K / 64) / nsplit >= 4), "vr2_gemv_bf16: M=%ld > 128 rows needs K %% 64 == 0 and >= 4 chunks per part", (long)M);
  VR2_REQUIRE(K % 32 == 0 && N % 4 == 0, "vr2_gemv_bf16: K=%ld %% 32 and N=%ld %% 4 must be 0", (long)K, (long)N);
  VR2_REQUIRE(ldx % 8 == 0 && ldo % 4 == 0, "vr2_gemv_bf16: strides must keep 16-byte alignment");
  VR2_REQUIRE(nsplit >= 1 && (K / 32) >= (int64_t)nsplit * 4, "vr2_gemv_bf16: nsplit=%d too large for K=%ld", nsplit, (long)K);
  const int n_tiles = (int)((N + 15) / 16);
  dim3 grid((unsigned)((n_tiles + 1) / 2), (unsigned)nsplit);
  launch_gemv<0>(grid, (hipStream_t)stream, (const bf16*)x, (int)ldx, (int)M, (const bf16*)Wpacked, n_tiles, (int)N, (int)K,
                 (float*)out_f32, (long)split_stride, (int)ldo, (bf16*)nullptr);
  return vr2_check_launch("vr2_gemv_bf16");
}

extern "C" int vr2_gemv_gateup_silu_bf16(const void* x, int64_t ldx, int64_t M, const void* Wpacked, int64_t I,
                                         int64_t K, void* act_bf16, int64_t ldo, void* stream) {
  VR2_REQUIRE(M >= 1 && M <= 256, "vr2_gemv_gateup_silu_bf16: M=%ld must be in [1,256]", (long)M);
  VR2_REQUIRE(M <= 128 || (K % 64 == 0 && K >= 256), "vr2_gemv_gateup_silu_bf16: M=%ld > 128 rows needs K %% 64 == 0", (long)M);
  VR2_REQUIRE(K % 32 == 0 && K >= 128 && I % 16 == 0, "vr2_gemv_gateup_silu_bf16: K=%ld %% 32 (>=128) and I=%ld %% 16 must be 0", (long)K, (long)I);
  VR2_REQUIRE(ldx % 8 == 0 && ldo % 4 == 0, "vr2_gemv_gateup_silu_bf16: strides must keep alignment");
  const int it = (int)(I / 16);
  dim3 grid((unsigned)it, 1);
  launch_gemv<1>(grid, (hipStream_t)stream, (const bf16*)x, (int)ldx, (int)M, (const bf16*)Wpacked, 2 * it, (int)I, (int)K,
                 (float*)nullptr, 0L, (int)ldo, (bf16*)act_bf16);
  return vr2_check_launch("vr2_gemv_gateup_silu_bf16");
}
