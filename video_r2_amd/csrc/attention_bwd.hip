// Attention backward for the LM update path, flash-style: P is recomputed from Q, K and the forward's
// log-sum-exp, nothing of size S x S is stored.  Two kernels, both one wave per SIMD (4 waves, the whole register file):
//   * vr2_attn_bwd_dkv : key-centric.  A workgroup holds 128 keys of one KV head -- every wave 32 keys, K / V fragments and the
//                        dK^T / dV^T accumulators of those keys in registers -- and sweeps (q heads of the GQA group) x (64-row
//                        query tiles that can see them): S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.  Prompt keys
//                        are seen by every completion row of all G sequences, so that sweep is SPLIT over several workgroups
//                        (host work list) and dK / dV are fp32 buffers the workgroups add into (whole 256-byte row segments
//                        per wave-instruction, the shape the memory-side float atomics run at full rate with).
//   * vr2_attn_bwd_dq  : query-centric.  A workgroup holds 128 query rows of one head (32 per wave: Q / dO fragments and dQ^T
//                        in registers) and sweeps the 64-key tiles it can see: S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T.
// Unified row space: rows [0,S) are the prompt, rows S + g*C + j completion token j of sequence g; Q/K/V/O/dO are row-major
// [T][heads*128] bf16.  Per-row constants come TRANSPOSED from vr2_attn_delta: rowc[0][head][t] = -LSE * log2(e),
// rowc[1][head][t] = delta = rowsum(dO o O), so that p = exp2(scale*log2(e) * s + rowc[0]) needs no row maximum.
//
// The swept tiles (Q and dO, or K and V: 64 rows x 256 B) arrive by direct-to-LDS loads (one 1-KiB piece = 4 rows per wave
// instruction) in a ring, counted vmcnt, one barrier per tile.  LDS image of a tile: plain 256-byte rows whose 16-byte chunks are
// XOR-swizzled with ((row & 3) << 2) | ((row >> 2) & 3) -- applied to the per-lane SOURCE address of the load and again to the
// read address -- so that ONE image serves the row reads (ds_read_b128, operand of S / dP) and the transposed reads
// (ds_read_b64_tr_b16, operand of the dK / dV / dQ products) without bank conflicts.  All MFMAs are
// v_mfma_f32_16x16x32_bf16.  MFMA row i of a 16-row block b of a 32-row half stands for tile row (i & 3) + 8 (i >> 2) + 4 b:
// the accumulators of blocks 0 and 1 then hold, per lane, 8 CONSECUTIVE rows 8 (lane >> 4) .. + 7 of the half, which is exactly
// the k order the next MFMA's B operand wants, so P and dS feed it from registers.
#include <type_traits>
#include "attn_tile.h"

namespace {

struct KeyWork { int k_row0, nk, q1_row0, q1_len, q2_row0, q2_len, head0, nheads; };  // q1: causal (key row <= query row); q2: all visible

// delta[t][h] = sum_d dO[t][h][d] * O[t][h][d];  rowc[0][h][t] = -lse[t][h] * log2(e), rowc[1][h][t] = delta
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ O, const bf16* __restrict__ dO, long ld, int n_heads,
                                                         const float* __restrict__ lse, float* __restrict__ rowc, long T) {
  const int t = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int h = wave; h < n_heads; h += 4) {
    const long off = (long)t * ld + (long)h * HD + lane * 2;
    float v = bf2f(O[off]) * bf2f(dO[off]) + bf2f(O[off + 1]) * bf2f(dO[off + 1]);
    v = wave_sum(v);
    if (lane == 0) {
      rowc[(long)h * T + t] = -lse[(long)t * n_heads + h] * LOG2E;
      rowc[((long)n_heads + h) * T + t] = v;
    }
  }
}

// ---------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const KeyWork* __restrict__ work, const bf16* __restrict__ Q, long ldq,
                                                           const bf16* __restrict__ K, const bf16* __restrict__ V, long ldk,
                                                           const bf16* __restrict__ dO, const float* __restrict__ rowc, long T,
                                                           float* __restrict__ dK, float* __restrict__ dV, int n_heads, int group,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const KeyWork w = work[blockIdx.x];
  const int kvh = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int nh = w.nheads > 0 ? w.nheads : group, head_first = kvh * group + w.head0;
  const int n1 = (w.q1_len + 63) >> 6, n2 = (w.q2_len + 63) >> 6, per_head = n1 + n2, nsteps = nh * per_head;
  bf16x8 kf[2][4], vf[2][4];
  bool kvalid[2];
  int krow[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int slot = wave * 32 + s * 16 + fr;
    kvalid[s] = slot < w.nk;
    krow[s] = w.k_row0 + (kvalid[s] ? slot : w.nk - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[s][ks] = *reinterpret_cast<const bf16x8*>(K + (long)krow[s] * ldk + kvh * HD + ks * 32 + fq * 8);
      vf[s][ks] = *reinterpret_cast<const bf16x8*>(V + (long)krow[s] * ldk + kvh * HD + ks * 32 + fq * 8);
    }
  }
  // consume the fragments here: the compiler waits for these loads now instead of inside the loop, where its (uncounted-for)
  // s_waitcnt vmcnt would also drain the ring's direct-to-LDS loads
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(kf[s][ks]), "v"(vf[s][ks]));
  f32x4 accK[2][8], accV[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) { accK[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float c2 = scale * LOG2E;

  struct Step { int head, r0, rl, q0; bool causal; };
  auto decode = [&](int st) {
    Step d;
    const int hq = st / per_head, t = st - hq * per_head;
    d.head = head_first + hq;
    d.causal = t < n1;
    d.r0 = d.causal ? w.q1_row0 : w.q2_row0;
    d.rl = d.causal ? w.q1_len : w.q2_len;
    d.q0 = (d.causal ? t : t - n1) * 64;
    return d;
  };
  auto issue = [&](int st) {   // 9 direct-to-LDS loads per wave
    const Step d = decode(st);
    char* stg = smem + (st % KV_NSTAGE) * KV_STAGE_B;
    const int rmax = d.rl - 1 - d.q0;   // rows past the range re-read its last row (masked out of P)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pi = wave * 4 + j;
      int row = pi * 4 + (lane >> 4);
      const int sc = (lane & 15) ^ (((lane >> 4) << 2) | j);   // (row & 3) = lane >> 4, (row >> 2) & 3 = j
      row = row < rmax ? row : rmax;
      const long off = (long)(d.r0 + d.q0 + row) * ldq + (long)d.head * HD + sc * 8;
      vr2_dma16(Q + off, stg + pi * 1024);
      vr2_dma16(dO + off, stg + TILE_B + pi * 1024);
    }
    {
      int row = wave * 16 + (lane & 15);
      row = row < rmax ? row : rmax;
      const float* src = rowc + ((long)((lane >> 4) & 1) * n_heads + d.head) * T + d.r0 + d.q0 + row;
      if (lane < 32) vr2_dma4(src, stg + 2 * TILE_B + wave * 128);
    }
  };
  auto compute = [&](int st, auto masked_tag) {
    constexpr bool MASK = decltype(masked_tag)::value;
    const Step d = decode(st);
    const char* sQ = smem + (st % KV_NSTAGE) * KV_STAGE_B;
    const char* sD = sQ + TILE_B;
    const float* sR = reinterpret_cast<const float*>(sQ + 2 * TILE_B);
    // fragment reads run ahead of the MFMAs that use them (one wave per SIMD: nothing else hides the LDS latency): row
    // fragments one k-step (8 MFMAs) ahead, transposed fragments two feature blocks (8 MFMAs) ahead
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 sa[2][2], da[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) { sa[b][s] = f32x4{0.f, 0.f, 0.f, 0.f}; da[b][s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      bf16x8 fq_[2][2], fd_[2][2];   // [ring][b]
#pragma unroll
      for (int b = 0; b < 2; ++b) { fq_[0][b] = row_frag(sQ, h, b, 0, lane); fd_[0][b] = row_frag(sD, h, b, 0, lane); }
      bf16x8 tq[3], td[3];           // transposed fragments, ring of 3
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
#pragma unroll
          for (int b = 0; b < 2; ++b) { fq_[(ks + 1) & 1][b] = row_frag(sQ, h, b, ks + 1, lane); fd_[(ks + 1) & 1][b] = row_frag(sD, h, b, ks + 1, lane); }
        } else {
          td[0] = tr_frag(sD, h, 0, lane); tq[0] = tr_frag(sQ, h, 0, lane);
          td[1] = tr_frag(sD, h, 1, lane); tq[1] = tr_frag(sQ, h, 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads in front of the MFMAs (the scheduler would sink them to their uses)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            sa[b][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq_[ks & 1][b], kf[s][ks], sa[b][s], 0, 0, 0);
            da[b][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd_[ks & 1][b], vf[s][ks], da[b][s], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      // lane: key slot fr of both key sets (column), tile rows 32 h + 8 fq + 4 b + r
      bf16x8 pb[2], dsb[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int rb = 32 * h + 8 * fq + 4 * b;
        const f32x4 nl = *reinterpret_cast<const f32x4*>(sR + (rb >> 4) * 32 + (rb & 15));
        const f32x4 dl = *reinterpret_cast<const f32x4*>(sR + (rb >> 4) * 32 + 16 + (rb & 15));
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(c2 * sa[b][s][r] + nl[r]);
            if (MASK) {
              const int ql = d.q0 + rb + r;
              const bool ok = kvalid[s] && ql < d.rl && (!d.causal || krow[s] <= d.r0 + ql);
              p = ok ? p : 0.f;
            }
            pb[s][4 * b + r] = f2bf(p);
            dsb[s][4 * b + r] = f2bf(p * (da[b][s][r] - dl[r]));
          }
      }
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        if (db + 2 < 8) { td[(db + 2) % 3] = tr_frag(sD, h, db + 2, lane); tq[(db + 2) % 3] = tr_frag(sQ, h, db + 2, lane); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          accV[s][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(td[db % 3], pb[s], accV[s][db], 0, 0, 0);
          accK[s][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq[db % 3], dsb[s], accK[s][db], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (nsteps > 0) issue(0);
  if (nsteps > 1) issue(1);
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 2 < nsteps) issue(st + 2);
    const Step d = decode(st);
    const bool masked = (d.causal && d.r0 + d.q0 < w.k_row0 + w.nk - 1) || (d.q0 + 64 > d.rl) || w.nk < 128;
    if (masked) compute(st, std::true_type{});
    else compute(st, std::false_type{});
  }
  // epilogue: every wave transposes its 32 key rows through LDS and adds whole 256-byte row segments
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float* stg = reinterpret_cast<float*>(smem) + wave * 32 * OUT_STRIDE;
  const int nrows = (w.nk - wave * 32) < 32 ? (w.nk - wave * 32) : 32;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* dst = (pass == 0 ? dK : dV) + (long)(w.k_row0 + wave * 32) * ldk + kvh * HD;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        f32x4 v = pass == 0 ? accK[s][db] : accV[s][db];
        if (pass == 0) { v[0] *= scale; v[1] *= scale; v[2] *= scale; v[3] *= scale; }
        *reinterpret_cast<f32x4*>(stg + (s * 16 + fr) * OUT_STRIDE + db * 16 + 4 * fq) = v;
      }
    for (int rr = 0; rr < nrows; ++rr) {
      unsafeAtomicAdd(dst + (long)rr * ldk + lane, stg[rr * OUT_STRIDE + lane]);
      unsafeAtomicAdd(dst + (long)rr * ldk + 64 + lane, stg[rr * OUT_STRIDE + 64 + lane]);
    }
  }
}

// ---------------------------------------------------------------- dQ
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnWork* __restrict__ work, const bf16* __restrict__ Q, long ldq,
                                                          const bf16* __restrict__ KA, const bf16* __restrict__ VA,
                                                          const bf16* __restrict__ KB, const bf16* __restrict__ VB, long ldk,
                                                          const bf16* __restrict__ dO, const float* __restrict__ rowc, long T,
                                                          bf16* __restrict__ dQ, int n_heads, int group, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnWork w = work[blockIdx.x];
  const int head = blockIdx.y, kvh = head / group;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 qf[2][4], df[2][4];
  float nls[2], dl[2];
  int vis_lim[2];
  long lastq = (long)w.q_causal0 + (w.nq - 1) + 1;
  const int bvis = (int)(lastq < w.kb_len ? lastq : w.kb_len);
  const int kend = w.ka_len + (bvis > 0 ? bvis : 0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int qi = wave * 32 + s * 16 + fr;
    const bool qvalid = qi < w.nq;
    const long qrow = w.q_row0 + (qvalid ? qi : w.nq - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[s][ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * ldq + (long)head * HD + ks * 32 + fq * 8);
      df[s][ks] = *reinterpret_cast<const bf16x8*>(dO + qrow * ldq + (long)head * HD + ks * 32 + fq * 8);
    }
    nls[s] = rowc[(long)head * T + qrow];
    dl[s] = rowc[((long)n_heads + head) * T + qrow];
    const long lim = (long)w.ka_len + (long)w.q_causal0 + qi + 1;
    vis_lim[s] = qvalid ? (int)(lim < kend ? lim : kend) : 0;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {   // wait for the loads here, not inside the loop (see attn_bwd_dkv_kernel)
    asm volatile("" ::"v"(nls[s]), "v"(dl[s]));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[s][ks]), "v"(df[s][ks]));
  }
  f32x4 acc[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[s][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c2 = scale * LOG2E;
  const int nsteps = (kend + 63) >> 6;
  // the wave's lowest fully visible limit: tiles that end below it need no mask
  int min_lim = vis_lim[0] < vis_lim[1] ? vis_lim[0] : vis_lim[1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(min_lim, o, 64); min_lim = x < min_lim ? x : min_lim; }

  auto issue = [&](int st) {   // 8 direct-to-LDS loads per wave
    char* stg = smem + (st % DQ_NSTAGE) * DQ_STAGE_B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pi = wave * 4 + j;
      int key = st * 64 + pi * 4 + (lane >> 4);
      const int sc = (lane & 15) ^ (((lane >> 4) << 2) | j);
      key = key < kend - 1 ? key : kend - 1;
      const bool inA = key < w.ka_len;
      const long off = (long)(inA ? w.ka_row0 + key : w.kb_row0 + key - w.ka_len) * ldk + kvh * HD + sc * 8;
      vr2_dma16((inA ? KA : KB) + off, stg + pi * 1024);
      vr2_dma16((inA ? VA : VB) + off, stg + TILE_B + pi * 1024);
    }
  };
  auto compute = [&](int st, auto masked_tag) {
    constexpr bool MASK = decltype(masked_tag)::value;
    const char* sK = smem + (st % DQ_NSTAGE) * DQ_STAGE_B;
    const char* sV = sK + TILE_B;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 sa[2][2], da[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) { sa[b][s] = f32x4{0.f, 0.f, 0.f, 0.f}; da[b][s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      bf16x8 fk[2][2], fv[2][2];   // [ring][b]: row fragments one k-step ahead of their MFMAs
#pragma unroll
      for (int b = 0; b < 2; ++b) { fk[0][b] = row_frag(sK, h, b, 0, lane); fv[0][b] = row_frag(sV, h, b, 0, lane); }
      bf16x8 tk[4];                // transposed fragments, ring of 4: three feature blocks (6 MFMAs) ahead
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
#pragma unroll
          for (int b = 0; b < 2; ++b) { fk[(ks + 1) & 1][b] = row_frag(sK, h, b, ks + 1, lane); fv[(ks + 1) & 1][b] = row_frag(sV, h, b, ks + 1, lane); }
        } else {
          tk[0] = tr_frag(sK, h, 0, lane); tk[1] = tr_frag(sK, h, 1, lane); tk[2] = tr_frag(sK, h, 2, lane);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads in front of the MFMAs (the scheduler would sink them to their uses)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            sa[b][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks & 1][b], qf[s][ks], sa[b][s], 0, 0, 0);
            da[b][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[ks & 1][b], df[s][ks], da[b][s], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      // lane: query slot fr of both query sets (column), keys 64 st + 32 h + 8 fq + 4 b + r
      bf16x8 dsb[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(c2 * sa[b][s][r] + nls[s]);
            if (MASK) p = (st * 64 + 32 * h + 8 * fq + 4 * b + r) < vis_lim[s] ? p : 0.f;
            dsb[s][4 * b + r] = f2bf(p * (da[b][s][r] - dl[s]));
          }
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        if (db + 3 < 8) tk[(db + 3) & 3] = tr_frag(sK, h, db + 3, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[s][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tk[db & 3], dsb[s], acc[s][db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (nsteps > 0) issue(0);
  for (int st = 0; st < nsteps; ++st) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 1 < nsteps) issue(st + 1);
    if (st * 64 + 64 > min_lim) compute(st, std::true_type{});
    else compute(st, std::false_type{});
  }
  // epilogue: the wave's 32 query rows go through LDS and leave as whole 256-byte bf16 rows
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  bf16* stg = reinterpret_cast<bf16*>(smem) + wave * 32 * OUT_STRIDE_BF;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      const f32x4 v = acc[s][db];
      bf16x4 o = {f2bf(v[0] * scale), f2bf(v[1] * scale), f2bf(v[2] * scale), f2bf(v[3] * scale)};
      *reinterpret_cast<bf16x4*>(stg + (s * 16 + fr) * OUT_STRIDE_BF + db * 16 + 4 * fq) = o;
    }
  store_rows_bf16(stg, dQ + (long)(w.q_row0 + wave * 32) * ldq + (long)head * HD, ldq, w.nq - wave * 32, lane);
}

}  // namespace

#define STREAM ((hipStream_t)stream)

extern "C" int vr2_attn_delta(const void* O_bf16, const void* dO_bf16, int64_t ld, int64_t T, int64_t n_heads, int64_t head_dim,
                              const void* lse_f32, void* rowc_f32, void* stream) {
  VR2_REQUIRE(head_dim == 128 && T >= 1, "vr2_attn_delta: head_dim 128 only");
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)T), dim3(256), 0, STREAM, (const bf16*)O_bf16, (const bf16*)dO_bf16, (long)ld,
                     (int)n_heads, (const float*)lse_f32, (float*)rowc_f32, (long)T);
  return vr2_check_launch("vr2_attn_delta");
}

extern "C" int vr2_attn_bwd_dq(const void* work_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* KA, const void* VA,
                               const void* KB, const void* VB, int64_t ldk, const void* dO, const void* rowc_f32, int64_t T,
                               void* dQ, int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* stream) {
  VR2_REQUIRE(head_dim == 128, "vr2_attn_bwd_dq: head_dim 128 only (the ViT's 80-wide heads are passed zero-padded to 128)");
  VR2_REQUIRE(n_work >= 1 && n_heads % group == 0 && ldq % 8 == 0 && ldk % 8 == 0, "vr2_attn_bwd_dq: bad arguments");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DQ_NSTAGE * DQ_STAGE_B);
    attr = true;
  }
  static_assert(DQ_NSTAGE * DQ_STAGE_B >= 4 * 32 * OUT_STRIDE_BF * 2, "dq: the ring holds the output staging region");
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)n_work, (unsigned)n_heads), dim3(256), DQ_NSTAGE * DQ_STAGE_B, STREAM,
                     (const AttnWork*)work_i32x8, (const bf16*)Q, (long)ldq, (const bf16*)KA, (const bf16*)VA, (const bf16*)KB,
                     (const bf16*)VB, (long)ldk, (const bf16*)dO, (const float*)rowc_f32, (long)T, (bf16*)dQ, (int)n_heads,
                     (int)group, scale);
  return vr2_check_launch("vr2_attn_bwd_dq");
}

extern "C" int vr2_attn_bwd_dkv(const void* keywork_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* K, const void* V,
                                int64_t ldk, const void* dO, const void* rowc_f32, int64_t T, void* dK_f32, void* dV_f32,
                                int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* stream) {
  void* dK = dK_f32; void* dV = dV_f32;   // fp32 [T][ldk], ZEROED by the caller: work items add into them
  VR2_REQUIRE(head_dim == 128, "vr2_attn_bwd_dkv: head_dim 128 only (the ViT's 80-wide heads are passed zero-padded to 128)");
  VR2_REQUIRE(n_work >= 1 && n_heads % group == 0 && ldq % 8 == 0 && ldk % 8 == 0, "vr2_attn_bwd_dkv: bad arguments");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, KV_NSTAGE * KV_STAGE_B);
    attr = true;
  }
  static_assert(KV_NSTAGE * KV_STAGE_B >= 4 * 32 * OUT_STRIDE * 4, "dkv: the ring holds the output staging region");
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)n_work, (unsigned)(n_heads / group)), dim3(256), KV_NSTAGE * KV_STAGE_B, STREAM,
                     (const KeyWork*)keywork_i32x8, (const bf16*)Q, (long)ldq, (const bf16*)K, (const bf16*)V, (long)ldk,
                     (const bf16*)dO, (const float*)rowc_f32, (long)T, (float*)dK, (float*)dV, (int)n_heads, (int)group, scale);
  return vr2_check_launch("vr2_attn_bwd_dkv");
}
