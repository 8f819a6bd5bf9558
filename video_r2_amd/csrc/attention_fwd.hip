// Flash-style attention forward for head_dim 128 (the LM: causal prefill, teacher-forced scoring of G completions against a
// shared prompt prefix, the training forward with log-sum-exp; HF modeling_qwen2_5_vl.py:641-689, K12 / K15 / K17 feed).
// The structure of attention_bwd.hip's dQ kernel: a workgroup = 4 waves = up to 128 query rows of one head, 32 per wave (two
// 16-row sets: Q fragments and the O^T accumulators in registers, one wave per SIMD); the 64-key K / V tiles the rows can see
// arrive by direct-to-LDS loads in a 2-deep ring (one barrier per tile; two workgroups per CU); scores are computed transposed,
// S^T = K Q^T, so a lane holds 16 keys of ONE query row of each set: the running row maximum / sum need 2 cross-lane steps, and
// the exponentiated block is already the B operand of O^T += V^T P^T (V^T fragments by ds_read_b64_tr_b16 from the same
// row-major image).  The row maximum is kept in the log2 domain (scores * scale * log2 e), p = exp2(s - m).
// Work items are vr2_attn_tok's (attention.hip): q_row0, nq <= 128, keys = [A: ka_len rows, all visible | B: kb_len rows,
// causal: row j visible to query i iff j <= q_causal0 + i].
#include <type_traits>
#include "attn_tile.h"

namespace {

constexpr int FW_STAGE_B = 2 * TILE_B;   // K tile | V tile
constexpr int FW_NSTAGE = 2;
constexpr int FW_LDS = FW_NSTAGE * FW_STAGE_B;

__global__ __launch_bounds__(256, 2) void attn_fwd128_kernel(const AttnWork* __restrict__ work, const bf16* __restrict__ Q, long ldq,
                                                          const bf16* __restrict__ KA, const bf16* __restrict__ VA,
                                                          const bf16* __restrict__ KB, const bf16* __restrict__ VB, long ldk,
                                                          bf16* __restrict__ O, long ldo, int group, float scale,
                                                          float* __restrict__ lse_out, int n_heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnWork w = work[blockIdx.x];
  const int head = blockIdx.y, kvh = head / group;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 qf[2][4];
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
    for (int ks = 0; ks < 4; ++ks) qf[s][ks] = *reinterpret_cast<const bf16x8*>(Q + qrow * ldq + (long)head * HD + ks * 32 + fq * 8);
    const long lim = (long)w.ka_len + (long)w.q_causal0 + qi + 1;
    vis_lim[s] = qvalid ? (int)(lim < kend ? lim : kend) : 0;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)   // wait for the loads here, not inside the loop (see attn_bwd_dkv_kernel)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[s][ks]));
  f32x4 acc[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[s][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float c2 = scale * LOG2E;
  const int nsteps = (kend + 63) >> 6;
  int min_lim = vis_lim[0] < vis_lim[1] ? vis_lim[0] : vis_lim[1];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(min_lim, o, 64); min_lim = x < min_lim ? x : min_lim; }

  auto issue = [&](int st) {   // 8 direct-to-LDS loads per wave
    char* stg = smem + (st % FW_NSTAGE) * FW_STAGE_B;
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
    const char* sK = smem + (st % FW_NSTAGE) * FW_STAGE_B;
    const char* sV = sK + TILE_B;
    // S^T for the whole 64-key tile: sa[h][b][s], lane = query slot fr of set s, keys 64 st + 32 h + 8 fq + 4 b + r
    f32x4 sa[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) sa[h][b][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fk[2][2];   // [ring][b]: row fragments one k-step (4 MFMAs) ahead
    bf16x8 tv[4];      // transposed V fragments, ring of 4
#pragma unroll
    for (int b = 0; b < 2; ++b) fk[0][b] = row_frag(sK, 0, b, 0, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // i = 4 h + ks
      const int h = i >> 2, ks = i & 3;
      if (i < 7) {
#pragma unroll
        for (int b = 0; b < 2; ++b) fk[(i + 1) & 1][b] = row_frag(sK, (i + 1) >> 2, b, (i + 1) & 3, lane);
      } else {
        tv[0] = tr_frag(sV, 0, 0, lane); tv[1] = tr_frag(sV, 0, 1, lane); tv[2] = tr_frag(sV, 0, 2, lane);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads in front of the MFMAs (the scheduler would sink them to their uses)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) sa[h][b][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[i & 1][b], qf[s][ks], sa[h][b][s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // online softmax per query slot (column): 16 keys in the lane, 4 lane quarters
    bf16x8 pb[2][2];   // [h][s]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float mx = -INFINITY;   // of the raw scores: scale * log2 e > 0 commutes with the maximum
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (MASK) sa[h][b][s][r] = (st * 64 + 32 * h + 8 * fq + 4 * b + r) < vis_lim[s] ? sa[h][b][s][r] : -INFINITY;
            mx = fmaxf(mx, sa[h][b][s][r]);
          }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // the reference point m of p = exp2(s - m) only moves when the row maximum has grown by more than 2^8 over it (any m gives
      // the same O = acc / l; p <= 256 keeps bf16 P and the fp32 sums far from their limits): the rescale of the 32 accumulator
      // registers (AGPR read, multiply, write back) then runs on the first tile and rarely afterwards
      const float mxs = mx * c2;
      const float m_new = (mxs > m_run[s] + 8.0f) ? mxs : m_run[s];
      const float m_use = fmaxf(m_new, -1e30f);           // all keys masked so far: exp2(-inf - m_use) = 0, no NaN
      const bool moved = m_new != m_run[s];
      const float alpha = moved ? __builtin_amdgcn_exp2f(m_run[s] - m_use) : 1.0f;
      m_run[s] = m_new;
      float sum = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sa[h][b][s][r], c2, -m_use));
            sum += p;
            pb[h][s][4 * b + r] = f2bf(p);
          }
      l_run[s] = l_run[s] * alpha + sum;
      if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
        for (int db = 0; db < 8; ++db) { acc[s][db][0] *= alpha; acc[s][db][1] *= alpha; acc[s][db][2] *= alpha; acc[s][db][3] *= alpha; }
      }
    }
    // O^T += V^T P^T
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // i = 8 h + db
      const int h = i >> 3, db = i & 7;
      if (i + 3 < 16) tv[(i + 3) & 3] = tr_frag(sV, (i + 3) >> 3, (i + 3) & 7, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tv[i & 3], pb[h][s], acc[s][db], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // 2-deep ring, two workgroups per CU (<= 256 registers: the other workgroup's MFMAs run under this one's softmax VALU work and
  // the other way round): tile st + 1 is loaded while tile st is computed
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
  for (int s = 0; s < 2; ++s) {
    float l = l_run[s];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int qi = wave * 32 + s * 16 + fr;
    if (lse_out && fq == 0 && qi < w.nq)   // natural-log LSE of the scaled scores, for the backward
      lse_out[(long)(w.q_row0 + qi) * n_heads + head] = (l > 0.f) ? (m_run[s] + __log2f(l)) * 0.6931471805599453f : INFINITY;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      const f32x4 v = acc[s][db];
      bf16x4 o = {f2bf(v[0] * inv), f2bf(v[1] * inv), f2bf(v[2] * inv), f2bf(v[3] * inv)};
      *reinterpret_cast<bf16x4*>(stg + (s * 16 + fr) * OUT_STRIDE_BF + db * 16 + 4 * fq) = o;
    }
  }
  store_rows_bf16(stg, O + (long)(w.q_row0 + wave * 32) * ldo + (long)head * HD, ldo, w.nq - wave * 32, lane);
}

}  // namespace

// called by vr2_attn_tok (attention.hip) for head_dim 128
int vr2_launch_attn_fwd128(const void* work_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* KA, const void* VA,
                           const void* KB, const void* VB, int64_t ldk, void* O, int64_t ldo, int64_t n_heads, int64_t group,
                           float scale, void* lse_out, void* stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FW_LDS);
    attr = true;
  }
  static_assert(FW_LDS >= 4 * 32 * OUT_STRIDE_BF * 2, "the ring holds the output staging region");
  hipLaunchKernelGGL(attn_fwd128_kernel, dim3((unsigned)n_work, (unsigned)n_heads), dim3(256), FW_LDS, (hipStream_t)stream,
                     (const AttnWork*)work_i32x8, (const bf16*)Q, (long)ldq, (const bf16*)KA, (const bf16*)VA, (const bf16*)KB,
                     (const bf16*)VB, (long)ldk, (bf16*)O, (long)ldo, (int)group, scale, (float*)lse_out, (int)n_heads);
  return vr2_check_launch("vr2_attn_tok(128)");
}
