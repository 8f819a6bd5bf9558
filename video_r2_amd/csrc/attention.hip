// Flash-style attention for every attention shape on the rollout path, one kernel body:
//   * ViT windowed / per-frame varlen, non-causal, head_dim 80   (HF modeling_qwen2_5_vl.py:244-291, K5)
//   * LM GQA causal prefill, head_dim 128                         (HF :641-689, K12)
//   * teacher-forced scoring of G completions against a SHARED prompt prefix KV + own causal suffix (K15 feed)
//   * decode: one new token per sequence against shared-prefix + per-sequence suffix KV, split over key
//     chunks with a log-sum-exp combine (K13)
//
// Layout of one wave: 16 "query slots".  TOK mode: slots = 16 consecutive query rows of one head.  GRP mode
// (decode): slots = the q heads that share one KV head (GQA group <= 16) of ONE token, so K/V are streamed
// once per KV head.  Scores are computed transposed, S^T = K * Q^T with v_mfma_f32_16x16x16_bf16 (A = 16 keys
// from LDS, B = the wave's Q fragment held in registers), which leaves every lane holding 4 keys of ONE query
// slot: the row max/sum is 3 in-lane ops + 2 cross-lane shuffles, and the exponentiated block is already the
// B operand of O^T += V^T * P^T.  V^T fragments come from the row-major V tile with ds_read_b64_tr_b16
// (hardware transpose read), so K and V are staged HBM->LDS exactly once, coalesced, never transposed in
// software.  Softmax statistics and the output accumulator are fp32; P is rounded to bf16 for the MFMA.
#include "common.h"

namespace {

struct AttnWork {   // TOK mode work item, 8 x int32 (built on the host, see host/vision_plan.py)
  int q_row0;       // first query row in Q / out
  int nq;           // queries in this tile (<= 16 * waves)
  int ka_row0;      // source A (fully visible keys): first row in KA/VA
  int ka_len;
  int kb_row0;      // source B (causal keys): first row in KB/VB
  int kb_len;
  int q_causal0;    // query i of the tile sees B key j iff j <= q_causal0 + i   (non-causal: >= kb_len)
  int pad;
};

constexpr int KT = 64;  // keys per LDS tile
#ifndef VR2_KTS
#define VR2_KTS 16
#endif
#ifndef VR2_KTP
#define VR2_KTP 64
#endif

template <int HD> struct Strides {
  static constexpr int K = HD + 8;                   // bf16 elements; 16-byte pad keeps the b64 K reads conflict-free
  static constexpr int V = (HD == 128) ? 144 : HD;   // 72-dword / 40-dword rows keep the tr_b16 reads conflict-free
};

__device__ __forceinline__ s16x4 tr_read(const bf16* p) {
  typedef __attribute__((ext_vector_type(4))) __bf16 vb4;
  vb4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(p));
  return __builtin_bit_cast(s16x4, t);
}

// K/V tile staging split in two (issue early / write late): load_tile puts the tile's 16-byte chunks into registers (global
// loads only), store_tile writes them to LDS.  attn_loop issues the loads of tile t+1 right after tile t's LDS image is
// complete, so their HBM/L2 latency runs under tile t's MFMAs instead of in front of tile t+1.
template <int HD, int NTHREADS, int KT_>
struct TileRegs {
  static constexpr int CPR = HD / 8;                                   // 16-byte chunks per row
  static constexpr int NCH = (KT_ * CPR + NTHREADS - 1) / NTHREADS;    // chunks per thread
  bf16x8 k[NCH], v[NCH];
};

template <int HD, int NTHREADS, int KT_ = KT>
__device__ __forceinline__ void load_tile(TileRegs<HD, NTHREADS, KT_>& t, const bf16* KA, const bf16* VA, const bf16* KB,
                                          const bf16* VB, long ldk, int hoff, int key0, int ka_row0, int ka_len, int kb_row0,
                                          int kend, int tid) {
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int i = 0; i < TileRegs<HD, NTHREADS, KT_>::NCH; ++i) {
    const int c = tid + i * NTHREADS;
    const int kr = c / CPR, ch = c % CPR;
    const int key = key0 + kr;
    bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < KT_ * CPR && key < kend) {
      const bool inA = key < ka_len;
      const long row = inA ? (long)(ka_row0 + key) : (long)(kb_row0 + key - ka_len);
      const long off = row * ldk + hoff + ch * 8;
      kv = *reinterpret_cast<const bf16x8*>((inA ? KA : KB) + off);
      vv = *reinterpret_cast<const bf16x8*>((inA ? VA : VB) + off);
    }
    t.k[i] = kv;
    t.v[i] = vv;
  }
}

template <int HD, int NTHREADS, int KT_ = KT>
__device__ __forceinline__ void store_tile(const TileRegs<HD, NTHREADS, KT_>& t, bf16* sK, bf16* sV, int tid) {
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int i = 0; i < TileRegs<HD, NTHREADS, KT_>::NCH; ++i) {
    const int c = tid + i * NTHREADS;
    if (c < KT_ * CPR) {
      const int kr = c / CPR, ch = c % CPR;
      *reinterpret_cast<bf16x8*>(sK + kr * Strides<HD>::K + ch * 8) = t.k[i];
      *reinterpret_cast<bf16x8*>(sV + kr * Strides<HD>::V + ch * 8) = t.v[i];
    }
  }
}

// One wave processes key range [kbeg, kend) for its 16 query slots.  qf: Q fragments.  vis_lim: per-lane causal
// limit in CONCATENATED key index (key visible iff key < vis_lim && key < kend).
// NTHREADS threads stage a tile together; NTHREADS == 64 = one wave with an LDS region of its own: no workgroup barrier (the
// other waves of the workgroup run their own loops with their own trip counts), a wave's DS operations execute in issue order.
// DEPTH tiles are in flight in registers (DEPTH register sets, the loop unrolled DEPTH times): the decode bodies are latency-bound
// walks -- one wave (suffix) or one workgroup (shared prefix) per CU streaming its keys with nothing else resident -- and with ONE
// tile in flight a walk moves a tile per memory round trip (0.8 us per 16-key tile at 128 sequences, round 3's sweep).
template <int HD, int NTHREADS, int KT_ = KT, int DEPTH = 1>
__device__ __forceinline__ void attn_loop(const s16x4* qf, float scale, int kbeg, int kend, int vis_lim, bf16* sK, bf16* sV,
                                          const bf16* KA, const bf16* VA, const bf16* KB, const bf16* VB, long ldk, int hoff,
                                          int ka_row0, int ka_len, int kb_row0, float& m_run, float& l_run, f32x4* acc) {
  constexpr int NKS = HD / 16;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int NB = KT_ / 16;   // 16-key blocks per LDS tile
  const int tid = NTHREADS == 64 ? lane : (int)threadIdx.x;
  TileRegs<HD, NTHREADS, KT_> tr[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d == 0 || kbeg + d * KT_ < kend)
      load_tile<HD, NTHREADS, KT_>(tr[d], KA, VA, KB, VB, ldk, hoff, kbeg + d * KT_, ka_row0, ka_len, kb_row0, kend, tid);
  for (int kbase = kbeg; kbase < kend; kbase += DEPTH * KT_) {
#pragma unroll
   for (int d = 0; d < DEPTH; ++d) {
    const int key0 = kbase + d * KT_;
    if (key0 >= kend) break;                 // (uniform over the staging threads)
    if constexpr (NTHREADS == 64) __builtin_amdgcn_wave_barrier(); else __syncthreads();  // previous tile fully consumed
    store_tile<HD, NTHREADS, KT_>(tr[d], sK, sV, tid);
    if constexpr (NTHREADS == 64) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    if (key0 + DEPTH * KT_ < kend)   // the tile DEPTH steps ahead: its loads fly under the MFMAs of this tile and the DEPTH - 1 next
      load_tile<HD, NTHREADS, KT_>(tr[d], KA, VA, KB, VB, ldk, hoff, key0 + DEPTH * KT_, ka_row0, ka_len, kb_row0, kend, tid);
    f32x4 s[NB];
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      const bf16* kp = sK + (b * 16 + fr) * Strides<HD>::K + fq * 4;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const s16x4 kf = *reinterpret_cast<const s16x4*>(kp + ks * 16);
        a = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf[ks], a, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + b * 16 + fq * 4 + r;
        const bool ok = (key < vis_lim) && (key < kend);
        a[r] = ok ? a[r] * scale : -INFINITY;
        mx = fmaxf(mx, a[r]);
      }
      s[b] = a;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = (m_new == -INFINITY) ? 1.0f : __expf(m_run - m_new);
    float psum = 0.f;
    s16x4 pb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      bf16x4 pv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (s[b][r] == -INFINITY) ? 0.f : __expf(s[b][r] - m_new);
        psum += p;
        pv[r] = f2bf(p);
      }
      pb[b] = __builtin_bit_cast(s16x4, pv);
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < NKS; ++db) {
      f32x4 o = acc[db];
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        // V^T fragment: 16-lane group fq reads the 4x16 block rows key 16b+4fq+{0..3}, cols 16db..16db+15
        const bf16* vp = sV + (b * 16 + fq * 4 + (fr >> 2)) * Strides<HD>::V + db * 16 + (fr & 3) * 4;
        const s16x4 vf = tr_read(vp);
        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, pb[b], o, 0, 0, 0);
      }
      acc[db] = o;
    }
   }
  }
}

// ---------------------------------------------------------------- TOK mode: prefill / vision / scoring
template <int HD>
__global__ __launch_bounds__(256) void attn_tok_kernel(const AttnWork* __restrict__ work, const bf16* __restrict__ Q, long ldq,
                                                       const bf16* __restrict__ KA, const bf16* __restrict__ VA,
                                                       const bf16* __restrict__ KB, const bf16* __restrict__ VB, long ldk,
                                                       bf16* __restrict__ O, long ldo, int group, float scale,
                                                       float* __restrict__ lse_out, int n_heads) {
  __shared__ __attribute__((aligned(16))) bf16 sK[KT * Strides<HD>::K];
  __shared__ __attribute__((aligned(16))) bf16 sV[KT * Strides<HD>::V];
  constexpr int NKS = HD / 16;
  const AttnWork w = work[blockIdx.x];
  const int head = blockIdx.y, kvh = head / group;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int qi = wave * 16 + fr;            // query index inside the tile (this lane's slot)
  const bool qvalid = qi < w.nq;
  const long qrow = w.q_row0 + (qvalid ? qi : 0);
  s16x4 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    qf[ks] = *reinterpret_cast<const s16x4*>(Q + qrow * ldq + (long)head * HD + ks * 16 + fq * 4);
  // keys needed by this workgroup: all of A plus the causal part of B visible to its last query
  long lastq = (long)w.q_causal0 + (w.nq - 1) + 1;
  const int bvis = (int)(lastq < w.kb_len ? lastq : w.kb_len);
  const int kend = w.ka_len + (bvis > 0 ? bvis : 0);
  long lim = (long)w.ka_len + (long)w.q_causal0 + qi + 1;
  const int vis_lim = qvalid ? (int)(lim < kend ? lim : kend) : 0;
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 acc[NKS];
#pragma unroll
  for (int i = 0; i < NKS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  attn_loop<HD, 256>(qf, scale, 0, kend, vis_lim, sK, sV, KA, VA, KB, VB, ldk, kvh * HD, w.ka_row0, w.ka_len, w.kb_row0,
                     m_run, l_run, acc);
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (!qvalid) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (lse_out && fq == 0) lse_out[qrow * n_heads + head] = (l > 0.f) ? m_run + __logf(l) : INFINITY;  // for the backward
  bf16* op = O + qrow * ldo + (long)head * HD + fq * 4;
#pragma unroll
  for (int db = 0; db < NKS; ++db) {
    bf16x4 o = {f2bf(acc[db][0] * inv), f2bf(acc[db][1] * inv), f2bf(acc[db][2] * inv), f2bf(acc[db][3] * inv)};
    *reinterpret_cast<bf16x4*>(op + db * 16) = o;
  }
}

// ---------------------------------------------------------------- GRP mode: decode, split over key chunks
// Sequence g belongs to prompt group p = g / seqs_per_group.  Its keys: rows [0, prefix_len[p]) of group p's prefix cache
// (KP + p * prefix_group_stride, stored ONCE per prompt), then rows g*rows_per_seq + [0, step+1) of the suffix cache.
// Partial slots per (g, head): [0, npc) prefix chunks, [npc, npc + nsc) suffix chunks; po [G][nH][npc+nsc][HD], pml [..][2].
//
// Prefix chunks, grid (npc, n_kv_heads, P * slot_blocks), 4 waves: the seqs_per_group sequences of a prompt group share the
// prefix keys, so ONE workgroup stages a key chunk once and serves seqs_per_group x group query slots (G=8 sequences x 7 q
// heads of one KV head = 56 slots on the 4 waves' 64): prefix K/V traffic and staging work drop by seqs_per_group.
template <int HD, int DP>
__device__ __forceinline__ void decode_prefix_body(bf16* sK, bf16* sV, int c, int kvh, int bz, const bf16* __restrict__ Q, long ldq,
                                                   const bf16* __restrict__ KP, const bf16* __restrict__ VP, long ldk,
                                                   const int* __restrict__ prefix_len_dev, int seqs_per_group,
                                                   long prefix_group_stride, int chunk, int nslot, int slot_blocks, int group,
                                                   int n_heads, float scale, float* __restrict__ po, float* __restrict__ pml) {
  constexpr int KTP = VR2_KTP;
  constexpr int NKS = HD / 16;
  const int grp = bz / slot_blocks, sb = bz % slot_blocks;
  const int prefix_len = prefix_len_dev[grp];
  const int kbeg = c * chunk;
  if (kbeg >= prefix_len) return;                       // uniform over the workgroup
  int kend = kbeg + chunk; if (kend > prefix_len) kend = prefix_len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int slot = sb * 64 + wave * 16 + fr;            // (sequence in group, q head in GQA group)
  const bool qvalid = slot < seqs_per_group * group;
  const int g = grp * seqs_per_group + (qvalid ? slot / group : 0);
  const int head = kvh * group + (qvalid ? slot % group : 0);
  s16x4 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    qf[ks] = *reinterpret_cast<const s16x4*>(Q + (long)g * ldq + (long)head * HD + ks * 16 + fq * 4);
  KP += (long)grp * prefix_group_stride;
  VP += (long)grp * prefix_group_stride;
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 acc[NKS];
#pragma unroll
  for (int i = 0; i < NKS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  attn_loop<HD, 256, KTP, DP>(qf, scale, kbeg, kend, qvalid ? kend : 0, sK, sV, KP, VP, KP, VP, ldk, kvh * HD, 0, prefix_len, 0, m_run,
                     l_run, acc);
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (!qvalid) return;
  const long pslot = ((long)g * n_heads + head) * nslot + c;
  if (fq == 0) { pml[pslot * 2] = m_run; pml[pslot * 2 + 1] = l; }
  float* op = po + pslot * HD + fq * 4;
#pragma unroll
  for (int db = 0; db < NKS; ++db) *reinterpret_cast<float4*>(op + db * 16) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
}

// Suffix chunks, grid (nsc, n_kv_heads, G), one wave: the keys a sequence generated itself (unique per sequence); slots =
// the q heads of one KV head.
template <int HD, int DS>
__device__ __forceinline__ void decode_suffix_body(bf16* sK, bf16* sV, int c, int kvh, int g, const bf16* __restrict__ Q, long ldq,
                                                   const bf16* __restrict__ KS, const bf16* __restrict__ VS, long ldk,
                                                   long rows_per_seq, const int* __restrict__ step, int chunk, int npc, int nslot,
                                                   int group, int n_heads, float scale, float* __restrict__ po,
                                                   float* __restrict__ pml, const int* __restrict__ finished) {
  constexpr int KTS = VR2_KTS;   // 16-key LDS tiles (9 KiB per wave), loaded one tile ahead: 44.2 / 38.7 / 36.8 us per layer for
                                 // 64- / 32- / 16-key tiles at 128 sequences, C = 1024, 256-key chunks
  constexpr int NKS = HD / 16;
  const int total = step[0] + 1;
  const int kbeg = c * chunk;
  if (kbeg >= total) return;
  int kend = kbeg + chunk; if (kend > total) kend = total;
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const bool qvalid = fr < group;
  const int head = kvh * group + (qvalid ? fr : 0);
  if (finished && finished[g]) {
    // the sequence has produced its EOS: every later token is the pad id whatever the model computes (generation/utils.py:2928-2929),
    // so its own keys are not walked any more; the chunk's slot says "empty" (weight 0 in the combine, stale values never read)
    if (qvalid && fq == 0) {
      const long pslot = ((long)g * n_heads + head) * nslot + npc + c;
      pml[pslot * 2] = -INFINITY; pml[pslot * 2 + 1] = 0.f;
    }
    return;
  }
  s16x4 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
    qf[ks] = *reinterpret_cast<const s16x4*>(Q + (long)g * ldq + (long)head * HD + ks * 16 + fq * 4);
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 acc[NKS];
#pragma unroll
  for (int i = 0; i < NKS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  attn_loop<HD, 64, KTS, DS>(qf, scale, kbeg, kend, qvalid ? kend : 0, sK, sV, KS, VS, KS, VS, ldk, kvh * HD, (int)((long)g * rows_per_seq),
                    total, 0, m_run, l_run, acc);
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (!qvalid) return;
  const long pslot = ((long)g * n_heads + head) * nslot + npc + c;
  if (fq == 0) { pml[pslot * 2] = m_run; pml[pslot * 2 + 1] = l; }
  float* op = po + pslot * HD + fq * 4;
#pragma unroll
  for (int db = 0; db < NKS; ++db) *reinterpret_cast<float4*>(op + db * 16) = make_float4(acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
}

// ONE launch for both: workgroups [0, n_suffix_blocks) take four suffix items each (one per wave, wave-local LDS tiles, no
// workgroup barrier), the rest are prefix workgroups.  The two parts read disjoint caches and write disjoint partial slots; in
// one grid they share the chip instead of running one after the other (prefix alone: P x n_kv x chunks short workgroups;
// suffix alone: a tail of single waves), and a launch boundary per layer goes away.
template <int HD, int DS, int DP>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16* __restrict__ Q, long ldq, const bf16* __restrict__ KP,
                                                          const bf16* __restrict__ VP, const bf16* __restrict__ KS,
                                                          const bf16* __restrict__ VS, long ldk, const int* __restrict__ prefix_len_dev,
                                                          int seqs_per_group, long prefix_group_stride, long rows_per_seq,
                                                          const int* __restrict__ step, int schunk, int pchunk, int npc, int nsc,
                                                          int n_suffix_blocks, int n_kv, int G, int slot_blocks, int group,
                                                          int n_heads, float scale, float* __restrict__ po, float* __restrict__ pml,
                                                          const int* __restrict__ finished) {
  constexpr int KTP = VR2_KTP, KTS = VR2_KTS;
  static_assert(KTP >= 4 * KTS, "four suffix waves share the prefix tile's LDS");
  __shared__ __attribute__((aligned(16))) bf16 sK[KTP * Strides<HD>::K];
  __shared__ __attribute__((aligned(16))) bf16 sV[KTP * Strides<HD>::V];
  const int nslot = npc + nsc;
  if ((int)blockIdx.x < n_suffix_blocks) {
    const int wave = threadIdx.x >> 6;
    const int item = (int)blockIdx.x * 4 + wave;           // (g, kvh, c), c fastest: the live chunks of a sequence sit together
    if (item >= nsc * n_kv * G) return;
    const int c = item % nsc, kvh = (item / nsc) % n_kv, g = item / (nsc * n_kv);
    decode_suffix_body<HD, DS>(sK + wave * KTS * Strides<HD>::K, sV + wave * KTS * Strides<HD>::V, c, kvh, g, Q, ldq, KS, VS, ldk, rows_per_seq,
                           step, schunk, npc, nslot, group, n_heads, scale, po, pml, finished);
  } else {
    const int b = (int)blockIdx.x - n_suffix_blocks;
    const int c = b % npc, kvh = (b / npc) % n_kv, bz = b / (npc * n_kv);
    decode_prefix_body<HD, DP>(sK, sV, c, kvh, bz, Q, ldq, KP, VP, ldk, prefix_len_dev, seqs_per_group, prefix_group_stride, pchunk, nslot,
                           slot_blocks, group, n_heads, scale, po, pml);
  }
}

// combine the chunk partials: out[g][head][:] bf16.  One wave per (head, sequence); the chunks that are live at this step are
// prefix chunks [0, ceil(prefix_len / chunk)) and suffix chunks npc + [0, ceil((step+1) / chunk)).  Pass 1: lanes take one
// chunk each -> global max; pass 2: per-chunk weights exp(m_c - M) go through LDS, every lane accumulates its 2 output dims over
// the chunks with 8 independent loads in flight (the serial one-chunk-at-a-time loop was 10 us at 64 sequences).
__global__ __launch_bounds__(64) void attn_combine_kernel(const float* __restrict__ po, const float* __restrict__ pml, int npc,
                                                          int nslot, int HD, bf16* __restrict__ out, long ldo, int n_heads,
                                                          const int* __restrict__ prefix_len_dev, int seqs_per_group,
                                                          const int* __restrict__ step, int chunk, int pchunk) {
  __shared__ float wgt[64];
  __shared__ int cidx[64];
  const int head = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
  const long base = ((long)g * n_heads + head) * nslot;
  int livep = (prefix_len_dev[g / seqs_per_group] + pchunk - 1) / pchunk;
  if (livep > npc) livep = npc;
  int lives = (step[0] + 1 + chunk - 1) / chunk;
  if (lives > nslot - npc) lives = nslot - npc;
  const int nlive = livep + lives;
  const int d = lane * 2;
  if (nlive <= 8) {
    // the usual case (a handful of live chunks): every lane reads all the chunks' (m, l) pairs itself -- the same addresses in every
    // lane, one broadcast load each -- and its two output dims of every chunk in the same batch of loads: ONE memory round trip, no
    // LDS, no shuffles (the two-pass form below waits for the maxima before it issues the value loads)
    float2 ml[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = j < livep ? j : npc + (j - livep);
      const bool on = j < nlive;
      ml[j] = on ? *reinterpret_cast<const float2*>(pml + (base + c) * 2) : make_float2(-INFINITY, 0.f);
      v[j] = (on && d < HD) ? *reinterpret_cast<const float2*>(po + (base + c) * HD + d) : make_float2(0.f, 0.f);
    }
    float Mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) Mx = fmaxf(Mx, ml[j].x);
    float Ls = 0.f, A0 = 0.f, A1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {       // chunk order, as the general path: the sums do not depend on which path ran
      const float wv = (ml[j].x == -INFINITY) ? 0.f : __expf(ml[j].x - Mx);
      Ls += ml[j].y * wv;
      if (wv != 0.f) { A0 += wv * v[j].x; A1 += wv * v[j].y; }
    }
    if (d < HD) {
      const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
      typedef __attribute__((ext_vector_type(2))) __bf16 vb2;
      vb2 ov = {f2bf(A0 * inv), f2bf(A1 * inv)};
      *reinterpret_cast<vb2*>(out + (long)g * ldo + (long)head * HD + d) = ov;
    }
    return;
  }
  float M = -INFINITY;
  for (int j = lane; j < nlive; j += 64) {
    const int c = j < livep ? j : npc + (j - livep);
    M = fmaxf(M, pml[(base + c) * 2]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
  float L = 0.f, O0 = 0.f, O1 = 0.f;
  for (int j0 = 0; j0 < nlive; j0 += 64) {
    const int j = j0 + lane;
    float wv = 0.f;
    int c = 0;
    if (j < nlive) {
      c = j < livep ? j : npc + (j - livep);
      const float m = pml[(base + c) * 2], l = pml[(base + c) * 2 + 1];
      wv = (m == -INFINITY) ? 0.f : __expf(m - M);
      L += l * wv;
    }
    __syncthreads();
    wgt[lane] = wv;
    cidx[lane] = c;
    __syncthreads();
    const int nb = (nlive - j0) < 64 ? (nlive - j0) : 64;
    if (d < HD) {
      int u = 0;
      for (; u + 8 <= nb; u += 8) {
        float2 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float2*>(po + (base + cidx[u + e]) * HD + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float ww = wgt[u + e]; if (ww != 0.f) { O0 += ww * v[e].x; O1 += ww * v[e].y; } }
      }
      for (; u < nb; ++u) {
        const float2 v = *reinterpret_cast<const float2*>(po + (base + cidx[u]) * HD + d);
        const float ww = wgt[u];
        if (ww != 0.f) { O0 += ww * v.x; O1 += ww * v.y; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) L += __shfl_xor(L, o, 64);
  if (d < HD) {
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    typedef __attribute__((ext_vector_type(2))) __bf16 vb2;
    vb2 ov = {f2bf(O0 * inv), f2bf(O1 * inv)};
    *reinterpret_cast<vb2*>(out + (long)g * ldo + (long)head * HD + d) = ov;
  }
}

}  // namespace

extern "C" int vr2_attn_tok(const void* work_i32x8, int64_t n_work, const void* Q, int64_t ldq, const void* KA,
                            const void* VA, const void* KB, const void* VB, int64_t ldk, void* O, int64_t ldo,
                            int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* lse_out_f32, void* stream) {
  VR2_REQUIRE(n_work >= 1 && n_heads >= 1 && group >= 1 && n_heads % group == 0, "vr2_attn_tok: bad head configuration");
  VR2_REQUIRE(head_dim == 80 || head_dim == 128, "vr2_attn_tok: head_dim %ld not built (80 and 128 are)", (long)head_dim);
  VR2_REQUIRE(ldq % 4 == 0 && ldk % 8 == 0 && ldo % 4 == 0, "vr2_attn_tok: strides must keep 8/16-byte alignment");
  dim3 grid((unsigned)n_work, (unsigned)n_heads);
  hipStream_t s = (hipStream_t)stream;
  if (head_dim == 128) {   // the LM: 128 query rows per workgroup (attention_fwd.hip)
    VR2_REQUIRE(ldq % 8 == 0, "vr2_attn_tok: head_dim 128 needs ldq %% 8 == 0");
    return vr2_launch_attn_fwd128(work_i32x8, n_work, Q, ldq, KA, VA, KB, VB, ldk, O, ldo, n_heads, group, scale, lse_out_f32, stream);
  }
  hipLaunchKernelGGL(attn_tok_kernel<80>, grid, dim3(256), 0, s, (const AttnWork*)work_i32x8, (const bf16*)Q, (long)ldq,
                       (const bf16*)KA, (const bf16*)VA, (const bf16*)KB, (const bf16*)VB, (long)ldk, (bf16*)O, (long)ldo,
                       (int)group, scale, (float*)lse_out_f32, (int)n_heads);
  return vr2_check_launch("vr2_attn_tok");
}

extern "C" int vr2_attn_decode(const void* Q, int64_t ldq, const void* Kprefix, const void* Vprefix, const void* Ksuffix,
                               const void* Vsuffix, int64_t ldk, const void* prefix_len_i32, int64_t seqs_per_group,
                               int64_t prefix_group_stride, int64_t rows_per_seq,
                               const void* step_i32, int64_t chunk, int64_t prefix_chunk, int64_t n_prefix_chunks,
                               int64_t n_suffix_chunks, int64_t G,
                               int64_t n_heads, int64_t group, int64_t head_dim, float scale, void* part_o_f32,
                               void* part_ml_f32, void* out_bf16, int64_t ldo, const void* finished_i32, void* stream) {
  VR2_REQUIRE(head_dim == 128, "vr2_attn_decode: head_dim %ld not built (128 is)", (long)head_dim);
  VR2_REQUIRE(group >= 1 && group <= 16 && n_heads % group == 0, "vr2_attn_decode: GQA group %ld must be in [1,16]", (long)group);
  VR2_REQUIRE(chunk % 64 == 0 && prefix_chunk % 64 == 0 && prefix_chunk >= 64 && n_prefix_chunks >= 1 && n_suffix_chunks >= 1 &&
                  G >= 1 && seqs_per_group >= 1 && G % seqs_per_group == 0,
              "vr2_attn_decode: chunk must be a multiple of 64, G a multiple of seqs_per_group");
  VR2_REQUIRE(ldq % 4 == 0 && ldk % 8 == 0, "vr2_attn_decode: strides must keep alignment");
  hipStream_t s = (hipStream_t)stream;
  const int npc = (int)n_prefix_chunks, nslot = (int)(n_prefix_chunks + n_suffix_chunks);
  const int slot_blocks = (int)((seqs_per_group * group + 63) / 64);
  const unsigned n_kv = (unsigned)(n_heads / group);
  const int P = (int)(G / seqs_per_group);
  const int n_suffix_items = (int)n_suffix_chunks * (int)n_kv * (int)G, n_suffix_blocks = (n_suffix_items + 3) / 4;
  const int n_prefix_blocks = npc * (int)n_kv * P * slot_blocks;
  // tiles in flight per suffix wave / per prefix workgroup (vr2_tune_set keys 3 / 4; defaults = the measured best)
  const int ds = vr2_tune_get(VR2_TUNE_ATTN_SUFFIX_DEPTH), dp = vr2_tune_get(VR2_TUNE_ATTN_PREFIX_DEPTH);
#define VR2_DEC_LAUNCH(DS_, DP_)                                                                                                          \
  hipLaunchKernelGGL((attn_decode_kernel<128, DS_, DP_>), dim3((unsigned)(n_suffix_blocks + n_prefix_blocks)), dim3(256), 0, s, (const bf16*)Q, \
                     (long)ldq, (const bf16*)Kprefix, (const bf16*)Vprefix, (const bf16*)Ksuffix, (const bf16*)Vsuffix, (long)ldk,          \
                     (const int*)prefix_len_i32, (int)seqs_per_group, (long)prefix_group_stride, (long)rows_per_seq,                       \
                     (const int*)step_i32, (int)chunk, (int)prefix_chunk, npc, (int)n_suffix_chunks, n_suffix_blocks, (int)n_kv, (int)G,  \
                     slot_blocks, (int)group, (int)n_heads, scale, (float*)part_o_f32, (float*)part_ml_f32, (const int*)finished_i32)
  if (ds <= 1 && dp <= 1) VR2_DEC_LAUNCH(1, 1);
  else if (ds == 2 && dp <= 1) VR2_DEC_LAUNCH(2, 1);
  else if (ds == 3 && dp <= 1) VR2_DEC_LAUNCH(3, 1);
  else if (ds == 4 && dp <= 1) VR2_DEC_LAUNCH(4, 1);
  else if (ds <= 1 && dp == 2) VR2_DEC_LAUNCH(1, 2);
  else if (ds == 2 && dp == 2) VR2_DEC_LAUNCH(2, 2);
  else if (ds == 3 && dp == 2) VR2_DEC_LAUNCH(3, 2);
  else if (ds == 4 && dp == 2) VR2_DEC_LAUNCH(4, 2);
  else if (ds == 2 && dp == 3) VR2_DEC_LAUNCH(2, 3);
  else if (ds == 3 && dp == 3) VR2_DEC_LAUNCH(3, 3);
  else VR2_DEC_LAUNCH(4, 3);
#undef VR2_DEC_LAUNCH
  if (vr2_check_launch("vr2_attn_decode(prefix + suffix)")) return -1;
  hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)n_heads, (unsigned)G), dim3(64), 0, s, (const float*)part_o_f32,
                     (const float*)part_ml_f32, npc, nslot, (int)head_dim, (bf16*)out_bf16, (long)ldo, (int)n_heads,
                     (const int*)prefix_len_i32, (int)seqs_per_group, (const int*)step_i32, (int)chunk, (int)prefix_chunk);
  return vr2_check_launch("vr2_attn_decode(combine)");
}
