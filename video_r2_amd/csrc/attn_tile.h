// LDS tile image and fragment reads shared by the head_dim-128 attention kernels (attention_fwd.hip, attention_bwd.hip).
// A swept tile = 64 rows x 256 B (128 bf16 features), filled by direct-to-LDS loads (one 1-KiB piece = 4 rows per wave
// instruction) as plain 256-byte rows whose 16-byte chunks are XOR-swizzled with ((row & 3) << 2) | ((row >> 2) & 3) --
// applied to the per-lane SOURCE address of the load and again to the read address -- so that ONE image serves the row reads
// (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16) of v_mfma_f32_16x16x32_bf16 operands without bank conflicts.
// MFMA row i of a 16-row block b of a 32-row half stands for tile row (i & 3) + 8 (i >> 2) + 4 b: the accumulators of blocks 0
// and 1 then hold, per lane, 8 CONSECUTIVE rows 8 (lane >> 4) .. + 7 of the half, which is exactly the k order the next MFMA's
// B operand wants, so P / dS feed it from registers.
#pragma once
#include "common.h"

namespace {

struct AttnWork { int q_row0, nq, ka_row0, ka_len, kb_row0, kb_len, q_causal0, pad; };

constexpr int HD = 128;
constexpr int TILE_B = 64 * 256;                 // one swept tile: 64 rows x 256 B
constexpr int ROWC_B = 512;                      // dkv: 4 waves x [16 x -lse*log2e | 16 x delta] fp32
constexpr int KV_STAGE_B = 2 * TILE_B + ROWC_B;  // dkv ring stage: Q tile | dO tile | row constants
constexpr int KV_NSTAGE = 3;
constexpr int DQ_STAGE_B = 2 * TILE_B;           // dq ring stage: K tile | V tile
constexpr int DQ_NSTAGE = 2;
constexpr int OUT_STRIDE = 132;                  // fp32 words per staged output row (528 B: conflict-free 16-byte writes)
constexpr int OUT_STRIDE_BF = 136;               // bf16 elements per staged output row (272 B)
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// A operand read by rows: MFMA row i = tile row 32 h + (i & 3) + 8 (i >> 2) + 4 b, k = features 32 ks + 8 fq .. + 7
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int h, int b, int ks, int lane) {
  const int i = lane & 15, fq = lane >> 4;
  const int row = 32 * h + (i & 3) + 8 * (i >> 2) + 4 * b;
  return *reinterpret_cast<const bf16x8*>(tile + img_off(row, ks * 4 + fq));
}
// A operand read transposed: MFMA row i = feature 16 db + i, k = tile rows 32 h + 8 fq .. + 7 (two 4-row blocks)
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int h, int db, int lane) {
  typedef __attribute__((ext_vector_type(4))) __bf16 vb4;
  const int fq = lane >> 4, a = (lane & 15) >> 2, p = lane & 3;
  const int row = 32 * h + 8 * fq + a, ch = db * 2 + (p >> 1);
  const vb4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(tile + img_off(row, ch) + 8 * (p & 1)));
  const vb4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) vb4*)(tile + img_off(row + 4, ch) + 8 * (p & 1)));
  bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return r;
}

// epilogue of the query-centric kernels: the wave's (up to) 32 staged bf16 rows leave as whole 256-byte rows, 4 rows per
// wave instruction (16 bytes per lane)
__device__ __forceinline__ void store_rows_bf16(const bf16* stg, bf16* dst, long ld, int nrows, int lane) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int rr = g * 4 + (lane >> 4);
    if (rr < nrows) *reinterpret_cast<bf16x8*>(dst + (long)rr * ld + (lane & 15) * 8) = *reinterpret_cast<const bf16x8*>(stg + rr * OUT_STRIDE_BF + (lane & 15) * 8);
  }
}

}  // namespace
