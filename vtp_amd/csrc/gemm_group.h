// Argument records of the grouped weight-gradient launch (vtp_gemm_tn_grouped): shared by the 8-phase kernel (gemm8p.hip) and the
// one-wave-per-SIMD kernel (gemm4w_tn.hip).
#pragma once
#include "gemm_common.h"

namespace vtp {

struct GroupProblem {  // 64-bit fields: written by the host as an int64 tensor
  const bf16* A;
  const bf16* B;
  float* C;
  float* colsum;
  long lda, ldb, ldc;
  long M, N;
  long c_grp, c_pre;
  long tile0;       // first tile of this problem in the launch's tile list
  long accumulate;  // 1: C += result, 0: C = result
  long pad[3];
};
struct GroupItem {  // one workgroup of an item-list launch (vtp_gemm_tn_grouped_items): 8 x int32, written by the host
  int tile;           // index in the launch's tile list
  int kbeg, kcount;   // its K range (kbeg a multiple of 64, kcount of 8)
  int nparts, part;   // workgroups sharing the tile, and this one's slot among them
  int pad[3];
};
struct GroupArgs {
  const GroupProblem* probs;
  float* part;
  int* ticket;
  int nprob, ntiles, splits, K, k_split;  // splits: slices per tile (uniform launches) | partial-sum slots per tile (item lists)
  unsigned long long* timing;
  const GroupItem* items;  // null: uniform tiles x splits geometry
};

}  // namespace vtp
