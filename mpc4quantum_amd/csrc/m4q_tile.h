// m4q_tile.h - the two sweeps of one QP solve on fp64 matrix-core tiles (v_mfma_f64_4x4x4_4b_f64), real arithmetic.
//
// Why.  In the DPP layout of m4q_device.h one 16-lane row owns one member and a lane owns one matrix column: at d = 3 that
// fills 9 (8 traceless) of 16 lanes, every cross-lane sum is a chain of unit-multiplier FMAs, and the fp64 pipe spends
// ~2,400 cycles per horizon index of the backward sweep on four members.  v_mfma_f64_4x4x4_4b_f64 runs FOUR independent
// 4x4x4 products per instruction at the same pipe rate (16.7 cycles, tools/ubench_mfma.hip) with every slot useful when
// the matrices are cut into 4x4 tiles - and an n = 8 matrix (the traceless coordinates of a qutrit, m4q_mpc.h) is exactly
// 2 x 2 tiles.  Reductions come for free: a product with a replicated operand IS the broadcast.
//
// Lane map of the instruction on gfx950, found by experiment (tools/mfma_layout.hip, profiles/r03_mfma_layout.txt):
//     block (= member) mb = (lane >> 2) & 3,   r = lane >> 4,   q = lane & 3          lane = 16 r + 4 mb + q
//     A operand:  A[i][k]  in lane (r = k, q = i)        B operand:  B[k][j]  in lane (r = k, q = j)
//     C / D:      D[i][j]  in lane (r = i, q = j)
// So with every 4x4 tile held "naturally" - X[r][q] in lane (r, q) of the member's 16 lanes - one instruction computes
//     mm(X, Y, C) = C + X^T Y                      (the A operand reads its tile transposed)
// and nothing ever has to be moved between lanes:
//   * P is symmetric and the recursion only needs P S = P^T S and S^T (P S): both are X^T Y forms of natural tiles;
//   * a vector is kept ROW-REPLICATED (v[4K + r] in every q of tile K): then  rowrep(M^T v) = sum_K mm(M[K][I], v[K]),
//     colrep(M^T v) = sum_K mm(v[K], M[K][J])  and a dot product  sum_K mm(x[K], y[K])  lands REPLICATED in all 16 lanes;
//   * an outer product b k^T is an elementwise product of a row-replicated b and a column-replicated k.
// The only values that cross lanes outside an MFMA are the m x m matrix G = R + B^H P B and h (m <= 3): they come out of one
// product as a natural tile and are spread to the member's lanes through 16 doubles of LDS.
//
// Scope: the BACKWARD sweep of the clipped Riccati solve of the closed loop (riccati_backward of m4q_mpc.h, same arithmetic: Joseph
// form, Delta_t = -B_t u_t, xbar_{t+1}) for real order-1 models on NS coordinates with a constant target over the window; the
// implementation is m4q_tile3.h (operands fetched four horizon indices at a time).  The rollout, and everything else
// (M4Q_QP_REF_LQR, the exact box QP, time-varying targets, order-2 libraries, complex models), stays on the DPP rows: a rollout on
// tiles was built in round 3 and lost (3,060 against 1,320 cycles per index: one short dependent chain per index, every MFMA
// waiting for the one before; profiles/r03_tile_log.txt), as did the per-index form of this sweep inside the kernel
// (profiles/r04_tile_chain.txt).
//
// Reference arithmetic: mpc4quantum/lqr.py:28-79 (+ Delta, optimize.py:41), mpc4quantum/linearize.py:43-70.
#pragma once
#include "m4q_mpc.h"

namespace m4q {

struct TileGeo {
  int mb, r, q;
  __device__ __forceinline__ TileGeo() {
    const int lane = threadIdx.x;
    r = lane >> 4;
    mb = (lane >> 2) & 3;
    q = lane & 3;
  }
};

// C + X^T Y on the four members' tiles at once
__device__ __forceinline__ double mm(double x, double y, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, c, 0, 0, 0); }

// Hand-over between the closed-loop state machine (DPP rows: member = lane >> 4) and the tile sweep (member = (lane >> 2) & 3):
// per member 4 words of LDS, and the 16 doubles through which G = R + B^H P B and h reach all of a member's lanes.
constexpr int TILE_IO_WORDS = 4;        // [0] flags (1 running) | [1] xbm byte offset | [2] ubm byte offset | [3] exact mode: working-set byte offset
constexpr int TILE_GB_DOUBLES = 18;     // the G / h tile of one member: 16 doubles at a pitch of 18 - every lane of a member reads the
                                        // same entry, and at pitch 16 (128 bytes) members 0 / 2 and 1 / 3 read the same banks: 46 % of the
                                        // tile kernel's LDS-active cycles were bank conflicts (profiles/r04_tile_chain.txt)
constexpr int TILE_LDS_BYTES = 4 * (TILE_IO_WORDS * 4 + TILE_GB_DOUBLES * 8);
// exact mode (pinned sweep on tiles): per member also the working-set view's byte offset (word [3]) and the first control's band
// (a member's block sits at a constant distance behind its gains block: one address register serves both)
constexpr int TILE_PIN_DOUBLES = TILE_GB_DOUBLES;     // [0, 3) lo0 | [3, 6) hi0 | spare
constexpr int TILE_PIN_LDS_BYTES = 4 * TILE_PIN_DOUBLES * 8;
constexpr int TILE_PIN_OFFSET = 4 * TILE_GB_DOUBLES + 4 * TILE_IO_WORDS / 2;      // in doubles, from a member's gb block

}  // namespace m4q
