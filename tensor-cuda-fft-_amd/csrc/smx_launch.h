// smx_launch.h -- pieces shared by the kernel translation units (smx_decim.hip, smx_fourstep.hip):
// the LDS declaration, the workgroup -> work item map and the launch-in-rounds helper.
#pragma once
#include "smx_kernels.h"

namespace smx {

// One LDS array only (guide: a second __shared__ object can de-pipeline the loop).
// 2 x 32 KiB exchange buffers; 2 workgroups per CU fit in the 160 KiB LDS.
// (the unpack exchange publishes at most 32 slots per thread per round = the same 64 KiB)
// (+ 2 KiB so that the staged filter tile of the NB == 1 kernels, WL_ELEMS, fits behind the first buffer)
#define SMX_LDS_DECL __shared__ cf lds[EX + WL_ELEMS]
static_assert(WL_ELEMS >= EX, "the second exchange buffer lives in the same space");

// ---- workgroup -> (batch row, d-tile, residue chunk, residue rotation) ---------------------------
// Blocks are dealt round-robin over the 8 XCDs (bid % 8), each with its own L2.  Within one tile
// every row a workgroup touches has the same address bits [7..13] (at D = 256: d-tile -> bits 7-9,
// residue -> bits 10-13), so the naive b-major order makes all workgroups of an XCD hit the same L2 channel
// slot at the same time.  map == 2 hands each XCD all d-tiles and a spread of residue phases (all 64
// (d-tile pair, residue) combinations once per 64 workgroups) and batch rows 8 apart: measured
// +11 % read and +15 % write bandwidth on the same access pattern (tools/probe_stride.hip).
// Placement only affects speed: every mapping is a bijection onto the same work items.
struct WgItem { int b, dt, c, rot; };
__device__ __forceinline__ WgItem wg_map(int bid, int B, int ndt, int nsplit, int lc, int map) {
  WgItem w;
  const int per = B * nsplit;                 // (b, c) pairs per d-tile
  // map == 3 | a << 8 | b << 16: rotation lattice rot = (a l2 + b dt) mod lc, for tools/rot_scan.py
  const int ra = (map >> 8) & 0xff, rb = (map >> 16) & 0xff;
  map &= 0xff;
  if ((map == 2 || map == 3) && per % 8 == 0 && (B % 8 == 0 || B == 1 || 8 % B == 0)) {
    const int x = bid & 7, l = bid >> 3;
    w.dt = l % ndt;
    const int l2 = l / ndt;                   // 0 .. per/8 - 1
    if (B % 8 == 0) { const int g = B / 8; w.b = x + 8 * (l2 % g); w.c = l2 / g; }
    else { const int g = 8 / B; w.b = x % B; w.c = (x / B) + g * l2; }      // B in {1,2,4}: XCDs share rows
    w.rot = map == 3 ? (l2 * ra + w.dt * rb) % lc : (l2 + (lc >> 1) * (w.dt & 1)) % lc;
    return w;
  }
  w.c = bid % nsplit;
  const int wg = bid / nsplit;
  w.b = wg / ndt; w.dt = wg % ndt;
  w.rot = map == 1 ? (int)(((unsigned)bid * 7u) % (unsigned)lc) : 0;
  return w;
}

// ---- launch helpers ----------------------------------------------------------------------------
static inline int n_wg(const DecimArgs& a) { return a.g.B * ((a.g.D + DT - 1) / DT); }

// The streaming kernels are launched in rounds of `a.round` workgroups (512 = 2 per CU, all resident):
// the kernel boundary keeps every round's read phase and write phase chip-wide in step.  One launch of
// 1024 workgroups lets the second round's reads run into the first round's writes, and mixed traffic is
// slower on this HBM.  Measured gain is small (1-2 % at (64,4096,512), (128,4096,256) and C3); the
// four-band kernels (one workgroup per CU) are faster in a single launch and keep that.
template <typename F>
static inline hipError_t for_rounds(const DecimArgs& a, int total, F launch, bool single = false) {
  const int round = a.round > 0 && !single ? a.round : total;
  for (int b0 = 0; b0 < total; b0 += round) {
    DecimArgs r = a;
    r.bid0 = b0;
    launch(r, dim3(total - b0 < round ? total - b0 : round));
  }
  return hipGetLastError();
}

}  // namespace smx
