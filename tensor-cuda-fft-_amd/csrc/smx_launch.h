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

// ---- the streamed tensors as raw buffers (round 4) -----------------------------------------------
// One batch row of x / y / g / grad_x is described to the hardware as a raw buffer (base + byte count), and a tile
// access is buffer_load / buffer_store_dwordx2 with ONE per-thread 32-bit offset (row t L + r, this thread's channel
// pair) and the row pitch of the thread's 16 rows, u 16 L D 4 bytes, as the instruction's SCALAR offset.  Against
// global_load with a 64-bit address per row this removes 16 v_lshl_add_u64 + the 64-bit row arithmetic per tile
// (about 30 of the loop's 530 vector instructions in round 3), halves the SGPRs the row pitches occupy, and states
// the streaming policy in the instruction itself: round 3's __builtin_nontemporal_store lost its hint on 12 of the
// 16 stores of a tile somewhere in the optimiser (llvm-objdump of the shipped kernel: 4 x `nt`, 12 x plain).
// Zero-padded rows (PAD: x / y hold R < N rows): the row pitch moves into the per-thread offset, and the buffer's
// range check (offset >= R D 4 bytes: loads return 0, stores are dropped) is the predicate.
// Needs R D 4 < 2^31 (make_plan sends larger batch rows to the direct plan).
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int BUF_NT = 2;          // cache policy operand of the buffer intrinsics: slc = `nt` on gfx950
struct RowBuf {
  __amdgpu_buffer_rsrc_t rs;       // batch row b: base + b R D floats, R D 4 bytes
  unsigned vo;                     // (t L D + d) 4: this thread's channel pair in row t L
  unsigned su;                     // 16 L D 4: bytes between a thread's consecutive rows
  unsigned rowb;                   // D 4
};
// in_range = false (a lane whose channel pair lies past D): every offset the lane forms is 2^31 + (an offset inside
// the batch row) -- in [2^31, 2^32), past any buffer and short of wrapping: its stores are dropped and its loads
// return 0 without a branch around the tile code
__device__ __forceinline__ RowBuf row_buf(const float* row0, const Geom& g, int t, int d, bool in_range = true) {
  RowBuf rb;
  rb.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row0), 0, (int)((unsigned)g.R * (unsigned)g.D * 4u),
                                            0x00020000);
  rb.rowb = (unsigned)g.D * 4u;
  rb.vo = in_range ? ((unsigned)t * (unsigned)g.L * (unsigned)g.D + (unsigned)d) * 4u : 0x80000000u;
  rb.su = 16u * (unsigned)g.L * rb.rowb;                 // (uniform: it is the instructions' scalar offset)
  return rb;
}
// rows u = U0 .. U0+CNT-1 of tile r: row (t + 16 u) L + r
template <int U0, int CNT, bool PAD>
__device__ __forceinline__ void load_rows(const RowBuf& rb, int r, cf (&v)[16]) {
  const unsigned vr = rb.vo + (unsigned)r * rb.rowb;
#pragma unroll
  for (int u = U0; u < U0 + CNT; ++u) {
    const u32x2 w = PAD ? __builtin_amdgcn_raw_buffer_load_b64(rb.rs, vr + (unsigned)u * rb.su, 0, BUF_NT)
                        : __builtin_amdgcn_raw_buffer_load_b64(rb.rs, vr, (unsigned)u * rb.su, BUF_NT);
    const unsigned wx = w.x, wy = w.y;      // (bit_cast straight from a vector ELEMENT reads element 0 twice: clang 22)
    float fx = __builtin_bit_cast(float, wx), fy = __builtin_bit_cast(float, wy);
    // two scalars from here on: left as <2 x float> the optimiser turns the first butterflies into v_pk_add_f32,
    // which issue at half the rate of the scalar adds with two waves per SIMD (tools/probe_valu.hip, round 3)
    asm("" : "+v"(fx));
    asm("" : "+v"(fy));
    v[u] = mk(fx, fy);
  }
}
// plain (wave-uniform: 0, 1, 2 or 4 -- DecimArgs::st_plain): the thread's first `plain` rows go out with the default
// write-back policy, the others with the streaming hint.  About 64 MiB of an output tensor written back through
// L2 / Infinity Cache costs nothing (it drains under the next launch's reads); all 16 rows streaming were 9 % slower
// per C2 step, all 16 cached 13 % (profiles/r04_store_policy.txt).  Four scalar branches per tile.
template <bool PAD>
__device__ __forceinline__ void store_rows(const RowBuf& rb, int r, const cf (&v)[16], int plain) {
  const unsigned vr = rb.vo + (unsigned)r * rb.rowb;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    u32x2 w;
    const float fx = v[u].x, fy = v[u].y;
    w.x = __builtin_bit_cast(unsigned, fx); w.y = __builtin_bit_cast(unsigned, fy);
    const unsigned vo = PAD ? vr + (unsigned)u * rb.su : vr, so = PAD ? 0u : (unsigned)u * rb.su;
    if (u < 4 && u < plain) __builtin_amdgcn_raw_buffer_store_b64(w, rb.rs, vo, so, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(w, rb.rs, vo, so, BUF_NT);
  }
}
#else
// host pass of hipcc: the kernels' bodies are parsed but never emitted -- declarations only
struct RowBuf { unsigned vo, su, rowb; };
__device__ RowBuf row_buf(const float* row0, const Geom& g, int t, int d, bool in_range = true);
template <int U0, int CNT, bool PAD> __device__ void load_rows(const RowBuf& rb, int r, cf (&v)[16]);
template <bool PAD> __device__ void store_rows(const RowBuf& rb, int r, const cf (&v)[16], int plain);
#endif

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
