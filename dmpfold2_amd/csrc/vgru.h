// Declarations shared by the three translation units of the vertical GRU (vgru.hip: split-f16 products, the fast mode;
// vgru_f32.hip: the reference's float32 instructions; vgru_x3.hip: full-width operands as three bf16 pieces, the default).
#pragma once
#include "common.h"

namespace dmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 vg_f16x8 __attribute__((ext_vector_type(8)));

// Per-context constants of the step kernel (baked into the hipGraph nodes) ...
struct VStatic {
  const uint4* wx[2];       // [layer]: input weight pieces   [2][3][KQ][512] x 16 bytes (KQ = 4 / 64)
  const uint4* wh[2];       // [layer]: hidden weight pieces  [2][3][64][512] x 16 bytes
  const float* bias[2];     // [layer]: [4][512]: r (b_ir+b_hr), z (b_iz+b_hz), b_in, b_hn
  float inv_scale[2];
  float* hT[2][2];          // [layer][parity] float32 state [128][Lb][4]
  uint16_t* hH[2][2];       // [layer][parity] f16 pieces of 1024*state [2][64][Lb][8]
};
constexpr int VG_TB = 32;              // columns per tile


// column pitch of a member's state: whole 32-column tiles
__host__ __device__ inline int vgru_pitch(int L) { return (L + VG_TB - 1) / VG_TB * VG_TB; }

typedef unsigned vg_u32x4 __attribute__((ext_vector_type(4)));

constexpr int VG_MAX_MEMBERS = 8;                         // contexts served by one group launch

// The state of a group lives in the LEADER's buffers as one wide alignment: member m owns the column tiles
// [tile0, tile0 + ceil(L / 32)), the column pitch is 32 x (number of tiles of the group).  A step kernel therefore
// derives every address of its main loop from its kernel arguments; the record below (device memory, cold at
// every kernel start) is only needed for what happens after the loop: is the tile active at this row, and which
// residue codes does layer 0 read.
struct VMember { const uint8_t* msa; int N, L, tile0, pad; };
struct VGroupRec { int t0, t_end, nmem, pad; VMember mem[VG_MAX_MEMBERS]; };
static_assert(sizeof(VGroupRec) <= 256, "dmp_ctx_create sizes vgru_run for 256 bytes");

typedef float vp_f32x4 __attribute__((ext_vector_type(4)));
constexpr int VP_WL0_SLOTS = 4 * 4 * 6 * 64;                 // [wave][k-step][piece x gate][lane] 16-byte slots: 98 304 B
constexpr int VP_RED_SLOTS = 4 * 14 * 64;                    // [wave][accumulator][lane] float4: 57 344 B
constexpr int VP_TAB_FLOATS = 3 * 24 * 16;                   // [gate][code][row] one-hot input terms: 4 608 B
constexpr int VP_LDS_BYTES = (VP_WL0_SLOTS + VP_RED_SLOTS) * 16 + VP_TAB_FLOATS * 4;     // 160 256 of 163 840
constexpr int VP_MAX_XCD_TILES = 64;                           // (+ 1.1 KB of static LDS: the XCD's tile table)
constexpr int VP_GRID = 256;                                 // one workgroup per CU, 32 per XCD
struct VPSync { unsigned count[8]; unsigned pad[24]; unsigned flag[8][32]; };              // zeroed before every launch
static_assert(sizeof(VPSync) == 128 + 1024, "VPSync layout");
// polls of a row barrier before it gives up (about 0.3 us each: a quarter of a second).  The FIRST barrier of a launch is
// also the residency wait - its 256 workgroups become resident as the kernels of other engines leave the CUs, and a
// float32 convolution at L ~ 2048 runs for 0.1-0.2 s - so it gets eight times the bound (two seconds); later rows only
// wait for workgroups that are known to be running.
constexpr unsigned VP_BARRIER_SPINS = 800000u;
constexpr unsigned VP_BARRIER_SPINS_FIRST = 8u * VP_BARRIER_SPINS;


// vgru_f32.hip: rows [t_lo, t_hi) of the group set up on `lead` in float32 (option "vgru_f32")
int vgru_f32_group_steps(dmp_ctx* lead, int t_lo, int t_hi, hipStream_t s);
int vgru_f32_kernel_attrs(dmp_ctx* c);
// vgru_x3.hip: the same rows with every float32 operand as three exact bf16 pieces (option "vgru_f32" = 2)
int vgru_x3_group_steps(dmp_ctx* lead, int t_lo, int t_hi, hipStream_t s);
int vgru_x3_kernel_attrs(dmp_ctx* c);

}  // namespace dmp
