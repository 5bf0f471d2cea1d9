// ht_block.hpp -- the two-body rows of k_solve resolved a BLOCK at a time (round 5).
//
// LimitLinear::Iter / LimitAngular::Iter (physics.h:289-307, 251-265) read the momenta their predecessors wrote, which is what made the two-body phases a
// chain of levels (one LDS round trip of the momenta and ~35 issued instructions per row, ~240 clocks per row on the critical path).  But a row depends on the
// rows before it only LINEARLY: with M0 the momenta before a block of consecutive rows,
//     x_j = (T_j - c_j . M0) k_j  +  sum_{i<j} G(j,i) d_i,      d_i = clamp(x_i, lo_i, hi_i),      G(j,i) = -k_j c_j . D_i
// (c_j: the row's velocity coefficients, k_j its 1 / effective mass, D_i: what a unit impulse of row i adds to the momenta; G(j,i) = 0 unless the rows share a
// body).  So a block's rows sit ONE PER LANE: every lane forms its c_j . M0 side by side with the others, then the impulses are resolved in the reference's row
// order -- clamp, v_readlane, one multiply-add on the lanes behind: 27 clocks per row measured (tools/probe/block_probe.hip) -- and the momenta are brought up to
// date once per block.  Same rows, same order, same clamps; another association order of the same sums (G d in place of c . (D d)).
//
// Couplings live in registers, statically indexed: a block has at most 32 rows, lane m's row needs G(m, i) for i < m only, so the unused half of a lane's 32
// coupling registers holds a SECOND block stored backwards (row p of it on lane 31 - p, its coupling to row i in register 31 - i), and lanes 32..63 hold two more
// blocks the same way: 32 registers serve 128 rows.  Block Q of a set: Q = 0 lanes 0..31 forwards, 1 lanes 31..0, 2 lanes 32..63, 3 lanes 63..32.
// While a block is resolved only the lanes still waiting are enabled (the enabled set shrinks by one lane per row): a lane that is through keeps its impulse, and
// the registers of the block it shares its lanes with are never applied to it.
#pragma once
#define BLK_LROWS 30      // rows of a block of two-body linear rows: ten triples

// one row: the lane LN clamps (its value is final now), leaves the enabled set, and the lanes behind it take its impulse through coupling register g<GN>
#define BLK_STEP(LN, SH, GN) \
	"v_med3_f32 %[imp], %[x], %[lo], %[hi]\n\t" \
	SH "\n\t" \
	"v_readlane_b32 %[s], %[imp], " #LN "\n\t" \
	"v_fmac_f32 %[x], %[s], %[g" #GN "]\n\t"
#define BLK_SHL_LO "s_lshl_b32 exec_lo, exec_lo, 1"
#define BLK_SHR_LO "s_lshr_b32 exec_lo, exec_lo, 1"
#define BLK_SHL_HI "s_lshl_b32 exec_hi, exec_hi, 1"
#define BLK_SHR_HI "s_lshr_b32 exec_hi, exec_hi, 1"
#define BLK_16(SH, l0, l1, l2, l3, l4, l5, l6, l7, l8, l9, l10, l11, l12, l13, l14, l15) \
	BLK_STEP(l0, SH, 0) BLK_STEP(l1, SH, 1) BLK_STEP(l2, SH, 2) BLK_STEP(l3, SH, 3) BLK_STEP(l4, SH, 4) BLK_STEP(l5, SH, 5) BLK_STEP(l6, SH, 6) BLK_STEP(l7, SH, 7) \
	BLK_STEP(l8, SH, 8) BLK_STEP(l9, SH, 9) BLK_STEP(l10, SH, 10) BLK_STEP(l11, SH, 11) BLK_STEP(l12, SH, 12) BLK_STEP(l13, SH, 13) BLK_STEP(l14, SH, 14) BLK_STEP(l15, SH, 15)
#define BLK_ENTER(ELO, EHI) "s_mov_b64 %[save], exec\n\ts_mov_b32 exec_lo, " ELO "\n\ts_mov_b32 exec_hi, " EHI "\n\t"
#define BLK_LEAVE "s_mov_b64 exec, %[save]"

// A statement can stop early: its control word cw (the same value on every lane) goes to a scalar register of the statement's own, and a scalar test in front of a group of
// rows leaves when the block has no row there (PARTIAL statements only: a full statement carries no tests).  The exec shifts and the tests write SCC: named as a clobber.
#define BLK_CW "v_readfirstlane_b32 %[m], %[cw]\n\t"
#define BLK_EXIT(BIT) "s_bitcmp0_b32 %[m], " #BIT "\n\ts_cbranch_scc1 9f\n\t"
#define BLK_16P(SH, l0, l1, l2, l3, l4, l5, l6, l7, l8, l9, l10, l11, l12, l13, l14, l15) BLK_CW \
	BLK_STEP(l0, SH, 0) BLK_STEP(l1, SH, 1) BLK_STEP(l2, SH, 2) BLK_STEP(l3, SH, 3) BLK_EXIT(1) BLK_STEP(l4, SH, 4) BLK_STEP(l5, SH, 5) BLK_STEP(l6, SH, 6) BLK_STEP(l7, SH, 7) \
	BLK_EXIT(2) BLK_STEP(l8, SH, 8) BLK_STEP(l9, SH, 9) BLK_STEP(l10, SH, 10) BLK_STEP(l11, SH, 11) BLK_EXIT(3) BLK_STEP(l12, SH, 12) BLK_STEP(l13, SH, 13) BLK_STEP(l14, SH, 14) BLK_STEP(l15, SH, 15) "9:\n\t"
// Rows 16*HALF .. 16*HALF+15 of block Q (angular rows).  x: (T - c.M0) k of this lane's row, updated in place; lo / hi: its impulse limits less what it has
// accumulated; imp: receives the row's impulse (lanes outside the block keep what they hold); G: the lane's coupling registers.  PARTIAL: cw bit g (1..3) = the block has
// a row among the statement's rows 4g .. 4g+3.
template <int Q, int HALF, bool PARTIAL>
__device__ __forceinline__ void blk_resolve16(float &x, const float lo, const float hi, float &imp, const float (&G)[32], const int cw)
{
	int s, m; long long save;
#define BLK_G(k) (G[(Q & 1) ? 31 - (16 * HALF + (k)) : 16 * HALF + (k)])
#define BLK_GS [g0] "v"(BLK_G(0)), [g1] "v"(BLK_G(1)), [g2] "v"(BLK_G(2)), [g3] "v"(BLK_G(3)), [g4] "v"(BLK_G(4)), [g5] "v"(BLK_G(5)), [g6] "v"(BLK_G(6)), [g7] "v"(BLK_G(7)), \
	  [g8] "v"(BLK_G(8)), [g9] "v"(BLK_G(9)), [g10] "v"(BLK_G(10)), [g11] "v"(BLK_G(11)), [g12] "v"(BLK_G(12)), [g13] "v"(BLK_G(13)), [g14] "v"(BLK_G(14)), [g15] "v"(BLK_G(15))
#define BLK_OPS : [imp] "+v"(imp), [x] "+v"(x), [s] "=&s"(s), [save] "=&s"(save) : [lo] "v"(lo), [hi] "v"(hi), BLK_GS : "scc"
#define BLK_OPSP : [imp] "+v"(imp), [x] "+v"(x), [s] "=&s"(s), [m] "=&s"(m), [save] "=&s"(save) : [lo] "v"(lo), [hi] "v"(hi), [cw] "v"(cw), BLK_GS : "scc"
#define BLK_BOTH(ELO, EHI, SH, ...) { if constexpr (PARTIAL) asm volatile(BLK_ENTER(ELO, EHI) BLK_16P(SH, __VA_ARGS__) BLK_LEAVE BLK_OPSP); else asm volatile(BLK_ENTER(ELO, EHI) BLK_16(SH, __VA_ARGS__) BLK_LEAVE BLK_OPS); }
	if constexpr (Q == 0 && HALF == 0) BLK_BOTH("-1", "0", BLK_SHL_LO, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
	if constexpr (Q == 0 && HALF == 1) BLK_BOTH("0xffff0000", "0", BLK_SHL_LO, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
	if constexpr (Q == 1 && HALF == 0) BLK_BOTH("-1", "0", BLK_SHR_LO, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16)
	if constexpr (Q == 1 && HALF == 1) BLK_BOTH("0x0000ffff", "0", BLK_SHR_LO, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0)
	if constexpr (Q == 2 && HALF == 0) BLK_BOTH("0", "-1", BLK_SHL_HI, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47)
	if constexpr (Q == 2 && HALF == 1) BLK_BOTH("0", "0xffff0000", BLK_SHL_HI, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63)
	if constexpr (Q == 3 && HALF == 0) BLK_BOTH("0", "-1", BLK_SHR_HI, 63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48)
	if constexpr (Q == 3 && HALF == 1) BLK_BOTH("0", "0x0000ffff", BLK_SHR_HI, 47, 46, 45, 44, 43, 42, 41, 40, 39, 38, 37, 36, 35, 34, 33, 32)
#undef BLK_BOTH
#undef BLK_OPS
#undef BLK_OPSP
#undef BLK_GS
#undef BLK_G
}

// ---- two-body LINEAR rows: blocks of 30 = ten groups of three (a joint's x, y, z rows or a contact's normal and two friction rows) ----
// A statement of joint triples only is fifteen plain steps.  The GENERAL statement serves contacts and the block's last, partly filled, half.  The friction rows' limits
// follow the normal row's impulse sum (physics.h:292): behind the head of a contact triple the two lanes after it set their limits to -+ mu * (the normal row's new sum),
// written as the row-by-row sweep writes it: lim = (mu * sum) * (1 / dt), hi = lim * dt, lo = (-lim) * dt, each less the row's own sum.  cw: bit T (0..4) = triple T of the
// statement is a contact, bit 8 + T = the block has triple T (scalar tests on the control word: a joint's head costs two scalar instructions, the statement ends at the first triple
// the block does not have); mp: the position of this lane's head row if the lane holds a friction row, else 255; fms: the head row's sum before this sweep.
#define BLK_HEAD(LN, SH, GN, T, TP, POS) BLK_EXIT(TP) BLK_STEP(LN, SH, GN) \
	"s_bitcmp1_b32 %[m], " #T "\n\t" \
	"s_cbranch_scc0 1f\n\t" \
	"v_add_f32 %[t0], %[s], %[fms]\n\t" \
	"v_mul_f32 %[t0], %[mu], %[t0]\n\t" \
	"v_mul_f32 %[t0], %[idt], %[t0]\n\t" \
	"v_mul_f32 %[t1], %[dt], %[t0]\n\t" \
	"v_mul_f32_e64 %[t0], -%[t0], %[dt]\n\t" \
	"v_cmp_eq_u32_e32 vcc, " #POS ", %[mp]\n\t" \
	"v_sub_f32 %[t1], %[t1], %[own]\n\t" \
	"v_sub_f32 %[t0], %[t0], %[own]\n\t" \
	"v_cndmask_b32 %[hi], %[hi], %[t1], vcc\n\t" \
	"v_cndmask_b32 %[lo], %[lo], %[t0], vcc\n\t" \
	"1:\n\t"
#define BLK_15G(SH, P0, P3, P6, P9, P12, l0, l1, l2, l3, l4, l5, l6, l7, l8, l9, l10, l11, l12, l13, l14) BLK_CW \
	BLK_HEAD(l0, SH, 0, 0, 8, P0) BLK_STEP(l1, SH, 1) BLK_STEP(l2, SH, 2) BLK_HEAD(l3, SH, 3, 1, 9, P3) BLK_STEP(l4, SH, 4) BLK_STEP(l5, SH, 5) BLK_HEAD(l6, SH, 6, 2, 10, P6) BLK_STEP(l7, SH, 7) BLK_STEP(l8, SH, 8) \
	BLK_HEAD(l9, SH, 9, 3, 11, P9) BLK_STEP(l10, SH, 10) BLK_STEP(l11, SH, 11) BLK_HEAD(l12, SH, 12, 4, 12, P12) BLK_STEP(l13, SH, 13) BLK_STEP(l14, SH, 14) "9:\n\t"
#define BLK_15(SH, P0, P3, P6, P9, P12, l0, l1, l2, l3, l4, l5, l6, l7, l8, l9, l10, l11, l12, l13, l14) \
	BLK_STEP(l0, SH, 0) BLK_STEP(l1, SH, 1) BLK_STEP(l2, SH, 2) BLK_STEP(l3, SH, 3) BLK_STEP(l4, SH, 4) BLK_STEP(l5, SH, 5) BLK_STEP(l6, SH, 6) BLK_STEP(l7, SH, 7) \
	BLK_STEP(l8, SH, 8) BLK_STEP(l9, SH, 9) BLK_STEP(l10, SH, 10) BLK_STEP(l11, SH, 11) BLK_STEP(l12, SH, 12) BLK_STEP(l13, SH, 13) BLK_STEP(l14, SH, 14)
// Rows 15*HALF .. 15*HALF+14 of linear block Q.  lo / hi are updated in place for friction rows; own: this lane's impulse sum before the sweep; idt, dt: 1 / dt and dt
// (wave-uniform).
template <int Q, int HALF, bool GENERAL>
__device__ __forceinline__ void blk_resolve15(float &x, float &lo, float &hi, float &imp, const float (&G)[32], const float fms, const float mu, const float own, const int mp,
                                              const int idt_bits, const int dt_bits, const int cw)
{
	int s, m; long long save; float t0, t1;
#define BLK_G(k) (G[(Q & 1) ? 31 - (15 * HALF + (k)) : 15 * HALF + (k)])
#define BLK_GS [g0] "v"(BLK_G(0)), [g1] "v"(BLK_G(1)), [g2] "v"(BLK_G(2)), [g3] "v"(BLK_G(3)), [g4] "v"(BLK_G(4)), [g5] "v"(BLK_G(5)), [g6] "v"(BLK_G(6)), [g7] "v"(BLK_G(7)), \
	  [g8] "v"(BLK_G(8)), [g9] "v"(BLK_G(9)), [g10] "v"(BLK_G(10)), [g11] "v"(BLK_G(11)), [g12] "v"(BLK_G(12)), [g13] "v"(BLK_G(13)), [g14] "v"(BLK_G(14))
#define BLK_OPSG : [imp] "+v"(imp), [x] "+v"(x), [lo] "+v"(lo), [hi] "+v"(hi), [s] "=&s"(s), [m] "=&s"(m), [save] "=&s"(save), [t0] "=&v"(t0), [t1] "=&v"(t1) \
	: [fms] "v"(fms), [mu] "v"(mu), [own] "v"(own), [mp] "v"(mp), [idt] "s"(idt_bits), [dt] "s"(dt_bits), [cw] "v"(cw), BLK_GS : "vcc", "scc"
#define BLK_OPS : [imp] "+v"(imp), [x] "+v"(x), [s] "=&s"(s), [save] "=&s"(save) : [lo] "v"(lo), [hi] "v"(hi), BLK_GS : "scc"
#define BLK_BOTH(ELO, EHI, SH, ...) { if constexpr (GENERAL) asm volatile(BLK_ENTER(ELO, EHI) BLK_15G(SH, __VA_ARGS__) BLK_LEAVE BLK_OPSG); else asm volatile(BLK_ENTER(ELO, EHI) BLK_15(SH, __VA_ARGS__) BLK_LEAVE BLK_OPS); }
	if constexpr (Q == 0 && HALF == 0) BLK_BOTH("-1", "0", BLK_SHL_LO, 0, 3, 6, 9, 12, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14)
	if constexpr (Q == 0 && HALF == 1) BLK_BOTH("0xffff8000", "0", BLK_SHL_LO, 15, 18, 21, 24, 27, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29)
	if constexpr (Q == 1 && HALF == 0) BLK_BOTH("-1", "0", BLK_SHR_LO, 0, 3, 6, 9, 12, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17)
	if constexpr (Q == 1 && HALF == 1) BLK_BOTH("0x0001ffff", "0", BLK_SHR_LO, 15, 18, 21, 24, 27, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2)
	if constexpr (Q == 2 && HALF == 0) BLK_BOTH("0", "-1", BLK_SHL_HI, 0, 3, 6, 9, 12, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46)
	if constexpr (Q == 2 && HALF == 1) BLK_BOTH("0", "0xffff8000", BLK_SHL_HI, 15, 18, 21, 24, 27, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61)
	if constexpr (Q == 3 && HALF == 0) BLK_BOTH("0", "-1", BLK_SHR_HI, 0, 3, 6, 9, 12, 63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49)
	if constexpr (Q == 3 && HALF == 1) BLK_BOTH("0", "0x0001ffff", BLK_SHR_HI, 15, 18, 21, 24, 27, 48, 47, 46, 45, 44, 43, 42, 41, 40, 39, 38, 37, 36, 35, 34)
#undef BLK_BOTH
#undef BLK_OPS
#undef BLK_OPSG
#undef BLK_GS
#undef BLK_G
}

// ---- the momenta brought up to date: a block's impulses summed per body ----
// A block's rows touch up to 64 (row, side) pairs = EDGES.  The prologue sorts a block's edges by body, one edge per lane (blk_edge word below); a sweep then pulls
// every edge's contribution from its row's lane (ds_bpermute), sums the runs of equal bodies with a segmented scan on DPP (the participation of a lane in every step is
// a bit of its edge word, turned into a multiplier 0 / 1: all lanes stay enabled, a DPP operand must not come from a disabled lane, and the last lane of a body's run
// adds the total to the body's momenta in LDS.  A fixed order of the same sums: bit-reproducible, unlike LDS float atomics (which also cost ~190 clocks of the CU's LDS
// per wave instruction: tools/probe/block_probe.hip).
// edge word: bits 0-5 the row's lane, 6 side (1 = rb1), 7 valid, 8 last of its body's run, 9-14 takes part in the scan steps (row_shr 1, 2, 4, 8, row_bcast15, row_bcast31), 16-23 body
#define BLK_E_SIDE 0x40
#define BLK_E_VALID 0x80
#define BLK_E_TAIL 0x100
struct blk_scan_mul { float m1, m2, m4, m8, m15, m31; };
__device__ __forceinline__ blk_scan_mul blk_scan_multipliers(unsigned e)
{
	blk_scan_mul m;
	m.m1 = (float)((e >> 9) & 1u); m.m2 = (float)((e >> 10) & 1u); m.m4 = (float)((e >> 11) & 1u);
	m.m8 = (float)((e >> 12) & 1u); m.m15 = (float)((e >> 13) & 1u); m.m31 = (float)((e >> 14) & 1u);
	return m;
}
// Inclusive segmented sums of three values at once: v += m * (the value `shift` lanes below), the lane read by a DPP operand of the multiply-add itself.  A lane whose
// source lies outside its row of 16 (row_shr) or whose row the step does not address (row_bcast) is left alone by the instruction; the three instruction streams fill each
// other's two wait states between a write and a DPP read of the same register, and the statement opens with the two a freshly computed input needs.
__device__ __forceinline__ void blk_seg_scan3(float &a, float &b, float &c, const blk_scan_mul &m)
{
#define BLK_SCAN(M, CTRL) \
	"v_fmac_f32_dpp %0, %0, %[" M "] " CTRL " bank_mask:0xf\n\t" \
	"v_fmac_f32_dpp %1, %1, %[" M "] " CTRL " bank_mask:0xf\n\t" \
	"v_fmac_f32_dpp %2, %2, %[" M "] " CTRL " bank_mask:0xf\n\t"
	asm volatile("s_nop 1\n\t"
	             BLK_SCAN("m1", "row_shr:1 row_mask:0xf") BLK_SCAN("m2", "row_shr:2 row_mask:0xf") BLK_SCAN("m4", "row_shr:4 row_mask:0xf") BLK_SCAN("m8", "row_shr:8 row_mask:0xf")
	             BLK_SCAN("m15", "row_bcast:15 row_mask:0xa") BLK_SCAN("m31", "row_bcast:31 row_mask:0xc")
	             : "+v"(a), "+v"(b), "+v"(c) : [m1] "v"(m.m1), [m2] "v"(m.m2), [m4] "v"(m.m4), [m8] "v"(m.m8), [m15] "v"(m.m15), [m31] "v"(m.m31));
#undef BLK_SCAN
}
__device__ __forceinline__ float blk_pull(int src_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
