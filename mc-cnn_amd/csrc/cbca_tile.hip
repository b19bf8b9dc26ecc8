// Cross-based cost aggregation (adcensus.cu:343-377) for real-scene arm statistics: column strips streamed through an LDS
// ring, supports sorted by height inside a step, vertical run sharing.
//
// The reference's region sum is ONE serial chain of fp32 additions per output -- rows ascending, inside a row x ascending --
// and that order is kept (cbca.hip, parity rule).  What can be shared without touching the order: the horizontal run of
// support row yy in column x,  [x - min(L0[yy,x], L1[yy,xp]), x + min(R0[yy,x], R1[yy,xp])],  depends on (d, yy, x)
// only, not on the output row y.  So the outputs (y, x), (y+1, x), ... of one COLUMN walk the same runs, each from its own
// first row to its own last row.  A lane therefore owns an ITEM = one column x 4 consecutive output rows: it walks the rows
// from the topmost first row to the bottommost last row of its four outputs, fetches every run value ONCE from LDS and
// adds it to all four accumulators; an accumulator is reset to +0.0 at its output's first row and read out after its last
// row -- what it collected outside its own rows never reaches a result.  The lanes of a wave walk rows of different run
// lengths in lockstep under a shrinking EXEC mask (a lane whose run is over is masked off).  Per tap and wave: one
// ds_read_b32, one compare, four additions.
//
// A block walks a strip of TW output columns of ONE disparity plane top to bottom in steps of TH rows.  An LDS ring of
// TH + 2A rows holds the step's window: values (TW + the arm halo on either side), per pixel of the TW columns the run
// (4 * left, length) and (up, down).  The TH rows the next step adds are fetched into registers before the step's work and
// committed to the ring after it: every row of the plane is read once per strip and memory latency hides behind the
// additions.  A wave runs at the pace of its tallest item and longest runs, and real scenes mix 3 x 3 supports with flat
// regions of (2 L1 - 1)^2 taps: the step's items are SORTED by height (counting sort in LDS, a few instructions per item)
// and the waves of the block pull 64-item chunks, tallest first, from a shared counter.  Two classes need no walk at all:
// items of four minimal 3 x 3 supports (six rows x three values, unrolled) and items whose four outputs are all three rows
// tall (row r of the item's six feeds outputs max(0, r - 2) .. min(3, r): runs are looked up, nothing else).  Results go
// to an LDS tile and leave as full rows.
//
// None of that bookkeeping depends on the volume: mc_predict aggregates a pair 2 + 16 times per direction (main.lua:998-1001,
// 1033-1039) and keeps the first pass's work in a PLAN -- per (plane, region, step) the sorted item table and the class
// counts, per voxel the combined run (2 bytes) and vertical arms (1 byte) as committed to the ring -- which the other passes
// read a step ahead with the rows (3.5 bytes per voxel of extra traffic for a third of the instructions).
#include "cbca_common.h"
#include <algorithm>
#include <type_traits>

namespace mc {

#define TPROF(slot) do { } while (0)

namespace {

template <int A, int TW, int TH, int MODE = 0>
struct TileGeo {
	static constexpr int AH = (A + 3) & ~3;           // horizontal halo, a multiple of 4 columns (16-byte rows)
#define MC_TILE_SWPAD 4   // (measured: 0 / 4 -> 2.885 / 2.835 ms per launch; 12 no longer fits three blocks per CU)
	// staged columns; the long-arm instance pads its rows so that the row stride is not a multiple of the 32 LDS banks: lanes of a chunk
	// read the same columns at different rows (runs that start at the same image edge), which a stride of 160 words puts into one bank
	// (only the plan-reading instance, which has no sort histogram in LDS, has the bytes for it: three blocks per CU)
	static constexpr int SW = TW + 2 * AH + ((A > 4 && MODE == 2) ? MC_TILE_SWPAD : 0);
	static constexpr int RR = TH + 2 * A;             // ring rows: the window of one step
	static constexpr int NI = TW * (TH / 4);          // items of a step: column x 4 rows
	static constexpr int NG = NI / 64;                // groups of 64 consecutive items (one wave's worth)
	static constexpr int NKEY = 32;                   // sort keys: 0 nothing to compute, 1 four 3 x 3 supports, 2 four three-row outputs, height + 1 otherwise (<= 2A + 5)
	static constexpr int V_BYTES = RR * SW * 4;
	static constexpr int M_BYTES = (RR * TW * 2 + 15) & ~15;   // runs: output columns only (a run is looked up in the item's own column)
	static constexpr int UD_BYTES = (RR * TW + 15) & ~15;      // up | down << 4 per pixel, 0xff = no output here
	static constexpr int OUT_BYTES = TH * TW * 4;
	static constexpr int TAB_BYTES = NI * 2 + 16;          // sorted items (column | row group << 8 | key << 11), then the step's class counts (nz, nfast, ngen, ntall): one plan entry
	static constexpr int ENT_BYTES = TAB_BYTES;            // plan entry of one (plane, region, step)
	static constexpr int MISC_BYTES = (MODE == 2 ? 0 : NG * NKEY * 4) + 16;   // items per group and key (sort); chunk counter
	static constexpr int LDS_BYTES = V_BYTES + M_BYTES + UD_BYTES + OUT_BYTES + TAB_BYTES + MISC_BYTES;
};

// The taps of one support row for the four accumulators of a lane.  Lane-private run: n values from LDS address p.  The
// lanes of a wave hold runs of different lengths, so tap t is added under  EXEC &= (t < n)  (v_cmpx narrows EXEC, the run
// is a prefix, so EXEC only ever shrinks inside a row): one compare and four additions per tap and wave, a lane whose run
// is over is simply masked off (its accumulators keep their values: no operand at all), and the row ends -- s_cbranch_execz
// after every group -- when the wave's longest run does.  Values are fetched nine at a time (0..8 up front: most rows
// end there); rows with longer runs fetch 9..17 and 18..26 when they get there.  hipcc cannot express the EXEC narrowing,
// hence the assembly; it waits for its own LDS reads (lgkmcnt(0)) before it uses them and restores EXEC.
#define MC_ADD1(v) " v_add_f32 %[s0], %[s0], %[" #v "]\n"
#define MC_ADD2(v) MC_ADD1(v) " v_add_f32 %[s1], %[s1], %[" #v "]\n"
#define MC_ADD3(v) MC_ADD2(v) " v_add_f32 %[s2], %[s2], %[" #v "]\n"
#define MC_ADD4(v) MC_ADD3(v) " v_add_f32 %[s3], %[s3], %[" #v "]\n"
#define MC_TAP(K, t, v) "v_cmpx_lt_u32_e32 vcc, " #t ", %[n]\n" MC_ADD##K(v)
#define MC_LOAD9(o) "ds_read_b32 %[v0], %[p] offset:" #o "+0\n ds_read_b32 %[v1], %[p] offset:" #o "+4\n ds_read_b32 %[v2], %[p] offset:" #o "+8\n" \
                    " ds_read_b32 %[v3], %[p] offset:" #o "+12\n ds_read_b32 %[v4], %[p] offset:" #o "+16\n ds_read_b32 %[v5], %[p] offset:" #o "+20\n" \
                    " ds_read_b32 %[v6], %[p] offset:" #o "+24\n ds_read_b32 %[v7], %[p] offset:" #o "+28\n ds_read_b32 %[v8], %[p] offset:" #o "+32\n"
#define MC_EXIT "s_nop 1\n s_cbranch_execz .Ltaps_done_%=\n"
#define MC_TAPS_9(K) "s_mov_b64 %[sv], exec\n" MC_LOAD9(0) "s_waitcnt lgkmcnt(0)\n" \
	MC_TAP(K, 0, v0) MC_TAP(K, 1, v1) MC_TAP(K, 2, v2) MC_EXIT MC_TAP(K, 3, v3) MC_TAP(K, 4, v4) MC_EXIT \
	MC_TAP(K, 5, v5) MC_TAP(K, 6, v6) MC_TAP(K, 7, v7) MC_TAP(K, 8, v8)
#define MC_TAPS_27(K) MC_TAPS_9(K) MC_EXIT MC_LOAD9(36) "s_waitcnt lgkmcnt(0)\n" \
	MC_TAP(K, 9, v0) MC_TAP(K, 10, v1) MC_TAP(K, 11, v2) MC_TAP(K, 12, v3) MC_EXIT MC_TAP(K, 13, v4) MC_TAP(K, 14, v5) MC_TAP(K, 15, v6) MC_TAP(K, 16, v7) \
	MC_TAP(K, 17, v8) MC_EXIT MC_LOAD9(72) "s_waitcnt lgkmcnt(0)\n" \
	MC_TAP(K, 18, v0) MC_TAP(K, 19, v1) MC_TAP(K, 20, v2) MC_TAP(K, 21, v3) MC_TAP(K, 22, v4) MC_EXIT MC_TAP(K, 23, v5) MC_TAP(K, 24, v6) MC_TAP(K, 25, v7) \
	MC_TAP(K, 26, v8)
#define MC_TAPS_END ".Ltaps_done_%=:\n s_mov_b64 exec, %[sv]\n"
#define MC_TAPS_TMPS [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [v4] "=&v"(v4), [v5] "=&v"(v5), [v6] "=&v"(v6), [v7] "=&v"(v7), \
	[v8] "=&v"(v8), [sv] "=&s"(sv)
// pa: LDS byte address of the run's first value.  LONG: runs up to 27 values (arms <= 13), else up to 9 (arms <= 4).
template <bool LONG>
__device__ __forceinline__ void tile_taps(unsigned pa, int n, float (&sum)[4])
{
	float v0, v1, v2, v3, v4, v5, v6, v7, v8;
	unsigned long long sv;
	if (LONG)
		asm volatile(MC_TAPS_27(4) MC_TAPS_END : [s0] "+v"(sum[0]), [s1] "+v"(sum[1]), [s2] "+v"(sum[2]), [s3] "+v"(sum[3]), MC_TAPS_TMPS
		             : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else
		asm volatile(MC_TAPS_9(4) MC_TAPS_END : [s0] "+v"(sum[0]), [s1] "+v"(sum[1]), [s2] "+v"(sum[2]), [s3] "+v"(sum[3]), MC_TAPS_TMPS
		             : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
// ... for one, two, three accumulators (rows of the three-row class, which feed a fixed set of outputs)
template <bool LONG>
__device__ __forceinline__ void tile_taps1(unsigned pa, int n, float &a)
{
	float v0, v1, v2, v3, v4, v5, v6, v7, v8;
	unsigned long long sv;
	if (LONG) asm volatile(MC_TAPS_27(1) MC_TAPS_END : [s0] "+v"(a), MC_TAPS_TMPS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else asm volatile(MC_TAPS_9(1) MC_TAPS_END : [s0] "+v"(a), MC_TAPS_TMPS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
template <bool LONG>
__device__ __forceinline__ void tile_taps2(unsigned pa, int n, float &a, float &b)
{
	float v0, v1, v2, v3, v4, v5, v6, v7, v8;
	unsigned long long sv;
	if (LONG) asm volatile(MC_TAPS_27(2) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), MC_TAPS_TMPS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else asm volatile(MC_TAPS_9(2) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), MC_TAPS_TMPS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
template <bool LONG>
__device__ __forceinline__ void tile_taps3(unsigned pa, int n, float &a, float &b, float &c)
{
	float v0, v1, v2, v3, v4, v5, v6, v7, v8;
	unsigned long long sv;
	if (LONG) asm volatile(MC_TAPS_27(3) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), [s2] "+v"(c), MC_TAPS_TMPS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else asm volatile(MC_TAPS_9(3) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), [s2] "+v"(c), MC_TAPS_TMPS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
// ... with the first nine values already in registers (requested a row ahead by the caller, who also waits for them): the
// plan-reading instance, which has the registers for it
#define MC_TAPS_9P(K) "s_mov_b64 %[sv], exec\n" \
	MC_TAP(K, 0, v0) MC_TAP(K, 1, v1) MC_TAP(K, 2, v2) MC_EXIT MC_TAP(K, 3, v3) MC_TAP(K, 4, v4) MC_EXIT \
	MC_TAP(K, 5, v5) MC_TAP(K, 6, v6) MC_TAP(K, 7, v7) MC_TAP(K, 8, v8)
#define MC_TAPS_27P(K) MC_TAPS_9P(K) MC_EXIT MC_LOAD9(36) "s_waitcnt lgkmcnt(0)\n" \
	MC_TAP(K, 9, v0) MC_TAP(K, 10, v1) MC_TAP(K, 11, v2) MC_TAP(K, 12, v3) MC_EXIT MC_TAP(K, 13, v4) MC_TAP(K, 14, v5) MC_TAP(K, 15, v6) MC_TAP(K, 16, v7) \
	MC_TAP(K, 17, v8) MC_EXIT MC_LOAD9(72) "s_waitcnt lgkmcnt(0)\n" \
	MC_TAP(K, 18, v0) MC_TAP(K, 19, v1) MC_TAP(K, 20, v2) MC_TAP(K, 21, v3) MC_TAP(K, 22, v4) MC_EXIT MC_TAP(K, 23, v5) MC_TAP(K, 24, v6) MC_TAP(K, 25, v7) \
	MC_TAP(K, 26, v8)
#define MC_TAPS_VALS [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7]), \
	[v8] "+v"(v[8]), [sv] "=&s"(sv)
typedef const float __attribute__((address_space(3))) *lds_cfp;
__device__ __forceinline__ void tile_load9(float (&v)[9], unsigned pa)
{
	const lds_cfp p = (lds_cfp)(__UINTPTR_TYPE__)pa;   // (LDS addresses are 32-bit)
#pragma unroll
	for (int t = 0; t < 9; ++t) v[t] = p[t];
}
template <bool LONG>
__device__ __forceinline__ void tile_taps_p(float (&v)[9], unsigned pa, int n, float (&sum)[4])
{
	unsigned long long sv;
	if (LONG)
		asm volatile(MC_TAPS_27P(4) MC_TAPS_END : [s0] "+v"(sum[0]), [s1] "+v"(sum[1]), [s2] "+v"(sum[2]), [s3] "+v"(sum[3]), MC_TAPS_VALS
		             : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else
		asm volatile(MC_TAPS_9P(4) MC_TAPS_END : [s0] "+v"(sum[0]), [s1] "+v"(sum[1]), [s2] "+v"(sum[2]), [s3] "+v"(sum[3]), MC_TAPS_VALS
		             : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
template <bool LONG>
__device__ __forceinline__ void tile_taps1_p(float (&v)[9], unsigned pa, int n, float &a)
{
	unsigned long long sv;
	if (LONG) asm volatile(MC_TAPS_27P(1) MC_TAPS_END : [s0] "+v"(a), MC_TAPS_VALS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else asm volatile(MC_TAPS_9P(1) MC_TAPS_END : [s0] "+v"(a), MC_TAPS_VALS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
template <bool LONG>
__device__ __forceinline__ void tile_taps2_p(float (&v)[9], unsigned pa, int n, float &a, float &b)
{
	unsigned long long sv;
	if (LONG) asm volatile(MC_TAPS_27P(2) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), MC_TAPS_VALS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else asm volatile(MC_TAPS_9P(2) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), MC_TAPS_VALS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
template <bool LONG>
__device__ __forceinline__ void tile_taps3_p(float (&v)[9], unsigned pa, int n, float &a, float &b, float &c)
{
	unsigned long long sv;
	if (LONG) asm volatile(MC_TAPS_27P(3) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), [s2] "+v"(c), MC_TAPS_VALS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
	else asm volatile(MC_TAPS_9P(3) MC_TAPS_END : [s0] "+v"(a), [s1] "+v"(b), [s2] "+v"(c), MC_TAPS_VALS : [n] "v"(n), [p] "v"(pa) : "vcc", "memory");
}
#undef MC_TAP
#undef MC_LOAD9
#undef MC_EXIT


}  // namespace

// head of the plan (256 bytes): who wrote it.  A pass that reads a plan stands down unless it was written for exactly this
// problem, direction and arm class (a reading call without its writing call, or on a cached scratch of another shape)
// (words 32 .. 37 of the 64: the first eight are the head of the texture route's list, LH_*, which shares the area -- ADVICE r4: with overlapping
// heads, clearing the one invalidated the other)
constexpr int PLAN_HDR = 256;
enum { PH_MAGIC = 32, PH_D = 33, PH_H = 34, PH_W = 35, PH_DIR = 36, PH_ARM = 37 };
static_assert(PLAN_HDR == LH_WORDS * 4 && PH_MAGIC >= 8, "the plan's head and the list's head share 256 bytes, not words");
constexpr uint32_t PH_MAGIC_VALUE = 0x504c414eu;
__device__ __forceinline__ bool plan_valid(const CbcaArgs &P, int arm_class)
{
	const uint32_t *__restrict__ h = (const uint32_t *)P.plan;
	return h[PH_MAGIC] == PH_MAGIC_VALUE && h[PH_D] == (uint32_t)P.D && h[PH_H] == (uint32_t)P.H && h[PH_W] == (uint32_t)P.W &&
	       h[PH_DIR] == (uint32_t)(P.direction + 1) && h[PH_ARM] == (uint32_t)arm_class;
}

// item i = (column c, row group g): outputs window rows A + 4g .. A + 4g + 3 of column c; height = rows from the topmost
// first row to the bottommost last row of its outputs that have a partner.  Window rows 0 .. (ring rows) - 1, ring slot of
// window row w = (base + w) mod RR.
template <int A, int TW, int RR>
__device__ __forceinline__ cb_u32 tile_item_rows(const unsigned char *__restrict__ UDl, int base, int c, int g, int (&s0)[4], int (&e0)[4], int &top, int &bot)
{
	top = 1 << 20; bot = -1;
	int slot = base + A + 4 * g;
	cb_u32 udall = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j, ++slot) {
		slot = slot >= RR ? slot - RR : slot;
		const cb_u32 ud = UDl[slot * TW + c];
		udall |= ud << (8 * j);
		const bool ok = ud != 0xffu;
		const int up = (int)(ud & 15u), dn = (int)(ud >> 4);
		s0[j] = ok ? 4 * g + j + A - up : 1 << 20;    // window row of the output's first / last support row
		e0[j] = ok ? 4 * g + j + A + dn : -1;
		top = min(top, s0[j]);
		bot = max(bot, e0[j]);
	}
	return udall;
}

// The first- / last-row events of a walk row (round 5): four compares into four scalar mask pairs FIRST, then the eight selects.  hipcc compiles the
// plain form to  v_cmp (vcc), s_nop 1, two v_cndmask  per accumulator -- on gfx950 a VALU instruction that reads a mask must stand two instructions
// behind the VALU compare that wrote it, and with every compare going through vcc nothing can fill the gap: 16 instructions per event kind and
// row, 4 of them s_nop.  Here the gaps fill themselves: 12 instructions (+ one s_nop 1 in front: the compiler's hazard recogniser does not look into the
// statement, and `ibit` may reach it fresh from a v_readfirstlane -- an SGPR written by a VALU instruction must be two states old when a VALU compare reads it).
// gfx950 only, like the rest of the library (Makefile: --offload-arch=gfx950).
__device__ __forceinline__ void tile_events_start(cb_u32 ibit, const cb_u32 (&sbit)[4], float (&sum)[4], int (&cb)[4], int Pn)
{
	unsigned long long m0, m1, m2, m3;
	asm volatile("s_nop 1\n v_cmp_eq_u32_e64 %[m0], %[ib], %[b0]\n v_cmp_eq_u32_e64 %[m1], %[ib], %[b1]\n v_cmp_eq_u32_e64 %[m2], %[ib], %[b2]\n v_cmp_eq_u32_e64 %[m3], %[ib], %[b3]\n"
	             " v_cndmask_b32_e64 %[s0], %[s0], 0, %[m0]\n v_cndmask_b32_e64 %[c0], %[c0], %[pn], %[m0]\n"
	             " v_cndmask_b32_e64 %[s1], %[s1], 0, %[m1]\n v_cndmask_b32_e64 %[c1], %[c1], %[pn], %[m1]\n"
	             " v_cndmask_b32_e64 %[s2], %[s2], 0, %[m2]\n v_cndmask_b32_e64 %[c2], %[c2], %[pn], %[m2]\n"
	             " v_cndmask_b32_e64 %[s3], %[s3], 0, %[m3]\n v_cndmask_b32_e64 %[c3], %[c3], %[pn], %[m3]\n"
	             : [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [s0] "+v"(sum[0]), [s1] "+v"(sum[1]), [s2] "+v"(sum[2]), [s3] "+v"(sum[3]),
	               [c0] "+v"(cb[0]), [c1] "+v"(cb[1]), [c2] "+v"(cb[2]), [c3] "+v"(cb[3])
	             : [ib] "s"(ibit), [b0] "v"(sbit[0]), [b1] "v"(sbit[1]), [b2] "v"(sbit[2]), [b3] "v"(sbit[3]), [pn] "v"(Pn));
}
__device__ __forceinline__ void tile_events_end(cb_u32 ibit, const cb_u32 (&ebit)[4], const float (&sum)[4], float (&res)[4], int (&ce)[4], int Pn)
{
	unsigned long long m0, m1, m2, m3;
	asm volatile("s_nop 1\n v_cmp_eq_u32_e64 %[m0], %[ib], %[b0]\n v_cmp_eq_u32_e64 %[m1], %[ib], %[b1]\n v_cmp_eq_u32_e64 %[m2], %[ib], %[b2]\n v_cmp_eq_u32_e64 %[m3], %[ib], %[b3]\n"
	             " v_cndmask_b32_e64 %[r0], %[r0], %[s0], %[m0]\n v_cndmask_b32_e64 %[c0], %[c0], %[pn], %[m0]\n"
	             " v_cndmask_b32_e64 %[r1], %[r1], %[s1], %[m1]\n v_cndmask_b32_e64 %[c1], %[c1], %[pn], %[m1]\n"
	             " v_cndmask_b32_e64 %[r2], %[r2], %[s2], %[m2]\n v_cndmask_b32_e64 %[c2], %[c2], %[pn], %[m2]\n"
	             " v_cndmask_b32_e64 %[r3], %[r3], %[s3], %[m3]\n v_cndmask_b32_e64 %[c3], %[c3], %[pn], %[m3]\n"
	             : [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [r0] "+v"(res[0]), [r1] "+v"(res[1]), [r2] "+v"(res[2]), [r3] "+v"(res[3]),
	               [c0] "+v"(ce[0]), [c1] "+v"(ce[1]), [c2] "+v"(ce[2]), [c3] "+v"(ce[3])
	             : [ib] "s"(ibit), [b0] "v"(ebit[0]), [b1] "v"(ebit[1]), [b2] "v"(ebit[2]), [b3] "v"(ebit[3]), [pn] "v"(Pn),
	               [s0] "v"(sum[0]), [s1] "v"(sum[1]), [s2] "v"(sum[2]), [s3] "v"(sum[3]));
}

// One work unit of a step (cbca_tile_kernel's chunk phase; window of the step = ring rows base .. base + TH + 2A - 1 mod RR):
// units [0, usplit) are split units of 16 tall items, one OUTPUT per lane; the others are chunks of 64 items of the sorted table,
// tallest first: the general walk, then the class of four three-row outputs (from item ngen on), then the class of four 3 x 3
// supports (from item nfast on).  Results go to the step's tile OUTl.
template <int A, int TW, int SW, int RR, bool PIPE>
__device__ __forceinline__ void tile_unit(int unit, int lane, float *__restrict__ Vl, const unsigned short *__restrict__ Ml,
                                          const unsigned char *__restrict__ UDl, float *__restrict__ OUTl, const unsigned short *__restrict__ TABl,
                                          int base, int nz, int nfast, int ngen, int nsplit, int usplit)
{
	constexpr int AH = (A + 3) & ~3;
	if (unit < usplit) {
		const int it = unit * 16 + (lane >> 2), j = lane & 3;
		const bool hasi = it < nsplit;
		const cb_u32 ent = TABl[hasi ? it : 0];
		const int c = (int)(ent & 0xffu), g = (int)((ent >> 8) & 7u);
		const int E = __builtin_amdgcn_readfirstlane((int)(ent >> 11)) - 1;   // lane 0: the unit's tallest item, key = height + 1
		// the item's rows (as every lane of the item walks them: one LDS address per item and row), this lane's output inside them
		int s0[4], e0[4], top, bot;
		tile_item_rows<A, TW, RR>(UDl, base, c, g, s0, e0, top, bot);
		const int sj = j == 0 ? s0[0] : j == 1 ? s0[1] : j == 2 ? s0[2] : s0[3];
		const int ej = j == 0 ? e0[0] : j == 1 ? e0[1] : j == 2 ? e0[2] : e0[3];
		const bool outp = hasi && ej >= 0;
		const int ext = hasi ? bot - top + 1 : 0;
		// rows of the walk that belong to this output: bits sj - top .. ej - top
		cb_u32 mine = outp ? ((2u << (ej - top)) - (1u << (sj - top))) : 0u;
		asm volatile("" : "+v"(mine));
		cb_u32 extbit = 1u << ext;
		asm volatile("" : "+v"(extbit));
		const cb_u32 Ebit = 1u << E;
		int slot = hasi ? base + top : 0;
		slot = slot >= RR ? slot - RR : slot;
		const cb_u32 cv = (cb_u32)(size_t)Vl + (cb_u32)(c + AH) * 4u;
		float sum = 0.0f;
		int Pn = 0;
		__builtin_amdgcn_s_setprio(2);
		cb_u32 mnext = ext > 0 ? (cb_u32)Ml[slot * TW + c] : 0u;
		for (cb_u32 ibit = 1u; ibit < Ebit; ibit <<= 1) {
			const cb_u32 m = mnext;
			const int n = (mine & ibit) ? (int)(m >> 8) : 0;
			const cb_u32 pa = __umul24((cb_u32)slot, (cb_u32)(SW * 4)) + cv - (m & 0xffu);
			slot = slot + 1 == RR ? 0 : slot + 1;
			const cb_u32 mr = Ml[slot * TW + c];
			mnext = (ibit << 1) < extbit ? mr : 0u;
			tile_taps1<(A > 4)>(pa, n, sum);
			Pn += n;
		}
		__builtin_amdgcn_s_setprio(0);
		if (outp) OUTl[(4 * g + j) * TW + c] = sum / (float)Pn;
		return;
	}
	const int i0 = nsplit + (unit - usplit) * 64;   // first item of the chunk
	const int idx = i0 + lane;
	const bool has = idx < nz;
	const cb_u32 ent = TABl[has ? idx : 0];
	const int c = (int)(ent & 0xffu), g = (int)((ent >> 8) & 7u);
	if (i0 >= nfast) {
		// every item of the chunk is four 3 x 3 supports: six rows x three values, nine additions per output in the
		// reference's order, no lengths to look at
		int slot = base + A + 4 * g - 1;
		slot = slot >= RR ? slot - RR : slot;
		float fs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
		for (int r = 0; r < 6; ++r) {   // window row r is row r - j of output j: outputs max(0, r - 2) .. min(3, r)
			const float *__restrict__ p = Vl + slot * SW + c + AH - 1;
			const float v0 = p[0], v1 = p[1], v2 = p[2];
			slot = slot + 1 == RR ? 0 : slot + 1;
#pragma unroll
			for (int j = 0; j < 4; ++j)
				if (j <= r && r <= j + 2) { fs[j] += v0; fs[j] += v1; fs[j] += v2; }
		}
#pragma unroll
		for (int j = 0; j < 4; ++j)
			if (has) OUTl[(4 * g + j) * TW + c] = fs[j] / 9.0f;
		return;
	}
	if (i0 >= ngen) {
		// every item of the chunk has four outputs of three rows each (the minimal class's items at its end included): window
		// row r of the item's six is row r - j of output j, so it feeds outputs max(0, r - 2) .. min(3, r) -- no first / last
		// rows to watch, only the runs to look up
		int slot = base + A + 4 * g - 1;
		slot = slot >= RR ? slot - RR : slot;
		const cb_u32 cv = (cb_u32)(size_t)Vl + (cb_u32)(c + AH) * 4u;
		float fs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
		int nr[6];
		if constexpr (PIPE) {   // the six run words first, then every row's values requested while the row before it is summed
			cb_u32 pa[6];
#pragma unroll
			for (int r = 0; r < 6; ++r) {
				const cb_u32 m = has ? (cb_u32)Ml[slot * TW + c] : 0u;
				nr[r] = (int)(m >> 8);
				pa[r] = __umul24((cb_u32)slot, (cb_u32)(SW * 4)) + cv - (m & 0xffu);
				slot = slot + 1 == RR ? 0 : slot + 1;
			}
			float va[9], vb[9];
			tile_load9(va, pa[0]);
			tile_load9(vb, pa[1]);
			tile_taps1_p<(A > 4)>(va, pa[0], nr[0], fs[0]);
			tile_load9(va, pa[2]);
			tile_taps2_p<(A > 4)>(vb, pa[1], nr[1], fs[0], fs[1]);
			tile_load9(vb, pa[3]);
			tile_taps3_p<(A > 4)>(va, pa[2], nr[2], fs[0], fs[1], fs[2]);
			tile_load9(va, pa[4]);
			tile_taps3_p<(A > 4)>(vb, pa[3], nr[3], fs[1], fs[2], fs[3]);
			tile_load9(vb, pa[5]);
			tile_taps2_p<(A > 4)>(va, pa[4], nr[4], fs[2], fs[3]);
			tile_taps1_p<(A > 4)>(vb, pa[5], nr[5], fs[3]);
		} else {
#pragma unroll
			for (int r = 0; r < 6; ++r) {
				const cb_u32 m = has ? (cb_u32)Ml[slot * TW + c] : 0u;
				nr[r] = (int)(m >> 8);
				const cb_u32 pa = __umul24((cb_u32)slot, (cb_u32)(SW * 4)) + cv - (m & 0xffu);
				slot = slot + 1 == RR ? 0 : slot + 1;
				if (r == 0) tile_taps1<(A > 4)>(pa, nr[r], fs[0]);
				else if (r == 1) tile_taps2<(A > 4)>(pa, nr[r], fs[0], fs[1]);
				else if (r == 2) tile_taps3<(A > 4)>(pa, nr[r], fs[0], fs[1], fs[2]);
				else if (r == 3) tile_taps3<(A > 4)>(pa, nr[r], fs[1], fs[2], fs[3]);
				else if (r == 4) tile_taps2<(A > 4)>(pa, nr[r], fs[2], fs[3]);
				else tile_taps1<(A > 4)>(pa, nr[r], fs[3]);
			}
		}
#pragma unroll
		for (int j = 0; j < 4; ++j)
			if (has) OUTl[(4 * g + j) * TW + c] = fs[j] / (float)(nr[j] + nr[j + 1] + nr[j + 2]);
		return;
	}
	int s0[4], e0[4], top, bot;
	tile_item_rows<A, TW, RR>(UDl, base, c, g, s0, e0, top, bot);
	const int ext = has ? bot - top + 1 : 0;
	// lane 0 holds the chunk's tallest item -- except that heights 1 and 2 share a key and that the six-row classes are sorted
	// behind every other class
	const int E = max(max(__builtin_amdgcn_readfirstlane(ext), 2), i0 + 64 > ngen ? 6 : 0);
	if (E >= 14) __builtin_amdgcn_s_setprio(2);   // a tall chunk is the step's critical path (one wave, a chain of thousands of instructions)
	// first / last row of output j, counted from the item's top, as one-hot words (0: no such output): the walk below tests
	// them against the row's bit with one compare each
	cb_u32 sbit[4], ebit[4];
	bool outj[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		outj[j] = has && e0[j] >= 0;
		sbit[j] = outj[j] ? 1u << (s0[j] - top) : 0u;
		ebit[j] = outj[j] ? 1u << (e0[j] - top) : 0u;
	}
	float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	int cb[4] = {0, 0, 0, 0}, ce[4] = {1, 1, 1, 1};   // taps of the wave-row walk before the output's first row / up to its last row
	int Pn = 0;
	const cb_u32 cv = (cb_u32)(size_t)Vl + (cb_u32)(c + AH) * 4u;   // LDS byte address of the item's column in ring slot 0
	int slot = has ? base + top : 0;
	slot = slot >= RR ? slot - RR : slot;
	const cb_u32 evmask = sbit[0] | sbit[1] | sbit[2] | sbit[3] | ebit[0] | ebit[1] | ebit[2] | ebit[3];   // rows at which an output starts or ends
	cb_u32 extbit = 1u << ext;   // (heights <= 2 A + 4 < 32)
	asm volatile("" : "+v"(extbit));   // (opaque: stays one compare per row)
	const cb_u32 Ebit = 1u << E;
	// Long-arm instance (round 5, A/B on one box, profiles/r05_ab_tile.txt: 95.5 -> 93.3 -> 92.9 ms of aggregation per realistic 1000 x 1500 x 256 pair; the
	// short-arm instance measured 0.5 % slower with either, so it keeps the plain forms): (i) the rows at which ANY lane of the chunk has an event, once
	// per chunk (an OR over the wave, DPP) -- the per-row test becomes one scalar AND instead of v_and + v_cmp + two scalar instructions; (ii) the
	// events as tile_events_start / tile_events_end (no wait states)
	constexpr bool EVFAST = A > 4;
	cb_u32 evrows = 0;
	if constexpr (EVFAST) {
		cb_u32 t = evmask;
		t |= (cb_u32)__builtin_amdgcn_update_dpp(0, (int)t, DPP_QUAD_1032, 0xF, 0xF, false);
		t |= (cb_u32)__builtin_amdgcn_update_dpp(0, (int)t, DPP_QUAD_2301, 0xF, 0xF, false);
		t |= (cb_u32)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_HALF_MIRROR, 0xF, 0xF, false);
		t |= (cb_u32)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_MIRROR, 0xF, 0xF, false);
		t |= (cb_u32)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST15, 0xA, 0xF, false);
		t |= (cb_u32)__builtin_amdgcn_update_dpp(0, (int)t, DPP_ROW_BCAST31, 0xC, 0xF, false);
		evrows = (cb_u32)__builtin_amdgcn_readlane((int)t, 63);
	}
#define MC_TILE_ANYEV(ibit) (EVFAST ? (evrows & (ibit)) != 0u : (bool)__any(evmask & (ibit)))
	if constexpr (PIPE) {
		// the walk, two rows in flight: while row i is summed the values of row i + 1 are on their way (requested through the
		// run word that was fetched during row i - 1) and so is the run word of row i + 2
		int slotC = slot, slotN = slot + 1 == RR ? 0 : slot + 1;
		cb_u32 mC = ext > 0 ? (cb_u32)Ml[slotC * TW + c] : 0u;
		cb_u32 mN = 2u < extbit ? (cb_u32)Ml[slotN * TW + c] : 0u;   // (1 < ext)
		float va[9], vb[9];
		tile_load9(va, __umul24((cb_u32)slotC, (cb_u32)(SW * 4)) + cv - (mC & 0xffu));
		auto walk_row = [&](float (&cur)[9], float (&nxt)[9], cb_u32 ibit) {
			const cb_u32 paN = __umul24((cb_u32)slotN, (cb_u32)(SW * 4)) + cv - (mN & 0xffu);
			tile_load9(nxt, paN);
			const int slotNN = slotN + 1 == RR ? 0 : slotN + 1;
			const cb_u32 mr = Ml[slotNN * TW + c];
			const cb_u32 mNN = (ibit << 2) < extbit ? mr : 0u;   // (i + 2 < ext)
			const int n = (int)(mC >> 8);
			const cb_u32 pa = __umul24((cb_u32)slotC, (cb_u32)(SW * 4)) + cv - (mC & 0xffu);
			const bool anyev = MC_TILE_ANYEV(ibit);
			if (anyev) {
				if constexpr (EVFAST) tile_events_start(ibit, sbit, sum, cb, Pn);
				else
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const bool st = sbit[j] == ibit;
					sum[j] = st ? 0.0f : sum[j];
					cb[j] = st ? Pn : cb[j];
				}
			}
			tile_taps_p<(A > 4)>(cur, pa, n, sum);
			Pn += n;
			if (anyev) {
				if constexpr (EVFAST) tile_events_end(ibit, ebit, sum, res, ce, Pn);
				else
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const bool en = ebit[j] == ibit;
					res[j] = en ? sum[j] : res[j];
					ce[j] = en ? Pn : ce[j];
				}
			}
			slotC = slotN; mC = mN; slotN = slotNN; mN = mNN;
		};
		for (cb_u32 ibit = 1u; ibit < Ebit; ibit <<= 2) {
			walk_row(va, vb, ibit);
			if ((ibit << 1) < Ebit) walk_row(vb, va, ibit << 1);
		}
	} else {
		cb_u32 mnext = ext > 0 ? (cb_u32)Ml[slot * TW + c] : 0u;
		for (cb_u32 ibit = 1u; ibit < Ebit; ibit <<= 1) {   // row i of the walk, as its (wave-uniform) bit 1 << i
			const cb_u32 m = mnext;
			const int n = (int)(m >> 8);
			const cb_u32 pa = __umul24((cb_u32)slot, (cb_u32)(SW * 4)) + cv - (m & 0xffu);
			slot = slot + 1 == RR ? 0 : slot + 1;
			const cb_u32 mr = Ml[slot * TW + c];   // the next row's run travels while this row is summed
			mnext = (ibit << 1) < extbit ? mr : 0u;   // (i + 1 < ext)
			const bool anyev = MC_TILE_ANYEV(ibit);   // most rows of a tall chunk start / end no output: the per-output tests are skipped
			if (anyev) {
				if constexpr (EVFAST) tile_events_start(ibit, sbit, sum, cb, Pn);
				else
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const bool st = sbit[j] == ibit;   // the output's first row: its chain starts from +0.0 here
					sum[j] = st ? 0.0f : sum[j];
					cb[j] = st ? Pn : cb[j];
				}
			}
			tile_taps<(A > 4)>(pa, n, sum);
			Pn += n;
			if (anyev) {
				if constexpr (EVFAST) tile_events_end(ibit, ebit, sum, res, ce, Pn);
				else
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const bool en = ebit[j] == ibit;   // the output's last row
					res[j] = en ? sum[j] : res[j];
					ce[j] = en ? Pn : ce[j];
				}
			}
		}
	}
#pragma unroll
	for (int j = 0; j < 4; ++j)
		if (outj[j]) OUTl[(4 * g + j) * TW + c] = res[j] / (float)(ce[j] - cb[j]);
	if (E >= 14) __builtin_amdgcn_s_setprio(0);
}

// P.gx x P.gy regions of TW columns x P.rb rows (a multiple of TH); a block takes one (region, plane).
// The order of a step's items depends on the pair's arms and the plane only -- not on the volume -- and a pair is aggregated many
// times over (main.lua:998-1001, 1033-1039: 2 + 16 iterations per direction on Middlebury).  MODE 1: the launch sorts and also
// writes every step's table + class counts and its own rows' combined runs / vertical arms to P.plan; MODE 2: it reads them a
// step ahead, together with the volume's rows -- no packed lengths, no sort pass, two barriers per step instead of four;
// MODE 0: no plan (adcensus.cbca called on its own).
template <int A, int TW, int TH, int NWAVES, bool NT, int MODE>
__global__ void __launch_bounds__(64 * NWAVES, NWAVES == 8 ? 6 : (NWAVES == 16 ? 4 : 1)) cbca_tile_kernel(const CbcaArgs P)
{
	using G = TileGeo<A, TW, TH, MODE>;
	constexpr int AH = G::AH, SW = G::SW, RR = G::RR, NI = G::NI, NKEY = G::NKEY;
	constexpr int NTHREADS = 64 * NWAVES;
	constexpr int VOL_AUX = NT ? 2 : 0;
	constexpr int NG = G::NG;
	constexpr int KT_ROWS = 14;        // "tall": items of this many rows and more
#define MC_TILE_NSPLIT 64   // (measured: 0 / 64 / 128 / 256 -> 2.90 / 2.86 / 2.89 / 3.08 ms per launch on the realistic pair)
	constexpr int NSPLIT_MAX = MC_TILE_NSPLIT;   // a step's tall items are taken apart into single outputs if there are at most this many
	constexpr bool PIPE = MODE == 2;   // values requested a row ahead: 10 more registers, which only the plan-reading instance has
	static_assert(TH % 4 == 0 && TW % 64 == 0 && TW <= 256 && TH / 4 <= 256 && 2 * A + 5 < NKEY && A <= 15 && NG % NWAVES == 0, "tile geometry");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float *__restrict__ Vl = (float *)smem;
	unsigned short *__restrict__ Ml = (unsigned short *)(smem + G::V_BYTES);
	unsigned char *__restrict__ UDl = smem + G::V_BYTES + G::M_BYTES;
	float *__restrict__ OUTl = (float *)(smem + G::V_BYTES + G::M_BYTES + G::UD_BYTES);
	unsigned short *__restrict__ TABl = (unsigned short *)(smem + G::V_BYTES + G::M_BYTES + G::UD_BYTES + G::OUT_BYTES);
	cb_u32 *__restrict__ GHl = (cb_u32 *)(smem + G::V_BYTES + G::M_BYTES + G::UD_BYTES + G::OUT_BYTES + G::TAB_BYTES);   // [group][key]
	cb_u32 *__restrict__ CTRl = GHl + (MODE == 2 ? 0 : NG * NKEY);   // [0] next chunk

	if (P.route == CR_PLANNED_TILE13) {   // (mc_predict: also a texture route whose list is unusable -- flat regions next to the texture)
		if (!cbca_gate_planned(P.flags, P.route, (const uint32_t *)P.plan, P.D, P.H, P.W, P.direction, P.lean_rb)) return;
	} else if (!cbca_gate(P.flags, P.route)) return;   // (the pair's arms call for another kernel)
	if (MODE == 2 && !plan_valid(P, A)) return;   // (not this problem's plan)
	if (MODE == 1 && blockIdx.x == 0 && threadIdx.x == 0) {
		uint32_t *h = (uint32_t *)P.plan;
		h[PH_D] = (uint32_t)P.D; h[PH_H] = (uint32_t)P.H; h[PH_W] = (uint32_t)P.W; h[PH_DIR] = (uint32_t)(P.direction + 1); h[PH_ARM] = (uint32_t)A;
		h[PH_MAGIC] = PH_MAGIC_VALUE;
	}
	const int tid0 = threadIdx.x;
	const int tid = tid0, lane = tid & 63;
	const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: per-row store descriptors stay in SGPRs
	const int H = P.H, W = P.W;
	const int HWi = H * W;
	// block -> (region, plane): the blocks of one XCD (blockIdx % 8) walk all planes of a region before the next region, so
	// the region's packed lengths (left: identical for every d, right: windows shifted by d) are fetched from HBM once per
	// XCD and served by its L2 for the other planes
	const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
	const int region = (s / P.nd) * 8 + xcd;
	const int d = P.d0 + s % P.nd;
	if (region >= P.gx * P.gy) return;
	const int cx = region % P.gx, cy = region / P.gx;
	const int tx0 = cx * TW, ys = cy * P.rb, ye = min(H, ys + P.rb);
	const int sh = d * P.direction;
	const int sx0 = tx0 - AH, yr0 = ys - A;   // image column / row of the ring's column 0 / relative row 0
	const cb_u32 OOB = 0x80000000u;
	const int plane_bytes = HWi * 4;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(P.vin + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(P.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(P.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	// outputs with a partner: columns [lo, lo + span) (adcensus.cu:353); the others are copied through
	const int lo = max(0, -sh);
	const cb_u32 span = (cb_u32)max(0, min(W, W - sh) - lo);
	const bool edge_tile = tx0 < lo || tx0 + TW > lo + (int)span;   // (block-uniform)
	constexpr int ENT = G::ENT_BYTES;
	const int nsteps = (ye - ys + TH - 1) / TH;
	const __amdgpu_buffer_rsrc_t rplan = __builtin_amdgcn_make_buffer_rsrc(
		MODE ? (void *)((char *)P.plan + PLAN_HDR + ((size_t)d * (size_t)(P.gx * P.gy) + (size_t)region) * (size_t)P.spr * ENT) : nullptr, 0, MODE ? P.spr * ENT : 0, 0x00020000);

	// ... and behind the tables, per plane, the combined runs (2 bytes per pixel) and vertical arms (1 byte) exactly as the
	// first pass committed them to its ring: rows of Wp = W rounded up to whole tiles
	const int Wp = P.wp;
	const __amdgpu_buffer_rsrc_t rpm = __builtin_amdgcn_make_buffer_rsrc(MODE ? (void *)((char *)P.plan + P.plan_m + (size_t)d * H * Wp * 2) : nullptr, 0,
	                                                                     MODE ? H * Wp * 2 : 0, 0x00020000);
	const __amdgpu_buffer_rsrc_t rpu = __builtin_amdgcn_make_buffer_rsrc(MODE ? (void *)((char *)P.plan + P.plan_ud + (size_t)d * H * Wp) : nullptr, 0,
	                                                                     MODE ? H * Wp : 0, 0x00020000);

	if (MODE != 2)
		for (int q = tid; q < NG * NKEY; q += NTHREADS) GHl[q] = 0;
	if (tid == 0) CTRl[0] = 0;

	// ---- rows in flight: TH rows of values (all staged columns) and packed lengths (output columns) per thread -------
	constexpr int UPR = SW / 4, APR = TW / 4;   // 4-column units per row
	// (index arithmetic of the requests / commits: 24-bit multiplies are full rate, 32-bit ones a quarter)
#define TMUL(a, b) __mul24((a), (b))
#define TQDIV(q) ((int)(__umul24((cb_u32)(q), UPR_RCP) >> 20))
	constexpr cb_u32 UPR_RCP = ((1u << 20) + UPR - 1) / UPR;
	static_assert([] { for (cb_u32 q = 0; q < 4096; ++q) if (((q * UPR_RCP) >> 20) != q / UPR) return false; return true; }(), "q / UPR by multiplication");
	constexpr int NV = (TH * UPR + NTHREADS - 1) / NTHREADS, NA = (TH * APR + NTHREADS - 1) / NTHREADS;
	struct Rows { cb_u4 v[NV], a[NA], b[NA]; cb_u2 m[NA]; cb_u32 ud[NA]; };   // (a, b: MODE 0 / 1; m, ud: MODE 2)
	const int ylast = min(H, ye + A);   // rows from here on are never needed
	// relative rows rr0 .. rr0 + nrows - 1 (relative row rr = image row yr0 + rr) -> registers; rows outside the image: zeros
	auto fetch_rows = [&](Rows &R, int rr0, int nrows, int tid) {
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const int q = tid + k * NTHREADS;
			const int r = TQDIV(q), u = q - r * UPR;
			const int y = yr0 + rr0 + r, x = sx0 + 4 * u;
			const bool rok = r < nrows && y >= 0 && y < ylast;
			// one 16-byte load wherever the unit starts: columns left of the image read the end of the row above (or, before the
			// plane, nothing), columns right of it the start of the row below (or, behind the plane, nothing) -- no run ever
			// includes a column outside the image, so what those words hold is never an operand
			R.v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, rok ? (cb_u32)(TMUL(y, W) + x) * 4u : OOB, 0, VOL_AUX);
		}
#pragma unroll
		for (int k = 0; k < NA; ++k) {
			const int q = tid + k * NTHREADS;
			const int r = q / APR, u = q - r * APR;
			const int y = yr0 + rr0 + r, x = tx0 + 4 * u;
			const bool rok = r < nrows && y >= 0 && y < ylast;
			if constexpr (MODE == 2) {   // the combined runs and vertical arms as the first pass committed them
				const bool in = rok && x < Wp;
				const cb_u32 pix = (cb_u32)(TMUL(y, Wp) + x);
				R.m[k] = __builtin_amdgcn_raw_buffer_load_b64(rpm, in ? pix * 2u : OOB, 0, 0);
				R.ud[k] = __builtin_amdgcn_raw_buffer_load_b32(rpu, in ? pix : OOB, 0, 0);
			} else {
				const int base = TMUL(y, W) + x;
				// the padded scratch makes any in-row start readable; columns outside the image / the shifted range are masked at commit
				R.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
				R.b[k] = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
			}
		}
	};
	// ... -> ring slots (slot0 = ring slot of relative row rr0)
	auto commit_rows = [&](const Rows &R, int rr0, int slot0, int nrows, int tid) {
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const int q = tid + k * NTHREADS;
			const int r = TQDIV(q), u = q - r * UPR;
			if (r >= nrows) continue;
			int slot = slot0 + r;
			slot = slot >= RR ? slot - RR : slot;
			*(cb_u4 *)(Vl + TMUL(slot, SW) + 4 * u) = R.v[k];
		}
#pragma unroll
		for (int k = 0; k < NA; ++k) {
			const int q = tid + k * NTHREADS;
			const int r = q / APR, u = q - r * APR;
			if (r >= nrows) continue;
			int slot = slot0 + r;
			slot = slot >= RR ? slot - RR : slot;
			const int y = yr0 + rr0 + r, x = tx0 + 4 * u;
			const bool yin = y >= 0 && y < ylast;   // (rows from ylast on are never looked at: like rows outside the image, no outputs, no runs)
			cb_u2 m4;
			cb_u32 ud4;
			if constexpr (MODE == 2) {
				const bool in = yin && x < Wp;
				m4 = in ? R.m[k] : cb_u2{0u, 0u};
				ud4 = in ? R.ud[k] : 0xffffffffu;
			} else {
				const cb_u32 spanr = yin ? span : 0u;   // pixel exists, partner inside the image (adcensus.cu:353)
				const cb_u32 mm[4] = {bytemin4(R.a[k].x, R.b[k].x), bytemin4(R.a[k].y, R.b[k].y), bytemin4(R.a[k].z, R.b[k].z), bytemin4(R.a[k].w, R.b[k].w)};
				cb_u32 run[4], ud[4];
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					const bool ok = (cb_u32)(x + t - lo) < spanr;
					const cb_u32 l = mm[t] & 0xffu, rr = (mm[t] >> 8) & 0xffu;
					run[t] = ok ? ((4u * l) | ((l + rr + 1u) << 8)) : 0u;
					ud[t] = ok ? (((mm[t] >> 16) & 15u) | ((mm[t] >> 20) & 0xf0u)) : 0xffu;
				}
				m4 = cb_u2{run[0] | (run[1] << 16), run[2] | (run[3] << 16)};
				ud4 = ud[0] | (ud[1] << 8) | (ud[2] << 16) | (ud[3] << 24);
				if constexpr (MODE == 1) {   // the block's own rows -> plan (halo rows belong to the regions above / below)
					const bool own = y >= ys && y < ye && x < Wp;
					__builtin_amdgcn_raw_buffer_store_b64(m4, rpm, own ? (cb_u32)(y * Wp + x) * 2u : OOB, 0, 0);
					__builtin_amdgcn_raw_buffer_store_b32(ud4, rpu, own ? (cb_u32)(y * Wp + x) : OOB, 0, 0);
				}
			}
			*(cb_u2 *)(Ml + slot * TW + 4 * u) = m4;
			*(cb_u32 *)(UDl + slot * TW + 4 * u) = ud4;
		}
	};

	// plan entry of step s -> registers -> the item table (MODE 2); threads 0 .. NI/4 - 1 carry four items each, the next two the header
	cb_u2 tabr = {0u, 0u};
	auto fetch_tab = [&](int s, int tid) {
		tabr = __builtin_amdgcn_raw_buffer_load_b64(rplan, (tid <= NI / 4 + 1 && s < nsteps) ? (cb_u32)(s * ENT + tid * 8) : OOB, 0, 0);
	};
	auto commit_tab = [&](int tid) {
		if (tid <= NI / 4 + 1) *(cb_u2 *)(TABl + 4 * tid) = tabr;
	};

	Rows R;
	for (int rr0 = 0; rr0 < RR; rr0 += TH) {   // the first step's window
		const int nrows = min(TH, RR - rr0);
		fetch_rows(R, rr0, nrows, tid);
		commit_rows(R, rr0, rr0, nrows, tid);
	}
	if (MODE == 2) {
		fetch_tab(0, tid);
		commit_tab(tid);
	}
	__syncthreads();

	int base = 0;   // ring slot of the step's first window row
	fetch_rows(R, RR, ys + TH < ye ? TH : 0, tid);   // the rows the second step adds
	if (MODE == 2) fetch_tab(1, tid);
	int sidx = 0;   // step of the region
	for (int y0 = ys, rrn = RR; y0 < ye; y0 += TH, rrn += TH, ++sidx) {
		const bool more = y0 + TH < ye;
		TPROF(0);
		TPROF(1);

		// ---- outputs without a partner are copied through (adcensus.cu:353-354): whole columns, and only in tiles that reach
		// beyond [lo, lo + span).  (Pixels outside the image never leave the tile: the row stores below are clipped.)
		if (edge_tile) {
			for (int q = tid0; q < TH * TW; q += NTHREADS) {
				const int r = q / TW, cc = q - r * TW;
				if ((cb_u32)(tx0 + cc - lo) >= span) {
					int slot = base + A + r;
					slot = slot >= RR ? slot - RR : slot;
					OUTl[q] = Vl[slot * SW + AH + cc];
				}
			}
		}

		int nz, nfast, ngen, ntall;   // items [0, ngen): general walk (the first ntall of them KT_ROWS rows or taller), [ngen, nfast): three-row class, [nfast, nz): four 3 x 3 supports
		if constexpr (MODE != 2) {
			// ---- items sorted by height (tallest first): counting sort ------------------------------------------------
			// key of an item: 0 = nothing to compute, 1 = its four supports are all the minimal 3 x 3, 2 = its four outputs all reach
			// exactly one row up and down (six rows, row r feeds outputs max(0, r - 2) .. min(3, r): no per-lane bookkeeping), else
			// height + 1 (heights 1 and 2 share key 3).  Ranks inside
			// a group of 64 consecutive columns, per-group counts, one scan: neighbouring columns of one class stay neighbouring
			// lanes, so a chunk's LDS reads are mostly consecutive words.
			const int wv = wvs;
			int lanes = tid0 & 63;
			asm volatile("" : "+v"(lanes));   // (opaque per step, as tidc below)
			constexpr int IPT = NG / NWAVES;
			cb_u32 keyrank[IPT];
#pragma unroll
			for (int k = 0; k < IPT; ++k) {
				const int gi = wv + k * NWAVES, i = gi * 64 + lanes;
				const int c = i % TW, g = i / TW;
				int s0[4], e0[4], top, bot;
				const cb_u32 udall = tile_item_rows<A, TW, RR>(UDl, base, c, g, s0, e0, top, bot);
				int key = bot >= top ? max(bot - top + 2, 3) : 0;
				if (udall == 0x11111111u) {   // all four outputs reach one row up and down: 3 x 3 each if the six rows' runs are (1, 1)
					bool mini = true;
					int slot = base + A + 4 * g - 1;
					slot = slot >= RR ? slot - RR : slot;
#pragma unroll
					for (int r = 0; r < 6; ++r) {
						mini = mini && Ml[slot * TW + c] == 0x0304u;
						slot = slot + 1 == RR ? 0 : slot + 1;
					}
					key = mini ? 1 : 2;
				}
				// rank inside (group, key): one LDS atomic per item (the order among equal keys is whatever the LDS serves -- lane
				// order in practice -- and changes no result: every output is computed by exactly one lane, whichever it is)
				const cb_u32 rank = atomicAdd(&GHl[gi * NKEY + key], 1u);
				keyrank[k] = (cb_u32)key | (rank << 8);
			}
			TPROF(2);
			__syncthreads();
			TPROF(3);
			// counts -> positions, in every wave for itself (no second barrier): lane l owns key l.  Keys descending, inside a key
			// group after group.
			cb_u32 posbase[IPT];
			{
				cb_u32 total = 0, pre[IPT];
#pragma unroll
				for (int k = 0; k < IPT; ++k) pre[k] = 0;
#pragma unroll
				for (int gi = 0; gi < NG; ++gi) {
					const cb_u32 n = lane < NKEY ? GHl[gi * NKEY + lane] : 0u;
#pragma unroll
					for (int k = 0; k < IPT; ++k) pre[k] = gi == wv + k * NWAVES ? total : pre[k];
					total += n;
				}
				// inclusive scan over the keys (lanes 0 .. 31): DPP row shifts, then row 0's sum into row 1
				cb_u32 sc = total;
				sc += (cb_u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x111, 0xF, 0xF, false);   // row_shr:1
				sc += (cb_u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x112, 0xF, 0xF, false);   // row_shr:2
				sc += (cb_u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x114, 0xF, 0xF, false);   // row_shr:4
				sc += (cb_u32)__builtin_amdgcn_update_dpp(0, (int)sc, 0x118, 0xF, 0xF, false);   // row_shr:8
				sc += (cb_u32)__builtin_amdgcn_update_dpp(0, (int)sc, DPP_ROW_BCAST15, 0xA, 0xF, false);
				const cb_u32 all = (cb_u32)__builtin_amdgcn_readlane((int)sc, NKEY - 1);
				const cb_u32 above = all - sc;   // items of larger keys
				nz = NI - __builtin_amdgcn_readlane((int)total, 0);           // items with at least one output to compute
				nfast = nz - __builtin_amdgcn_readlane((int)total, 1);        // ... of which the last ones are four 3 x 3 supports each
				ngen = nfast - __builtin_amdgcn_readlane((int)total, 2);      // ... preceded by the three-row class
				ntall = (int)(all - (cb_u32)__builtin_amdgcn_readlane((int)sc, KT_ROWS));   // keys > KT_ROWS: heights >= KT_ROWS
#pragma unroll
				for (int k = 0; k < IPT; ++k) posbase[k] = above + pre[k];
			}
#pragma unroll
			for (int k = 0; k < IPT; ++k) {
				const int gi = wv + k * NWAVES, i = gi * 64 + lanes;
				const int c = i % TW, g = i / TW;
				const cb_u32 key = keyrank[k] & 0xffu, rank = keyrank[k] >> 8;
				const cb_u32 first = (cb_u32)__builtin_amdgcn_ds_bpermute((int)(key * 4u), (int)posbase[k]);
				TABl[first + rank] = (unsigned short)(c | (g << 8) | (key << 11));   // (TH / 4 <= 8 row groups, keys < 32)
			}
			TPROF(4);
			__syncthreads();
			TPROF(5);
			if constexpr (MODE == 1) {   // the finished table (and the two counts behind it) -> this step's plan entry
				int tw = tid0;
				asm volatile("" : "+v"(tw));
				cb_u2 ent2 = *(const cb_u2 *)(TABl + 4 * (tw <= NI / 4 ? tw : 0));
				if (tw == NI / 4) ent2 = cb_u2{(cb_u32)nz, (cb_u32)nfast};
				if (tw == NI / 4 + 1) ent2 = cb_u2{(cb_u32)ngen, (cb_u32)ntall};
				__builtin_amdgcn_raw_buffer_store_b64(ent2, rplan, tw <= NI / 4 + 1 ? (cb_u32)(sidx * ENT + tw * 8) : OOB, 0, 0);
			}
		} else {
			const cb_u32 *hdr = (const cb_u32 *)(TABl + NI);
			nz = max(0, min(__builtin_amdgcn_readfirstlane((int)hdr[0]), NI));   // (clamped: an entry nobody wrote must not turn into a long loop)
			nfast = max(0, min(__builtin_amdgcn_readfirstlane((int)hdr[1]), nz));
			ngen = max(0, min(__builtin_amdgcn_readfirstlane((int)hdr[2]), nfast));
			ntall = max(0, min(__builtin_amdgcn_readfirstlane((int)hdr[3]), nz));
		}
		// A step's tallest chunk is one wave walking a chain of thousands of instructions while the block's other waves wait at
		// the barrier.  Where a step has only a few tall items (flat regions entering the tile) they are taken apart instead: a
		// unit of 16 items, one OUTPUT per lane -- the same rows in the same order, one accumulator each, 0.6 of the chain, on four
		// times the waves.  (Where most of the step is tall the chunks balance by themselves and the shared walk is the cheaper one.)
		const int nsplit = (A > 4 && ntall <= NSPLIT_MAX) ? ntall : 0;
		const int usplit = (nsplit + 15) >> 4;
		const int nunits = usplit + ((nz - nsplit + 63) >> 6);

		// ---- work units, tallest first: split units of 16 items, then chunks of 64 items -----------------------------------
		for (;;) {
			int unit = 0;
			if (lane == 0) unit = (int)atomicAdd(&CTRl[0], 1u);
			unit = __builtin_amdgcn_readfirstlane(unit);
			if (unit >= nunits) break;
			tile_unit<A, TW, SW, RR, PIPE>(unit, lane, Vl, Ml, UDl, OUTl, TABl, base, nz, nfast, ngen, nsplit, usplit);
		}
		TPROF(6);
		__syncthreads();
		TPROF(7);

		// ---- the rows the next step adds replace the oldest ones; results leave as rows; the rows of the step after next are
		// requested.  Order matters on gfx9, where stores and loads share one in-order counter: the committed rows were
		// requested a whole step ago and are waited for BEFORE this step's stores are issued, and the stored values keep their
		// registers until the new loads are out -- so nothing here waits for a store to complete.
		int tidc = tid0;
		asm volatile("" : "+v"(tidc));   // (opaque: the per-thread index arithmetic of commit / fetch is redone per step instead of living in ~40 registers)
		// (no branch around the commit / the requests: on a path that skips them hipcc's wait-count model keeps the previous
		// loads pending and protects their registers with full waits in the middle of the next step)
		commit_rows(R, rrn, base, more ? TH : 0, tidc);   // relative row rrn = RR + k TH lives in slot (k TH) mod RR = base: the oldest rows go
		if (MODE == 2) commit_tab(tidc);                  // ... and the next step's item table
		TPROF(8);
		// a wave stores whole rows through a descriptor that ends with the row: the words of a last unit that lie beyond the
		// image (W not a multiple of 4) are dropped by the range check, rows beyond the region get an empty descriptor -- no branch
		constexpr int OPR = TW / 4;
		constexpr int NO = TH / NWAVES;
		static_assert(OPR <= 64 && TH % NWAVES == 0, "a wave stores whole rows");
		cb_f4 ov[NO];
#pragma unroll
		for (int k = 0; k < NO; ++k) {
			const int r = wvs * NO + k;
			const int y = y0 + r;
			const int x = tx0 + 4 * lane;
			const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void *)(P.vout + (size_t)d * HWi), 0, y < ye ? (y + 1) * W * 4 : 0, 0x00020000);
			ov[k] = *(const cb_f4 *)(OUTl + r * TW + 4 * (lane < OPR ? lane : 0));
			__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(ov[k].x), __float_as_uint(ov[k].y), __float_as_uint(ov[k].z), __float_as_uint(ov[k].w)}, rrow,
			                                       lane < OPR ? (cb_u32)(y * W + x) * 4u : OOB, 0, VOL_AUX);
		}
		if (MODE != 2)
			for (int q = tid; q < NG * NKEY; q += NTHREADS) GHl[q] = 0;
		if (tid == 0) CTRl[0] = 0;
		fetch_rows(R, rrn + TH, y0 + 2 * TH < ye ? TH : 0, tidc);
		if (MODE == 2) fetch_tab(sidx + 2, tidc);
#pragma unroll
		for (int k = 0; k < NO; ++k) asm volatile("" :: "v"(ov[k]));   // (the stored values keep their registers until here)
		base += TH;
		base = base >= RR ? base - RR : base;
		TPROF(9);
		__syncthreads();
		TPROF(10);
	}
}


// regions: strips of TW columns x row ranges of a multiple of TH rows; enough of them for ~5 regions per XCD, a multiple of 8
// where a nearby row split gives one
static void tile_regions(int H, int W, int TW, int TH, int &gx, int &gy, int &rb)
{
	gx = (int)cdiv(W, TW);
	const int steps = (int)cdiv(H, TH);
	// (round 4, one box: 80 / 160 regions per plane instead of 40 -- KITTI size 2.72 -> 2.75 / 2.76 ms of aggregation per pair, 1000x1500 97 -> 103 ms: more halo rows, no better balance)
#define MC_TILE_REGIONS 40
	gy = std::max(1, std::min(steps, (int)cdiv(MC_TILE_REGIONS, gx)));
	for (int t = gy; t < gy + 8 && t <= steps; ++t)
		if ((gx * t) % 8 == 0) { gy = t; break; }
	rb = (int)cdiv(steps, gy) * TH;
	gy = (int)cdiv(H, rb);
}

// the product's geometries (128 x 16 tiles for either arm class) share one plan layout
constexpr int PLAN_TW = 128, PLAN_TH = 16;
// [header (PLAN_HDR) | item tables: D x regions x steps per region entries | combined runs (D, H, Wp) u16 | vertical arms (D, H, Wp) u8], Wp = W rounded up to whole tiles
struct PlanLayout { size_t m, ud, total; int wp; };
static PlanLayout plan_layout(int D, int H, int W)
{
	int gx, gy, rb;
	tile_regions(H, W, PLAN_TW, PLAN_TH, gx, gy, rb);
	PlanLayout L;
	L.wp = gx * PLAN_TW;   // rows of whole tiles: a tile's 128 pixels of a row are two / one aligned 128-byte lines
	L.m = (PLAN_HDR + (size_t)D * gx * gy * (rb / PLAN_TH) * TileGeo<4, PLAN_TW, PLAN_TH>::ENT_BYTES + 255) / 256 * 256;
	L.ud = L.m + ((size_t)D * H * L.wp * 2 + 255) / 256 * 256;
	L.total = L.ud + (size_t)D * H * L.wp;
	return L;
}
size_t cbca_lean2x_bytes(int D, int H, int W);   // cbca_lean.hip: what the texture route's two-pass records take of the same area
int cbca_lean_rows(int D, int H, int W, int rb, bool two_pass);
size_t cbca_plan_bytes(int D, int H, int W) { return std::max(plan_layout(D, H, W).total, cbca_lean2x_bytes(D, H, W)); }

template <int A, int TW, int TH, int NWAVES, int MODE>
static int cbca_tiles_launch_mode(CbcaArgs P, bool nt, hipStream_t st)
{
	using G = TileGeo<A, TW, TH, MODE>;
	const int64_t blocks = (int64_t)cdiv((int64_t)P.gx * P.gy, 8) * 8 * P.nd;
	if (blocks > 0x7fffffff) {
		set_error("cbca_tiles: %lld blocks", (long long)blocks);
		return MC_EINVAL;
	}
	auto kern_nt = cbca_tile_kernel<A, TW, TH, NWAVES, true, MODE>;
	auto kern = cbca_tile_kernel<A, TW, TH, NWAVES, false, MODE>;
	if (G::LDS_BYTES > 64 * 1024) {   // (per device, cheap: on every launch that needs it, checked -- as mean2d does)
		const hipError_t e = hipFuncSetAttribute(nt ? (const void *)kern_nt : (const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
		if (e != hipSuccess) {
			set_error("cbca_tile: hipFuncSetAttribute(%d bytes of LDS): %s", G::LDS_BYTES, hipGetErrorString(e));
			return (int)e;
		}
	}
	if (nt) hipLaunchKernelGGL(kern_nt, dim3((unsigned)blocks), dim3(64 * NWAVES), G::LDS_BYTES, st, P);
	else hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NWAVES), G::LDS_BYTES, st, P);
	return check_launch("cbca_tile");
}

template <int A, int TW, int TH, int NWAVES>
static int cbca_tiles_launch(CbcaArgs P, bool nt, int plan_mode, hipStream_t st)
{
	tile_regions(P.H, P.W, TW, TH, P.gx, P.gy, P.rb);
	P.spr = P.rb / TH;
	const PlanLayout L = plan_layout(P.D, P.H, P.W);
	P.plan_m = L.m; P.plan_ud = L.ud; P.wp = L.wp;
	if constexpr (TW == PLAN_TW && TH == PLAN_TH) {
		static_assert(TileGeo<A, TW, TH>::ENT_BYTES == TileGeo<4, PLAN_TW, PLAN_TH>::ENT_BYTES, "one plan layout");
		if (plan_mode == 1) return cbca_tiles_launch_mode<A, TW, TH, NWAVES, 1>(P, nt, st);
		if (plan_mode == 2) return cbca_tiles_launch_mode<A, TW, TH, NWAVES, 2>(P, nt, st);
	}
	return cbca_tiles_launch_mode<A, TW, TH, NWAVES, 0>(P, nt, st);   // (other geometries: no plan)
}

// arm_class 4: every arm <= 4 (L1 <= 5); 13: every arm <= 13 (L1 <= 14).  route >= 0 (the caller does not know the arms): the
// launch stands down unless cbca_pack's route word equals it.  cfg.plan / cfg.plan_mode: see the kernel.
int cbca_tiles(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int arm_class, int route,
               hipStream_t st, const CbcaCfg &cfg)
{
	const int d0 = cfg.nd > 0 ? cfg.d0 : 0, nd = cfg.nd > 0 ? cfg.nd : D;
	CbcaArgs P;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	P.p0 = cs.p0; P.p1 = cs.p1;
	P.vin = vin; P.vout = vout;
	P.D = D; P.H = H; P.W = W; P.direction = direction;
	P.d0 = d0; P.nd = nd;
	P.rb = 0; P.gx = P.gy = 0; P.spr = 0;   // (regions: set by the launcher)
	P.lean_rb = 0;
	P.flags = route >= 0 ? cs.flag : nullptr;
	P.route = route;
	const int pm = cfg.plan ? cfg.plan_mode : 0;
	P.plan = pm ? cfg.plan : nullptr;
	if (route == CR_PLANNED_TILE13) {
		if (!pm || arm_class != 13) { set_error("cbca_tiles: the planned route needs the plan area and the long-arm instance"); return MC_EINVAL; }
		P.lean_rb = cbca_lean_rows(D, H, W, cfg.lean_rb, true);   // (the list head the gate looks at: the two-pass records')
	}
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)nd * H * W * 4 > ((int64_t)768 << 20);
	// cfg.variant selects the tile geometry (test / tuning hook; 0 = the product's choice)
	if (arm_class <= 4) {
		switch (cfg.variant) {
		case 1: return cbca_tiles_launch<4, 128, 32, 8>(P, nt, pm, st);
		case 2: return cbca_tiles_launch<4, 128, 32, 4>(P, nt, pm, st);
		case 3: return cbca_tiles_launch<4, 256, 16, 8>(P, nt, pm, st);
		default: return cbca_tiles_launch<4, 128, 16, 4>(P, nt, pm, st);
		}
	}
	switch (cfg.variant) {
	case 1: return cbca_tiles_launch<13, 128, 32, 8>(P, nt, pm, st);
	case 2: return cbca_tiles_launch<13, 128, 16, 4>(P, nt, pm, st);
	case 3: return cbca_tiles_launch<13, 64, 16, 4>(P, nt, pm, st);
	default: return cbca_tiles_launch<13, 128, 16, 8>(P, nt, pm, st);
	}
}

}  // namespace mc
