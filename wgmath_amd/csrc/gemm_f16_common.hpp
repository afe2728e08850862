// Shared by the f16 GEMM translation units (gemm_f16.hip: the shipped 16x16x32 kernel + launcher; gemm_f16_t128.hip: the 128 x 128
// kernel; gemm_f16_generic.hip: the generic fallback).
#pragma once
#include "wg_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

namespace wgf16 {


#ifndef WG_ABLATE
#define WG_ABLATE 0 // timing experiments only: 1 = no barrier, 2 = no DMA, 4 = no LDS reads, 8 = (unused), 16 = no lane swaps, 32 = no epilogue stores (bitmask); results are garbage
#endif

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
// The same 16 bytes at an address in GLOBAL memory that is only element-aligned (an operand view at an odd offset or with an odd leading dimension):
// the target runs the memory pipeline in unaligned-access mode -- one global_load / global_store_dwordx4 either way, at the aligned rate for 4-byte offsets and
// 0.9 of it for 2-byte ones (tools/cpp/unaligned_probe.hip, unaligned_dma_probe.hip; profiles/r06_unaligned_probe.txt) -- and the type says what is really known.
typedef half8_t __attribute__((aligned(2))) half8_u;
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define WG_AS1 __attribute__((address_space(1)))
#define WG_AS3 __attribute__((address_space(3)))

constexpr int BM = 256, BN = 256, BKH = 32; // K advances in half-steps of 32
constexpr int HA_BYTES = BM * BKH * 2;      // 16 KiB
constexpr int HB_BYTES = BN * BKH * 2;
constexpr int HSTAGE_BYTES = HA_BYTES + HB_BYTES; // 32 KiB
constexpr int NSLOT = 5;                    // 5 x 32 KiB = all 160 KiB of LDS

// Calibrated shares across XCDs (gemm_f16.hip, "balance"). The XCDs of one chip differ by a few per cent in speed under the power cap,
// hardware deals every XCD the same number of workgroups, and with a handful of tiles per CU whole tiles are too coarse to even that
// out. So a FAST XCD ("taker") starts with one extra unit per CU -- the first `p` stages of K of a tile that belongs to a SLOW XCD
// ("giver"), raw f32 accumulators written to a scratch tile -- and the giver's workgroup for that tile, in its last round, starts from
// those accumulators instead of zeros and runs the remaining stages: the same k-ordered accumulation chain in the same registers, bit
// for bit the unsplit result. Workgroup id b is unit b / 8 of XCD b % 8's list: `pre_cnt` prefix units, then the XCD's own tiles
// (ids = x mod 8, in order) of which up to two ranges are suffix units. Everything is decided on the host from measured per-XCD rates.
struct BalancePlan {
    uint32_t on, epoch;          // flag value of this launch's finished prefix units
    uint32_t *flags;             // [pair]
    float *part;                 // [pair][256 x 256] raw accumulators, tile-local column-major (the tail split's format)
    // Every field is a 32-bit word ON PURPOSE: the kernel indexes these arrays with a run-time slot number straight out of its argument
    // block, and with 1- and 2-byte elements hipcc (ROCm 7.2) builds scalar loads on a byte-granular base (kernarg + x) -- whose two
    // low address bits the scalar memory unit ignores: slot 7 then read slot 6's entry (found on the GPU, not in the source).
    uint32_t len[8];             // units of slot x (workgroups with b / 8 >= len[b % 8] exit)
    uint32_t pre_cnt[8], pre_src[8], pre_p[8], pre_slot[8], pre_pair0[8]; // taker: p stages of the tiles 8 (pre_slot + j) + pre_src, j < pre_cnt; pair pre_pair0 + j
    uint32_t suf_lo[8][2], suf_cnt[8][2], suf_p[8][2], suf_pair0[8][2];   // giver: own tiles [suf_lo, suf_lo + suf_cnt) start at stage suf_p from pair suf_pair0 + ...
};

// workgroup id b -> its unit (false: none, the workgroup exits). mode 0: whole tile; 1: prefix, stages [0, ns); 2: suffix, stages [kb, kb + ns).
// `stages` = K / 64. Shared by the kernel and by the host-side check of the planner (wg_debug_f16_balance_plan).
__host__ __device__ inline bool bal_decode(const BalancePlan &bp, uint32_t b, uint32_t stages, uint32_t &tile, uint32_t &mode, uint32_t &kb, uint32_t &ns,
                                           uint32_t &pair) {
    const uint32_t x = b & 7u, c = b >> 3;
    mode = 0; kb = 0; ns = 0; pair = 0; tile = b;
    if (c >= bp.len[x]) return false;
    const uint32_t npre = bp.pre_cnt[x];
    if (c < npre) {
        mode = 1; ns = bp.pre_p[x]; pair = bp.pre_pair0[x] + c;
        tile = 8u * (bp.pre_slot[x] + c) + bp.pre_src[x];
        return true;
    }
    const uint32_t t = c - npre;
    tile = 8u * t + x;
    for (int r = 0; r < 2; ++r)
        if (t - bp.suf_lo[x][r] < (uint32_t)bp.suf_cnt[x][r]) {
            mode = 2; kb = bp.suf_p[x][r]; ns = stages - kb; pair = bp.suf_pair0[x][r] + (t - bp.suf_lo[x][r]);
        }
    return true;
}

// N-panels with arrival counters (the M-sharded Gemm of comm.hip as ONE launch per step): the output is written panel by panel into the slots of a
// staging cube [panel][rank][np x ldc] (col_stride = all ranks' rows of a column; slot_rows = the rows of the ranks in front of this one) -- or, with
// col_stride == ldc and slot_rows == 0, simply into the columns of one matrix. The tiles are numbered panel by panel, the epilogue stores write through
// to memory, and every WAVE that has its stores acknowledged adds one to counters[p] (system scope, fire and forget): panel p is complete in memory
// when the word has grown by 4 x its tile count. A copy engine's stream waits on exactly that (hipStreamWaitValue32) while the kernel is still working
// on the next panels.
// Widths: n_main panels of `cols` columns, then a TAIL of 1 .. 8 panels given in tile columns (tail_tn) -- the narrower panels that let the last
// exchanges of a step hide under less and less compute (the only exposed exchange is the last panel's); the very last panel takes whatever is left of N.
constexpr int kPanelTail = 8;
struct PanelArgs {
    uint32_t cols;       // columns per main panel (a multiple of 256); 0: off
    uint32_t npanels, tiles; // n_main + tail panels; tiles of a main panel (tiles_m * cols / 256)
    uint32_t n_main;     // panels of full width
    uint64_t tail_tn;    // tile columns of the tail panels, 8 bits each, panel n_main + q in bits [8 q, 8 q + 8) (the last one: ceil of what is left);
                         // packed, not an array: a dynamically indexed member would move the by-value kernel argument into scratch
    uint64_t col_stride, slot_rows; // elements
    uint32_t *counters;  // [npanels] waves finished, running totals over every launch (never reset between launches: a waiter of an earlier
                         // launch may not have looked yet)
};

struct GemmArgs {
    const _Float16 *a; uint32_t lda; uint64_t a_batch;
    const _Float16 *b; uint32_t ldb; uint64_t b_batch;
    _Float16 *c; uint32_t ldc; uint64_t c_batch;
    uint32_t M, N, K;
    uint32_t tiles_m, tiles_n;
    // split-K: grid.y = nmats * nsplit, workgroup (z, s) covers K range s and writes an f32 slab of `part` ([z][s][N][M])
    uint32_t nsplit, k_per_split;
    float *part;
    float alpha, beta; // out = alpha * acc + beta * out (wg_gemm_ex)
    // 1: the result is stored past the caches (`sc1 nt`): chosen by the launcher when the operands fit the 256 MiB Infinity Cache and the
    // result on top of them would not -- the result then leaves the operands alone there (8192^3: +1.2 ... 2.8 %, profiles/r03_evidence.md section 8)
    uint32_t c_stream;
    // 1: op(A)'s LDS-DMA pieces carry the non-temporal hint (gemm_f16_t128.hip: one tile column, i.e. A is read once, and A too large for the Infinity Cache anyway)
    uint32_t a_nt;
    // "tail split" launches of the 16x16x32 kernel: tile id = tile_base + blockIdx.x; with tail_tiles > 0 the workgroup (tile, split)
    // writes its f32 partial tile to part[(split * tail_tiles + blockIdx.x) * 65536 + col_local * 256 + row_local]
    uint32_t tile_base, tail_tiles;
    // dynamic tile scheduler of the 16x16x32 kernel (nullptr: tile = blockIdx.x): 8 queue words, one per XCD, 128 bytes apart;
    // sched_tiles tiles are dealt out (gemm_f16.hip: m16_acquire_tile)
    unsigned long long *sched; uint32_t sched_tiles;
    // per-XCD rate measurement (nullptr: off): calib[2 x] += main-loop time of a tile that ran on XCD x (100 MHz ticks), calib[2 x + 1] += its stages
    unsigned long long *calib;
    BalancePlan bal;
    PanelArgs panel;
#ifdef WG_F16_TRACE
    uint32_t trace_tiles; // timing experiment: records in `part` (gemm_f16.hip)
#endif
};

#ifndef WG_NN_NOSWAP
#define WG_NN_NOSWAP 1  // 1: Gemm's (column-major) A fragments without v_permlane16_swap (round 6; layout: "NN A" in gemm_f16.hip). 0: round 5's paired reads + lane-row swaps
#endif

// LDS-DMA: 16 bytes per lane from `gsrc` (per-lane) to LDS byte address `lds_dst` + 16*lane (`lds_dst` wave-uniform).
// Issued through inline asm ON PURPOSE: hipcc cannot tell that the DMA into stage t+1 does not alias the ds_reads of
// stage t (one LDS array, no alias scopes) and would put `s_waitcnt vmcnt(0)` in front of the first ds_read of every
// K-step, serialising HBM latency with the MFMAs. Hidden from its scoreboard, the DMA stays in flight during the whole
// K-step; the kernel waits for it itself (wait_dma) right before the barrier that publishes the stage.
// Address form: wave-uniform 64-bit base in SGPRs + one 32-bit per-lane byte offset + immediate (cheaper to issue than a 64-bit
// per-lane address: ~36 vs ~60 cycles among MFMAs).
template <int IMM>
static __device__ __forceinline__ void glds16s(uint32_t voff, const void *sbase, uint32_t lds_dst) {
    if (WG_ABLATE & 2) return;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%c4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst), "i"(IMM));
}
static __device__ __forceinline__ void wait_dma_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's DMA pieces are still in flight (they retire in issue order)
template <int N>
static __device__ __forceinline__ void wait_dma_keep() { asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(N) : "memory"); }
static __device__ __forceinline__ short4_t lds_tr(const char *p) {
    if (WG_ABLATE & 4) { short4_t v; asm volatile("" : "=v"(v)); return v; }
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((WG_AS3 short4_t *)p);
}
static __device__ __forceinline__ half8_t lds_h8(const char *p) {
    if (WG_ABLATE & 4) { half8_t v; asm volatile("" : "=v"(v)); return v; }
    return *reinterpret_cast<const half8_t *>(p);
}
// the same reads from a 32-bit LDS byte address (a running value kept in a VGPR)
static __device__ __forceinline__ short4_t lds_tr_at(uint32_t addr) {
    if (WG_ABLATE & 4) { short4_t v; asm volatile("" : "=v"(v)); return v; }
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((WG_AS3 short4_t *)(uintptr_t)addr);
}
static __device__ __forceinline__ half8_t lds_h8_at(uint32_t addr) {
    if (WG_ABLATE & 4) { half8_t v; asm volatile("" : "=v"(v)); return v; }
    return *(WG_AS3 const half8_t *)(uintptr_t)addr;
}
static __device__ __forceinline__ half8_t cat(short4_t lo, short4_t hi) {
    short8_t v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
    return __builtin_bit_cast(half8_t, v);
}

// workgroup id -> tile, general form. Hardware deals ids round-robin to the 8 XCDs; XCD x = id % 8 takes a contiguous range of the
// order index o, and o walks strips of 4 tile rows column by column: the ~32 workgroups an XCD runs at a time then cover a compact
// 4 x 8 patch of the output (4 A panels + 8 B panels through its L2) whatever the tile counts -- in particular the tiles that share
// an A panel of a tall-skinny product (tiles_n = 4: four of them) run together instead of tiles_m ids apart (131072 x 1024 x 8192:
// 9.2 GB fetched with the column-major order, A read four times).
static __device__ __forceinline__ void tile_strips(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t &tm, uint32_t &tn) {
    const uint32_t nwg = tiles_m * tiles_n;
    const uint32_t q = nwg / 8u, r = nwg % 8u, xcd = bid % 8u, local = bid / 8u;
    const uint32_t o = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + local;
    const uint32_t strip = o / (4u * tiles_n);
    const uint32_t within = o - strip * 4u * tiles_n;
    const uint32_t h = min(4u, tiles_m - 4u * strip);
    tn = within / h;
    tm = 4u * strip + (within - tn * h);
}

// workgroup id -> (tile_m, tile_n) of the 256 x 256 kernels
static __device__ __forceinline__ void tile_of(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t &tm, uint32_t &tn) {
    if ((tiles_m % 16u) == 0 && (tiles_n % 16u) == 0) {
        // 256 consecutive ids = one 16x16 super-tile; hardware deals ids round-robin to the 8 XCDs: XCD x gets a 4x8 patch
        const uint32_t super = bid >> 8, within = bid & 255u;
        const uint32_t xcd = within & 7u, local = within >> 3;
        const uint32_t sm = super % (tiles_m / 16u), sn = super / (tiles_m / 16u); // (a serpentine over the super-tile columns measured the same: r03 evidence 8)
        tm = sm * 16u + (xcd & 3u) * 4u + (local & 3u);
        tn = sn * 16u + (xcd >> 2) * 8u + (local >> 2);
    } else {
        tile_strips(bid, tiles_m, tiles_n, tm, tn);
    }
}


// launch wrapper of gemm_f16_generic.hip (grid / arguments prepared by wgk_gemm_f16)
int generic_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g);
// gemm_f16_t128.hip: tm x 128 tiles, tm = 128 or 256 (g.tiles_m / g.tiles_n count those), grid = (tiles, nmats * nsplit); K per split % 64 == 0
int t128_launch(wg_ctx *ctx, bool trans, dim3 grid, const GemmArgs &g, int tm = 128);
// ... with m2 contiguous along N (the row-major GemmTr): Gemm only, K % 64 == 0, one split
int t128_launch_nt(wg_ctx *ctx, dim3 grid, const GemmArgs &g, int tm = 128);

} // namespace wgf16
