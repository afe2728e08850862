// f16 Gemm (extension -- the reference has no f16 kernel): out = m1 * m2 (NN) or m1^T * m2 (TN), column-major,
// f16 operands, f32 accumulation on the matrix cores, ONE rounding (RNE) to f16 at the end.
//
// This file: the shipped kernel (v_mfma_f32_16x16x32_f16, description below), the tail-split reduce and the launcher that picks
// between it, the previous-generation 32x32x16 kernel and the generic fallback (both in gemm_f16_generic.hip).
// Bound: MFMA (2.5 PFLOP/s dense peak); in practice the 1400 W package power cap (profiles/r01_evidence.md).
#include "gemm_f16_common.hpp"

namespace wgf16 {
namespace {

// ===============================================================================================================
// 16x16x32 variant (v_mfma_f32_16x16x32_f16), the shipped f16 kernel. Same block tile (256 x 256) and wave tile
// (128 x 128) as the 32x32x16 kernel above, but:
//   * 8 x 8 MFMA tiles of 16 x 16 per wave, a half-step (32 k) is ONE MFMA k-step: half the accumulator register traffic
//     per flop (4 instead of 16 accumulator registers per MFMA for half the flops) -- the chip is power-limited on this
//     kernel and clocks ~7 % higher for it -- and 64 issue gaps of 16 cycles per half-step with at most one LDS read, lane
//     swap or DMA instruction in each;
//   * B (k-contiguous) is staged per FULL stage (64 k): a DMA piece is 8 rows x 128 bytes, i.e. whole 128-byte lines of
//     global memory (64-byte row pieces cost the L2/TCP request slot of a full line: measured +7 %). LDS plan:
//     A ring 4 x 16 KiB (half-stages) + B ring 3 x 32 KiB (full stages) = 160 KiB;
//   * the LDS-DMA is `s_mov m0` one slot ahead + `global_load_lds` (nothing else in this kernel touches M0).
// Lane (i16 = lane&15, kg = lane>>4) feeds MFMA row/column i16 with k = 8 kg .. 8 kg + 7; C/D: column i16, rows 4 kg + r.
// MFMA tile t = 2 p + tb of the wave covers rows 32 p + 8 (i>>2) + 4 tb + (i&3), i = MFMA row: a lane's registers of a
// tile pair are 8 consecutive rows of C (one 16-byte store), and for NN the two tiles of a pair are the two 8-byte
// halves of the same 16-byte LDS units (below).
// LDS layouts. ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md,
// LDS); with G = {0,3,2,1}:
//   B: [row][64 k] = 128-byte rows; logical 16-byte chunk c (0..7) of a row sits at position c ^ (G((row>>2)&3) | ((row>>1)&1)<<2):
//      every lane group touches 16 distinct (row&1, position) bank quads -- conflict-free;
//   TN op(A): [row][32 k] = 64-byte rows per half-stage, chunk c at c ^ G((row>>3)&3) (A's rows are permuted as above).
//   NN A (m-contiguous): 256-byte blocks [k/4][m/32][4 k][32 m], read with ds_read_b64_tr_b16: a lane row (16 lanes) takes 4 k x 16 m of ONE k-group, lane
//      4 krow + a the 8 bytes (4 m) of 16-byte unit a of k row krow. A 32-lane pass holds two lane rows = two k-groups; reading the SAME 8-byte half of
//      every unit they would hit the same banks. Round 5 read the two tiles of a pair in one instruction (lane rows 0/1 the low/high halves) and put
//      the k-groups back on their lane rows with 16 v_permlane16_swap per half-step. Now (WG_NN_NOSWAP): every instruction reads ONE tile, lane row r
//      k-group r, and the two rows of a pass still cover all 64 banks because
//        * the tiles of a pair interleave by unit PARITY: tile tb holds rows 8 a + 4 (tb ^ (a & 1)) + e, i.e. half tb ^ (a & 1) of unit a, and
//        * the DMA pieces of ODD k-groups swap neighbouring units (unit a lands at position a ^ 1; a piece is one unit per lane, so that is only a
//          different per-lane global offset): in an odd k-group's block tile tb occupies exactly the 8-byte halves it does NOT occupy in an even one.
//      Lane (kg, i16) of the result then holds, for tile 2 p + tb, rows 32 p + 8 kg + 4 (tb ^ (kg & 1)) + r: still 8 consecutive rows per pair, the two
//      tiles' order exchanged in odd lane rows (one select per register in the epilogue, once per tile, against 16 swaps per 32 k). Same k order in
//      every MFMA, same accumulation chain per element: bit-identical to the swapped form and to GemmTr.
// Pipeline (H = half-step, s = H>>1 its stage). Fragments of half-step H+1 are read during H into the other register set; a
// half-step's synchronisation (lgkmcnt(0) + counted vmcnt + barrier) sits a few MFMA slots before its end, after its last fragment
// read and DMA piece, so the LDS slot of A(H) / B(s) is free from the start of H / of 2s+1. NN: during H, A(H+4) -> A slot H&3 (four
// half-stage slots, a barrier every half-step); TN: A in two full-stage slots, a barrier per stage. B: the first half of B(s+3)
// during odd H, the second half of B(s+2) during even H -> B slot (stage % 3). Everything a barrier must publish was issued >= 2
// half-steps before it. Ring offsets, global bases and LDS read addresses are running values updated in otherwise empty MFMA gaps
// (see "Issue model" in m16_tile).
// ===============================================================================================================
#ifndef WG_NN_KEEP
#define WG_NN_KEEP 16   // NN: DMA pieces allowed in flight at a half-step's counted wait (8 / 12 / 20 measured: no gain, evidence 3b'')
#endif
#ifndef WG_NN_DSTRIDE
#define WG_NN_DSTRIDE 8 // NN: a DMA piece every this many MFMA slots (>= 2) ...
#endif
#ifndef WG_NN_DOFF
#define WG_NN_DOFF 3    // ... starting at this slot (>= 1): 4 A pieces, then 4 B pieces
#endif
#ifndef WG_NN_SYNC_SLOT
#define WG_NN_SYNC_SLOT 60 // NN: lgkmcnt(0) + counted vmcnt + barrier after this MFMA slot
#endif
#ifndef WG_TN_SYNC_SLOT
#define WG_TN_SYNC_SLOT 56 // staged pipeline: likewise
#endif
#ifndef WG_F16_UNPEELED
#define WG_F16_UNPEELED 1 // 1: ONE loop over all S stages; for the last three the DMA cursors stay parked on the last stage (their pieces land in
                          //    LDS slots nobody reads any more). 0: the last three stages peeled into six half-steps that issue fewer / no pieces
#endif
#ifndef WG_NN_STAGED
#define WG_NN_STAGED 0  // 1: NN on the staged pipeline (A in two full-stage slots, one barrier per stage): measured slower (8192^3 -0.7 %, 32768^3 -4 %)
#endif
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef uintx4 __attribute__((aligned(2))) uintx4_u; // (in global memory at any element-aligned address: gemm_f16_common.hpp half8_u)

// compile-time loop: the body sees its index as a constant expression (no reliance on the unroller's size thresholds)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int M16_A_RING = 4, M16_B_RING = 3;
constexpr int M16_BS_BYTES = BN * 64 * 2;              // one full stage of B: 256 rows x 128 bytes = 32 KiB
constexpr int M16_B_BASE = M16_A_RING * HA_BYTES;      // 64 KiB
static_assert(M16_B_BASE + M16_B_RING * M16_BS_BYTES == 160 * 1024, "LDS plan");

__device__ __forceinline__ void m16_set_m0(uint32_t lds_dst) {
    if (WG_ABLATE & 2) return;
    asm volatile("s_mov_b32 m0, %0" ::"s"(lds_dst));
}
// LDS destination = M0 + IMM + 16 * lane; the instruction's immediate offset moves the GLOBAL address too, so the per-piece
// voff registers are biased by -IMM (see M16_BIAS). Needs >= 1 instruction since m16_set_m0.
#ifndef WG_F16_DMA_MUBUF
#define WG_F16_DMA_MUBUF 0 // 1: buffer_load_dwordx4 ... offen lds (descriptor + 32-bit offset) instead of global_load_lds_dwordx4 (64-bit scalar base)
#endif
template <int IMM>
__device__ __forceinline__ void m16_dma_imm(uint32_t voff, const void *sbase) {
    if (WG_ABLATE & 2) return;
    if constexpr (WG_F16_DMA_MUBUF) {
        // raw buffer descriptor over "everything from sbase on": stride 0, num_records 2^32 - 1 (offsets stay below 2^31: off_ok), gfx9 raw-buffer flags
        const uintx4 srd = { (uint32_t)(uintptr_t)sbase, (uint32_t)((uintptr_t)sbase >> 32) & 0xffffu, 0xffffffffu, 0x00020000u };
        asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:%c2 lds" ::"v"(voff), "s"(srd), "i"(IMM));
    } else {
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%c2" ::"v"(voff), "s"(sbase), "i"(IMM));
    }
}
__device__ __forceinline__ void m16_dma(int q, uint32_t voff, const void *sbase) { // piece q (0..3) of a group of four
    if (q == 0) m16_dma_imm<0>(voff, sbase);
    else if (q == 1) m16_dma_imm<1024>(voff, sbase);
    else if (q == 2) m16_dma_imm<2048>(voff, sbase);
    else m16_dma_imm<3072>(voff, sbase);
}
constexpr uint32_t M16_BIAS = 3072; // scalar bases are lowered by this many bytes, voff of piece q raised by BIAS - 1024 q

// One 256 x 256 tile (tile id `bid`), K range of this workgroup's split. `smem` is the kernel's 160 KiB LDS array.
// Balance units (BalancePlan; workgroup-uniform arguments): unit_mode 1 = PREFIX, stages [0, unit_ns) of the tile, raw f32 accumulators to
// scratch tile `unit_pair`, then its flag; 2 = SUFFIX, stages [unit_kb, unit_kb + unit_ns) on top of the accumulators of prefix `unit_pair`
// (waits for its flag; should that never come -- its workgroup was never scheduled -- the whole tile is computed here from zeros).
template <bool TRANS_A>
__device__ __forceinline__ void m16_tile(const GemmArgs &g, const uint32_t bid, char *const smem, const uint32_t unit_mode = 0, uint32_t unit_kb = 0,
                                         uint32_t unit_ns = 0, const uint32_t unit_pair = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    const int aq = i16 >> 2, bb = i16 & 3;
    const int gq = (4 - aq) & 3; // G(aq)

#ifdef WG_F16_TRACE
    // timing experiment: wave 0 records s_memrealtime (100 MHz) at 5 points + its hardware id into g.part[blockIdx.x * 8 ..]
    uint64_t tr_t[5];
    tr_t[0] = __builtin_amdgcn_s_memrealtime();
#define WG_TRACE_POINT(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WG_TRACE_POINT(i) do { } while (0)
#endif
#if defined(WG_F16_TRACE) && WG_F16_TRACE >= 2
    // fine trace: every wave splits each end-of-half-step into counted-DMA wait / barrier wait / probe latency (shader cycles, s_memtime);
    // the sums + the loop's span go to the second 64 bytes of the tile's trace record (see tools/f16_trace.py)
    uint32_t q_a = 0, q_b = 0, q_c = 0, q_d = 0, s_vm = 0, s_bar = 0, s_probe = 0, m_vm = 0, m_bar = 0, n_adv = 0, loop_t0 = 0, loop_t1 = 0;
#endif
    uint32_t tm, tn, panel = 0;
    uint32_t panel_c0 = 0, panel_np = 0; // first column / columns of this tile's panel (paneled launches)
    if (g.panel.cols) { // tiles are numbered panel by panel (PanelArgs); within a panel the usual XCD-aware order on the panel's own grid
        const uint32_t id = bid + g.tile_base;
        const uint32_t mtn = g.panel.cols / 256u, main_tiles = g.panel.n_main * g.panel.tiles;
        uint32_t loc, ptn, tn0;
        if (id < main_tiles) {
            panel = __builtin_amdgcn_readfirstlane(id / g.panel.tiles); // (the division runs on the vector unit)
            loc = id - panel * g.panel.tiles; ptn = mtn; tn0 = panel * mtn;
        } else { // the tail: at most kPanelTail narrower panels, widths in tile columns
            loc = __builtin_amdgcn_readfirstlane(id - main_tiles);
            tn0 = g.panel.n_main * mtn;
            const uint32_t n_tail = g.panel.npanels - g.panel.n_main;
            uint32_t q = 0;
            uint64_t tail = g.panel.tail_tn;
            ptn = (uint32_t)tail & 0xffu;
            while (q + 1u < n_tail && loc >= g.tiles_m * ptn) { loc -= g.tiles_m * ptn; tn0 += ptn; ++q; tail >>= 8; ptn = (uint32_t)tail & 0xffu; }
            panel = g.panel.n_main + q;
        }
        tile_of(loc, g.tiles_m, ptn, tm, tn);
        tn += tn0;
        panel_c0 = tn0 * 256u;
        panel_np = panel + 1u == g.panel.npanels ? g.N - panel_c0 : ptn * 256u;
    } else tile_of(bid + g.tile_base, g.tiles_m, g.tiles_n, tm, tn);
    panel = __builtin_amdgcn_readfirstlane(panel); // (all of these are workgroup-uniform; said explicitly: they end up in scalar operands of the DMA asm)
    panel_c0 = __builtin_amdgcn_readfirstlane(panel_c0); panel_np = __builtin_amdgcn_readfirstlane(panel_np);
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    const uint32_t m0 = tm * BM, n0 = tn * BN;
    const uint32_t z = blockIdx.y / g.nsplit, split = blockIdx.y % g.nsplit;
    uint32_t k_begin = split * g.k_per_split;
    uint32_t K_loc = split + 1u == g.nsplit ? g.K - k_begin : g.k_per_split; // (the last split takes whatever is left, K remainder included)
    bool from_partial = false; // workgroup-uniform
    if (unit_mode == 2) {
        // consumer side of the hand-off (guide, G16): ONE lane polls relaxed, ONE agent-scope acquire, barrier, then plain loads by everyone
        if (threadIdx.x == 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            uint32_t ok = 1;
            // (system scope: an agent-scope load is served by THIS XCD's L2, which may still hold the flag's line with last launch's epoch --
            // measured: 32 suffix units polling ~45 us each until the line happened to be replaced)
            while (__hip_atomic_load(g.bal.flags + unit_pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != g.bal.epoch) {
                __builtin_amdgcn_s_sleep(16);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 100000ull) { ok = 0; break; } // 1 ms: the prefix ran at the start of the launch or never will
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *reinterpret_cast<volatile uint32_t *>(smem) = ok;
        }
        __syncthreads();
        from_partial = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t *>(smem)) != 0;
        __syncthreads(); // everyone has read the word before the first DMA piece may land on it
    }
    // K remainder (K % 64, a multiple of 8): the k-values past the last whole stage of this workgroup's range are multiplied FIRST -- they
    // are the accumulators' initial value (rem_block below) -- by the unit that starts the tile's accumulation chain: a prefix unit,
    // a whole tile, the last split. A suffix unit continues a chain that already holds them.
    uint32_t rem = 0, rem_k0 = 0;
    if (unit_mode == 1) { k_begin = 0; K_loc = unit_ns * 64u; rem = g.K & 63u; rem_k0 = g.K - rem; }
    else if (unit_mode == 2 && from_partial) { k_begin = unit_kb * 64u; K_loc = (g.K & ~63u) - k_begin; }
    else {
        if (unit_mode == 2) { k_begin = 0; K_loc = g.K; } // a suffix unit without its prefix: the whole tile
        rem = K_loc & 63u; K_loc -= rem; rem_k0 = k_begin + K_loc;
    }
    const _Float16 *A = g.a + z * g.a_batch + (TRANS_A ? (uint64_t)k_begin : (uint64_t)k_begin * g.lda);
    const _Float16 *B = g.b + z * g.b_batch + k_begin;
    _Float16 *C = g.c + z * g.c_batch;
    const bool paneled = g.panel.cols != 0; // workgroup-uniform
    if (paneled) { // the epilogue indexes columns globally: C + col * ldc; panel p's columns start at its own base: the cube's columns are col_stride
        // apart, this rank's slot of panel p starts slot_rows * np(p) elements into it (C already points slot_rows * cols into panel 0)
        // (64-bit products run on the vector unit even when uniform: back to scalar registers by hand, C is pinned in SGPRs below)
        const uint64_t off = (uint64_t)panel_c0 * (g.panel.col_stride - g.ldc) + g.panel.slot_rows * (uint64_t)panel_np - g.panel.slot_rows * (uint64_t)g.panel.cols;
        C += ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
    }
    uint32_t *panel_counter = g.panel.counters + panel;
    // Pin the epilogue's kernel arguments in SGPRs NOW: left alone, the compiler loads them (s_load, also an lgkmcnt event,
    // returning out of order with LDS reads) right in front of the main loop, and every counted LDS wait of the first
    // half-step then degrades to lgkmcnt(0).
    float *part = unit_mode ? g.bal.part + (uint64_t)unit_pair * 65536u : g.part;
    float alpha = g.alpha, beta = g.beta;
    uint32_t ldc = g.ldc, c_stream = g.c_stream;
    uint32_t *bal_flag = g.bal.flags + unit_pair;
    uint32_t bal_epoch = g.bal.epoch;
    unsigned long long *calib = g.calib;
    asm volatile("" : "+s"(C), "+s"(part), "+s"(alpha), "+s"(beta), "+s"(ldc), "+s"(c_stream), "+s"(bal_flag), "+s"(bal_epoch), "+s"(calib));

    // ---- DMA addressing; ragged tiles: rows past the end are clamped to the last valid one (results discarded by the epilogue) ----
    // A half-stage = 16 pieces of 1 KiB, wave stages P = 4 wave + q; B full stage = 32 pieces, wave stages P = 8 wave + q.
    uint32_t a_voff[TRANS_A ? 8 : 4], b_voff[8];
    const _Float16 *a_base, *b_base = B + (uint64_t)n0 * g.ldb;
    if constexpr (TRANS_A) a_base = A + (uint64_t)m0 * g.lda; else a_base = A + m0;
    // K remainder (see the prologue): where its k-values start, and the loop's stage 0 becomes the remainder -- every base one stage back
    const uint32_t rem_dk = rem_k0 - k_begin;
    const char *rem_a = TRANS_A ? (const char *)(a_base + rem_dk) : (const char *)(a_base + (uint64_t)rem_dk * g.lda);
    const char *rem_b = (const char *)(b_base + rem_dk);
    if (rem) {
        if constexpr (TRANS_A) a_base -= 64; else a_base -= (uint64_t)64u * g.lda;
        b_base -= 64;
    }
    if constexpr (TRANS_A) { // op(A) rows are k-contiguous: full stages like B -- rows 8P..8P+7, 128 bytes each, P = 8 wave + q
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t P = 8u * wave + q;
            const uint32_t row = 8u * P + (lane >> 3);
            const uint32_t f = ((4u - ((row >> 3) & 3u)) & 3u) | (((row >> 1) & 1u) << 2); // A's rows are permuted: G index (row>>3)&3
            const uint32_t chunk = (lane & 7u) ^ f;
            const uint32_t ra = min(row, g.M - 1u - m0);
            a_voff[q] = (ra * g.lda + 8u * chunk) * 2u + (M16_BIAS - 1024u * (q & 3));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { // blocks 4P..4P+3: kq = P>>1, mblk = 4 (P&1) + (lane>>4), k row (lane>>2)&3, 16-byte unit lane&3
            const uint32_t P = 4u * wave + q;
            const uint32_t unit = WG_NN_NOSWAP ? (lane & 3u) ^ ((uint32_t)wave & 1u) : (lane & 3u); // (the piece's k-group within the half-stage is `wave`: odd ones swap neighbouring units)
            const uint32_t mpiece = min(128u * (P & 1) + 32u * (lane >> 4) + 8u * unit, g.M - 8u - m0); // M % 8 == 0
            a_voff[q] = ((4u * (P >> 1) + ((lane >> 2) & 3)) * g.lda + mpiece) * 2u + (M16_BIAS - 1024u * q);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { // rows 8P..8P+7, 128 bytes each: lane -> row 8P + (lane>>3), position lane&7
        const uint32_t P = 8u * wave + q;
        const uint32_t row = 8u * P + (lane >> 3);
        const uint32_t f = ((4u - ((row >> 2) & 3u)) & 3u) | (((row >> 1) & 1u) << 2);
        const uint32_t chunk = (lane & 7u) ^ f;
        const uint32_t rb = min(row, g.N - 1u - n0);
        b_voff[q] = (rb * g.ldb + 8u * chunk) * 2u + (M16_BIAS - 1024u * (q & 3));
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    // NN: + slot * 16 KiB + q * 1 KiB (4 half-stage slots); TN: + slot * 32 KiB + q * 1 KiB (2 full-stage slots)
    const uint32_t lds_a_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * (TRANS_A ? 8192 : 4096));
    const uint32_t lds_b_wave = __builtin_amdgcn_readfirstlane(lds_base + M16_B_BASE + wave * 8192); // + slot * 32 KiB + q * 1 KiB
    auto a_src = [&](uint32_t H) -> const _Float16 * { // NN: global base of half-stage H;  TN: of full stage H
        const char *p0;
        if constexpr (TRANS_A) p0 = (const char *)(a_base + H * 64u); else p0 = (const char *)(a_base + (uint64_t)(H * BKH) * g.lda);
        return (const _Float16 *)(p0 - M16_BIAS);
    };
    auto b_src = [&](uint32_t stage) -> const _Float16 * { return (const _Float16 *)((const char *)(b_base + 64u * stage) - M16_BIAS); };

    // ---- per-lane LDS read addresses (bytes from the start of the LDS) for half-step parity 0 / 1 of a stage ----
    uint32_t vbaseA[2], vbaseB[2];
    uint32_t vbaseA1 = 0; // NN without swaps: the odd tiles' base (the even tiles': vbaseA[0]); the staged pipeline's second half-stage: + HA_BYTES
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        const uint32_t chunk = (uint32_t)(((kg ^ gq) | ((hs ^ (bb >> 1)) << 2)) * 16);
        vbaseB[hs] = lds_base + M16_B_BASE + ((uint32_t)128 * wn + i16) * 128u + chunk;
        // TN: row 8 aq + bb of the even tile of a pair (odd tile: + 4 rows = + 512 bytes), chunk as B
        if constexpr (TRANS_A) vbaseA[hs] = lds_base + (128u * wm + 8u * aq + bb) * 128u + chunk;
    }
    if constexpr (!TRANS_A) {
        if constexpr (WG_NN_NOSWAP) {
            // lane row kg reads k-group kg (blocks kq = 2 kg + h, mblk = 4 wm + p); lane 4 krow + a: k row krow, unit a -- stored at position a ^ (kg & 1) --
            // and of it the 8-byte half tb ^ (a & 1) for tile tb
            const uint32_t a = (uint32_t)i16 & 3u, krow = (uint32_t)i16 >> 2;
            const uint32_t common = lds_base + (uint32_t)kg * 4096u + (4u * wm) * 256u + krow * 64u + (a ^ ((uint32_t)kg & 1u)) * 16u;
            vbaseA[0] = common + (a & 1u) * 8u;
            vbaseA1 = common + ((a & 1u) ^ 1u) * 8u;
        } else {
            // lane row kg reads k-group (kg&2) + ins, 8-byte half (kg&1) of unit i16 of block (kq = 2*kgroup + h, mblk = 4 wm + p)
            vbaseA[0] = lds_base + (uint32_t)((kg & 2) * 2) * 2048u + (4u * wm) * 256u + (uint32_t)i16 * 16u + (uint32_t)(kg & 1) * 8u;
        }
        vbaseA[1] = vbaseA[0] + HA_BYTES; // the second half-stage of a full stage (staged pipeline only)
    }
    // NN without swaps: odd lane rows hold the tiles of a pair in exchanged order (rows 32 p + 8 kg + 4 (tb ^ (kg & 1)) + r): see the epilogue
    const bool odd_row = !TRANS_A && WG_NN_NOSWAP && (kg & 1);

    floatx4 acc[8][8]; // [t][u]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.f;
    if (from_partial) {
        // the prefix unit's accumulators (written in the tail split's tile-local format by the same lane -> element map): this unit
        // continues their k-ordered chain. Loaded by inline asm straight into the accumulator registers and waited for here: a load the
        // compiler can see makes its wait-count pass put an `s_waitcnt vmcnt(n)` in front of the first MFMA of the LOOP (where it would
        // drain this kernel's DMA pipeline in every iteration), and 64 loads through VGPRs spill.
        const uint32_t pvoff = ((128u * wn + i16) * 256u + 128u * wm + 8u * kg) * 4u;
        const uint32_t pv_even = pvoff + (odd_row ? 16u : 0u), pv_odd = pvoff + (odd_row ? 0u : 16u); // rows + 0..3 / + 4..7 of the 8: which tile of the pair holds them
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t vo0 = pv_even + (uint32_t)u * 16384u, vo1 = pv_odd + (uint32_t)u * 16384u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // sc0 sc1: the scratch tile is rewritten by another XCD every launch; read it past this XCD's L2 as well
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%c3 sc0 sc1" : "=a"(acc[2 * p][u]) : "v"(vo0), "s"(part), "i"(p * 128));
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%c3 sc0 sc1" : "=a"(acc[2 * p + 1][u]) : "v"(vo1), "s"(part), "i"(p * 128));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    uintx4 a_r[2][8]; // [register set][M tile]  (bit patterns of 8 halves)
    half8_t b_f[2][8];

    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using yes = std::true_type;
    using no = std::false_type;
    using km1 = std::integral_constant<int, -1>;
    using km2 = std::integral_constant<int, -2>;
    using k0 = std::integral_constant<int, 0>;
    using k4 = std::integral_constant<int, 4>;
    using k8 = std::integral_constant<int, 8>;
    using k12 = std::integral_constant<int, 12>;
    using k16 = std::integral_constant<int, 16>;

    const uint32_t S = K_loc / 64u + (rem ? 1u : 0u); // stages of the loop (the K remainder, if any, is its stage 0); the launcher guarantees >= 3 whole ones
    uint32_t st = 0;                // current stage

    // ---- Issue model (measured: profiles/r02_evidence.md 3d). With one wave per SIMD, the 16 cycles of a 16x16x32 MFMA hide up to ~3
    // single-issue instructions of the same wave; every instruction beyond that in one MFMA-to-MFMA gap costs ~4.8 cycles. Recomputing
    // ring slots, DMA bases and LDS read addresses at the top of a half-step, behind the waits and the barrier, put ~18 instructions in
    // two gaps: ~80 of 1190 cycles per half-step. So every such quantity is a RUNNING value, updated in place in a gap of its own after
    // its last use in the half-step, and the synchronisation sits in an otherwise empty gap near the end (all fragment reads and all
    // DMA pieces of the half-step are issued by then, so the counts mean what they would at the very end; the MFMAs behind it only
    // touch registers). `asm volatile("" : "+s"(x))` pins an update to its slot. tools/gap_hist.py lists the fillers per gap.
    uint32_t va, vb;                                     // VGPRs: where this half-step's A / B fragment reads start
    uint32_t va1 = 0;                                    // NN without swaps: ... of the odd tiles' A reads (va: the even tiles')
    const char *ga, *gb, *ga2 = nullptr;                 // SGPR pairs: global bases (less M16_BIAS) of the A / B pieces issued this half-step
    uint32_t oR = 0, oD = 2u * M16_BS_BYTES;             // B ring (3 stages): byte offset of the slot read / DMA'd this half-step
    uint32_t rR = 1u << 14, rD = 0;                      // NN, 4 half-stage slots of A: offset of the slot read / DMA'd this half-step
    uint32_t oA = 0, oAD = 0;                            // staged pipeline, 2 full-stage slots of A: likewise
    uint32_t la = 0, lb = 0;                             // M0 values of the next group of four pieces
    const uint64_t a_step = (uint64_t)BKH * g.lda * 2u;  // NN: bytes between two half-stages of A in global memory (< 4 GiB: launcher)
    // Unpeeled loop: how far the cursors move after this stage's pieces -- 0 once the next piece would lie past the last stage
    // (recomputed when st advances). a_inc0 / a_inc1: ga after the even / odd half-step's A pieces; b_inc: gb after the even half-step's.
    constexpr bool STAGED = TRANS_A || WG_NN_STAGED;
    const uint32_t a_full = TRANS_A ? 128u : (uint32_t)a_step * (STAGED ? 2u : 1u);
    uint32_t a_inc0 = 0, a_inc1 = 0, b_inc = 0;
    auto ring3 = [](uint32_t o) -> uint32_t { o += (uint32_t)M16_BS_BYTES; return o == 3u * M16_BS_BYTES ? 0u : o; };

    // the fragment-producing operations of one half-step, in the order the next half-step consumes them
    constexpr int kOps = TRANS_A ? 16 : WG_NN_NOSWAP ? 24 : 40;
    auto frag = [&](int op, int set) {
        auto rb = [&](int u) { b_f[set][u] = lds_h8_at(vb + u * 2048); };
        if constexpr (TRANS_A) {
            auto ra = [&](int t) { a_r[set][t] = __builtin_bit_cast(uintx4, lds_h8_at(va + (t & 1) * 512 + (t >> 1) * 4096)); };
            if (op == 0) ra(0);
            else if (op <= 8) rb(op - 1);
            else ra(op - 8);
        } else if constexpr (WG_NN_NOSWAP) {
            // tr(p, i): transpose read i = 2 tb + h of pair p: the 4 k of half h (block kq = 2 kg + h) of this lane row's k-group, for tile 2 p + tb, dwords 2 h, 2 h + 1
            auto tr = [&](int p, int i) {
                const int tb = i >> 1, h = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr_at((tb ? va1 : va) + h * 2048 + p * 256));
                a_r[set][2 * p + tb][2 * h] = v[0];
                a_r[set][2 * p + tb][2 * h + 1] = v[1];
            };
            // in the order the next half-step's MFMAs want them: tiles 0, 1 | B 0..3 | tiles 2, 3 | B 4..7 | tiles 4..7
            if (op < 4) tr(0, op);
            else if (op < 8) rb(op - 4);
            else if (op < 12) tr(1, op - 8);
            else if (op < 16) rb(op - 8);
            else if (op < 20) tr(2, op - 16);
            else tr(3, op - 20);
        } else {
            // tr(p, i): transpose read i = 2 h + ins of pair p lands in tile 2p+ins, dwords 2h, 2h+1;
            // sw(p, i): lane-row swap of dword i of the pair's two tiles
            auto tr = [&](int p, int i) {
                const int h = i >> 1, ins = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr_at(va + (2 * ins + h) * 2048 + p * 256));
                a_r[set][2 * p + ins][2 * h] = v[0];
                a_r[set][2 * p + ins][2 * h + 1] = v[1];
            };
            auto sw = [&](int p, int i) {
                if (WG_ABLATE & 16) return;
                const uintx2 r = __builtin_amdgcn_permlane16_swap(a_r[set][2 * p][i], a_r[set][2 * p + 1][i], false, false);
                a_r[set][2 * p][i] = r[0];
                a_r[set][2 * p + 1][i] = r[1];
            };
            if (op < 4) tr(0, op);
            else if (op < 8) rb(op - 4);
            else if (op < 12) tr(1, op - 8);
            else if (op < 16) sw(0, op - 12);
            else if (op < 20) rb(op - 12);
            else if (op < 24) tr(2, op - 20);
            else if (op < 28) sw(1, op - 24);
            else if (op < 32) tr(3, op - 28);
            else if (op < 36) sw(2, op - 32);
            else sw(3, op - 36);
        }
    };
    // fragment op of slot j: TN one read every 3 slots (0..45); NN 3 ops in every 4 slots (0..52: the last read in slot 41, then swaps)
    auto frag_slot = [&](auto jc, int set) {
        constexpr int j = decltype(jc)::value;
        if constexpr (TRANS_A) { if constexpr ((j % 3) == 0 && (j / 3) < kOps) frag(j / 3, set); }
        else { if constexpr ((j & 3) != 3 && 3 * (j >> 2) + (j & 3) < kOps) frag(3 * (j >> 2) + (j & 3), set); }
    };
    // lgkmcnt(0) [+ counted vmcnt + barrier]: all fragment reads of the half-step have returned, at most KEEP pieces still fly, publish
    auto sync = [&](auto keep_c) {
        constexpr int KEEP = decltype(keep_c)::value;
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0) (also tells the compiler's wait-count pass that no LDS read is pending across the back-edge)
        if constexpr (KEEP >= 0) {
#if defined(WG_F16_TRACE) && WG_F16_TRACE >= 2
            q_a = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
            wait_dma_keep<(KEEP >= 0 ? KEEP : 0)>();
#if defined(WG_F16_TRACE) && WG_F16_TRACE >= 2
            q_b = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
            if (!(WG_ABLATE & 1)) __builtin_amdgcn_s_barrier();
#if defined(WG_F16_TRACE) && WG_F16_TRACE >= 2
            q_c = (uint32_t)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xc07f); // the three timestamps have returned: no scalar load is pending during the rest of the half-step
            q_d = (uint32_t)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            {
                const uint32_t dv = q_b - q_a, db = q_c - q_b;
                s_vm += dv; s_bar += db; s_probe += 2u * (q_d - q_c); // two round trips of the probe itself
                m_vm = dv > m_vm ? dv : m_vm; m_bar = db > m_bar ? db : m_bar; ++n_adv;
            }
#endif
        }
    };

    // ---- NN (default): A in four half-stage slots. Half-step H = 2 st + HS on register set HS: 64 MFMAs, one per slot; the fragment ops
    // of half-step H+1 (A slot (H+1) & 3; B stage of H+1) fill set HS^1. DMA: the 4 pieces of A(H+4) -> slot H & 3 at slots DO + DS p
    // (p < 4), then 4 pieces of B (even H: second half of B(st+2), odd H: first half of B(st+3)); M0 once per group of four, one slot
    // ahead. Counted wait: a half-step issues its A pieces before its B pieces, and only the A pieces issued two half-steps ago must
    // have landed: keep WG_NN_KEEP = 16 (20 is valid too: the 4 B pieces behind them are needed one half-step later; no gain measured).
    //   rD  += 16 KiB mod 64 KiB at slot DO+3 (M0 written at DO-1)     oR  moves on at slot 26 of even H (last B read: slot 25)
    //   vb  = base[parity] + oR at slot 27                             ga  += 32 k rows after the last A piece
    //   oD  moves on after the B M0 of even H                           rR, va  at slots 47 / 54 (last A read: slot 41)
    //   gb  += 64 k after the even half-step's last B piece             st  at slot 56 of odd H
    // sync_c: >= 0: lgkmcnt(0) + counted wait keeping that many pieces + barrier after slot WG_NN_SYNC_SLOT; -2: nothing
    auto half_step_nn = [&](auto hs_c, auto a_dma, auto b_dma, auto has_next, auto sync_c) {
        constexpr int HS = decltype(hs_c)::value, SYNC = decltype(sync_c)::value;
        constexpr bool ADMA = decltype(a_dma)::value, BDMA = decltype(b_dma)::value;
        constexpr int DS = WG_NN_DSTRIDE, DO = WG_NN_DOFF;
        static_assert(DO >= 1 && DS >= 2 && DO + 7 * DS < WG_NN_SYNC_SLOT && 52 < WG_NN_SYNC_SLOT && WG_NN_SYNC_SLOT < 63, "every piece and fragment op is issued before the counted wait");
        static_for<64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 3, u = j & 7;
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a_r[HS][t]), b_f[HS][u], acc[t][u], 0, 0, 0);
            if constexpr (decltype(has_next)::value) frag_slot(jc, HS ^ 1);
            if constexpr (ADMA && j == DO - 1) m16_set_m0(lds_a_wave + rD);
            if constexpr (BDMA && j == DO + 4 * DS - 4) { lb = lds_b_wave + oD + (HS == 0 ? 4096u : 0u); asm volatile("" : "+s"(lb)); }
            if constexpr (BDMA && j == DO + 4 * DS - 1) m16_set_m0(lb);
            if constexpr (j >= DO && ((j - DO) % DS) == 0 && (j - DO) / DS < 8) {
                constexpr int pi = (j - DO) / DS, q = pi & 3;
                if constexpr (pi < 4) { if constexpr (ADMA) m16_dma_imm<1024 * q>(a_voff[q], ga); }
                else { if constexpr (BDMA) m16_dma_imm<1024 * q>(b_voff[(HS == 0 ? 4 : 0) + q], gb); }
            }
            if constexpr (j == DO + 3) { rD = (rD + 0x4000u) & 0xffffu; asm volatile("" : "+s"(rD)); }
            if constexpr (j == 26 && HS == 0) { oR = ring3(oR); asm volatile("" : "+s"(oR)); }
            if constexpr (j == 27) { vb = vbaseB[HS] + oR; asm volatile("" : "+v"(vb)); } // the next half-step reads parity HS
            if constexpr (j == DO + 3 * DS + 3) {
                if constexpr (WG_F16_UNPEELED) ga += (HS == 0 ? a_inc0 : a_inc1); else ga += a_step;
                asm volatile("" : "+s"(ga));
            }
            if constexpr (j == DO + 4 * DS + 3 && HS == 0) { oD = ring3(oD); asm volatile("" : "+s"(oD)); }
            if constexpr (j == 47) { rR = (rR + 0x4000u) & 0xffffu; asm volatile("" : "+s"(rR)); }
            if constexpr (WG_NN_NOSWAP && j == 53) { va1 = vbaseA1 + rR; asm volatile("" : "+v"(va1)); }
            if constexpr (j == 54) { va = vbaseA[0] + rR; asm volatile("" : "+v"(va)); }
            if constexpr (j == 55 && HS == 1) { ++st; asm volatile("" : "+s"(st)); }
            if constexpr (WG_F16_UNPEELED && HS == 1 && j == 56) { a_inc0 = st + 3u <= S ? a_full : 0u; asm volatile("" : "+s"(a_inc0)); } // A(2 st + 5) exists
            if constexpr (WG_F16_UNPEELED && HS == 1 && j == 57) { a_inc1 = st + 4u <= S ? a_full : 0u; asm volatile("" : "+s"(a_inc1)); } // A(2 st + 6) exists
            if constexpr (WG_F16_UNPEELED && HS == 1 && j == 58) { b_inc = st + 4u <= S ? 128u : 0u; asm volatile("" : "+s"(b_inc)); }     // B(st + 3) exists
            if constexpr (j == WG_NN_SYNC_SLOT && SYNC >= -1) sync(sync_c);
            if constexpr (j == WG_NN_SYNC_SLOT + 1 && HS == 0) {
                if constexpr (WG_F16_UNPEELED) gb += b_inc; else gb += 128;
                asm volatile("" : "+s"(gb));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- The staged pipeline (TN; NN with -DWG_NN_STAGED=1: measured slower, 32768^3 -4 %, see the evidence file): A lives in two
    // full-stage slots like B's three, so ONE counted wait + barrier per stage suffices, after slot WG_TN_SYNC_SLOT of the EVEN half-step:
    // it publishes stage st+1 (read by the odd half-step that follows and the even one after it) and releases the slots of stage st
    // (refilled from the odd half-step on). There, only the 4 + 4 newest pieces (the two halves of a B stage) may still be in flight --
    // the full stage of A issued in the odd half-step before must have landed (it goes ahead of B in the in-order vmcnt queue). Even
    // half-steps issue 4 pieces (second half of B(st+2)), odd ones 12 (A(st+2), then the first half of B(st+3)), one every 4 slots, M0
    // one slot ahead of each group of four.
    //   oA toggles after the even half-step's last A read, oAD after the odd half-step's second A M0; oR / oD / gb as above;
    //   ga [ga2: NN, the stage's second half-stage] += 64 k after the odd half-step's last A piece.
    // sync_c: >= 0: lgkmcnt(0) + counted wait + barrier; -1: lgkmcnt(0) only; -2: nothing (last half-step of the tile)
    auto half_step_s = [&](auto hs_c, auto a_dma, auto b_dma, auto has_next, auto sync_c) {
        constexpr int HS = decltype(hs_c)::value, SYNC = decltype(sync_c)::value;
        constexpr bool ADMA = decltype(a_dma)::value, BDMA = decltype(b_dma)::value;
        constexpr int nA = ADMA ? 8 : 0, nB = BDMA ? 4 : 0;
        constexpr int DS = 4, DO = TRANS_A ? 2 : 3;       // NN: pieces in the slots without a fragment op
        constexpr int LASTB = TRANS_A ? 24 : 25;          // slot of the last B fragment read (of the last A read: 45 / 41)
        constexpr uint32_t A_GROUP = TRANS_A ? 4096u : (uint32_t)HA_BYTES; // LDS distance of the second group of four A pieces
        static_assert(DO + DS * 11 < WG_TN_SYNC_SLOT && 53 < WG_TN_SYNC_SLOT && WG_TN_SYNC_SLOT < 64, "every piece and fragment op is issued before the wait");
        static_for<64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 3, u = j & 7;
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a_r[HS][t]), b_f[HS][u], acc[t][u], 0, 0, 0);
            if constexpr (decltype(has_next)::value) frag_slot(jc, HS ^ 1);
            // M0 values: computed one slot before the write that uses them
            if constexpr (ADMA && j == DO - 2) { la = lds_a_wave + oAD; asm volatile("" : "+s"(la)); }
            if constexpr (ADMA && j == DO + 4 * DS - 2) { la += A_GROUP; asm volatile("" : "+s"(la)); }
            if constexpr (BDMA && j == DO + DS * nA - 2) { lb = lds_b_wave + oD + (HS == 0 ? 4096u : 0u); asm volatile("" : "+s"(lb)); }
            if constexpr (j + 1 >= DO && ((j + 1 - DO) % DS) == 0 && ((j + 1 - DO) / DS) < nA + nB && (((j + 1 - DO) / DS) & 3) == 0) {
                constexpr int n = (j + 1 - DO) / DS;
                if constexpr (n < nA) m16_set_m0(la); else m16_set_m0(lb);
            }
            if constexpr (j >= DO && ((j - DO) % DS) == 0 && ((j - DO) / DS) < nA + nB) {
                constexpr int n = (j - DO) / DS;
                if constexpr (n < nA) {
                    if constexpr (TRANS_A) m16_dma_imm<1024 * (n & 3)>(a_voff[n], ga);
                    else m16_dma_imm<1024 * (n & 3)>(a_voff[n & 3], n < 4 ? ga : ga2);
                } else m16_dma_imm<1024 * ((n - nA) & 3)>(b_voff[(HS == 0 ? 4 : 0) + (n - nA)], gb);
            }
            if constexpr (HS == 0 && j == DO + (TRANS_A ? 2 : 3)) { oD = ring3(oD); asm volatile("" : "+s"(oD)); }            // even: B M0 written at slot DO - 1
            if constexpr (HS == 0 && j == DO + DS * 3 + (TRANS_A ? 1 : 4)) {                                                // even: last B piece at DO + 12
                if constexpr (WG_F16_UNPEELED) gb += b_inc; else gb += 128;
                asm volatile("" : "+s"(gb));
            }
            if constexpr (HS == 1 && j == DO + 4 * DS + 1) { oAD ^= (uint32_t)M16_BS_BYTES; asm volatile("" : "+s"(oAD)); }  // odd: second A M0 at DO + 15
            if constexpr (HS == 1 && j == (TRANS_A ? DO + DS * 7 + 1 : 38)) {                                                // odd: last A piece at DO + 28
                if constexpr (WG_F16_UNPEELED) ga += a_inc1; else if constexpr (TRANS_A) ga += 128; else ga += 2u * a_step;
                asm volatile("" : "+s"(ga));
            }
            if constexpr (!TRANS_A && HS == 1 && j == 41) {
                if constexpr (WG_F16_UNPEELED) ga2 += a_inc1; else ga2 += 2u * a_step;
                asm volatile("" : "+s"(ga2));
            }
            if constexpr (HS == 0 && j == LASTB + 1) { oR = ring3(oR); asm volatile("" : "+s"(oR)); }
            if constexpr (j == LASTB + 4) { vb = vbaseB[HS] + oR; asm volatile("" : "+v"(vb)); }                             // the next half-step reads parity HS
            if constexpr (HS == 0 && j == 47) { oA ^= (uint32_t)M16_BS_BYTES; asm volatile("" : "+s"(oA)); }
            if constexpr (!TRANS_A && WG_NN_NOSWAP && j == 48) { va1 = vbaseA1 + (HS ? (uint32_t)HA_BYTES : 0u) + oA; asm volatile("" : "+v"(va1)); }
            if constexpr (j == 49) { va = vbaseA[HS] + oA; asm volatile("" : "+v"(va)); }
            if constexpr (HS == 1 && j == 52) { ++st; asm volatile("" : "+s"(st)); }
            if constexpr (WG_F16_UNPEELED && HS == 1 && j == 53) { a_inc1 = st + 4u <= S ? a_full : 0u; asm volatile("" : "+s"(a_inc1)); } // A(st + 3) exists
            if constexpr (WG_F16_UNPEELED && HS == 1 && j == 54) { b_inc = st + 4u <= S ? 128u : 0u; asm volatile("" : "+s"(b_inc)); }    // B(st + 3) exists
            if constexpr (j == WG_TN_SYNC_SLOT && SYNC >= -1) sync(sync_c);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // prologue (once per tile): A(0..3) [TN: stages 0, 1 of A], B(0), B(1) and the first half of B(2). Only what the first two
    // half-steps read -- A(0), B(0), A(1), issued first -- must have landed before the first barrier; the rest is issued in the order
    // the steady state's counted waits expect it to retire: A(2), B(1) (needed after half-step 0), A(3), B(2) first half (after half-step 1).
    auto pro_a = [&](int hh) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 0) { m16_set_m0(lds_a_wave + hh * HA_BYTES); asm volatile("s_nop 0"); }
            m16_dma(q, a_voff[q], a_src(hh));
        }
    };
    auto pro_b = [&](int sb, int nq) { // sb >= 0: stage sb of B;  sb = -1 / -2 (TN only): stage 0 / 1 of A
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q >= nq) break;
            if (sb >= 0) {
                if ((q & 3) == 0) { m16_set_m0(lds_b_wave + sb * M16_BS_BYTES + (q >> 2) * 4096); asm volatile("s_nop 0"); }
                m16_dma(q & 3, b_voff[q], b_src(sb));
            } else if constexpr (TRANS_A) {
                const int sa_ = -1 - sb;
                if ((q & 3) == 0) { m16_set_m0(lds_a_wave + sa_ * M16_BS_BYTES + (q >> 2) * 4096); asm volatile("s_nop 0"); }
                m16_dma(q & 3, a_voff[q], a_src(sa_));
            }
        }
    };
    if (rem) {
        // K remainder = stage 0 of the loop. Its k-values, zero-padded to a whole stage, go through the LDS by ordinary loads + stores in
        // exactly the image the DMA pieces of stage 0 would have written (same per-lane source addresses, same lane-linear destinations);
        // the DMA stream starts with stage 1 = the range's first 64 k (the bases were moved back by one stage above). No MFMA outside the
        // one loop, no second accumulation chain: the remainder is simply multiplied first.
        auto put = [&](uint32_t lds_dst, const char *src, bool valid) {
            uintx4 v = { 0u, 0u, 0u, 0u };
            if (valid) v = *reinterpret_cast<const uintx4_u *>(src);
            *(WG_AS3 uintx4 *)(uintptr_t)(lds_dst + 16u * (uint32_t)lane) = v;
        };
        if constexpr (TRANS_A) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t row = 8u * (8u * wave + q) + (lane >> 3);
                const uint32_t chunk = (lane & 7u) ^ (((4u - ((row >> 3) & 3u)) & 3u) | (((row >> 1) & 1u) << 2));
                put(lds_a_wave + (uint32_t)(q >> 2) * 4096u + (uint32_t)(q & 3) * 1024u, rem_a + (a_voff[q] - (M16_BIAS - 1024u * (q & 3))), 8u * chunk < rem);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t k_local = 32u * h + 4u * ((4u * wave + q) >> 1) + ((lane >> 2) & 3u);
                    put(lds_a_wave + (uint32_t)h * HA_BYTES + (uint32_t)q * 1024u, rem_a + (uint64_t)(32u * h) * g.lda * 2u + (a_voff[q] - (M16_BIAS - 1024u * q)), k_local < rem);
                }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t row = 8u * (8u * wave + q) + (lane >> 3);
            const uint32_t chunk = (lane & 7u) ^ (((4u - ((row >> 2) & 3u)) & 3u) | (((row >> 1) & 1u) << 2));
            put(lds_b_wave + (uint32_t)(q >> 2) * 4096u + (uint32_t)(q & 3) * 1024u, rem_b + (b_voff[q] - (M16_BIAS - 1024u * (q & 3))), 8u * chunk < rem);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0); // the loads above are the compiler's to wait for: keep them in front of the DMA pieces it cannot see
        if (!(WG_ABLATE & 2)) {
            if constexpr (TRANS_A) { pro_b(-2, 8); pro_b(1, 8); pro_b(2, 4); }
            else { pro_a(2); pro_b(1, 8); pro_a(3); pro_b(2, 4); }
        }
    } else if (!(WG_ABLATE & 2)) {
        if constexpr (TRANS_A) { // full stages of A as well: A(0), B(0) | A(1), B(1), first half of B(2)
            pro_b(-1, 8); pro_b(0, 8);
            pro_b(-2, 8); pro_b(1, 8); pro_b(2, 4);
        } else {
            pro_a(0); pro_b(0, 8); pro_a(1);
            pro_a(2); pro_b(1, 8); pro_a(3); pro_b(2, 4);
        }
    }
    wait_dma_keep<20>();
    __syncthreads();
    va = vbaseA[0]; va1 = vbaseA1; vb = vbaseB[0]; // half-step 0's own fragments: first half of stage 0
#pragma unroll
    for (int op = 0; op < kOps; ++op) frag(op, 0);
    const uint64_t calib_t0 = __builtin_amdgcn_s_memrealtime(); // per-XCD rate measurement (returns before the wait below)
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): nothing pending on entry to the loop either
    if (!(WG_ABLATE & 1)) __builtin_amdgcn_s_barrier(); // every wave has read A(0): half-step 0 may overwrite its slot with A(4)
    // running values as half-step 0 expects them: it reads the second half of stage 0 and issues A(4) / stage 2 pieces
    vb = vbaseB[1];
    gb = (const char *)b_src(2u);
    if constexpr (TRANS_A) { va = vbaseA[1]; ga = (const char *)a_src(2u); }
    else if constexpr (STAGED) { va = vbaseA[1]; va1 = vbaseA1 + (uint32_t)HA_BYTES; ga = (const char *)a_src(4u); ga2 = (const char *)a_src(5u); }
    else { va = vbaseA[0] + rR; va1 = vbaseA1 + rR; ga = (const char *)a_src(4u); }
    __builtin_amdgcn_sched_barrier(0);
    WG_TRACE_POINT(1);
#if defined(WG_F16_TRACE) && WG_F16_TRACE >= 2
    loop_t0 = (uint32_t)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);
#endif

    // No parity branches (two accumulator-modifying arms merging in a loop make the register allocator shuffle all 256
    // accumulators): stages are processed as pairs of half-steps in straight-line code. The last three stages would issue fewer
    // (then no) DMA pieces and wait with smaller counts; -DWG_F16_UNPEELED=0 peels them into six half-steps (the form measured
    // against in profiles/r02_evidence.md 3e).
    const uint32_t s_end = S - 3u;
    if constexpr (WG_F16_UNPEELED) {
        // The steady-state body for every stage: the pieces of the last three stages re-read the last stage (3 x 64 KiB of L2 hits per
        // tile) into LDS slots nobody reads any more -- by the same invariants that free those slots in the steady state -- so the
        // counted waits mean the same all the way. No accumulator ever crosses from the loop into differently allocated straight-line
        // code: the peeled form cost 450-600 v_accvgpr moves per tile there (outside a loop every MFMA result is a fresh value,
        // nothing ties it to its accumulator's register, all 256 AGPRs are taken, and the allocator rotates accumulators through
        // VGPRs). The last half-step's fragment reads are dead too. Keep this a `while`: as a do-while the allocator splits the
        // accumulators inside the loop. The workgroup must not end with pieces in flight: see the vmcnt(0) at every exit.
        a_inc0 = S >= 3u ? a_full : 0u; a_inc1 = S >= 4u ? a_full : 0u; b_inc = S >= 4u ? 128u : 0u;
        if constexpr (STAGED) {
            while (st < S) {
                half_step_s(c0{}, no{}, yes{}, yes{}, k8{});
                half_step_s(c1{}, yes{}, yes{}, yes{}, km1{}); // ++st inside
            }
        } else {
            while (st < S) {
                half_step_nn(c0{}, yes{}, yes{}, yes{}, std::integral_constant<int, WG_NN_KEEP>{});
                half_step_nn(c1{}, yes{}, yes{}, yes{}, std::integral_constant<int, WG_NN_KEEP>{}); // ++st inside
            }
        }
    } else if constexpr (STAGED) {
        while (st < s_end) {
            half_step_s(c0{}, no{}, yes{}, yes{}, k8{});
            half_step_s(c1{}, yes{}, yes{}, yes{}, km1{}); // ++st inside
        }
        half_step_s(c0{}, no{}, yes{}, yes{}, k8{});   // stage S-3: second half of B(S-1)
        half_step_s(c1{}, yes{}, no{}, yes{}, km1{});  //            A(S-1)
        half_step_s(c0{}, no{}, no{}, yes{}, k0{});    // stage S-2: A(S-1), B(S-1) must have landed
        half_step_s(c1{}, no{}, no{}, yes{}, km1{});
        half_step_s(c0{}, no{}, no{}, yes{}, km1{});   // stage S-1
        half_step_s(c1{}, no{}, no{}, no{}, km2{});
    } else {
        while (st < s_end) {
            half_step_nn(c0{}, yes{}, yes{}, yes{}, std::integral_constant<int, WG_NN_KEEP>{});
            half_step_nn(c1{}, yes{}, yes{}, yes{}, std::integral_constant<int, WG_NN_KEEP>{}); // ++st inside
        }
        half_step_nn(c0{}, yes{}, yes{}, yes{}, k16{}); // stage S-3: A(2S-2), second half of B(S-1)
        half_step_nn(c1{}, yes{}, no{}, yes{}, k12{});  //            A(2S-1)
        half_step_nn(c0{}, no{}, no{}, yes{}, k4{});    // stage S-2
        half_step_nn(c1{}, no{}, no{}, yes{}, k0{});
        half_step_nn(c0{}, no{}, no{}, yes{}, k0{});    // stage S-1
        half_step_nn(c1{}, no{}, no{}, no{}, km2{});
    }
    WG_TRACE_POINT(2);
#if defined(WG_F16_TRACE) && WG_F16_TRACE >= 2
    loop_t1 = (uint32_t)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
    if (calib) { // this tile's main-loop time and stage count to the XCD it really ran on (fire-and-forget atomics, one lane)
        const uint64_t calib_t1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (threadIdx.x == 0) {
            const uint32_t x = blockIdx.x & 7u; // the workgroup SLOT (hardware deals ids round-robin to the XCDs): what the plan hands work to
            atomicAdd(calib + 2u * x, (unsigned long long)(calib_t1 - calib_t0));
            atomicAdd(calib + 2u * x + 1u, (unsigned long long)S);
        }
    }

    // ---- epilogue: lane holds, per (pair p, N tile u), rows 32 p + 8 kg + 0..7 of column 16 u + i16 (tile 2 p: + 0..3, tile 2 p + 1: + 4..7; the other way
    // round in odd lane rows of the swap-free Gemm) ----
    const bool full_tile = (m0 + BM <= g.M) && (n0 + BN <= g.N); // workgroup-uniform
    const uint32_t row0 = m0 + 128u * wm + 8u * kg;
    if (g.tail_tiles > 0 || unit_mode == 1) { // tail split: raw f32 partial TILE (256 x 256, dense) of this split; gemm_f16_tail_reduce finishes the job
        float *P = unit_mode == 1 ? part : part + ((uint64_t)split * g.tail_tiles + bid) * 65536u; // (prefix unit: its pair's scratch tile)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t cl = 128u * wn + 16u * u + i16; // column within the tile
            float *pc = P + (uint64_t)cl * 256u + 128u * wm + 8u * kg;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 *d = reinterpret_cast<float4 *>(pc + 32 * p);
                d[odd_row ? 1 : 0] = make_float4(acc[2 * p][u][0], acc[2 * p][u][1], acc[2 * p][u][2], acc[2 * p][u][3]);
                d[odd_row ? 0 : 1] = make_float4(acc[2 * p + 1][u][0], acc[2 * p + 1][u][1], acc[2 * p + 1][u][2], acc[2 * p + 1][u][3]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (unit_mode == 1) { // producer side of the hand-off (guide, G16): every wave's stores done, ONE agent-scope release, then the flag
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the compiler may drop the wait behind buffer_wbl2: restated where it cannot)
                __hip_atomic_store(bal_flag, bal_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (g.nsplit > 1) { // split-K: raw f32 partial sums to this split's slab (dense, ld = M)
        float *P = part + ((uint64_t)z * g.nsplit + split) * ((uint64_t)g.M * g.N);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t col = n0 + 128u * wn + 16u * u + i16;
            if (!full_tile && col >= g.N) continue;
            float *pc = P + (uint64_t)col * g.M + row0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (!(full_tile || row0 + 32 * p < g.M)) continue;
                float4 *d = reinterpret_cast<float4 *>(pc + 32 * p);
                d[odd_row ? 1 : 0] = make_float4(acc[2 * p][u][0], acc[2 * p][u][1], acc[2 * p][u][2], acc[2 * p][u][3]);
                d[odd_row ? 0 : 1] = make_float4(acc[2 * p + 1][u][0], acc[2 * p + 1][u][1], acc[2 * p + 1][u][2], acc[2 * p + 1][u][3]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
#ifndef WG_PANEL_STORE
#define WG_PANEL_STORE 2 // write-through flavour of a paneled launch's stores: 1 = sc1 (agent scope), 2 = sc0 sc1 (system scope)
#endif
#ifndef WG_EPI_STORE
#define WG_EPI_STORE -1 // -1: by GemmArgs::c_stream (shipped); experiments: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt for every launch
#endif
    // (u, p) in the lane's own layout: rows row0 + 32 p .. + 7 of column n0 + 128 wn + 16 u + i16, alpha / beta applied, rounded once
    auto pack = [&](int u, int p, bool ok, const _Float16 *cc) -> half8_t {
        float r[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { // (odd lane rows of the swap-free Gemm: the pair's tiles in exchanged order)
            r[q] = odd_row ? acc[2 * p + 1][u][q] : acc[2 * p][u][q];
            r[4 + q] = odd_row ? acc[2 * p][u][q] : acc[2 * p + 1][u][q];
        }
        if (alpha != 1.f) {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] *= alpha;
        }
        if (beta != 0.f && ok) { // beta == 0 never reads C
            const half8_t c = *reinterpret_cast<const half8_u *>(cc + 32 * p);
#pragma unroll
            for (int q = 0; q < 8; ++q) r[q] = fmaf(beta, (float)c[q], r[q]);
        }
        half8_t v;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (_Float16)r[q];
        return v;
    };
    // (every hand-written store carries its own `s_nop 1`: a store of more than 64 bits reads its data registers AFTER it issues, the compiler's hazard
    // recognizer does not look inside an asm statement, and a read-out of the next pack into the same registers right behind the store changed the first
    // dword for lanes 12 .. 15 of every row -- seen in the continuous kernel, round 5)
    auto store8 = [&](_Float16 *dst, half8_t v) {
        if (paneled) { // write-through to memory: when the store is acknowledged a copy engine may read it
            if constexpr (WG_PANEL_STORE == 0) *reinterpret_cast<half8_u *>(dst) = v; // (timing experiments only: NOT visible to a copy engine in time)
            else if constexpr (WG_PANEL_STORE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        } else if (!(WG_ABLATE & 32)) {
            if constexpr (WG_EPI_STORE == -1) {
                if (c_stream) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                else *reinterpret_cast<half8_u *>(dst) = v;
            } else if constexpr (WG_EPI_STORE == 0) *reinterpret_cast<half8_u *>(dst) = v;
            else if constexpr (WG_EPI_STORE == 1) __builtin_nontemporal_store(v, reinterpret_cast<half8_u *>(dst));
            else if constexpr (WG_EPI_STORE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            else if constexpr (WG_EPI_STORE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
        }
    };
    // (A store instruction covers 16 columns x 64 bytes -- half lines -- and an idle CU gets a tile out four times faster when 16 consecutive lanes
    // cover 256 contiguous bytes (tools/cpp/store_probe.hip). Both ways to that layout were built and measured in the kernel -- a DPP exchange between
    // the halves of the 16-lane rows, and a round trip of the rounded tile through the idle LDS -- and neither paid: the stores of a tile boundary
    // are bound by what the chip takes from 256 CUs at once, not by their issue; 8192 x 8192 x 512 92.5 -> 97 us: profiles/r03_evidence.md section 9.)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t col = n0 + 128u * wn + 16u * u + i16;
        if (!full_tile && col >= g.N) continue;
        _Float16 *cc = C + (uint64_t)col * ldc + row0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!(full_tile || row0 + 32 * p < g.M)) continue; // 8 consecutive rows, all in or all out (M % 8 == 0)
            store8(cc + 32 * p, pack(u, p, true, cc));
        }
    }
#ifndef WG_F16_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing of this tile (parked DMA pieces, stores) is in flight when the workgroup ends
#endif
    if (paneled && lane == 0) // this wave's stores are in memory (vmcnt(0) above): one arrival per WAVE on the panel's counter, fire and forget --
        // no barrier, no returned value; the panel's exchange waits for 4 x its tile count (hipStreamWaitValue32 on the counter itself)
        __hip_atomic_fetch_add(panel_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef WG_F16_TRACE
    WG_TRACE_POINT(3);                       // all stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_TRACE_POINT(4);                       // all stores acknowledged
    if (threadIdx.x == 0 && g.nsplit == 1 && g.part) {
        uint64_t *o = (uint64_t *)g.part + (uint64_t)bid * 8u;
        for (int i = 0; i < 5; ++i) o[i] = tr_t[i];
        o[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID (4): wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13]
        o[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // XCC_ID (gfx940+)
    }
#if WG_F16_TRACE >= 2
    if (lane == 0 && g.nsplit == 1 && g.part) {
        uint32_t *o = (uint32_t *)((uint64_t *)g.part + (uint64_t)g.trace_tiles * 8u) + ((uint64_t)bid * 4u + wave) * 8u;
        o[0] = loop_t1 - loop_t0; o[1] = s_vm; o[2] = s_bar; o[3] = s_probe; o[4] = m_vm; o[5] = m_bar; o[6] = n_adv; o[7] = 0;
    }
#endif
#endif
}

// Dynamic tile scheduler. Hardware deals workgroup ids round-robin to the 8 XCDs, i.e. every XCD gets the same NUMBER of tiles -- and
// the XCDs of one chip do not run at the same speed under the power cap (measured per-tile times differ by up to 10 % between XCDs,
// which XCD is slow differs from box to box): at 32768^3 the fast XCDs idle for the last 3.5 % of the kernel. So the launch carries
// 12.5 % more workgroups than tiles and a workgroup TAKES its tile: from the bottom of its own XCD's queue (the same tile, in the same
// order, the static id -> tile map would have given that XCD: XCD x's t-th tile is id 8 t + x), and once that queue is empty from the TOP
// of the queue with the most tiles left (the tiles the victim would have reached last; consecutive steals are neighbours, so the
// thieves share panels through their own L2). One 64-bit word per XCD holds both ends -- low half: taken from the bottom, high half:
// taken from the top -- so one atomic add claims a tile exactly when low + high < the queue's length. Workgroups that find every queue
// empty exit. Returns the id to run as, or ~0u.
static __device__ uint32_t m16_acquire_tile(const GemmArgs &g) {
    const uint32_t x = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
    const uint32_t q = g.sched_tiles / 8u, r = g.sched_tiles % 8u;
    auto len = [&](uint32_t v) { return q + (v < r ? 1u : 0u); };
    unsigned long long old = atomicAdd(&g.sched[16u * x], 1ull);
    uint32_t lo = (uint32_t)old, hi = (uint32_t)(old >> 32);
    if (lo + hi < len(x)) return 8u * lo + x;
    for (int attempt = 0; attempt < 64; ++attempt) {
        uint32_t best = 0, victim = 8;
        for (uint32_t v = 0; v < 8u; ++v) {
            if (v == x) continue;
            const unsigned long long w = __hip_atomic_load(&g.sched[16u * v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t taken = (uint32_t)w + (uint32_t)(w >> 32), n = len(v);
            if (taken < n && n - taken > best) { best = n - taken; victim = v; }
        }
        if (victim == 8u) return ~0u;
        old = atomicAdd(&g.sched[16u * victim], 1ull << 32);
        lo = (uint32_t)old; hi = (uint32_t)(old >> 32);
        if (lo + hi < len(victim)) return 8u * (len(victim) - 1u - hi) + victim;
    }
    return ~0u;
}

template <bool TRANS_A>
__global__ __launch_bounds__(256, 1) void gemm_f16_m16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[160 * 1024];
    uint32_t bid = blockIdx.x;
    uint32_t mode = 0, kb = 0, ns = 0, pair = 0; // balance unit (BalancePlan); 0: a whole tile
    if (g.sched) { // workgroup-uniform
        if (threadIdx.x == 0) *reinterpret_cast<volatile uint32_t *>(smem) = m16_acquire_tile(g);
        __syncthreads();
        bid = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t *>(smem));
        __syncthreads(); // everyone has read the word before the tile's first DMA piece may land on it
        if (bid == ~0u) return;
    } else if (g.bal.on) { // calibrated shares: workgroup b is unit b / 8 of XCD b % 8's list (the hardware's deal; any unit may run anywhere)
        if (!bal_decode(g.bal, blockIdx.x, g.K / 64u, bid, mode, kb, ns, pair)) return;
    }
    m16_tile<TRANS_A>(g, bid, smem, mode, kb, ns, pair); // (ONE call site: the tile body must exist once in the kernel)
}


// The continuous form's 64 accumulator quads live in a[0:255] BY NAME: left to the compiler, accumulators that are live across a tile's epilogue inside
// an outer loop get copied to VGPRs wholesale at the inner loop's exit (on top of the fragments that must stay live there), spilled to scratch and shuffled
// inside the multiply loop (three cuts, ISA read each time: profiles/r05_evidence.md section 3). So the multiply (a tile's first one with C = 0) and the read-out are inline
// asm on fixed registers, each naming what it overwrites; the compiler keeps no value of its own in an AGPR in that kernel (checked in the ISA test).
#define WG_ACC_QUADS(X) \
    X(0, 0, 1, 2, 3) X(1, 4, 5, 6, 7) X(2, 8, 9, 10, 11) X(3, 12, 13, 14, 15) \
    X(4, 16, 17, 18, 19) X(5, 20, 21, 22, 23) X(6, 24, 25, 26, 27) X(7, 28, 29, 30, 31) \
    X(8, 32, 33, 34, 35) X(9, 36, 37, 38, 39) X(10, 40, 41, 42, 43) X(11, 44, 45, 46, 47) \
    X(12, 48, 49, 50, 51) X(13, 52, 53, 54, 55) X(14, 56, 57, 58, 59) X(15, 60, 61, 62, 63) \
    X(16, 64, 65, 66, 67) X(17, 68, 69, 70, 71) X(18, 72, 73, 74, 75) X(19, 76, 77, 78, 79) \
    X(20, 80, 81, 82, 83) X(21, 84, 85, 86, 87) X(22, 88, 89, 90, 91) X(23, 92, 93, 94, 95) \
    X(24, 96, 97, 98, 99) X(25, 100, 101, 102, 103) X(26, 104, 105, 106, 107) X(27, 108, 109, 110, 111) \
    X(28, 112, 113, 114, 115) X(29, 116, 117, 118, 119) X(30, 120, 121, 122, 123) X(31, 124, 125, 126, 127) \
    X(32, 128, 129, 130, 131) X(33, 132, 133, 134, 135) X(34, 136, 137, 138, 139) X(35, 140, 141, 142, 143) \
    X(36, 144, 145, 146, 147) X(37, 148, 149, 150, 151) X(38, 152, 153, 154, 155) X(39, 156, 157, 158, 159) \
    X(40, 160, 161, 162, 163) X(41, 164, 165, 166, 167) X(42, 168, 169, 170, 171) X(43, 172, 173, 174, 175) \
    X(44, 176, 177, 178, 179) X(45, 180, 181, 182, 183) X(46, 184, 185, 186, 187) X(47, 188, 189, 190, 191) \
    X(48, 192, 193, 194, 195) X(49, 196, 197, 198, 199) X(50, 200, 201, 202, 203) X(51, 204, 205, 206, 207) \
    X(52, 208, 209, 210, 211) X(53, 212, 213, 214, 215) X(54, 216, 217, 218, 219) X(55, 220, 221, 222, 223) \
    X(56, 224, 225, 226, 227) X(57, 228, 229, 230, 231) X(58, 232, 233, 234, 235) X(59, 236, 237, 238, 239) \
    X(60, 240, 241, 242, 243) X(61, 244, 245, 246, 247) X(62, 248, 249, 250, 251) X(63, 252, 253, 254, 255)
template <int I> struct AccQuad;
#define WG_ACC_QUAD_DEF(I, R0, R1, R2, R3)                                                                                                              \
    template <> struct AccQuad<I> {                                                                                                                     \
        static __device__ __forceinline__ void mfma(half8_t a, half8_t b) {                                                                             \
            asm volatile("v_mfma_f32_16x16x32_f16 a[" #R0 ":" #R3 "], %0, %1, a[" #R0 ":" #R3 "]" ::"v"(a), "v"(b) : "a" #R0, "a" #R1, "a" #R2, "a" #R3);  \
        }                                                                                                                                               \
        static __device__ __forceinline__ void mfma0(half8_t a, half8_t b) { /* the quad's first product of a tile: C = 0 instead of a zeroing pass */      \
            asm volatile("v_mfma_f32_16x16x32_f16 a[" #R0 ":" #R3 "], %0, %1, 0" ::"v"(a), "v"(b) : "a" #R0, "a" #R1, "a" #R2, "a" #R3);                     \
        }                                                                                                                                               \
        static __device__ __forceinline__ void read(float &x0, float &x1, float &x2, float &x3) {                                                       \
            asm volatile("v_accvgpr_read_b32 %0, a" #R0 "\n\tv_accvgpr_read_b32 %1, a" #R1 "\n\tv_accvgpr_read_b32 %2, a" #R2 "\n\tv_accvgpr_read_b32 %3, a" #R3 \
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));                                                                                      \
        }                                                                                                                                               \
    };
WG_ACC_QUADS(WG_ACC_QUAD_DEF)
#undef WG_ACC_QUAD_DEF

// ===============================================================================================================
// Gemm / GemmTr, continuous form (round 5): ONE workgroup per CU walks its tiles (id, id + grid, ...: the tiles the hardware's round-robin deal
// would have given that CU) and the LDS-DMA stream never stops -- where m16_tile parks its cursors for a tile's last three stages, these jump to
// the next tile, so when a tile's loop ends the next tile's first stages are in the LDS, its first fragments in registers, and the loop simply
// goes on: no prologue (2.8 us), no re-dispatch (0.7 us), and the stores of the finished tile drain under the next tile's first stage (stores and
// DMA pieces retire in order in vmcnt: the counted waits of the first stage after an epilogue allow the 32 stores on top of their pieces; the next
// one is the first that needs anything issued behind them). What it is for: short K on many tiles, where those 5 us are a fifth of a tile
// (8192^2 x 1024: profiles/r05_evidence.md section 3). Same tiles, same k order, same accumulation chains as m16_tile: bit-identical results.
// The half-steps are m16_tile's (half_step_nn / half_step_s, slot for slot: see there for the plan); what differs is marked.
// Restrictions (launcher): K % 64 == 0, K >= 256, beta == 0, no split, no panels (and the fast path's M % 8 == 0). Batches: the walk goes through the matrices' tiles in
// turn. Ragged tiles (M or N not a multiple of 256): rows past the end are clamped in the DMA offsets and skipped by the epilogue, as in m16_tile.
// ===============================================================================================================
// ALPHA1: alpha == 1 (the launcher's choice): the epilogue skips its 256 multiplies by alpha per tile (x * 1.0f is x: the same bits, a quarter fewer instructions in the
// one stretch of a tile where no MFMA runs).
template <bool TRANS_A, bool STREAM, bool ALPHA1>
__device__ __forceinline__ void m16_cont(const GemmArgs &g, char *const smem, const uint32_t walk_first, const int32_t walk_stride, const uint32_t walk_count) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    const int aq = i16 >> 2, bb = i16 & 3;
    const int gq = (4 - aq) & 3; // G(aq)
    const uint32_t S = g.K / 64u;                                   // stages per tile (>= 4)
    uint32_t tile = walk_first;                                      // this workgroup's current tile; its walk: walk_count tiles, walk_stride ids apart
    uint32_t tiles_left = walk_count;                                // ... counting the current one
    uint32_t rem_g = walk_count * S;                                 // stages this workgroup still has to multiply (all of its tiles)
    float alpha = g.alpha;
    uint32_t ldc = g.ldc;
    asm volatile("" : "+s"(alpha), "+s"(ldc));

    // ---- per-lane DMA offsets. a_int / b_int: a whole tile's; a_voff / b_voff: the current ones -- a ragged tile (last tile row / column of an M or N that is not a
    // multiple of 256) clamps the rows past the end to the last valid one, as m16_tile does (their results are never stored): set_voffs_*, per tile, at the slot where
    // the operand's cursor leaves the previous tile. ----
    constexpr int NA = TRANS_A ? 8 : 4;
    uint32_t a_int[NA], b_int[8], a_voff[NA], b_voff[8];
    const uint32_t row_l = 64u * wave + (lane >> 3);                  // piece q of a k-contiguous operand: row row_l + 8 q of the tile
    // Gemm's A: piece q starts at row 128 (q & 1) + mp_l (swap-free form: the pieces of odd k-groups -- a piece's k-group within its half-stage is `wave` -- swap neighbouring units)
    const uint32_t mp_l = 32u * (lane >> 4) + 8u * (WG_NN_NOSWAP && !TRANS_A ? (lane & 3u) ^ ((uint32_t)wave & 1u) : (lane & 3u));
    if constexpr (TRANS_A) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t row = row_l + 8u * q;
            const uint32_t fa = ((4u - ((row >> 3) & 3u)) & 3u) | (((row >> 1) & 1u) << 2); // A's rows are permuted: G index (row >> 3) & 3
            a_int[q] = (row * g.lda + 8u * ((lane & 7u) ^ fa)) * 2u + (M16_BIAS - 1024u * (q & 3));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { // blocks 4P..4P+3: kq = P>>1, mblk = 4 (P&1) + (lane>>4), k row (lane>>2)&3, 16-byte unit lane&3
            const uint32_t P = 4u * wave + q;
            a_int[q] = ((4u * (P >> 1) + ((lane >> 2) & 3)) * g.lda + 128u * (P & 1) + mp_l) * 2u + (M16_BIAS - 1024u * q);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const uint32_t row = row_l + 8u * q;
        const uint32_t fb = ((4u - ((row >> 2) & 3u)) & 3u) | (((row >> 1) & 1u) << 2);
        b_int[q] = (row * g.ldb + 8u * ((lane & 7u) ^ fb)) * 2u + (M16_BIAS - 1024u * (q & 3));
    }
    // lim: the last row the tile may read (GemmTr's A, B: M - 1 - m0, N - 1 - n0; Gemm's A, in pieces of 8 rows: M - 8 - m0); >= 255 / 248 in a whole tile: nothing moves
    auto set_voffs_a = [&](uint32_t lim) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const uint32_t r = TRANS_A ? row_l + 8u * q : 128u * (q & 1) + mp_l;
            const uint32_t over = r > lim ? r - lim : 0u;
            a_voff[q] = a_int[q] - over * (TRANS_A ? g.lda * 2u : 2u);
        }
    };
    auto set_voffs_b = [&](uint32_t lim) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t r = row_l + 8u * q;
            const uint32_t over = r > lim ? r - lim : 0u;
            b_voff[q] = b_int[q] - over * (g.ldb * 2u);
        }
    };
    const uint32_t lds_base = (uint32_t)(uintptr_t)(WG_AS3 char *)smem;
    const uint32_t lds_a_wave = __builtin_amdgcn_readfirstlane(lds_base + wave * (TRANS_A ? 8192 : 4096));
    const uint32_t lds_b_wave = __builtin_amdgcn_readfirstlane(lds_base + M16_B_BASE + wave * 8192);
    uint32_t vbaseA[2], vbaseB[2];
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        const uint32_t chunk = (uint32_t)(((kg ^ gq) | ((hs ^ (bb >> 1)) << 2)) * 16);
        vbaseB[hs] = lds_base + M16_B_BASE + ((uint32_t)128 * wn + i16) * 128u + chunk;
        if constexpr (TRANS_A) vbaseA[hs] = lds_base + (128u * wm + 8u * aq + bb) * 128u + chunk;
    }
    uint32_t vbaseA1 = 0; // swap-free Gemm: the odd tiles' base (m16_tile: "NN A")
    if constexpr (!TRANS_A) {
        if constexpr (WG_NN_NOSWAP) {
            const uint32_t a = (uint32_t)i16 & 3u, krow = (uint32_t)i16 >> 2;
            const uint32_t common = lds_base + (uint32_t)kg * 4096u + (4u * wm) * 256u + krow * 64u + (a ^ ((uint32_t)kg & 1u)) * 16u;
            vbaseA[0] = common + (a & 1u) * 8u;
            vbaseA1 = common + ((a & 1u) ^ 1u) * 8u;
        } else {
            vbaseA[0] = lds_base + (uint32_t)((kg & 2) * 2) * 2048u + (4u * wm) * 256u + (uint32_t)i16 * 16u + (uint32_t)(kg & 1) * 8u;
        }
        vbaseA[1] = vbaseA[0];
    }
    const bool odd_row = !TRANS_A && WG_NN_NOSWAP && (kg & 1); // these lanes hold the tiles of a pair in exchanged order
    // one stage of A in global memory: 128 bytes along its k-contiguous rows (GemmTr), 64 columns (Gemm; the half-stage a_half = 32 columns is its unit there)
    const uint64_t a_half = TRANS_A ? 64u : (uint64_t)BKH * g.lda * 2u;
    const uint64_t a_full = TRANS_A ? 128u : a_half;            // what a cursor step of A covers: a stage (GemmTr) / a half-stage (Gemm)
    // ---- tile -> global bases (scalar) ----
    // walk ids run through the matrices of a batch: id = matrix * tiles per matrix + tile (the per-tile launch's grid.y, flattened)
    const uint32_t tiles_per = g.tiles_m * g.tiles_n;
    auto sc64 = [](uint64_t v) -> uint64_t { // (64-bit products run on the vector unit: back to scalars by hand)
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    auto bases = [&](uint32_t id, const char *&a0, const char *&b0, uint32_t &m0, uint32_t &n0, uint64_t &c_off) {
        const uint32_t z = __builtin_amdgcn_readfirstlane(id / tiles_per), t = __builtin_amdgcn_readfirstlane(id - z * tiles_per);
        uint32_t tm, tn;
        tile_of(t, g.tiles_m, g.tiles_n, tm, tn);
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        m0 = tm * BM; n0 = tn * BN;
        const uint64_t ao = (TRANS_A ? (uint64_t)m0 * g.lda : (uint64_t)m0) + (uint64_t)z * g.a_batch, bo = (uint64_t)n0 * g.ldb + (uint64_t)z * g.b_batch; // elements
        a0 = (const char *)g.a + sc64(ao * 2u) - M16_BIAS;
        b0 = (const char *)g.b + sc64(bo * 2u) - M16_BIAS;
        c_off = sc64((uint64_t)z * g.c_batch);
    };
    const char *a0, *b0; // stage 0 of the current tile's operands (less M16_BIAS)
    uint32_t m0, n0;
    uint64_t c_off;      // its matrix's offset in the output (elements)
    bases(tile, a0, b0, m0, n0, c_off);
    auto lim_a = [&](uint32_t m0_) { return TRANS_A ? g.M - 1u - m0_ : g.M - 8u - m0_; };
    set_voffs_a(lim_a(m0)); set_voffs_b(g.N - 1u - n0);
    uint32_t nlim_a = 0, nlim_b = 0; // the next tile's (set when a tile starts, used where the cursors leave it)

    uintx4 a_r[2][8]; // (the accumulators: AccQuad<8 t + u>, a[0:255])
    half8_t b_f[2][8];
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    uint32_t st = 0;
    uint32_t va, vb, va1 = 0;
    const char *ga, *gb;
    uint32_t oR = 0, oD = 2u * M16_BS_BYTES, oA = 0, oAD = 0, la = 0, lb = 0; // B ring (3 stages); GemmTr: A's two full-stage slots
    uint32_t rR = 1u << 14, rD = 0;                                           // Gemm: A's four half-stage slots
    // Cursor steps after a stage's pieces. m16_tile: 0 once the next piece would lie past the tile's last stage; here: 0 once this WORKGROUP has nothing left to
    // fetch (rem_g), and at the step that leaves a tile -- behind the pieces of its stage S - 3 -- the distance to the next tile's first stage instead.
    uint32_t a_step0 = 0;                   // Gemm: after the even half-step's A pieces (on to half-stage 2 st + 5)
    uint64_t a_step1 = 0, b_step = 0;       // after the odd half-step's A pieces (on to A of stage st + 3) / the even half-step's B pieces (B(st + 3))
    uint32_t a_plain = 0, b_plain = 0;      // ... inside a tile, for the stage that starts next (a stage of A is less than 4 GiB: launcher)
    uint64_t a_cross = 0, b_cross = 0;      // ... out of the current tile: its own step + (first stage of the next tile - "stage S" of this one); set when a tile starts
    uint32_t cross_st = ~0u;                // the stage whose steps leave the tile: S - 3, if this workgroup has a next tile
    uint32_t after_store = 0;               // counted waits that still have to allow for the finished tile's 32 stores (in flight behind this tile's first pieces)
    auto ring3 = [](uint32_t o) -> uint32_t { o += (uint32_t)M16_BS_BYTES; return o == 3u * M16_BS_BYTES ? 0u : o; };
    constexpr int kOps = TRANS_A ? 16 : WG_NN_NOSWAP ? 24 : 40;
    auto frag = [&](int op, int set) {
        auto rb = [&](int u) { b_f[set][u] = lds_h8_at(vb + u * 2048); };
        if constexpr (TRANS_A) {
            auto ra = [&](int t) { a_r[set][t] = __builtin_bit_cast(uintx4, lds_h8_at(va + (t & 1) * 512 + (t >> 1) * 4096)); };
            if (op == 0) ra(0);
            else if (op <= 8) rb(op - 1);
            else ra(op - 8);
        } else if constexpr (WG_NN_NOSWAP) {
            auto tr = [&](int p, int i) { // m16_tile's: read i = 2 tb + h of pair p -> tile 2 p + tb, dwords 2 h, 2 h + 1
                const int tb = i >> 1, h = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr_at((tb ? va1 : va) + h * 2048 + p * 256));
                a_r[set][2 * p + tb][2 * h] = v[0];
                a_r[set][2 * p + tb][2 * h + 1] = v[1];
            };
            if (op < 4) tr(0, op);
            else if (op < 8) rb(op - 4);
            else if (op < 12) tr(1, op - 8);
            else if (op < 16) rb(op - 8);
            else if (op < 20) tr(2, op - 16);
            else tr(3, op - 20);
        } else {
            auto tr = [&](int p, int i) {
                const int h = i >> 1, ins = i & 1;
                const uintx2 v = __builtin_bit_cast(uintx2, lds_tr_at(va + (2 * ins + h) * 2048 + p * 256));
                a_r[set][2 * p + ins][2 * h] = v[0];
                a_r[set][2 * p + ins][2 * h + 1] = v[1];
            };
            auto sw = [&](int p, int i) {
                const uintx2 r = __builtin_amdgcn_permlane16_swap(a_r[set][2 * p][i], a_r[set][2 * p + 1][i], false, false);
                a_r[set][2 * p][i] = r[0];
                a_r[set][2 * p + 1][i] = r[1];
            };
            if (op < 4) tr(0, op);
            else if (op < 8) rb(op - 4);
            else if (op < 12) tr(1, op - 8);
            else if (op < 16) sw(0, op - 12);
            else if (op < 20) rb(op - 12);
            else if (op < 24) tr(2, op - 20);
            else if (op < 28) sw(1, op - 24);
            else if (op < 32) tr(3, op - 28);
            else if (op < 36) sw(2, op - 32);
            else sw(3, op - 36);
        }
    };
    auto frag_slot = [&](auto jc, int set) {
        constexpr int j = decltype(jc)::value;
        if constexpr (TRANS_A) { if constexpr ((j % 3) == 0 && (j / 3) < kOps) frag(j / 3, set); }
        else { if constexpr ((j & 3) != 3 && 3 * (j >> 2) + (j & 3) < kOps) frag(3 * (j >> 2) + (j & 3), set); }
    };
    // the end of a stage (odd half-step; four slots, at most three scalar instructions each): count, then the steps of the stage that starts next
    auto stage_end = [&](int k) {
        if (k == 0) { ++st; --rem_g; asm volatile("" : "+s"(st), "+s"(rem_g)); }
        if (k == 1) {
            const bool more = rem_g >= 4u;
            a_plain = more ? (uint32_t)a_full : 0u; b_plain = more ? 128u : 0u;
            asm volatile("" : "+s"(a_plain), "+s"(b_plain));
        }
        if (k == 2) { a_step1 = st == cross_st ? a_cross : (uint64_t)a_plain; asm volatile("" : "+s"(a_step1)); }
        if (k == 3) { b_step = st == cross_st ? b_cross : (uint64_t)b_plain; asm volatile("" : "+s"(b_step)); }
    };
    // lgkmcnt(0), the counted wait (KEEP pieces may fly; behind an epilogue: and the 32 stores issued in front of them), barrier
    auto sync = [&](auto keep_c) {
        constexpr int KEEP = decltype(keep_c)::value;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (after_store) { wait_dma_keep<KEEP + 32>(); --after_store; }
        else wait_dma_keep<KEEP>();
        __builtin_amdgcn_s_barrier();
    };
    // m16_tile's half_step_s (GemmTr): A in two full-stage slots, ONE counted wait + barrier per stage, in the even half-step
    auto half_step_tn = [&](auto hs_c, auto first_c) {
        constexpr int HS = decltype(hs_c)::value;
        constexpr bool FIRST = decltype(first_c)::value; // the tile's first half-step: every quad's first product, C = 0
        constexpr bool ADMA = HS == 1;
        constexpr int nA = ADMA ? 8 : 0, nB = 4;
        constexpr int DS = 4, DO = 2, LASTB = 24;
        static_for<64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 3, u = j & 7;
            if constexpr (FIRST) AccQuad<8 * t + u>::mfma0(__builtin_bit_cast(half8_t, a_r[HS][t]), b_f[HS][u]);
            else AccQuad<8 * t + u>::mfma(__builtin_bit_cast(half8_t, a_r[HS][t]), b_f[HS][u]);
            frag_slot(jc, HS ^ 1);
            if constexpr (ADMA && j == DO - 2) { la = lds_a_wave + oAD; asm volatile("" : "+s"(la)); }
            if constexpr (ADMA && j == DO + 4 * DS - 2) { la += 4096u; asm volatile("" : "+s"(la)); }
            if constexpr (j == DO + DS * nA - 2) { lb = lds_b_wave + oD + (HS == 0 ? 4096u : 0u); asm volatile("" : "+s"(lb)); }
            if constexpr (j + 1 >= DO && ((j + 1 - DO) % DS) == 0 && ((j + 1 - DO) / DS) < nA + nB && (((j + 1 - DO) / DS) & 3) == 0) {
                constexpr int n = (j + 1 - DO) / DS;
                if constexpr (n < nA) m16_set_m0(la); else m16_set_m0(lb);
            }
            if constexpr (j >= DO && ((j - DO) % DS) == 0 && ((j - DO) / DS) < nA + nB) {
                constexpr int n = (j - DO) / DS;
                if constexpr (n < nA) m16_dma_imm<1024 * (n & 3)>(a_voff[n], ga);
                else m16_dma_imm<1024 * ((n - nA) & 3)>(b_voff[(HS == 0 ? 4 : 0) + (n - nA)], gb);
            }
            if constexpr (HS == 0 && j == DO + 2) { oD = ring3(oD); asm volatile("" : "+s"(oD)); }
            if constexpr (HS == 0 && j == DO + DS * 3 + 1) { gb += b_step; asm volatile("" : "+s"(gb)); }  // even: after the last B piece -- on to B(st + 3), in the next tile if this one ends there
            if constexpr (HS == 0 && j == DO + DS * 3 + 2) { if (st == cross_st) set_voffs_b(nlim_b); }     // (a slot of its own: the compare and branch are two fillers)
            if constexpr (HS == 1 && j == DO + 4 * DS + 1) { oAD ^= (uint32_t)M16_BS_BYTES; asm volatile("" : "+s"(oAD)); }
            if constexpr (HS == 1 && j == DO + DS * 7 + 1) { ga += a_step1; asm volatile("" : "+s"(ga)); } // odd: after the last A piece -- on to A(st + 3)
            if constexpr (HS == 1 && j == DO + DS * 7 + 2) { if (st == cross_st) set_voffs_a(nlim_a); }
            if constexpr (HS == 0 && j == LASTB + 1) { oR = ring3(oR); asm volatile("" : "+s"(oR)); }
            if constexpr (j == LASTB + 4) { vb = vbaseB[HS] + oR; asm volatile("" : "+v"(vb)); }
            if constexpr (HS == 0 && j == 47) { oA ^= (uint32_t)M16_BS_BYTES; asm volatile("" : "+s"(oA)); }
            if constexpr (j == 49) { va = vbaseA[HS] + oA; asm volatile("" : "+v"(va)); }
            if constexpr (HS == 1 && j >= 52 && j <= 55) stage_end(j - 52);
            if constexpr (j == WG_TN_SYNC_SLOT) {
                if constexpr (HS == 0) sync(std::integral_constant<int, 8>{});
                else __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // m16_tile's half_step_nn (Gemm): A in four half-stage slots, a counted wait + barrier in every half-step
    auto half_step_nn = [&](auto hs_c, auto first_c) {
        constexpr int HS = decltype(hs_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int DS = WG_NN_DSTRIDE, DO = WG_NN_DOFF;
        static_for<64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 3, u = j & 7;
            if constexpr (FIRST) AccQuad<8 * t + u>::mfma0(__builtin_bit_cast(half8_t, a_r[HS][t]), b_f[HS][u]);
            else AccQuad<8 * t + u>::mfma(__builtin_bit_cast(half8_t, a_r[HS][t]), b_f[HS][u]);
            frag_slot(jc, HS ^ 1);
            if constexpr (j == DO - 1) m16_set_m0(lds_a_wave + rD);
            if constexpr (j == DO + 4 * DS - 4) { lb = lds_b_wave + oD + (HS == 0 ? 4096u : 0u); asm volatile("" : "+s"(lb)); }
            if constexpr (j == DO + 4 * DS - 1) m16_set_m0(lb);
            if constexpr (j >= DO && ((j - DO) % DS) == 0 && (j - DO) / DS < 8) {
                constexpr int pi = (j - DO) / DS, q = pi & 3;
                if constexpr (pi < 4) m16_dma_imm<1024 * q>(a_voff[q], ga);
                else m16_dma_imm<1024 * q>(b_voff[(HS == 0 ? 4 : 0) + q], gb);
            }
            if constexpr (j == DO + 3) { rD = (rD + 0x4000u) & 0xffffu; asm volatile("" : "+s"(rD)); }
            if constexpr (j == 26 && HS == 0) { oR = ring3(oR); asm volatile("" : "+s"(oR)); }
            if constexpr (j == 27) { vb = vbaseB[HS] + oR; asm volatile("" : "+v"(vb)); }
            if constexpr (j == DO + 3 * DS + 3) { // after the last A piece: on to half-stage 2 st + 5 (even) / the first one of stage st + 3 (odd: in the next tile if this one ends there)
                if constexpr (HS == 0) ga += a_step0; else ga += a_step1;
                asm volatile("" : "+s"(ga));
            }
            if constexpr (HS == 1 && j == DO + 3 * DS + 4) { if (st == cross_st) set_voffs_a(nlim_a); } // (a slot of its own, no fragment op, no piece)
            if constexpr (j == DO + 4 * DS + 3 && HS == 0) { oD = ring3(oD); asm volatile("" : "+s"(oD)); }
            if constexpr (j == 47) { rR = (rR + 0x4000u) & 0xffffu; asm volatile("" : "+s"(rR)); }
            if constexpr (WG_NN_NOSWAP && j == 53) { va1 = vbaseA1 + rR; asm volatile("" : "+v"(va1)); }
            if constexpr (j == 54) { va = vbaseA[0] + rR; asm volatile("" : "+v"(va)); }
            if constexpr (HS == 1 && j == 55) stage_end(0);
            if constexpr (HS == 1 && j == 56) { a_step0 = rem_g >= 3u ? (uint32_t)a_full : 0u; asm volatile("" : "+s"(a_step0)); }
            if constexpr (HS == 1 && j == 57) stage_end(1);
            if constexpr (HS == 1 && j == 58) stage_end(2);
            if constexpr (HS == 1 && j == WG_NN_SYNC_SLOT + 1) stage_end(3);
            if constexpr (j == WG_NN_SYNC_SLOT) sync(std::integral_constant<int, WG_NN_KEEP>{});
            if constexpr (j == WG_NN_SYNC_SLOT + 1 && HS == 0) { gb += b_step; asm volatile("" : "+s"(gb)); }
            if constexpr (j == WG_NN_SYNC_SLOT + 2 && HS == 0) { if (st == cross_st) set_voffs_b(nlim_b); }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // ---- prologue, first tile only (m16_tile's): what the first two half-steps read first, the rest in the order the steady state's counted waits expect ----
    auto pro_a = [&](int h) { // GemmTr: stage h;  Gemm: half-stage h
#pragma unroll
        for (int q = 0; q < (TRANS_A ? 8 : 4); ++q) {
            if ((q & 3) == 0) { m16_set_m0(lds_a_wave + (TRANS_A ? h * M16_BS_BYTES + (q >> 2) * 4096 : h * HA_BYTES)); asm volatile("s_nop 0"); }
            m16_dma(q & 3, a_voff[q], a0 + (uint64_t)h * a_full);
        }
    };
    auto pro_b = [&](int stage, int nq) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q >= nq) break;
            if ((q & 3) == 0) { m16_set_m0(lds_b_wave + stage * M16_BS_BYTES + (q >> 2) * 4096); asm volatile("s_nop 0"); }
            m16_dma(q & 3, b_voff[q], b0 + 128 * stage);
        }
    };
    if constexpr (TRANS_A) { pro_a(0); pro_b(0, 8); pro_a(1); pro_b(1, 8); pro_b(2, 4); }
    else { pro_a(0); pro_b(0, 8); pro_a(1); pro_a(2); pro_b(1, 8); pro_a(3); pro_b(2, 4); }
    wait_dma_keep<20>();
    __syncthreads();
    va = vbaseA[0]; va1 = vbaseA1; vb = vbaseB[0];
#pragma unroll
    for (int op = 0; op < kOps; ++op) frag(op, 0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    vb = vbaseB[1]; gb = b0 + 256; // stage 2
    if constexpr (TRANS_A) { va = vbaseA[1]; ga = a0 + 256; }
    else { va = vbaseA[0] + rR; va1 = vbaseA1 + rR; ga = a0 + 4u * a_full; }
    a_step0 = rem_g >= 3u ? (uint32_t)a_full : 0u;
    a_step1 = rem_g >= 4u ? a_full : 0u; b_step = rem_g >= 4u ? 128u : 0u; // (S >= 4: no tile ends within its first stage)
    asm volatile("" : "+s"(a_step0), "+s"(a_step1), "+s"(b_step));
    __builtin_amdgcn_sched_barrier(0);

#ifndef WG_CONT_STORE_PERM
#define WG_CONT_STORE_PERM 1 // the epilogue's packs go through one ds_bpermute_b32 per dword (128 per wave and tile) so that a column's four 16-byte pieces sit on four CONSECUTIVE
                             // lanes (lane = 4 c + piece) instead of lanes 16 apart: the same bytes per store instruction, which an idle CU gets out in 1.1 us instead of 3.4
                             // (tools/cpp/store_probe.hip pattern 5); in the walk's lock-step tiles -- all CUs storing at once -- worth 0.5-1.5 % at K <= 1024
                             // (profiles/r06_cont_store_perm_ab.txt). Bits unchanged. 0: the MFMA layout's own store pattern (A/B builds)
#endif
    // this lane's first row within a tile / its column within an N tile, as the stores see them (permuted form: lane = 4 * column + row group)
    const uint32_t row_w = WG_CONT_STORE_PERM ? 128u * wm + 8u * ((uint32_t)lane & 3u) : 128u * wm + 8u * kg;
    const uint32_t col_w = WG_CONT_STORE_PERM ? (uint32_t)lane >> 2 : (uint32_t)i16;
    const uint32_t perm_src = 4u * (16u * ((uint32_t)lane & 3u) + ((uint32_t)lane >> 2)); // ds_bpermute address: this lane takes the pack of lane 16 * row group + column
    while (true) {
        // the steps from this tile's "stage S" to the next tile's stage 0 (if this workgroup has a next tile)
        const uint32_t next = tile + (uint32_t)walk_stride;
        const bool has_next = tiles_left > 1u;
        const char *na0 = a0, *nb0 = b0;
        uint32_t nm0 = m0, nn0 = n0;
        uint64_t nc_off = c_off;
        if (has_next) bases(next, na0, nb0, nm0, nn0, nc_off);
        a_cross = a_full + (uint64_t)((int64_t)(na0 - a0) - (int64_t)((TRANS_A ? S : 2u * S) * a_full));
        b_cross = 128u + (uint64_t)((int64_t)(nb0 - b0) - (int64_t)S * 128);
        cross_st = has_next ? S - 3u : ~0u;
        nlim_a = lim_a(nm0); nlim_b = g.N - 1u - nn0;
        asm volatile("" : "+s"(a_cross), "+s"(b_cross), "+s"(cross_st), "+s"(nlim_a), "+s"(nlim_b));
        // stage 0: its first half-step multiplies every quad once with C = 0 (no zeroing pass: 256 VALU writes, 0.45 us of a tile with the matrix cores idle)
        if constexpr (TRANS_A) { half_step_tn(c0{}, std::true_type{}); half_step_tn(c1{}, std::false_type{}); }
        else { half_step_nn(c0{}, std::true_type{}); half_step_nn(c1{}, std::false_type{}); }
        while (st < S) {
            if constexpr (TRANS_A) { half_step_tn(c0{}, std::false_type{}); half_step_tn(c1{}, std::false_type{}); }
            else { half_step_nn(c0{}, std::false_type{}); half_step_nn(c1{}, std::false_type{}); } // ++st inside
        }
        // ---- epilogue of `tile`: the stores are NOT waited for; they drain under the next tile's first stage. ONE basic block (no alpha / beta / store-flavour
        // branches: beta == 0 is the launcher's condition, x * 1.0f is x) with a scheduling fence per pack: with branches between the packs the compiler brings all
        // 256 accumulators over to VGPRs at the loop's exit, on top of the live fragments, and spills (ISA of the first cut: 22 dwords, reloaded behind the stores).
        _Float16 *C = g.c + c_off;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // the last MFMA's result is out of the pipe before the first hand-written read
        const bool full_tile = m0 + BM <= g.M && n0 + BN <= g.N; // workgroup-uniform
        static_for<8>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const uint32_t col = n0 + 128u * wn + 16u * u + col_w;
            const bool col_ok = full_tile || col < g.N;
            _Float16 *cc = C + (uint64_t)(col_ok ? col : n0) * ldc + m0 + row_w;
            static_for<4>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const bool ok = col_ok && (full_tile || m0 + row_w + 32u * p < g.M); // 8 consecutive rows, all in or all out (M % 8 == 0)
                float r[8];
                AccQuad<8 * (2 * p) + u>::read(r[0], r[1], r[2], r[3]);
                AccQuad<8 * (2 * p + 1) + u>::read(r[4], r[5], r[6], r[7]);
                half8_t v;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (!ALPHA1) {
                        r[q] *= alpha;
                        asm volatile("" : "+v"(r[q])); // the f32 product, THEN one rounding to f16, as m16_tile's epilogue does it (in one expression the two
                                                       // become v_fma_mix*_f16, whose f16 results differ in the denormal range: 477 of 2^25 elements, alpha = -0.375)
                    }
                    v[q] = (_Float16)r[q];
                }
                if constexpr (!TRANS_A && WG_NN_NOSWAP) { // odd lane rows hold the pair's tiles in exchanged order (rows + 4..7 in tile 2 p): exchanged back on the PACKED
                                                          // halves -- two selects per tile instead of four (the epilogue is the tile boundary's cost: m16_cont's header)
                    const uintx4 w = __builtin_bit_cast(uintx4, v);
                    const uintx4 x = { odd_row ? w[2] : w[0], odd_row ? w[3] : w[1], odd_row ? w[0] : w[2], odd_row ? w[1] : w[3] };
                    v = __builtin_bit_cast(half8_t, x);
                }
                if constexpr (WG_CONT_STORE_PERM) { // (every lane is active here: the predicated store comes after)
                    uintx4 w = __builtin_bit_cast(uintx4, v);
#pragma unroll
                    for (int d = 0; d < 4; ++d) w[d] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)perm_src, (int)w[d]);
                    v = __builtin_bit_cast(half8_t, w);
                }
                _Float16 *dst = cc + 32 * p;
#ifndef WG_CONT_ABLATE
#define WG_CONT_ABLATE 0 // timing experiments only (results are garbage): 1 = the epilogue issues no stores; 2 = no read-out / convert either (the accumulators are simply overwritten)
#endif
                if (ok && !(WG_CONT_ABLATE & 1)) { // (a wave with no row or column inside the matrix skips the instruction: see the wait behind a ragged tile below)
                    if constexpr (STREAM) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if (!has_next) break;
        tile = next; --tiles_left; a0 = na0; b0 = nb0; m0 = nm0; n0 = nn0; c_off = nc_off;
        st = 0;
        after_store = (WG_CONT_ABLATE & 1) ? 0u : TRANS_A ? 1u : 2u; // GemmTr: the next stage's one wait;  Gemm: its two (the third one is the first that needs a piece issued behind the stores)
        if (!full_tile) { // a ragged tile: a wave whose rows or columns lie past the end issued fewer than 32 stores (none, if all of them do), and a counted wait that
                          // allowed for 32 would let that many pieces fly instead: wait the stores out here (edge tiles only: one tile row / column of the output)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            after_store = 0u;
        }
        asm volatile("" : "+s"(st), "+s"(after_store));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing (parked pieces, stores) is in flight when the workgroup ends
}

// (Walks of four tiles taken from the per-XCD tile queues -- one workgroup per walk, dispatched by the hardware to whichever CU frees up, single tiles for a queue's last
// four rounds -- were built for the long-K, many-round shapes the fixed walk loses on, and measured: bit-identical, and no faster than the per-tile launch with its tile
// queues (16384^2 x 8192 3058 -> 3131 / 3081 us, Gemm 3097 -> 3096; 12288^3 2498 -> 2502; walks over CONSECUTIVE queue entries, which take the XCD's 32 CUs off one L2
// patch, 5 % slower). A boundary inside a walk saves 3-5 us, 2 % of a K = 8192 tile, on three tiles of four of three quarters of the tiles: removed.
// profiles/r05_f16_chunked_walk_*.txt.)
template <bool TRANS_A, bool STREAM, bool ALPHA1>
__global__ __launch_bounds__(256, 1) void gemm_f16_m16c_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[160 * 1024];
    // the tiles the hardware's round-robin deal would have given this CU: blockIdx.x, + gridDim.x, ... below g.sched_tiles
    const uint32_t first = blockIdx.x;
    if (first >= g.sched_tiles) return;
    m16_cont<TRANS_A, STREAM, ALPHA1>(g, smem, first, (int32_t)gridDim.x, (g.sched_tiles - 1u - first) / gridDim.x + 1u);
}

// (Round 2's persistent form -- one workgroup per CU calling m16_tile per tile, a barrier in between -- measured 0.5-1 % slower than letting the hardware re-dispatch a
// workgroup per tile, profiles/r02_evidence.md section 3c, and is gone: what a walk gains is the overlap across the tile boundary, which is m16_cont above.)

// Tail split (tile quantisation): when the tile count is a little more than a multiple of the CU count, the last round would
// run on a nearly empty chip for a full tile time. The launcher then runs the full rounds normally and cuts the few tail tiles
// along K over the idle CUs (f32 partial tiles in the workspace); this kernel adds a tile's partials in ASCENDING split order
// (deterministic) and writes it with the usual alpha / beta / ragged-edge rules. grid = (tail tiles, 64): 4 columns per block.
__global__ __launch_bounds__(256) void gemm_f16_tail_reduce(GemmArgs g) {
    uint32_t tm, tn;
    tile_of(blockIdx.x + g.tile_base, g.tiles_m, g.tiles_n, tm, tn);
    const uint32_t row = tm * BM + 4u * (threadIdx.x & 63u), col = tn * BN + blockIdx.y * 4u + (threadIdx.x >> 6);
    if (row >= g.M || col >= g.N) return; // M % 8 == 0: the 4 rows are all in or all out
    const float *p = g.part + (uint64_t)blockIdx.x * 65536u + (uint64_t)(col - tn * BN) * 256u + (row - tm * BM);
    float4 s = *reinterpret_cast<const float4 *>(p);
    for (uint32_t i = 1; i < g.nsplit; ++i) {
        const float4 q = *reinterpret_cast<const float4 *>(p + (uint64_t)i * g.tail_tiles * 65536u);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    _Float16 *o = g.c + (uint64_t)col * g.ldc + row;
    struct alignas(2) h4 { _Float16 v[4]; }; // (C at any element-aligned address)
    if (g.alpha != 1.f) { s.x *= g.alpha; s.y *= g.alpha; s.z *= g.alpha; s.w *= g.alpha; }
    if (g.beta != 0.f) {
        const h4 t = *reinterpret_cast<const h4 *>(o);
        s.x = fmaf(g.beta, (float)t.v[0], s.x); s.y = fmaf(g.beta, (float)t.v[1], s.y);
        s.z = fmaf(g.beta, (float)t.v[2], s.z); s.w = fmaf(g.beta, (float)t.v[3], s.w);
    }
    const h4 r = { { (_Float16)s.x, (_Float16)s.y, (_Float16)s.z, (_Float16)s.w } };
    *reinterpret_cast<h4 *>(o) = r;
}


} // namespace
} // namespace wgf16

using namespace wgf16;

namespace {
// ---------------------------------------------------------------------------------------------------------------------------------
// Calibrated shares across XCDs ("balance", BalancePlan in gemm_f16_common.hpp). Measured (profiles/r03_evidence.md): under the power
// cap the eight XCDs of a chip run the same tile 2-4 % apart (the same XCDs every run), hardware deals every XCD the same number of
// workgroups, and at 8192^3 (4 tiles per CU) 3.7 % of the CU time is idle at the end. Every launch of the kernel over full rounds adds
// each tile's main-loop time to the accumulator of its workgroup slot (b % 8); a 128-byte snapshot of the accumulators travels to the
// host on a side stream now and then; the launcher turns the measured rates into prefix / suffix units of a few stages per CU.
// ---------------------------------------------------------------------------------------------------------------------------------
// MEASURED OUTCOME (profiles/r03_evidence.md section 1): the hand-off costs what it saves. A prefix unit costs its taker ~12 us (prologue,
// 256 KiB of raw accumulators written at the CU's store rate, flag), a suffix unit costs its giver ~10 us (flag, acquire, 256 KiB read past
// its L2) -- together more than the 15-30 us a CU of the slowest XCD is behind at 8192^3. A/B on two boxes: -0.7 ... -1.2 %. So the
// default is OFF (WG_TUNE_F16_BALANCE = 0: no calibration traffic, no plan); -1 lets the planner act where its cost model -- these
// measured overheads -- still predicts a gain (speed spreads above ~5 %), 1 is the tests' fixed pattern. The units themselves are
// bit-identical to the unsplit launch and stay tested.
constexpr double kBalTileOverhead = 5.0; // per tile outside the main loop (prologue, epilogue, re-dispatch), in stages of ~1.4 us
constexpr double kBalTakeOverhead = 8.5; // a prefix unit's extra cost to its taker
constexpr double kBalGiveOverhead = 7.0; // a suffix unit's extra cost to its giver
constexpr uint32_t kBalFlagsOffset = 2048, kBalMaxPairs = 1024, kBalDevBytes = kBalFlagsOffset + kBalMaxPairs * 4;

int bal_prepare(wg_ctx *ctx) { // device block + side stream on first use; harvest a finished snapshot, start the next one
    wg_ctx::F16Balance &b = ctx->bal;
    if (ctx->recording) return WG_OK; // (no allocation, no other-stream work inside a capture)
    if (!b.dev) {
        WG_HIP_TRY(hipMalloc((void **)&b.dev, kBalDevBytes));
        WG_HIP_TRY(hipMemset(b.dev, 0, kBalDevBytes));
        WG_HIP_TRY(hipHostMalloc((void **)&b.host, 16 * sizeof(unsigned long long), hipHostMallocDefault));
        WG_HIP_TRY(hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking));
        WG_HIP_TRY(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
    }
    if (b.inflight) {
        const hipError_t q = hipEventQuery(b.ev);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return WG_OK; }
        WG_HIP_TRY(q);
        b.inflight = false;
        double r[8], mean = 0;
        bool ok = true;
        for (int x = 0; x < 8; ++x) {
            const unsigned long long dt = b.host[2 * x] - b.prev[2 * x], ds = b.host[2 * x + 1] - b.prev[2 * x + 1];
            if (ds < 512) ok = false; // a few tiles per slot at least
            r[x] = ds ? (double)dt / (double)ds : 0.0;
            mean += r[x] / 8.0;
        }
        if (ok) { // (else: keep accumulating into the same interval)
            for (int x = 0; x < 8; ++x) b.prev[2 * x] = b.host[2 * x], b.prev[2 * x + 1] = b.host[2 * x + 1];
            for (int x = 0; x < 8; ++x) ok = ok && r[x] > 0.8 * mean && r[x] < 1.25 * mean;
            if (ok) {
                double m2 = 0;
                for (int x = 0; x < 8; ++x) {
                    const double rn = r[x] / mean;
                    b.rel[x] = b.valid ? 0.5 * b.rel[x] + 0.5 * rn : rn;
                    m2 += b.rel[x] / 8.0;
                }
                for (int x = 0; x < 8; ++x) b.rel[x] /= m2;
                b.valid = true;
                b.updates++;
            }
        }
    }
    // a snapshot per eligible launch until the rates have settled, then one in sixteen
    static thread_local uint32_t tick = 0;
    if (!b.inflight && (b.updates < 8 || (++tick & 15u) == 0)) {
        WG_HIP_TRY(hipMemcpyAsync(b.host, b.dev, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b.side));
        WG_HIP_TRY(hipEventRecord(b.ev, b.side));
        b.inflight = true;
    }
    return WG_OK;
}

// rel[x]: time per stage of slot x relative to the mean. Returns the number of pairs (0: nothing worth moving).
// `forced` (WG_TUNE_F16_BALANCE = 1, tests): a fixed pattern of takers and givers, two giving rounds included, whatever the size.
uint32_t bal_plan(const double rel[8], uint32_t tiles, uint32_t S, bool forced, BalancePlan &bp) {
    uint32_t n[8];
    double w[8], delta[8], sum_w = 0, sum_inv = 0;
    for (int x = 0; x < 8; ++x) {
        n[x] = tiles / 8u + ((uint32_t)x < tiles % 8u ? 1u : 0u);
        w[x] = ((double)S + kBalTileOverhead) * n[x] / 32.0; // stage-equivalents per CU
        sum_w += w[x];
        sum_inv += 1.0 / rel[x];
    }
    const double T = sum_w / sum_inv; // common finishing time if work could move freely
    for (int x = 0; x < 8; ++x) delta[x] = T / rel[x] - w[x];
    double want[8], rem[8]; // stages per CU a taker wants to add / a giver wants to shed (hand-off costs included)
    for (int x = 0; x < 8; ++x) {
        want[x] = delta[x] > 0 ? delta[x] - kBalTakeOverhead : 0.0;
        rem[x] = delta[x] < 0 ? -delta[x] + kBalGiveOverhead : 0.0;
    }
    if (forced) {
        const double s = (double)S, take[8] = { s / 3, 0, s / 2, s / 4, 0, 0, s / 4, 0 }, give[8] = { 0, s / 4, 0, 0, 0, s / 2, 0, s / 2 + s / 4 };
        for (int x = 0; x < 8; ++x) want[x] = take[x], rem[x] = give[x];
    }
    uint32_t rounds_used[8] = { 0 }, max_rounds[8], pairs = 0;
    for (int x = 0; x < 8; ++x) {
        // a giving round = up to 32 tiles at the end of the slot's list; at least as many plain tiles stay in front of every giving round
        const uint32_t r = n[x] / 64u;
        max_rounds[x] = r > 2u ? 2u : r;
        if (n[x] < 64u) max_rounds[x] = n[x] >= 4u ? 2u : (n[x] >= 2u ? 1u : 0u); // few tiles (tests, small products): rounds of n / 4
        bp.len[x] = n[x];
        bp.pre_cnt[x] = 0; bp.pre_src[x] = 0; bp.pre_p[x] = bp.pre_slot[x] = bp.pre_pair0[x] = 0;
        for (int r2 = 0; r2 < 2; ++r2) bp.suf_lo[x][r2] = bp.suf_p[x][r2] = bp.suf_pair0[x][r2] = 0, bp.suf_cnt[x][r2] = 0;
    }
    bool taken[8] = { false };
    for (;;) {
        int f = -1;
        for (int x = 0; x < 8; ++x)
            if (!taken[x] && want[x] >= 3.0 && (f < 0 || want[x] > want[f])) f = x;
        if (f < 0) break;
        taken[f] = true;
        int g = -1;
        for (int x = 0; x < 8; ++x)
            if (rem[x] >= 3.0 && rounds_used[x] < max_rounds[x] && (g < 0 || rem[x] > rem[g])) g = x;
        if (g < 0) break;
        double pd = want[f];
        if (pd > rem[g]) pd = rem[g];
        uint32_t p = (uint32_t)(pd + 0.5);
        if (p + 3u > S) p = S - 3u;
        if (p < 3u) continue;
        if (!forced && (double)p < kBalGiveOverhead + 1.0) continue; // the giver must come out ahead too
        const uint32_t round = n[g] >= 64u ? 32u : (n[g] >= 4u ? n[g] / 4u : 1u);     // tiles per giving round of this giver
        const uint32_t cnt = round < 32u ? round : 32u;                                // (a taker has 32 CUs)
        const uint32_t lo = n[g] - round * (rounds_used[g] + 1u);
        if (pairs + cnt > kBalMaxPairs) break;
        bp.pre_cnt[f] = cnt; bp.pre_src[f] = (uint32_t)g; bp.pre_p[f] = p; bp.pre_slot[f] = lo; bp.pre_pair0[f] = pairs;
        bp.len[f] = n[f] + cnt;
        const uint32_t r2 = rounds_used[g]++;
        bp.suf_lo[g][r2] = lo; bp.suf_cnt[g][r2] = cnt; bp.suf_p[g][r2] = p; bp.suf_pair0[g][r2] = pairs;
        pairs += cnt;
        rem[g] -= (double)p;
    }
    return pairs;
}

thread_local bool g_padding = false; // set while wgk_gemm_f16 runs on padded copies (see the staging branch)
int pad_copy(wg_ctx *ctx, _Float16 *dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd, const _Float16 *src, uint32_t ld_src,
             uint64_t src_batch, uint32_t rs, uint32_t cs, uint32_t nmats) {
    // zero-padded copy of a column-major block, any alignment on either side (transpose.hip: 16-byte accesses with a byte shift)
    return wgk_stage_copy(ctx, WG_F16, dst, ld_dst, dst_batch, rd, cd, src, ld_src, src_batch, rs, cs, nmats);
}
} // namespace

int wgk_gemm_f16(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats,
                 __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2, float alpha, float beta, const wgk_panels *panels) {
    if (M == 0 || N == 0 || nmats == 0) return panels ? WG_ERR_UNSUPPORTED : WG_OK;
    if (nmats > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: more than 65535 matrices in one call");
    GemmArgs g;
    g.a = (const _Float16 *)m1.ptr; g.lda = m1.ld; g.a_batch = m1.batch;
    g.b = (const _Float16 *)m2.ptr; g.ldb = m2.ld; g.b_batch = m2.batch;
    g.c = (_Float16 *)out; g.ldc = out_ld; g.c_batch = out_batch;
    g.M = M; g.N = N; g.K = K;
    g.alpha = alpha; g.beta = beta;
    {   // the result past the caches when it would push the operands out of the Infinity Cache (see GemmArgs::c_stream); beta != 0 reads C back
        const uint64_t MiB = 1ull << 20, ab = ((uint64_t)M * K + (uint64_t)K * N) * nmats * 2u, cb = (uint64_t)M * N * nmats * 2u;
        g.c_stream = (beta == 0.f && ab <= 256u * MiB && ab + cb > 256u * MiB) ? 1u : 0u;
#ifdef WG_FORCE_C_STREAM
        g.c_stream = WG_FORCE_C_STREAM; // experiment builds
#endif
        // (read by gemm_f16_t128.hip only: one column of 128-wide tiles. Gemm only: GemmTr's k-contiguous A arrives as 64-byte row pieces, two per line at different times)
        g.a_nt = (!trans && N <= 128u && (uint64_t)M * K * 2u >= 384u * MiB) ? 1u : 0u;
    }
    g.tile_base = 0; g.tail_tiles = 0;
    g.sched = nullptr; g.sched_tiles = 0;
    g.calib = nullptr; g.bal = BalancePlan{}; g.panel = PanelArgs{};

    auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
#ifndef WG_F16_SKINNY
#define WG_F16_SKINNY 1 // 0: f16 GemmTr with few columns never takes the streaming kernel (A/B builds)
#endif
    // GemmTr with N <= 16 on a matrix that is not launch-bound (an f16 GemvTr with a few right-hand sides; a weight matrix applied to a small batch): HBM-bound on m1,
    // the tiled kernels below spend 15/16 of a 128-column tile on nothing. The few-column streaming kernel (gemm_f32_skinny.hip, T = _Float16) reads m1 once.
    if (WG_F16_SKINNY && trans && !panels && N <= 16u && M >= 512u && M % 4u == 0 && K >= 256u && K % 8u == 0 && m1.ld % 8u == 0 && m2.ld % 8u == 0 &&
        out_ld % 4u == 0 && al16(m1.ptr) && al16(m2.ptr) && ((uintptr_t)out & 7) == 0 && (nmats == 1 || (m1.batch % 8u == 0 && m2.batch % 8u == 0 && out_batch % 4u == 0)) &&
        (uint64_t)M * K * 2u >= (16ull << 20) && (uint64_t)m1.ld * 32u * 2u < (1ull << 31) && (uint64_t)m2.ld * 32u * 2u < (1ull << 31))
        return wgk_gemm_f16_skinny(ctx, M, N, K, nmats, out, out_ld, out_batch, m1, m2, alpha, beta);
    // 32-bit DMA offsets within a tile: rows * ld * 2 bytes must stay below 2^31
    const bool off_ok = (uint64_t)m1.ld * 2u * (trans ? 256u : 32u) < (1ull << 31) && (uint64_t)m2.ld * 2u * 256u < (1ull << 31);
    // (N is free: B rows are clamped per column and the epilogues skip columns >= N)
    // K: any multiple of 8 with >= 3 whole stages (the 256 x 256 kernel: a K % 64 remainder is the accumulators' initial value, m16_tile);
    // from one whole stage on (the 128 x 128 kernel, same treatment of the remainder). Everything else is zero-padded along K by the staging branch below.
    const uint32_t krem = K % 64u;
    const bool k_big = K % 8u == 0 && K - krem >= 192u, k_small = K % 8u == 0 && K - krem >= 64u;
    const bool a_step_fits = trans || (uint64_t)m1.ld * 64u < (1ull << 32); // NN: a half-stage of A (32 k rows) apart in 32 bits (the DMA cursors' increments are SGPRs)
    // (Leading dimensions, base addresses and batch strides: anything element-aligned since round 6. The operands come in by LDS-DMA and the results leave in 16-byte
    // stores, and both take any element-aligned address on this target -- tools/cpp/unaligned_probe.hip, unaligned_dma_probe.hip: the aligned rate at 4-byte offsets, 0.9 of
    // it at 2-byte ones. Until then such views went through padded copies: 2048^3 with one odd leading dimension 35 us instead of 25, a C at an odd offset twice the time.)
    // (The DMA'd operands at 4-byte alignment, though: pieces that start 2 bytes off a dword cost the GemmTr of 8192^2 x 1024 146 us against 127 on a padded copy of A.)
    auto al4 = [](const void *p) { return ((uintptr_t)p & 3) == 0; };
    const bool a_al = al4(m1.ptr) && m1.ld % 2 == 0 && (nmats == 1 || m1.batch % 2 == 0), b_al = al4(m2.ptr) && m2.ld % 2 == 0 && (nmats == 1 || m2.batch % 2 == 0);
    const bool fast = (M % 8 == 0) && (k_big || k_small) && a_step_fits && off_ok && a_al && b_al;
    auto panels_ok = [&]() -> bool { // n_main panels of `cols`, then 1 .. 8 tail panels (<= 255 tile columns each) that end exactly at N
        if (!panels->cols || panels->cols % 256u || panels->n_tail < 1 || panels->n_tail > (uint32_t)kPanelTail || panels->n_main + panels->n_tail < 2) return false;
        uint64_t c0 = (uint64_t)panels->n_main * panels->cols;
        for (uint32_t q = 0; q + 1u < panels->n_tail; ++q) {
            if (panels->tail_cols[q] == 0 || panels->tail_cols[q] % 256u || panels->tail_cols[q] / 256u > 255u) return false;
            c0 += panels->tail_cols[q];
        }
        return c0 < N && N - c0 == panels->tail_cols[panels->n_tail - 1u] && (N - c0 + 255u) / 256u <= 255u;
    };
    if (panels && !(fast && k_big && nmats == 1 && alpha == 1.f && beta == 0.f && panels_ok() &&
                    (uint64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) >= (uint64_t)(ctx->compute_units > 0 ? ctx->compute_units : 256)))
        return WG_ERR_UNSUPPORTED; // (no message: the caller falls back to one launch per panel)
    if (fast) {
        g.tiles_m = (M + BM - 1) / BM;
        g.tiles_n = (N + BN - 1) / BN;
        const uint64_t tiles = (uint64_t)g.tiles_m * g.tiles_n;
        if (tiles > 0x7fffffffull) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: too many tiles");
        const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
        if (panels) {
            const uint32_t ptn = panels->cols / 256u;
            g.panel.cols = panels->cols; g.panel.n_main = panels->n_main; g.panel.npanels = panels->n_main + panels->n_tail; g.panel.tiles = g.tiles_m * ptn;
            uint32_t tn_left = g.tiles_n - panels->n_main * ptn;
            for (uint32_t q = 0; q < panels->n_tail; ++q) { // (validated above: multiples of 256 that, with the main panels, cover N exactly; the last may be ragged)
                const uint32_t t = q + 1u == panels->n_tail ? tn_left : panels->tail_cols[q] / 256u;
                g.panel.tail_tn |= (uint64_t)t << (8u * q);
                tn_left -= t;
            }
            g.panel.col_stride = panels->col_stride; g.panel.slot_rows = panels->slot_rows;
            g.panel.counters = panels->counters; // (never reset here: they only ever grow, the caller waits for its own running totals)
        }
        // Outputs with fewer 256 x 256 tiles than CUs: the 128 x 128 kernel (gemm_f16_t128.hip) fills the chip with four times as many
        // tiles instead of split-K partial slabs, at ~2/3 of the big kernel's rate per busy CU. Estimates from measured rates
        // (profiles/r01_evidence.md section 12; us per k of one tile: 256 x 256 0.0234 with 8 us per workgroup of prologue + epilogue;
        // 128 x 128 0.00875 alone on a CU, 0.0108 each when several share it, + 6 us; f32 partial slabs written at ~3.5 TB/s + 3 us,
        // reduced at ~7 TB/s + 4 us). WG_F16_TILE=128|256 forces the choice (tests, experiments).
        // (K % 64 != 0: the 128 x 128 kernel multiplies the remainder first, like the big one; it needs >= 64 whole k behind it)
        // 256 x 128 tiles, two workgroups per CU (gemm_f16_t128.hip, TM = 256): short K against the big kernel's per-tile costs. WG_F16_TILE=256128 forces it.
        // Model from the sweep (profiles/r04_evidence.md section 9; us per round of the chip): the big kernel 12.3 + 0.0213 K per round of 256 tiles; a PAIR of
        // co-resident 256 x 128 tiles per CU 6.1 + 0.0263 K (Gemm) / 5.7 + 0.0298 K (GemmTr: its k-contiguous A arrives as 64-byte row pieces, twice the L2
        // requests), a last partial round of at most one tile per CU 0.6 of that. Crossover K ~ 1300 (Gemm) / ~ 650 (GemmTr): 8192 x 8192 x 256 71 -> 51 us
        // (vendor 55), x 512 92 -> 78 (83), 6144 x 6144 x 512 76 -> 50 (52). Only from one round of 256 x 256 tiles on (fewer: the 128 x 128 logic below).
        // Round 5: where the product can take the continuous walk the big kernel's side of the comparison is that walk's model (it is ahead of the pairs on whole
        // rounds at every K: GemmTr 8192^2 x 256 54 -> 50 us, x 512 86 -> 71; the pairs keep ragged tile counts such as 6144^2 x 512).
        bool t256x128 = ctx->tuning[WG_TUNE_F16_TILE] == 256128;
        // (whether launch_tiles below would put the product on the continuous walk by its default rule: whole tiles and stages, more than one round)
        const bool cont_shape = !panels && krem == 0 && K >= 256u && K <= 4096u && tiles * nmats > (uint64_t)cus && g.beta == 0.f && // (K: the pairs are a short-K choice anyway)
                                ctx->tuning[WG_TUNE_F16_CONT] != 0 && ctx->tuning[WG_TUNE_F16_SCHED] < 0 && ctx->tuning[WG_TUNE_F16_BALANCE] != 1;
        // (GemmTr's cap was 768 while the big kernel's side of the comparison was the per-tile launch; against the walk's model the pairs only win ragged tile counts, at any
        // K up to here: 4608^2 x 1024, 324 tiles: walk + cut-up tail 66.6 us, pairs 51.5; profiles/r05_f16_tile_sweep.txt)
        if (ctx->tuning[WG_TUNE_F16_TILE] == 0 && !panels && tiles * nmats >= (uint64_t)cus && K <= 1536u) {
            double t_big = (double)((tiles * nmats + cus - 1) / cus) * (12.3 + 0.0213 * K);
            if (cont_shape) { // the continuous walk (launch_tiles below): 5 us + 5.7 + 0.0211 K per full round; a last partial round costs a whole one, or -- up to half a
                              // round of tiles, from 6 stages on -- the cut-up tail's two extra launches (8192^2 x 256 49.6 us, x 512 71.4, x 1024 114.5; 6144^2 x 512 67.2)
                const double per = 5.7 + 0.0211 * K;
                const uint64_t all = tiles * nmats;
                const uint32_t r = (uint32_t)(all % (uint64_t)cus);
                t_big = 5.0 + (double)(all / (uint64_t)cus) * per + (r == 0 ? 0.0 : (2u * r <= (uint32_t)cus && K >= 384u && nmats == 1 ? 25.0 + 0.0107 * K : per));
            }
            const double r2 = (double)((uint64_t)((M + 255u) / 256u) * ((N + 127u) / 128u) * nmats) / (2.0 * cus), fl = floor(r2), fr = r2 - fl;
            const double pair = trans ? 5.7 + 0.027 * K : 6.1 + 0.0263 * K; // (GemmTr's slope re-fitted in round 5 on 6144^2 x 768 / 1024 / 1536 and 4608^2 x 1024: 0.0262 .. 0.0277)
            t256x128 = fl * pair + (fr > 0.0 ? (fr <= 0.5 ? 0.6 : 1.0) * pair : 0.0) < 0.95 * t_big;
        }
        // Fewer 256 x 256 tiles than CUs, but about one 256 x 128 tile per CU (70 .. 100 % of them): that tile ALONE on its CU is ahead of both other families from K = 512
        // to 4096 (tools/f16_tile_sweep.py, profiles/r05_f16_tile_sweep.txt; 128 x 128 | 256 x 256 | 256 x 128, us): 4096 x 2048 x 2048 41.2 | 52.9 | 37.4, GemmTr 43.4 | 51.8 |
        // 36.5; x 4096 73.1 | 74.3 | 66.2; 3584 x 2048 x 2048 38.5 | 49.0 | 35.6; 2560^2 x 1024 GemmTr 22.2 | 35.6 | 19.6; 2048^3 x 2 matrices 42.3 | 53.3 | 38.3. At K = 8192 the
        // big tile is back in front (4096 x 2048 x 8192 140 | 119 | 130), at K = 512 the three are level.
        // (one or two matrices: 1024^3 x 8, the same tile counts, is 5 % faster on the 128 x 128 kernel)
        if (ctx->tuning[WG_TUNE_F16_TILE] == 0 && !panels && nmats <= 2u && tiles * nmats < (uint64_t)cus && K >= 512u && K <= 4096u) {
            const uint64_t tt = (uint64_t)((M + 255u) / 256u) * ((N + 127u) / 128u) * nmats;
            if (tt <= (uint64_t)cus && 10u * tt >= 7u * (uint64_t)cus) t256x128 = true;
        }
        if ((krem == 0 || K - krem >= 64u) && !panels && t256x128) {
            GemmArgs t = g;
            t.tiles_m = (M + 255u) / 256u;
            t.tiles_n = (N + 127u) / 128u;
            t.nsplit = 1; t.k_per_split = K; t.part = nullptr;
            const uint64_t tiles_t = (uint64_t)t.tiles_m * t.tiles_n;
            if (tiles_t <= 0x7fffffffull && nmats <= 65535u) return t128_launch(ctx, trans, dim3((uint32_t)tiles_t, nmats), t, 256);
        }
        if ((krem == 0 || K - krem >= 64u) && !panels) {
            const double out_bytes = (double)M * N * nmats * 4.0;
            auto slabs = [&](uint32_t ns) { return ns > 1 ? ns * out_bytes / 3.5e6 + 3.0 + 4.0 + ns * out_bytes / 7.0e6 : 0.0; };
            GemmArgs t = g;
            t.tiles_m = (M + 127u) / 128u;
            t.tiles_n = (N + 127u) / 128u;
            const uint64_t tiles128 = (uint64_t)t.tiles_m * t.tiles_n;
            // split-K only when even these tiles leave more than half of the CUs empty, and then >= 1024 k per split
            uint32_t ns = 1;
            if (tiles128 * nmats * 2u <= (uint64_t)cus) {
                ns = (uint32_t)((uint64_t)cus / (tiles128 * nmats));
                if (ns > K / 1024u) ns = K / 1024u; // (whole stages per split; the last split also takes the K % 64 remainder)
                while (ns > 1 && (double)ns * out_bytes > (double)(512ull << 20)) --ns;
                if (ns < 2) ns = 1;
            }
#ifdef WG_T128_FORCE_NS
            if (tiles128 * nmats <= (uint64_t)cus && K >= 1024u) ns = WG_T128_FORCE_NS; // experiment: K cut on a full round of 128 x 128 tiles
#endif
            bool want128 = false;
            if (!k_big) want128 = true; // fewer than the three whole stages the big kernel's DMA pipeline runs ahead
            else if (ctx->tuning[WG_TUNE_F16_TILE]) want128 = ctx->tuning[WG_TUNE_F16_TILE] == 128;
            else if (N <= 64u) want128 = true; // few columns: HBM-bound on op(A), and a 256-wide tile multiplies four times the padding (65536 x 8 x 4096: 140 -> 104 us, 131072 x 8 x 1024: 72 -> 41)
            else if (tiles * nmats < (uint64_t)cus) {
                const double w128 = (double)(tiles128 * nmats * ns) / cus, k128 = (double)(((K / 64u + ns - 1) / ns) * 64u);
                const double est128 = (w128 <= 1.0 ? k128 * 0.00875 : w128 * k128 * 0.0108) + 6.0 + slabs(ns);
                const uint32_t ns256 = wg_splitk_plan(tiles * nmats, (uint32_t)cus, K / BKH, 8, (uint64_t)M * N * nmats, 512ull << 20);
                const double k256 = (double)(((K / BKH + ns256 - 1) / ns256) * BKH);
                // (x 0.8, round 5: the 0.0234 us per k is the whole chip's, power-capped; fewer than 256 workgroups clock higher -- 1280 x 7168 x 5120, 140 tiles: 128 us by the
                // formula, 90 measured, and the 128 x 128 kernel it sent the product to takes 123; 4096 x 2048 x 2048 / 4096 / 8192 with two splits: 68 / 92 / 140 against 53 / 74 / 119)
                // (lightly split plans only: with K cut many ways across the chip's idle CUs the formula is, if anything, optimistic -- 256 x 256 x 8192: 25 us by it, 28 measured, and x 0.8 sent that
                // product and 384 x 1408 x 2816 / 1024 x 1408 x 6144 to this kernel at 1.2-1.7 x the 128 x 128 kernel's time for an hour of the round)
                // (... up to four splits, where the measured / formula ratio is 0.73-0.83: 4096 x 2048 x 2048 .. 8192 with two, 512 x 1408 x 6144 x 8 matrices with two -- 107 by the
                // formula, 88 measured, and the 128 x 128 kernel it went to without the factor takes 124 --, 2048^3 with four; from eight splits on it is 1.1-1.2)
                const double est256 = (ns256 <= 4u ? 0.8 : 1.0) * ((double)((tiles * nmats * ns256 + cus - 1) / cus) * (k256 * 0.0234 + 8.0) + slabs(ns256));
                want128 = est128 < est256;
            }
            if (want128 && tiles128 <= 0x7fffffffull) {
                t.nsplit = ns;
                t.k_per_split = ns > 1 ? ((K / 64u + ns - 1) / ns) * 64u : K;
                t.part = nullptr;
                if (ns > 1) {
                    t.nsplit = ns = (K - krem + t.k_per_split - 1) / t.k_per_split;
                    void *ws = nullptr;
                    if (int rc = wg_ctx_workspace(ctx, (size_t)ns * M * N * nmats * sizeof(float), &ws)) return rc;
                    t.part = (float *)ws;
                }
                if ((uint64_t)nmats * ns <= 65535) {
                    if (int rc = t128_launch(ctx, trans, dim3((uint32_t)tiles128, nmats * ns), t)) return rc;
                    if (ns > 1) return wg_splitk_reduce(ctx, t.part, ns, M, N, nmats, WG_F16, out, out_ld, out_batch, alpha, beta);
                    return WG_OK;
                }
            }
        }
        if (!k_big) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: K = %u with %u matrices does not fit the 128 x 128 kernel's launch", K, nmats);
        // split-K when the output has too few tiles for the chip (1 workgroup per CU): >= 8 half-steps (256 k) per split. Every split is a
        // whole number of stages and at least three of them (the DMA stream runs three stages ahead); the LAST one also takes the K % 64
        // remainder.
        uint32_t nsplit = panels ? 1u : wg_splitk_plan(tiles * nmats, (uint32_t)cus, K / BKH, 8, (uint64_t)M * N * nmats, 512ull << 20);
        const uint32_t stages = K / 64u;
        auto kps_of = [&](uint32_t ns) { return ((stages + ns - 1u) / ns) * 64u; };
        while (nsplit > 1) {
            const uint32_t kps = kps_of(nsplit), n = (K - krem + kps - 1u) / kps;
            if (n == nsplit && kps >= 192u && K - krem - (n - 1u) * kps >= 192u) break;
            nsplit = n < nsplit ? n : nsplit - 1u;
        }
        g.nsplit = nsplit;
        g.k_per_split = nsplit > 1 ? kps_of(nsplit) : K;
        g.part = nullptr;
        if (nsplit > 1) {
            void *ws = nullptr;
            if (int rc = wg_ctx_workspace(ctx, (size_t)nsplit * M * N * nmats * sizeof(float), &ws)) return rc;
            g.part = (float *)ws;
        }
        if ((uint64_t)nmats * nsplit > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: nmats * splits exceeds 65535");
        {
#ifdef WG_F16_TRACE
            uint64_t *trace = nullptr;
            if (nsplit == 1) {
                WG_HIP_TRY(hipMalloc((void **)&trace, tiles * nmats * 192));
                WG_HIP_TRY(hipMemset(trace, 0, tiles * nmats * 192));
                g.part = (float *)trace;
                g.trace_tiles = (uint32_t)(tiles * nmats);
            }
#endif
            // the launch of `ntiles` whole tiles (ids 0 .. ntiles - 1). From WG_F16_SCHED_ROUNDS rounds of the chip on, the workgroups
            // take their tiles from the per-XCD queues (m16_acquire_tile) and the launch carries an eighth more of them than tiles.
            // Stealing whole tiles evens the XCDs out to about half a tile per CU, and taking a tile costs ~1.5 us (an atomic and two
            // barriers ahead of the prologue): measured neutral at 8-16 rounds, -0.9 % at 4, +3.4 % at 64 (32768^3).
#ifndef WG_F16_SCHED_ROUNDS
#define WG_F16_SCHED_ROUNDS 16
#endif
            auto launch_tiles = [&](GemmArgs gm, uint32_t ntiles) -> int {
                uint32_t nwg = ntiles;
                const int sched_env = ctx->tuning[WG_TUNE_F16_SCHED]; // 0 / 1 force (tests), default: by size
                // The continuous tile walk (m16_cont): whole 256 x 256 tiles, whole stages, more than one round of tiles (one round: nothing to continue into).
                // One workgroup per CU; a tile's prologue, re-dispatch and store drain (~5 us) go under its neighbours' multiplies. GemmTr 8192 x 8192 x 256
                // 71 -> 50 us, x 640 104 -> 83, x 1024 131 -> 114 (vendor 112-114), x 2048 216 -> 200, x 4096 386 -> 369, 16384^2 x 1024 531 -> 447, 16384^2 x 4096
                // 1511 -> 1484; Gemm 8192^2 x 256 66 -> 47, x 1024 126 -> 111 (vendor 128), x 2048 210 -> 197, x 4096 383 -> 372, 16384^2 x 1024 518 -> 437.
                // 8192^3 717 -> 722 and 725 -> 717 (two boxes: nothing), 16384^2 x 8192 3090 -> 3145, 12288^3 2496 -> 2516: from K ~ 8192 on the XCDs' uneven speeds
                // (tile scheduler, calibrated shares) weigh more than the tile boundaries. A cut-up tail (below) follows the full rounds as before.
                // Below the tile scheduler's 16 rounds the walk still gains 1-3 % at K = 5120 ... 8192 (5120^3 214 -> 205, 8192^2 x 6144 563 -> 549, 8192^3 737 -> 728, Gemm
                // 760 -> 754; 8 rounds: 8192 x 16384 x 8192 1496 -> 1474, 131072 x 1024 x 8192 1563 -> 1552; 9 rounds: 12288^2 x 6144 1300 -> 1276).
                // WG_TUNE_F16_CONT: 0 never, 1 wherever it applies, -1 (default) K <= 4096, or K <= 8192 below 16 rounds of tiles, and neither the tile scheduler nor
                // the calibrated shares forced on.
                {
                    const int cont = ctx->tuning[WG_TUNE_F16_CONT];
                    const uint64_t all = (uint64_t)ntiles * nmats; // a batch: the walk goes through the matrices' tiles in turn (grid.y of the per-tile launch, flattened)
                    const bool applies = gm.nsplit == 1 && gm.panel.cols == 0 && gm.tail_tiles == 0 && gm.tile_base == 0 &&
                                         krem == 0 && K >= 256u && all > (uint64_t)cus && all <= 0x7fffffffull && gm.beta == 0.f;
                    const bool by_rule = (K <= 4096u || (K <= 8192u && all < (uint64_t)(WG_F16_SCHED_ROUNDS * cus))) && !ctx->uneven_xcds && sched_env < 0 && ctx->tuning[WG_TUNE_F16_BALANCE] != 1;
                    if (cont != 0 && applies && (cont == 1 || by_rule)) {
                        const dim3 grid((uint32_t)cus), block(256);
                        gm.sched = nullptr; gm.sched_tiles = (uint32_t)all;
                        auto go = [&](auto tr_c, auto st_c, auto a1_c) {
                            hipLaunchKernelGGL((gemm_f16_m16c_kernel<decltype(tr_c)::value, decltype(st_c)::value, decltype(a1_c)::value>), grid, block, 0, ctx->stream, gm);
                        };
                        auto by_alpha = [&](auto tr_c, auto st_c) { if (gm.alpha == 1.f) go(tr_c, st_c, std::true_type{}); else go(tr_c, st_c, std::false_type{}); };
                        auto by_stream = [&](auto tr_c) { if (gm.c_stream) by_alpha(tr_c, std::true_type{}); else by_alpha(tr_c, std::false_type{}); };
                        if (trans) by_stream(std::true_type{}); else by_stream(std::false_type{});
                        return WG_OK;
                    }
                }
                // (a stream whose missing CUs all come from one XCD: that XCD cannot keep up with an eighth of the tiles -- the others take them from 2 rounds on)
                const bool dyn = gm.nsplit == 1 && nmats == 1 && (sched_env >= 0 ? sched_env != 0 : ntiles >= (uint32_t)((ctx->uneven_xcds ? 2 : WG_F16_SCHED_ROUNDS) * cus));
                if (dyn) {
                    if (!ctx->tile_queues) {
                        if (ctx->recording) return wg_set_error(WG_ERR_WORKSPACE, "Gemm: the tile queues are needed while recording: run the call once outside the recording first");
                        WG_HIP_TRY(hipMalloc((void **)&ctx->tile_queues, 8 * 128));
                    }
                    WG_HIP_TRY(hipMemsetAsync(ctx->tile_queues, 0, 8 * 128, ctx->stream));
                    gm.sched = ctx->tile_queues; gm.sched_tiles = ntiles;
                    nwg = (ntiles + ntiles / 8u + 7u) & ~7u; // a fast XCD takes ~5 % more than its share; the surplus workgroups exit in ~2 us each
                }
                // Calibrated shares (bal_plan): few rounds of tiles on the whole chip. Below the scheduler's threshold whole tiles are too
                // coarse to steal; prefix / suffix units of a few stages per CU even the XCDs out from measured rates, bit-identically.
                // WG_TUNE_F16_BALANCE: 0 never, 1 whenever the shape allows (tests: made-up rates until real ones exist), -1 by size.
                const int bal_knob = ctx->tuning[WG_TUNE_F16_BALANCE];
                const uint32_t S = gm.K / 64u;
                if (bal_knob != 0 && !dyn && gm.panel.cols == 0 && gm.nsplit == 1 && nmats == 1 && gm.tail_tiles == 0 && gm.tile_base == 0 && cus == 256 && ntiles >= 16u && ntiles < 8u * 65535u && S >= 8u) {
                    if (int rc = bal_prepare(ctx)) return rc;
                    wg_ctx::F16Balance &b = ctx->bal;
                    if (b.dev && ntiles >= (uint32_t)cus) gm.calib = b.dev; // full rounds only: a slot's rate with the whole chip busy
                    const bool want = bal_knob == 1 || (bal_knob < 0 && b.valid && ntiles >= 2u * (uint32_t)cus);
                    if (want && b.dev && !ctx->recording) { // (a recorded launch would replay with this launch's flag epoch)
                        const uint32_t pairs = bal_plan(b.rel, ntiles, S, bal_knob == 1, gm.bal);
                        if (pairs) {
                            void *ws = nullptr;
                            if (int rc = wg_ctx_bal_workspace(ctx, (size_t)pairs * 65536u * sizeof(float), &ws)) return rc;
                            gm.bal.on = 1; gm.bal.epoch = ++b.epoch;
                            gm.bal.flags = (uint32_t *)((char *)b.dev + kBalFlagsOffset);
                            gm.bal.part = (float *)ws;
                            uint32_t mx = 0;
                            for (int x = 0; x < 8; ++x) mx = gm.bal.len[x] > mx ? gm.bal.len[x] : mx;
                            nwg = 8u * mx;
                        } else gm.bal = BalancePlan{};
                    }
                }
                if (trans) hipLaunchKernelGGL((gemm_f16_m16_kernel<true>), dim3(nwg, gm.nsplit * nmats), dim3(256), 0, ctx->stream, gm);
                else hipLaunchKernelGGL((gemm_f16_m16_kernel<false>), dim3(nwg, gm.nsplit * nmats), dim3(256), 0, ctx->stream, gm);
                return WG_OK;
            };
            // tail split: full rounds as they are, the few tiles of a nearly empty last round cut along K over the idle CUs
            uint32_t tail = 0, tail_split = 1, tail_kps = K;
#ifndef WG_F16_TAIL_SPLIT
#define WG_F16_TAIL_SPLIT 1
#endif
            if (WG_F16_TAIL_SPLIT && nsplit == 1 && nmats == 1 && tiles > (uint64_t)cus && !panels) {
                const uint32_t r = (uint32_t)(tiles % (uint64_t)cus);
                if (r > 0 && r * 2u <= (uint32_t)cus) {
                    uint32_t sp = (uint32_t)cus / r;
                    if (sp > stages / 3u) sp = stages / 3u; // >= 3 stages per split
                    while (sp >= 2) {
                        const uint32_t kps = ((stages + sp - 1) / sp) * 64u;
                        const uint32_t n = (K - krem + kps - 1) / kps, last = K - krem - (n - 1) * kps; // (the last split also takes the K % 64 remainder)
                        if (n >= 2 && last >= 192u && (size_t)n * r * 65536u * sizeof(float) <= (512ull << 20)) { tail = r; tail_split = n; tail_kps = kps; break; }
                        --sp;
                    }
                }
            }
            if (tail) {
                void *ws = nullptr;
                if (int rc = wg_ctx_workspace(ctx, (size_t)tail_split * tail * 65536u * sizeof(float), &ws)) return rc;
                const uint32_t full = (uint32_t)tiles - tail;
                if (int rc = launch_tiles(g, full)) return rc; // the full rounds
                GemmArgs gt = g; // the tail tiles, cut along K
                gt.tile_base = full; gt.tail_tiles = tail; gt.nsplit = tail_split; gt.k_per_split = tail_kps; gt.part = (float *)ws;
                if (trans) hipLaunchKernelGGL((gemm_f16_m16_kernel<true>), dim3(tail, tail_split), dim3(256), 0, ctx->stream, gt);
                else hipLaunchKernelGGL((gemm_f16_m16_kernel<false>), dim3(tail, tail_split), dim3(256), 0, ctx->stream, gt);
                hipLaunchKernelGGL(gemm_f16_tail_reduce, dim3(tail, 64), dim3(256), 0, ctx->stream, gt);
                WG_HIP_TRY(hipGetLastError());
                return WG_OK;
            }
            if (int rc = launch_tiles(g, (uint32_t)tiles)) return rc;
#ifdef WG_F16_TRACE
            if (trace) {
                WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
                std::vector<uint64_t> h(tiles * nmats * 24); // 8 u64 per tile, then 4 waves x 8 u32 per tile
                WG_HIP_TRY(hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost));
                if (FILE *f = fopen("/tmp/wg_f16_trace.bin", "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
                (void)hipFree(trace);
                g.part = nullptr;
            }
#endif
        }
        WG_HIP_TRY(hipGetLastError());
        if (nsplit > 1) return wg_splitk_reduce(ctx, g.part, nsplit, M, N, nmats, WG_F16, out, out_ld, out_batch, alpha, beta);
        return WG_OK;
    } else if (!g_padding && (uint64_t)M * N * K >= (1ull << 24) && K > 0 && nmats <= 65535u) {
        struct Guard { Guard() { g_padding = true; } ~Guard() { g_padding = false; } } guard; // the padded call must not pad again
        // Shapes the MFMA kernels do not take as they are (K % 8, fewer than one whole stage, M % 8; an op(A) or B that starts or steps 2 bytes off a dword): at
        // ~40 TFLOP/s the generic kernel below is 20x slower (4096 x 4096 x 4104: 3.2 ms against 0.14 ms). Stage zero-padded dense copies
        // of the operands (and, if the output does not qualify either, a padded output that is copied back) in the context's padding
        // scratch and run the same call on those: HBM-bound passes over a few MB against a GEMM that re-reads them hundreds of times.
        // Only what does not qualify is copied: op(A) when K or M is off, B when K is, the output when M is (N is free: columns are independent;
        // leading dimensions and alignments are free since round 6).
        const bool k_ok = k_big || k_small; // (else: zero-padded to whole stages -- at least one, which the 128 x 128 kernel takes)
        const uint32_t Mp = (M + 7u) & ~7u, Kp = k_ok ? K : ((K + 63u) & ~63u);
        const bool a_ok = k_ok && M == Mp && a_al;
        const bool b_ok = k_ok && b_al;
        const bool c_ok = M == Mp;
        const uint64_t a_elems = a_ok ? 0 : (uint64_t)Mp * Kp, b_elems = b_ok ? 0 : (uint64_t)Kp * N, c_elems = c_ok ? 0 : (uint64_t)Mp * N;
        // every region starts 16-byte aligned: element counts rounded up to 8
        const uint64_t a_sz = ((a_elems + 7u) & ~7ull), b_sz = ((b_elems + 7u) & ~7ull), c_sz = ((c_elems + 7u) & ~7ull);
        void *ws = nullptr;
        if (int rc = wg_ctx_pad_workspace(ctx, (size_t)((a_sz + b_sz + c_sz) * nmats * sizeof(_Float16)) + 16, &ws)) return rc;
        _Float16 *ap = (_Float16 *)ws, *bp = ap + a_sz * nmats, *cp = bp + b_sz * nmats;
        wgk_mat a2 = m1, b2 = m2;
        if (!a_ok) { // op(A) is M x K: stored M x K or, transposed, K x M
            if (trans) { if (int rc = pad_copy(ctx, ap, Kp, a_sz, Kp, Mp, (const _Float16 *)m1.ptr, m1.ld, m1.batch, K, M, nmats)) return rc; }
            else { if (int rc = pad_copy(ctx, ap, Mp, a_sz, Mp, Kp, (const _Float16 *)m1.ptr, m1.ld, m1.batch, M, K, nmats)) return rc; }
            a2 = wgk_mat{ ap, trans ? Kp : Mp, a_sz };
        }
        if (!b_ok) {
            if (int rc = pad_copy(ctx, bp, Kp, b_sz, Kp, N, (const _Float16 *)m2.ptr, m2.ld, m2.batch, K, N, nmats)) return rc;
            b2 = wgk_mat{ bp, Kp, b_sz };
        }
        if (c_ok) return wgk_gemm_f16(ctx, trans, M, N, Kp, nmats, out, out_ld, out_batch, a2, b2, alpha, beta);
        if (beta != 0.f) // the padded output starts as a copy of the old one
            if (int rc = pad_copy(ctx, cp, Mp, c_sz, Mp, N, (const _Float16 *)out, out_ld, out_batch, M, N, nmats)) return rc;
        if (int rc = wgk_gemm_f16(ctx, trans, Mp, N, Kp, nmats, (__half *)cp, Mp, c_sz, a2, b2, alpha, beta)) return rc;
        return pad_copy(ctx, (_Float16 *)out, out_ld, out_batch, M, N, cp, Mp, c_sz, M, N, nmats);
    } else {
        g.tiles_m = (M + 63) / 64;
        g.tiles_n = (N + 63) / 64;
        g.nsplit = 1; g.k_per_split = K; g.part = nullptr;
        if (g.tiles_n > 65535) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm: N too large for the generic f16 path");
        if (int rc = generic_launch(ctx, trans, dim3(g.tiles_m, g.tiles_n, nmats), g)) return rc;
    }
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

// Host-side check of the planner (tests/test_abi_and_host.py; no device needed): the plan for `tiles` whole tiles of `stages` stages from the
// relative slot rates rel8 (or the fixed test pattern), decoded for every workgroup id exactly as the kernel decodes it.
// units[5 i .. 5 i + 4] = (tile, mode, first stage, stages, pair) of the i-th workgroup that has a unit.
extern "C" int wg_debug_f16_balance_plan(const double *rel8, uint32_t tiles, uint32_t stages, int forced, uint32_t *units, uint32_t capacity, uint32_t *nunits,
                                          uint32_t *nworkgroups) {
    if (!rel8 || !nunits || (capacity && !units)) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_f16_balance_plan: NULL argument");
    BalancePlan bp = BalancePlan{};
    const uint32_t pairs = bal_plan(rel8, tiles, stages, forced != 0, bp);
    (void)pairs;
    uint32_t mx = 0, n = 0;
    for (int x = 0; x < 8; ++x) mx = bp.len[x] > mx ? bp.len[x] : mx;
    for (uint32_t b = 0; b < 8u * mx; ++b) {
        uint32_t tile, mode, kb, ns, pair;
        if (!bal_decode(bp, b, stages, tile, mode, kb, ns, pair)) continue;
        if (n < capacity) { units[5 * n] = tile; units[5 * n + 1] = mode; units[5 * n + 2] = kb; units[5 * n + 3] = mode ? ns : stages; units[5 * n + 4] = pair; }
        ++n;
    }
    *nunits = n;
    if (nworkgroups) *nworkgroups = 8u * mx;
    return WG_OK;
}

// What the calibration has measured so far on this context and how many launches ran with calibrated shares (bench.py reports it; tests
// check that a forced launch really took the balanced path).
extern "C" int wg_ctx_f16_balance_info(const wg_ctx *ctx, double *rel8, int *valid, uint32_t *updates, uint32_t *balanced_launches) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_f16_balance_info: ctx is NULL");
    if (rel8) for (int x = 0; x < 8; ++x) rel8[x] = ctx->bal.rel[x];
    if (valid) *valid = ctx->bal.valid ? 1 : 0;
    if (updates) *updates = ctx->bal.updates;
    if (balanced_launches) *balanced_launches = ctx->bal.epoch;
    return WG_OK;
}
