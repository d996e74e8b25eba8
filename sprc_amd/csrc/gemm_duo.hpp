// gemm_duo.hpp -- the "duo" GEMM kernel: 256 x 128 tiles on TWO free-running workgroups per CU, persistent over the tile list.
// Included by gemm_duo.hip (inside namespace sprc, after gemm_impl.hpp: it uses the shared epilogue).
//
// Why a second large-tile kernel.  The 256 x 256 anti-phase kernel owns a CU: while a tile sits in its prologue
// (3.4-4.8 k cycles) or its epilogue (11 k cycles with a 16-bit output, 23-32 k with the fp32 residual stream) the matrix pipe of that CU
// idles -- 21 % of a K = 1408 ViT layer -- and N = 1408 = 5.5 tiles pays MFMA work on padding columns.  Here a workgroup is half the size
// (4 waves, wave tile 128 x 64 as before, so the LDS bytes read per MFMA are the anti-phase kernel's) and a CU holds two of them, each with
// its own accumulators (2 x 128 of the SIMD's 512 registers) and its own 72-KB LDS ring: when one workgroup leaves its K loop the other one
// gets the whole matrix pipe, so an epilogue runs under the neighbour's K loop instead of beside an idle pipe.  N = 1408 / 4224 / 6144 /
// 768 / 9216 are whole multiples of 128: no padding columns.  The ragged last row panel (M = 128 x 257) is a masked tile of the same launch.
//   K-tile = 64 bytes of K (32 fp16): a stage is (256 + 128) rows x 64 B = 24 KB, ring of 3 -> 72 KB per workgroup, 144 of the CU's 160 KB.
//   LDS image of a DMA wave-instruction (1 KB) = 16 rows x 4 16-B slots, lane-linear; physical slot = logical slot ^ ((row >> 2) & 3):
//   each 16-lane group of a ds_read_b128 (lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) then covers all 16 slots of the 256-B bank row.
//   Per K-tile and wave: 12 ds_read_b128, 16 MFMA 32x32x16, 6 DMA instructions, ONE barrier; K-tiles t+1 and t+2 are in flight while t is
//   multiplied (counted vmcnt(6)); the stage of tile t is refilled with tile t+3 right after the barrier that ends its reads.
//   Persistent: grid = 2 x CUs, workgroup b takes tiles b, b + grid, ...  A launch starts both workgroups of a CU at the same moment, and
//   two identical workgroups stay in lock step (both in the K loop at half rate, then both in the epilogue: nothing gained); any initial
//   offset persists (the map offset -> W - E - offset is an involution), so ONE of the two sleeps (W + E) / 2 cycles before its first tile.
//   Which one: the parity of a per-CU arrival counter (key = XCC id, SE / SH / CU id of HW_ID; the counters only ever grow, a launch adds two
//   per CU, so no reset is needed; a wrong parity costs overlap, never correctness).
#pragma once

#ifndef SPRC_DUO_ABL
#define SPRC_DUO_ABL 0          // timing ablations (WRONG results): 1 = no refill loads in the K loop, 2 = no barrier in the K loop, 4 = one workgroup per CU
#endif
constexpr int DUO_BM = 256, DUO_BN = 128, DUO_KTB = 64, DUO_NS = 3;
constexpr int DUO_STAGE = (DUO_BM + DUO_BN) * DUO_KTB;      // 24576
constexpr int DUO_LDS = DUO_NS * DUO_STAGE;                 // 73728
constexpr int DUO_CTRS = 8 * 256;                           // arrival counters: XCC id x (SE | SH | CU id)

// The kernel never touches its by-value parameter: every use site reads the fields it needs through the (laundered) kernarg pointer, so
// that nothing but the K loop's own state is live across the K loop.  (With the parameter block read once at entry, its ~45 dwords
// stayed live across the persistent tile loop: 113 spilled SGPRs, v_readlane / v_writelane inside the K loop, 16 spilled VGPRs.)
typedef const __attribute__((address_space(4))) GemmParams* duo_kparams_t;
__device__ __forceinline__ duo_kparams_t duo_kparams() {
    duo_kparams_t kp = (duo_kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

template <typename T, typename OutT, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(GemmParams) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(sizeof(T) == 2, "16-bit operands");
    constexpr int TM = 4, TN = 2, KTB = DUO_KTB, STAGE = DUO_STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r32 = lane & 31, half = lane >> 5;
    int nt, total;
    {
        duo_kparams_t q = duo_kparams();
        nt = (int)(((int64_t)q->K * 2) / KTB);              // K-tiles of 64 bytes
        total = q->tiles_m * q->tiles_n * (q->dual ? 2 : 1);
    }

    // ---- stagger: the later arrival on this CU sleeps before its first tile ----
    if (duo_kparams()->duo_sleep > 0) {
        duo_kparams_t p = duo_kparams();
        int slot = 0;
        if (tid == 0) {
            const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID
            const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;    // HW_REG_XCC_ID[3:0]
            slot = atomicAdd(p->duo_ctr + xcc * 256 + ((hw >> 8) & 0xffu), 1) & 1;
        }
        slot = __builtin_amdgcn_readfirstlane(slot);        // (lane 0 of wave 0; the other waves wait at the first barrier)
        if (wave == 0 && slot) {
            const uint64_t t0 = __builtin_amdgcn_s_memtime();
            const int64_t dt = p->duo_sleep;
            while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < dt) __builtin_amdgcn_s_sleep(64);
        }
        __builtin_amdgcn_s_barrier();
    }

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const uint32_t sw = (uint32_t)((r32 >> 2) & 3);
    uint32_t addr_a[2], addr_b[2];                          // k-step k of a K-tile: logical slot 2k + half
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t ph = ((uint32_t)(2 * k + half) ^ sw) << 4;
        addr_a[k] = lds0 + (uint32_t)(wr * 128 + r32) * KTB + ph;
        addr_b[k] = lds0 + (uint32_t)(DUO_BM + wc * 64 + r32) * KTB + ph;
    }
    const uint32_t wave_base = (uint32_t)wave * 1024u;      // this wave's 1-KB share of a 4-KB (64-row) piece
    const int srow = lane >> 2;                             // staging: row of this lane inside a 16-row wave-instruction
    const uint32_t sslot = (uint32_t)((lane & 3) ^ ((lane >> 4) & 3)) << 4;   // logical slot behind this lane's physical slot

    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };

    for (int vb0 = blockIdx.x; vb0 < total; vb0 += gridDim.x) {
        int m0, n0;
        bool second;                                        // second product of a paired launch
        uint32_t pc_off[6];                                 // byte offset of this lane's 16-B chunk of piece j from the tile's first row
        __amdgpu_buffer_rsrc_t rs_a, rs_w;
        {
            duo_kparams_t p = duo_kparams();
            const int nwg = p->tiles_m * p->tiles_n;
            int vb = vb0;
            second = p->dual && vb >= nwg;
            if (second) vb -= nwg;
            const char* Wp = second ? p->W2 : p->W;
            const int a_offm = second ? p->a_off2 : p->a_off;
            tile_origin(vb, nwg, p->tiles_m, p->tiles_n, DUO_BM, DUO_BN, p->order, m0, n0);
            const int a_shift = p->a_shift, a_stride = p->a_stride, M = p->M, N = p->N;
            const int64_t lda_b = p->lda_b, ldw_b = p->ldw_b;
            const int64_t a_row0 = map_row_s(a_shift, a_stride, a_offm, m0);
            const bool plain_a = a_shift < 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int row = (j & 3) * 64 + wave * 16 + srow;
                uint32_t o;
                if (j < 4) {
                    const int am = min(m0 + row, M - 1);
                    if (plain_a) o = __umul24((uint32_t)(am - m0), (uint32_t)lda_b);
                    else o = (uint32_t)((map_row_s(a_shift, a_stride, a_offm, am) - a_row0) * lda_b);
                } else {
                    o = __umul24((uint32_t)(min(n0 + row, N - 1) - n0), (uint32_t)ldw_b);
                }
                pc_off[j] = o + sslot;
            }
            rs_a = make_rsrc(p->A + a_row0 * lda_b);
            rs_w = make_rsrc(Wp + (int64_t)n0 * ldw_b);
        }

        f32x16 acc[TM][TN];
        u32x4 fa[2][TM], fb[2][TN];                         // [k-step][row block]

        auto dma_piece = [&](auto S_, auto j_, int t) {
            constexpr int S = decltype(S_)::value, j = decltype(j_)::value;
            buffer_load_lds16(j < 4 ? rs_a : rs_w, smem + (wave_base + (uint32_t)(S * STAGE + j * 4096)), pc_off[j], t * KTB);
        };
        auto dma_tile = [&](auto S_, int t) { static_for<0, 6>([&](auto j_) { dma_piece(S_, j_, t); }); };
        auto read_a = [&](auto S_, auto k_, auto i_) {
            constexpr int S = decltype(S_)::value, k = decltype(k_)::value, i = decltype(i_)::value;
            fa[k][i] = lds_read128<S * STAGE + i * 32 * KTB>(addr_a[k]);
        };
        auto read_b = [&](auto S_, auto k_, auto i_) {
            constexpr int S = decltype(S_)::value, k = decltype(k_)::value, i = decltype(i_)::value;
            fb[k][i] = lds_read128<S * STAGE + i * 32 * KTB>(addr_b[k]);
        };
        auto reads = [&](auto S_, auto k_) {
            static_for<0, TM>([&](auto i_) { read_a(S_, k_, i_); });
            static_for<0, TN>([&](auto i_) { read_b(S_, k_, i_); });
        };
        auto mfma1 = [&](auto k_, auto x_) {
            constexpr int k = decltype(k_)::value, x = decltype(x_)::value, mi = x >> 1, ni = x & 1;
            acc[mi][ni] = mfma_frag<T>(fb[k][ni], fa[k][mi], acc[mi][ni]);
        };
        using std::integral_constant;
        typedef integral_constant<int, 0> I0;
        typedef integral_constant<int, 1> I1;

        // One K-tile.  On entry: fragments of (t, k-step 0) are in flight or landed, K-tile t is visible in stage S, K-tiles t+1 (and t+2)
        // are in flight.  STEADY: K-tiles t+1 .. t+3 exist (no run-time flags).
        auto iter = [&](auto S_, auto steady_, int t) {
            constexpr int S = decltype(S_)::value, S1 = (S + 1) % 3;
            constexpr bool STEADY = decltype(steady_)::value;
            typedef integral_constant<int, S1> SN;
            const bool n1 = STEADY || t + 1 < nt, n2 = STEADY || t + 2 < nt, n3 = STEADY || t + 3 < nt;
            reads(S_, I1{});                                 // (t, k-step 1)
            wait_lgkmcnt<6>();                               // (t, k-step 0) landed
            static_for<0, 8>([&](auto x_) { mfma1(I0{}, x_); });
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkmcnt<0>();                               // (t, k-step 1) landed: this wave is done with stage S
            if (n1) {
                if (n2 && !(SPRC_DUO_ABL & 1)) wait_vmcnt<6>();   // own pieces of K-tile t+1 landed; t+2 may fly
                else wait_vmcnt<0>();
                if constexpr (!(SPRC_DUO_ABL & 2)) barrier();   // K-tile t+1 visible to everyone; everyone is done with stage S
            }
            // k-step 1 of tile t: 8 MFMAs with the refill of stage S (K-tile t+3) and the reads of (t+1, k-step 0) in their shadow
            static_for<0, 8>([&](auto x_) {
                constexpr int x = decltype(x_)::value;
                mfma1(I1{}, x_);
                if constexpr (x < 6 && !(SPRC_DUO_ABL & 1)) { if (n3) dma_piece(S_, integral_constant<int, x>{}, t + 3); }
                if constexpr (x == 1) { if (n1) { read_a(SN{}, I0{}, I0{}); read_a(SN{}, I0{}, I1{}); } }
                if constexpr (x == 3) { if (n1) { read_a(SN{}, I0{}, integral_constant<int, 2>{}); read_a(SN{}, I0{}, integral_constant<int, 3>{}); } }
                if constexpr (x == 5) { if (n1) { read_b(SN{}, I0{}, I0{}); read_b(SN{}, I0{}, I1{}); } }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        // ---- prologue: K-tiles 0 .. 2 into stages 0 .. 2 ----
        barrier();                                           // the previous tile's epilogue strips are dead in every wave
        dma_tile(I0{}, 0);
        if (nt > 1) dma_tile(I1{}, 1);
        if (nt > 2) dma_tile(integral_constant<int, 2>{}, 2);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        asm volatile("" ::: "memory");
        if (nt > 2) wait_vmcnt<12>();
        else if (nt > 1) wait_vmcnt<6>();
        else wait_vmcnt<0>();
        barrier();
        reads(I0{}, I0{});
        {
            int t = 0;
            for (; t + 5 < nt; t += 3) {                     // K-tiles t .. t+2 all have three successors
                iter(I0{}, std::true_type{}, t);
                iter(I1{}, std::true_type{}, t + 1);
                iter(integral_constant<int, 2>{}, std::true_type{}, t + 2);
            }
            for (; t < nt; t += 3) {
                iter(I0{}, std::false_type{}, t);
                if (t + 1 < nt) iter(I1{}, std::false_type{}, t + 1);
                if (t + 2 < nt) iter(integral_constant<int, 2>{}, std::false_type{}, t + 2);
            }
        }
        barrier();                                           // every wave is done reading operand tiles: the epilogue reuses LDS

        GemmParams pe = *(const GemmParams*)duo_kparams();     // (generic view of the constant-address-space block: scalar loads after address-space inference)
        if (second) { pe.bias = pe.bias2; pe.c_off = pe.c_off2; }
        const bool vec_ok = (pe.N % 4 == 0) && (pe.ldc % 4 == 0) && (pe.resid == nullptr || pe.ldr % 4 == 0) &&
                            ((uintptr_t)pe.C % (4 * sizeof(OutT)) == 0) && ((uintptr_t)pe.bias % 16 == 0) && ((uintptr_t)pe.resid % 16 == 0);
        gemm_epilogue<T, OutT, ACT, false, TM, TN>(pe, acc, m0, n0, wr, wc, r32, half, vec_ok, smem, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores have left: the K loop's counted waits see loads only
    }
}
