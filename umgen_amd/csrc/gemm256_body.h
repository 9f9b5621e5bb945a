// 256 x 256 x 64 MFMA GEMM for the TAR / ego prefill stacks (replaces F.linear at module.py:184-190, 236-242 on [B*T*S, E] rows):
//     C[i][j] = sum_k P[i][k] Q[j][k]      P = weights [N features][K], Q = activations [R tokens][K], both K-contiguous, 16-bit
// Round 2's 128 x 128 kernels (gemm.hip) reach 700-800 TFLOP/s on these shapes: one barrier per k-tile with a full vmcnt(0) drain,
// operand feed saturated (TA busy 72-81 %).  This kernel is the deep-pipelined form of MI355X's playbook:
//   * one 512-thread workgroup per CU (persistent, walks its XCD's tile list), 8 waves = 2 (features) x 4 (tokens), each wave a
//     128 x 64 sub-tile = 8 x 4 MFMA tiles of v_mfma_f32_16x16x32 (128 accumulator VGPRs): twice the flops per LDS byte of the
//     128 x 128 form;
//   * operands go HBM -> LDS with global_load_lds (no staging registers), into a RING of 8 half-tile slots (128 rows x 64 k = 16 KB
//     each, XOR-swizzled on the source side so that ds_read_b128 fragment reads are conflict-free): 2 k-tiles x {P half 0/1, Q half 0/1};
//   * (round 4: the four phases below run as TWO of 32 MFMAs -- UMGEN_G256_PH2 -- with the same refill / wait points pairwise merged)
//   * a k-tile is 4 phases of 16 MFMAs (one quadrant of the wave's sub-tile x the whole k-tile).  Every phase issues ONE half-tile
//     refill into the slot whose last reader retired a phase earlier: the P halves of k-tile kt+1 in phases 0 / 1, the Q halves of
//     k-tile kt+2 in phases 2 / 3 -- every load has >= 3 phases (~1.5k cycles) to land, and the loads of an output tile's first
//     k-tiles are issued during the previous tile's last ones (the ring runs on across output tiles);
//   * ONE counted wait per k-tile (phase 3: s_waitcnt vmcnt(4) = "everything but the two newest refills has landed"), raw
//     s_barriers (no vmcnt(0) drain), fragment reads of phase p issued before the barrier that starts p's MFMA block;
//   * epilogues (bias, erf-GELU, 16-bit / fp32 store, fp32 residual read-modify-write) go through a private 4 KB LDS strip per
//     wave so that global memory sees whole 256-byte token-row pieces; no workgroup barrier inside the epilogue.
// Accumulation order of every output element: k ascending in steps of 32 inside the MFMA, the same for every tile position, so
// results do not depend on which other rows are in the launch (scenes stay batch-invariant).
#pragma once
#include <type_traits>

#include "kernels.h"

namespace umgen {

namespace {

constexpr int TM = 256, HK = 64;
constexpr int kSlot = 128 * HK * 2;        // one half-tile: 128 rows x 64 k x 2 B = 16 KB
constexpr int kRing = 8 * kSlot;           // 128 KB
constexpr int kStage = 4096;               // epilogue strip per wave
constexpr int kLds256 = kRing + 8 * kStage;   // 160 KB: one workgroup per CU

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }   // byte offset in a slot

struct Src { int ti, trow0, tmax; };   // an output tile's feature tile, first token row and last addressable token row (wave-uniform)

#ifndef UMGEN_GEMM256_STAGGER
#define UMGEN_GEMM256_STAGGER 1
#endif
constexpr bool STAGGER = UMGEN_GEMM256_STAGGER;
// Measurement builds only (tools/build_variant.sh; the shipped library has neither): UMGEN_G256_EPI = 1 keeps the epilogue's LDS pass but
// drops its global stores, 2 stores without the nontemporal hint, 3 drops the residual epilogue's reads; UMGEN_G256_STAMPS accumulates
// wall-clock ticks (100 MHz) of one workgroup's k-loops / first k-tiles / epilogues, read back by umgen_dbg_gemm_stamps.
#ifndef UMGEN_G256_EPI
#define UMGEN_G256_EPI 0
#endif
constexpr int EPI = UMGEN_G256_EPI;
#ifndef UMGEN_G256_FBALT
#define UMGEN_G256_FBALT 0
#endif
constexpr bool FBALT = UMGEN_G256_FBALT;
#ifndef UMGEN_G256_PH2
#define UMGEN_G256_PH2 1     // two phases of 32 MFMAs per k-tile -- 4 barriers instead of the 8 of the four-phase form (0): fragment reads 16 / 8 per
                             // phase, both P refills in the first, both Q refills and the counted wait in the second.  Same products in the same
                             // order (bit-identical outputs); 8 scenes' rows: q|k 936 -> 994, fc 1015 -> 1048, fc + GELU 833 -> 865, K = 3072
                             // projection 957 -> 1019, V^T 855 -> 908, 4096^3 1283 -> 1402 TFLOP/s (profiles/r04_gemm_bench_ph2.txt)
#endif
#ifdef UMGEN_G256_STAMPS
__device__ unsigned long long g256_stamps[16];
#endif

// The kernel's body as a device function, so that the decode engine's background workers (bg_worker.h) run the SAME code on their share of the
// tile lists: `nx` tile lists (one per XCD of the launch: 8 for a whole-chip launch), this workgroup walks list `xcd` at positions
// lb + i * nloc for i in [i_begin, i_end) -- the whole-chip kernel: (blockIdx.x & 7, 8, blockIdx.x >> 3, gridDim.x >> 3, 0, all).
// ARGS: `const GemmArgs` (the kernel's own argument, by value) or a reference into the CONSTANT address space (an op-list entry the host wrote before the
// launch: its fields are scalar loads issued where they are used, like a kernel argument's -- as register copies they cost the k-loop ~20 scalar registers
// it does not have).
#define UMGEN_AS4 __attribute__((address_space(4)))
template <int MODE, typename TT, typename ARGS>
__device__ __forceinline__ void gemm16_256_body(ARGS a, int nI, int nJ, int splitI, int tpf, int xcd, int nx, int lb, int nloc, int i_begin, int i_end) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // the launch's whole dynamic LDS (kLds256 bytes), from offset 0
    typedef typename Mma16<TT>::vec vec8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;
    const int frow = lane & 15, g = lane >> 4;
    const TT* P = reinterpret_cast<const TT*>(a.P);
    const TT* Q = reinterpret_cast<const TT*>(a.Q);
    const int nkt = a.K / HK;                 // even (launcher)
    // ---- this XCD's tile list (block b runs on XCD b % 8): splitI feature groups x (8 / splitI) token groups; inside a group the
    //      feature tiles of one token tile are consecutive, so the 32 workgroups of the XCD share activation tiles through its L2
    const int gJ = nx / splitI;
    const int hI = (nI + splitI - 1) / splitI, qJ = (nJ + gJ - 1) / gJ;
    const int i0 = (xcd % splitI) * hI, j0 = (xcd / splitI) * qJ;
    const int ni = max(0, min(nI, i0 + hI) - i0), nj = max(0, min(nJ, j0 + qJ) - j0);
    const int count = (int)min((long)ni * nj, (long)lb + (long)i_end * nloc);     // (list positions behind i_end belong to a later call)
    int t = lb + i_begin * nloc;
    if (t >= count) return;
    // a tile's operand rows are addressed from its (wave-uniform) tile indices at issue time: lane constants row0 / c8 + two integer
    // multiply-adds per request instead of eight offset registers per tile in flight (the k-loop runs at the register limit)
    const int row0 = wave * 8 + (lane >> 3);                       // row of this lane's 16-byte piece inside a 64-row segment group
    const int c8 = ((lane & 7) ^ (row0 & 7)) * 8;                  // its XOR-swizzled k chunk (the same for row0 + 64: 64 % 8 == 0)
    // GEMM_VT (V transposed per frame for the spatial attention, [frame][feature][S_pad tokens]): the token tiles are per FRAME -- tpf
    // tiles of 256 for the frame's a.Nj tokens, frame z = tj / tpf -- so that a tile's tokens are contiguous in the output
    auto make_src = [&](int tt) {
        const int ti = i0 + tt % ni, tj = j0 + tt / ni;
        if (MODE == GEMM_VT) {
            const int z = tj / tpf, tl = tj - z * tpf;
            return Src{ti, z * a.Nj + tl * TM, z * a.Nj + a.Nj - 1};
        }
        return Src{ti, tj * TM, a.Nj - 1};
    };
    auto issue = [&](int slot, bool isP, const Src& sr, int h, int k0) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const int row = h * 128 + sg * 64 + row0;
            const unsigned off = isP ? (unsigned)((sr.ti * TM + row) * a.ldp + c8) : (unsigned)(min(sr.trow0 + row, sr.tmax) * a.ldq + c8);
            __builtin_amdgcn_global_load_lds((const void*)((isP ? P : Q) + off + k0),
                                             (__attribute__((address_space(3))) void*)(lds + slot * kSlot + (wave + 8 * sg) * 1024), 16, 0, 0);
        }
    };
    // ring slot of (k-tile parity, kind): kind 0 / 1 = P half 0 / 1, 2 / 3 = Q half 0 / 1
    Src cur = make_src(t);
    // prologue: k-tile 0 entirely, the Q halves of k-tile 1 (its P halves follow in phases 0 / 1 of k-tile 0)
    issue(2, false, cur, 0, 0);
    issue(3, false, cur, 1, 0);
    issue(0, true, cur, 0, 0);
    issue(1, true, cur, 1, 0);
    issue(4 + 2, false, cur, 0, HK);
    issue(4 + 3, false, cur, 1, HK);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // STAGGER: the waves of the second feature half (one per SIMD, like those of the first) run one barrier interval behind: while one
    // group issues its fragment reads and its share of a refill, the other group's 16 MFMAs own the matrix pipe (MI355X playbook: the
    // wave role split is what lets LDS reads, LDS-DMA and MFMAs overlap inside one workgroup).  Both groups pass the same number
    // of barriers: this one here, its counterpart for the first group behind the last tile.
    // The stagger is per OUTPUT TILE: the second group takes its extra barrier at the top of every tile, the first group its counterpart
    // right behind the tile's last MFMA block -- so both groups enter the epilogue together.  (Staggered across tiles -- rounds 3 / 4a --
    // the second group's last phase waited for the first group's whole epilogue and then ran its own while the first group stood at the
    // next tile's first barrier: the two epilogues ran one behind the other, 2 x 2.9 us per tile in STORE mode, 2 x 6.5 with the GELU,
    // 2 x 10.9 with the residual read-modify-write; profiles/r04_gemm_stamps_before.txt.)

    unsigned char* stage = lds + kRing + wave * kStage;
#ifdef UMGEN_G256_STAMPS
    unsigned long long st_main = 0, st_k0 = 0, st_epi = 0, st_n = 0, st_t0 = 0, st_t1 = 0, st_c0 = 0, st_clk = 0;   // st_clk: shader-clock ticks (clock64) of the k-loops
#endif
    while (true) {
        const int tn = t + nloc;
        const bool has_next = tn < count;
        Src nxt = cur;
        if (has_next) nxt = make_src(tn);
        f32x4_t acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (STAGGER && wi == 1) __builtin_amdgcn_s_barrier();
#ifdef UMGEN_G256_STAMPS
        st_t0 = wall_clock64();
        st_c0 = clock64();
#endif
        // one k-tile = 4 phases; FX / FY: the registers of the token fragments of the first / second 32 tokens.  FBALT (build option, off:
        // measured on the shapes of the stacks, profiles/r04_gemm_bench_fbalt.txt -- 4096^3 +3 %, the K = 768 shapes -3 .. +3 %, no net gain): the first-token
        // fragments of k-tile kt + 1 are read in phase 3 of k-tile kt (into FY, free since phase 2) instead of its own phase 0, and the two
        // register sets swap roles every k-tile: 8 / 4 / 8 / 4 fragment reads per phase instead of 12 / 4 / 8 / 0 (the wave group that
        // reads shares the LDS with the LDS-DMA while the other group's 16 MFMAs run: 12 reads are 384 LDS clocks against 256 MFMA clocks).
        // Their slot (Q of k-tile kt + 1) is retired one phase earlier for it, by a counted wait in phase 2.
        auto ktile = [&](const int kt, vec8 (&FX)[2][2], vec8 (&FY)[2][2], auto preloaded_tag, auto preload_tag) {
#ifdef UMGEN_G256_STAMPS
            if (kt == 1) st_k0 += wall_clock64() - st_t0;
#endif
            const int par = kt & 1;
            const unsigned char* sP = lds + (par * 4 + wi) * kSlot;                      // this wave's P half (128 features)
            const unsigned char* sQ = lds + (par * 4 + 2 + (wj >> 1)) * kSlot + (wj & 1) * 64 * 128;   // its 64 tokens inside a Q half
            // refills of this k-tile: P halves of k-tile kt + 1 (phases 0, 1), Q halves of k-tile kt + 2 (phases 2, 3)
            const bool in1 = kt + 1 < nkt, in2 = kt + 2 < nkt;
            const bool do1 = in1 || has_next, do2 = in2 || has_next;
            const Src& s1 = in1 ? cur : nxt;
            const Src& s2 = in2 ? cur : nxt;
            const int k1 = (in1 ? kt + 1 : kt + 1 - nkt) * HK, k2 = (in2 ? kt + 2 : kt + 2 - nkt) * HK;
            const int par1 = par ^ 1;
            // (k-tile 0 of an output tile neither finds its fragments preloaded nor preloads: its phase-2 wait would sit right behind the
            //  previous tile's stores, which vmcnt counts too)
            constexpr bool preloaded = FBALT && decltype(preloaded_tag)::value;
            const bool preload = FBALT && decltype(preload_tag)::value && in1;
            vec8 fa[4][2];
            // ---------------- phase 0: quadrant (features 0..63, tokens 0..31) ----------------
            if (!preloaded) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) FX[n][kk] = *reinterpret_cast<const vec8*>(sQ + swz(n * 16 + frow, kk * 4 + g));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fa[m][kk] = *reinterpret_cast<const vec8*>(sP + swz(m * 16 + frow, kk * 4 + g));
#if UMGEN_G256_PH2
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) FY[n][kk] = *reinterpret_cast<const vec8*>(sQ + swz((2 + n) * 16 + frow, kk * 4 + g));
            if (do1) issue(par1 * 4 + 1, true, s1, 1, k1);
#endif
            if (do1) issue(par1 * 4 + 0, true, s1, 0, k1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired BEFORE the barrier: the other wave group's next refill may target this slot
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = Mma16<TT>::mfma(fa[m][kk], FX[n][kk], acc[m][n]);
#if !UMGEN_G256_PH2
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 1: quadrant (features 0..63, tokens 32..63) ----------------
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) FY[n][kk] = *reinterpret_cast<const vec8*>(sQ + swz((2 + n) * 16 + frow, kk * 4 + g));
            if (do1) issue(par1 * 4 + 1, true, s1, 1, k1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired BEFORE the barrier: the other wave group's next refill may target this slot
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][2 + n] = Mma16<TT>::mfma(fa[m][kk], FY[n][kk], acc[m][2 + n]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 2: quadrant (features 64..127, tokens 32..63) ----------------
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fa[m][kk] = *reinterpret_cast<const vec8*>(sP + swz((4 + m) * 16 + frow, kk * 4 + g));
            if (do2) issue(par * 4 + 2, false, s2, 0, k2);       // (the Q slots of this k-tile: their last reads were phase 1's)
#if UMGEN_G256_PH2
            if (do2) {
                issue(par * 4 + 3, false, s2, 1, k2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#endif
            if (preload) {     // the Q halves of k-tile kt + 1 (requested a k-tile ago) have landed: everything but this k-tile's 4 + 2 requests
                if (do2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired BEFORE the barrier: the other wave group's next refill may target this slot
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[4 + m][2 + n] = Mma16<TT>::mfma(fa[m][kk], FY[n][kk], acc[4 + m][2 + n]);
#if !UMGEN_G256_PH2
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 3: quadrant (features 64..127, tokens 0..31); the k-tile's one counted wait ----------------
            if (preload) {     // first-token fragments of k-tile kt + 1 into the registers phase 2 was the last to use
                const unsigned char* sQn = lds + (par1 * 4 + 2 + (wj >> 1)) * kSlot + (wj & 1) * 64 * 128;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) FY[n][kk] = *reinterpret_cast<const vec8*>(sQn + swz(n * 16 + frow, kk * 4 + g));
            }
            if (do2) {
                issue(par * 4 + 3, false, s2, 1, k2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all of k-tile kt + 1 has landed (only this k-tile's two Q refills may be in flight)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (FBALT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[4 + m][n] = Mma16<TT>::mfma(fa[m][kk], FX[n][kk], acc[4 + m][n]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        };
        vec8 fbA[2][2], fbB[2][2];
        ktile(0, fbA, fbB, std::false_type{}, std::false_type{});      // (nkt is even and >= 2: the launcher)
        ktile(1, fbB, fbA, std::false_type{}, std::true_type{});
        for (int kt = 2; kt < nkt; kt += 2) {
            ktile(kt, fbA, fbB, std::true_type{}, std::true_type{});
            ktile(kt + 1, fbB, fbA, std::true_type{}, std::true_type{});
        }
        if (STAGGER && wi == 0) __builtin_amdgcn_s_barrier();
#ifdef UMGEN_G256_STAMPS
        st_t1 = wall_clock64();
        st_main += st_t1 - st_t0;
        st_clk += clock64() - st_c0;
#endif
        // ---------------- epilogue of this wave's 128 features x 64 tokens (private LDS strip, no workgroup barrier) ----------------
        const int ti = i0 + t % ni, tj = j0 + t / ni;
        const int fbase = ti * TM + wi * 128;            // first feature of the wave
        const int tbase = tj * TM + wj * 64;             // first token of the wave
        // bias of this lane's 32 features (the fragment registers are free now); one wait for all eight loads
        if (MODE == GEMM_STORE) {
            float bv[8][4];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (a.bias) load4(a.bias + fbase + m * 16 + 4 * g, bv[m]);
                else { bv[m][0] = 0.f; bv[m][1] = 0.f; bv[m][2] = 0.f; bv[m][3] = 0.f; }
            }
            TT* out = reinterpret_cast<TT*>(a.out);
            auto run = [&](auto gelu_tag) {
                constexpr bool GELU = decltype(gelu_tag)::value;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[m][n][r] + bv[m][r];
                            o[r] = GELU ? gelu_fast(v) : v;
                        }
                        store4(reinterpret_cast<TT*>(stage + frow * 256 + (((m * 4 + g) ^ frow) << 3)), o);   // 8-byte granule p of token t at p ^ t
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int tl = it * 4 + (lane >> 4), q = lane & 15;
                        const int pos = ((2 * q) ^ tl) & ~1;
                        uint4 v = *reinterpret_cast<const uint4*>(stage + tl * 256 + pos * 8);
                        if (tl & 1) v = make_uint4(v.z, v.w, v.x, v.y);
                        const int token = tbase + n * 16 + tl;
                        if (token < a.Nj && (EPI != 1 || a.ldo < 0)) {
                            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                            if (EPI == 2) *reinterpret_cast<u32x4*>(out + (long)token * a.ldo + fbase + 8 * q) = u32x4{v.x, v.y, v.z, v.w};
                            else __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(out + (long)token * a.ldo + fbase + 8 * q));
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // strip reads done before the next n overwrites it
                }
            };
            if (a.gelu) run(std::true_type{}); else run(std::false_type{});
        } else if (MODE == GEMM_VT) {
            // V transposed: out[(z * Mi + feature) * ldo + token] -- the lanes of one accumulator register hold 16 consecutive tokens of
            // one feature already; 16 features x 64 tokens go through the strip (row pitch 144 B: the four feature groups of a request
            // fall into different banks) and leave as 16-byte pieces of 8 tokens.  Tokens behind the frame's last one are written as
            // zeros (the pad columns of a V^T row must be zero for the P.V product), pieces behind the padded row are skipped.
            TT* out = reinterpret_cast<TT*>(a.out);
            const int z = tj / tpf, tl = tj - z * tpf;
            const int tok0 = tl * TM + wj * 64;
            constexpr int PITCH = 144;
            float bv[8][4];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (a.bias) load4(a.bias + fbase + m * 16 + 4 * g, bv[m]);
                else { bv[m][0] = 0.f; bv[m][1] = 0.f; bv[m][2] = 0.f; bv[m][3] = 0.f; }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<TT*>(stage + (4 * g + r) * PITCH + (n * 16 + frow) * 2) = Cvt<TT>::from_f(acc[m][n][r] + bv[m][r]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int piece = it * 64 + lane, f = piece >> 3, pc = piece & 7;
                    uint4 v = *reinterpret_cast<const uint4*>(stage + f * PITCH + pc * 16);
                    const int tk = tok0 + pc * 8, valid = a.Nj - tk;      // tokens of this piece inside the frame
                    if (valid < 8) {
                        v.x &= (valid > 0 ? 0xffffu : 0u) | (valid > 1 ? 0xffff0000u : 0u);
                        v.y &= (valid > 2 ? 0xffffu : 0u) | (valid > 3 ? 0xffff0000u : 0u);
                        v.z &= (valid > 4 ? 0xffffu : 0u) | (valid > 5 ? 0xffff0000u : 0u);
                        v.w &= (valid > 6 ? 0xffffu : 0u) | (valid > 7 ? 0xffff0000u : 0u);
                    }
                    if (tk < a.ldo) {
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w},
                                                    reinterpret_cast<u32x4*>(out + ((long)z * a.Mi + fbase + m * 16 + f) * a.ldo + tk));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // strip reads done before the next m overwrites it
            }
        } else {   // GEMM_RESID (x += acc + bias) / GEMM_STORE_F32: fp32 rows, 64 features per pass
            float* out = reinterpret_cast<float*>(a.out);
            // bias of the 4 features this lane holds BEHIND the strip transposition (the same acc + bias, added there: 8 registers instead of 32)
            float4 b4[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                b4[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + fbase + hf * 64 + 4 * (lane & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
            // the read-modify-write's loads run TWO 16-token passes ahead of their use (the fragment registers are free now): the wave
            // waits for a memory round trip twice per tile instead of four times
            float4 xob[2][2][4];
            auto load_xo = [&](int n, float4 (&dst)[2][4]) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int token = min(tbase + n * 16 + it * 4 + (lane >> 4), a.Nj - 1);
                        dst[hf][it] = *reinterpret_cast<const float4*>(out + (long)token * a.ldo + fbase + hf * 64 + 4 * (lane & 15));
                    }
            };
            if (MODE == GEMM_RESID && EPI != 3) { load_xo(0, xob[0]); load_xo(1, xob[1]); }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float4 (&xo)[2][4] = xob[n & 1];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        const int m = hf * 4 + mm;
                        const float4 o = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
                        *reinterpret_cast<float4*>(stage + frow * 256 + (((mm * 4 + g) ^ frow) << 4)) = o;   // 16-byte granule p of token t at p ^ t
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int tl = it * 4 + (lane >> 4), q = lane & 15;
                        float4 v = *reinterpret_cast<const float4*>(stage + tl * 256 + ((q ^ tl) << 4));
                        v = make_float4(v.x + b4[hf].x, v.y + b4[hf].y, v.z + b4[hf].z, v.w + b4[hf].w);
                        const int token = tbase + n * 16 + tl;
                        if (token < a.Nj && (EPI != 1 || a.ldo < 0)) {
                            float* x = out + (long)token * a.ldo + fbase + hf * 64 + 4 * q;
                            if (MODE == GEMM_RESID && EPI != 3) {
                                const float4 c = xo[hf][it];
                                *reinterpret_cast<float4*>(x) = make_float4(c.x + v.x, c.y + v.y, c.z + v.z, c.w + v.w);   // x + (acc + bias), as every other residual epilogue
                            } else {
                                *reinterpret_cast<float4*>(x) = v;
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (MODE == GEMM_RESID && EPI != 3 && n + 2 < 4) load_xo(n + 2, xob[n & 1]);
            }
        }
#ifdef UMGEN_G256_STAMPS
        st_epi += wall_clock64() - st_t1;
        st_n += 1;
#endif
        if (!has_next) break;
        t = tn;
        cur = nxt;
    }
#ifdef UMGEN_G256_STAMPS
    if (xcd == 1 && lb == 1 && lane == 0 && (wave == 0 || wave == 4)) {
        atomicAdd(&g256_stamps[wi * 8 + 0], st_main);
        atomicAdd(&g256_stamps[wi * 8 + 1], st_k0);
        atomicAdd(&g256_stamps[wi * 8 + 2], st_epi);
        atomicAdd(&g256_stamps[wi * 8 + 3], st_n);
        atomicAdd(&g256_stamps[wi * 8 + 4], st_clk);
    }
#endif
}

}  // namespace

}  // namespace umgen
