// Background workers of the XCD-resident decode engine (oar_engine.hip, BG instantiation; round 6).
//
// At one scene per GPU the decode engine computes on ONE XCD at a time; on 4 XCD groups instead of 8 a step costs the same (441.3 vs 441.2 us,
// profiles/r06_engine_contention.txt) and neither a matrix-core nor a streaming load on the other four XCDs slows it.  HIP cannot keep a second
// kernel's workgroups off the engine's XCDs (block b lands on XCD b % 8 whatever the CU mask is, profiles/r03_cumask_probe.txt), and an engine
// workgroup owns its CU (512 threads, 252 VGPRs, 159 KB of LDS), so the next frame's TAR / ego pass over the history slots that are already known
// (UMGen.py:1484-1494; slots 0 .. T - 2 of the next window, engine.hip launch_prefix) runs INSIDE the engine's launches: the engine workgroups that
// landed on the four idle XCDs become WORKERS and execute the pass's kernels -- the same device functions the stand-alone kernels are made of
// (gemm256_body.h, attn_body.h, rowops_body.h, frame_body.h), so every output bit is the one the stand-alone launch produces -- as "virtual
// launches" out of an op list in HBM:
//   * an op = one kernel launch of the pass; its units (output tiles, attention blocks, rows) are dealt statically: worker w takes units
//     w, w + n_workers, ... (GEMM: positions of its XCD's tile list, so the 32 workers of an XCD share operand tiles through its L2 as the 32
//     workgroups of a stand-alone launch do);
//   * ops run one behind the other: a worker that has finished its units of op k publishes them (agent-scope release: the XCDs' L2s are not
//     coherent with each other) and arrives on arrive[k]; the last arrival publishes ops_done = k + 1, everybody starts op k + 1 when it sees that (agent-scope acquire);
//   * the pass spans ~2200 decode steps: a worker stops when the next batch of units would not fit before the expected end of this launch (the
//     duration of the previous launch's engine part minus a margin; unit times are refined on the device), stores (op, units done, arrived)
//     and picks up there in the next step's launch.  What is left behind the frame's last step is drained by a launch without an engine part.
#pragma once
#include "attn_body.h"
#include "frame_body.h"
#include "gemm256_body.h"
#include "rowops_body.h"

namespace umgen {

namespace {

// uniform (scalar-register) copy of a structure another launch or the host has written: per-lane loads of the same address, then one lane's value
template <typename S>
__device__ __forceinline__ S load_uniform(const S* p) {
    static_assert(sizeof(S) % 4 == 0, "load_uniform");
    S out;
    const unsigned* src = reinterpret_cast<const unsigned*>(p);
    unsigned* dst = reinterpret_cast<unsigned*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(S) / 4); ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return out;
}

// number of tiles on list `xcd` of a 256-tile GEMM with nx lists (the partition of gemm16_256_body)
__device__ __forceinline__ int gemm256_list_count(int nI, int nJ, int splitI, int nx, int xcd) {
    const int gJ = nx / splitI;
    const int hI = (nI + splitI - 1) / splitI, qJ = (nJ + gJ - 1) / gJ;
    const int i0 = (xcd % splitI) * hI, j0 = (xcd / splitI) * qJ;
    const int ni = max(0, min(nI, i0 + hI) - i0), nj = max(0, min(nJ, j0 + qJ) - j0);
    return ni * nj;
}

// The GEMM body as a FUNCTION of its own (not inlined into the engine kernel): it runs at the register limit (256 VGPRs, 97 SGPRs), and inlined next to
// the worker loop's live values the allocator spilled ~190 scalar and ~240 vector registers around and inside its k-loop.
template <int MODE, typename TT>
__device__ __noinline__ void bg_gemm(const GemmArgs* gp, int nI, int nJ, int splitI, int tpf, int xcd, int nx, int lb, int i_begin, int i_end) {
    typedef const UMGEN_AS4 GemmArgs& ArgsRef;
    ArgsRef g = *reinterpret_cast<const UMGEN_AS4 GemmArgs*>(reinterpret_cast<unsigned long long>(gp));
    gemm16_256_body<MODE, TT, ArgsRef>(g, nI, nJ, splitI, tpf, xcd, nx, lb, kEngGroup, i_begin, i_end);
}

// Every kind's units [done, done + n) of this worker's share as a function of its own as well: the worker loop stays a few dozen instructions, and the
// engine part of the kernel keeps the register allocation it has without the workers.
template <typename TT>
__device__ __noinline__ void bg_ln(const BgOp* op, int E, int widx, int nwk, int done, int n) {      // 8 rows per unit; l0 = row stride, l1 = rows
    const BgOpArgs o = load_uniform(&op->a);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int u = done;
    if (E <= 768) {      // four units at a time: a wave's four rows with all their loads in flight (the same arithmetic per row)
        for (; u + 4 <= done + n; u += 4) {
            long rows[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rows[r] = ((long)widx + (long)(u + r) * nwk) * 8 + wave;
            layernorm_rows<TT, 4, 12>(reinterpret_cast<const float*>(o.p0), o.l0, E, reinterpret_cast<const float*>(o.p1), reinterpret_cast<TT*>(o.p2), rows, o.l1, lane);
        }
    }
    for (; u < done + n; ++u) {
        const long row = ((long)widx + (long)u * nwk) * 8 + wave;
        if (row < o.l1) layernorm_row<TT>(reinterpret_cast<const float*>(o.p0), o.l0, E, reinterpret_cast<const float*>(o.p1), reinterpret_cast<TT*>(o.p2), row, lane);
    }
}
template <typename TT>
__device__ __noinline__ void bg_attn_s(const BgOp* op, int S, int S_pad, int H, int nq, int widx, int nwk, int done, int n) {   // i4 = npairs, i5 = blocks of the virtual grid; a unit = two blocks
    const BgOpArgs o = load_uniform(&op->a);
    const int tid = threadIdx.x, half = tid >> 8;
    for (int u = done; u < done + n; ++u) {
        const int vb = 2 * (widx + u * nwk) + half;
        const bool valid = vb < o.i5;
        attn_spatial_mfma_body<UMGEN_ATTN_QT, TT, false, true>(reinterpret_cast<const TT*>(o.p0), reinterpret_cast<const TT*>(o.p1), reinterpret_cast<TT*>(o.p2), S, S_pad, H, nq,
                                                                o.i4, valid ? vb : 0, tid & 255, half * 2 * kTileBytes, valid);
    }
}
template <typename TT, int TMAX>
__device__ __noinline__ void bg_attn_t(const BgOp* op, int Tn, int S, int H, int widx, int nwk, int done, int n) {   // i4.. = TemporalRange {t0, Tcap, write, q0}, p2 = cache; a unit = one block
    const BgOpArgs o = load_uniform(&op->a);
    const int tid = threadIdx.x;
    TemporalRange tr;
    tr.t0 = o.i4; tr.cache = o.p2; tr.Tcap = o.i5; tr.write = o.i6; tr.q0 = o.i7;
    for (int u = done; u < done + n; ++u) {
        attn_temporal_body<TT, 4, TMAX>(reinterpret_cast<const TT*>(o.p0), reinterpret_cast<TT*>(o.p1), Tn, S, H, tr, (long)widx + (long)u * nwk, tid);
        __syncthreads();      // the next block's rows land in the same LDS
    }
}
// BG_EMBED: i0 = stack, i1 = B, i2 = T, i3 = Tfull, i4 = t0; window tokens p0..p3, p4 = X, p5 = mapfeat; l1 = rows; 4 rows per unit
// BG_WARP:  i0 = stack, i1 = T, i2 = Tfull, i3 = t0; p0 = mapfeat, p1 = pose_diff, p2 = X, p3 = warped_last; l1 = cells; 4 cells per unit
__device__ __noinline__ void bg_embed_warp(const BgOpHead h, const BgOp* op, const BgQueue* q, int widx, int nwk, int done, int n) {
    const BgOpArgs o = load_uniform(&op->a);
    const EmbedTables tb = load_uniform(&q->tb);
    const int tid = threadIdx.x;
    for (int u = done; u < done + n; ++u) {
        const long r = ((long)widx + (long)u * nwk) * 4 + (tid >> 7);
        if (r >= o.l1) continue;
        if (h.kind == BG_EMBED) {
            WindowTokens w{reinterpret_cast<const int*>(o.p0), reinterpret_cast<const int*>(o.p1), reinterpret_cast<const int*>(o.p2), reinterpret_cast<const int*>(o.p3), h.i1, h.i2, h.i3, o.i4};
            embed_stack_row(h.i0, tb, w, reinterpret_cast<float*>(o.p4), reinterpret_cast<float*>(o.p5), r, tid & 127);
        } else {
            warp_map_cell(h.i0, tb, h.i1, reinterpret_cast<const float*>(o.p0), reinterpret_cast<const float*>(o.p1), reinterpret_cast<float*>(o.p2), reinterpret_cast<float*>(o.p3), h.i2,
                          h.i3, r, tid & 127);
        }
    }
}

// units [done, done + n) of this worker's share of op *op (h: its header words)
template <typename TT>
__device__ __forceinline__ void bg_run(const BgOpHead h, const BgOp* op, const BgQueue* q, int widx, int nwk, int done, int n) {
    switch (h.kind) {
        case BG_GEMM: {      // h.i0..i3 = nI, nJ, splitI, tpf
            const int xcd = widx / kEngGroup, lb = widx % kEngGroup, nx = nwk / kEngGroup;
            if (h.mode == GEMM_STORE) bg_gemm<GEMM_STORE, TT>(&op->g, h.i0, h.i1, h.i2, h.i3, xcd, nx, lb, done, done + n);
            else if (h.mode == GEMM_RESID) bg_gemm<GEMM_RESID, TT>(&op->g, h.i0, h.i1, h.i2, h.i3, xcd, nx, lb, done, done + n);
            else bg_gemm<GEMM_VT, TT>(&op->g, h.i0, h.i1, h.i2, h.i3, xcd, nx, lb, done, done + n);
            break;
        }
        case BG_LN: bg_ln<TT>(op, h.i0, widx, nwk, done, n); break;                                        // i0 = E
        case BG_ATTN_S: bg_attn_s<TT>(op, h.i0, h.i1, h.i2, h.i3, widx, nwk, done, n); break;            // i0..i3 = S, S_pad, H, nq
        case BG_ATTN_T:                                                                                    // i0..i2 = Tn, S, H; mode = query slots per head
            if (h.mode == 20) bg_attn_t<TT, 20>(op, h.i0, h.i1, h.i2, widx, nwk, done, n);
            else bg_attn_t<TT, 32>(op, h.i0, h.i1, h.i2, widx, nwk, done, n);
            break;
        case BG_EMBED:
        case BG_WARP: bg_embed_warp(h, op, q, widx, nwk, done, n); break;
        default: break;
    }
}

// worker `widx` of `nwk` (a multiple of 32: whole XCDs); t_k0 = kernel entry (100 MHz); drain: no engine part in this launch, run to the end.
// The worker's progress -- (op, units of it done | arrived << 31) -- LIVES in q->state[widx] (agent-scope accesses: L1 bypassed) and is read back at the
// top of every round: nothing but the queue pointer and two integers is live across a kernel body (the GEMM body alone needs 97 scalar registers;
// with the round's state in registers next to it the compiler spilled 220 of them into vector lanes and 280 VGPRs to scratch memory).
template <typename TT>
__device__ __forceinline__ void bg_worker(BgQueue* q, int widx, int nwk, unsigned long long t_k0, bool drain) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char wk_lds[];
    volatile int* ctl = reinterpret_cast<volatile int*>(wk_lds + kRing);      // (a GEMM epilogue strip: free between the bodies' calls)
    const int tid = threadIdx.x;
    if (nwk > kBgMaxWorkers) return;
    unsigned* const st = &q->state[widx][0];
    auto ld = [](const unsigned* p) { return (unsigned)__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
    auto stw = [](unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // Leaving the launch with units done in it: write this XCD's dirty L2 lines back now, under the engine part's last microseconds, instead of at the end of the kernel
    // (measured: no difference in the step time -- the engine launch is longer while workers run because they share HBM and the fabric with it, not because of this write-back).
    bool worked = false;
    auto leave = [&]() {
        if (!worked) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    for (;;) {
        const unsigned n_ops = ld(&q->n_ops), op = ld(st);
        if (op >= n_ops) { leave(); return; }
        const unsigned st1 = ld(st + 1);
        const int done = (int)(st1 & 0x7fffffffu);
        const bool arrived = (st1 >> 31) != 0u;
        const unsigned eng = ld(&q->engine_ticks), margin = ld(&q->margin_ticks);
        const unsigned long long deadline = drain ? ~0ull : t_k0 + (unsigned long long)(eng > margin ? eng - margin : 0u);
        const BgOpHead h = load_uniform(&q->ops[op].h);
        int mine;
        if (h.kind == BG_GEMM) {
            const int cnt = gemm256_list_count(h.i0, h.i1, h.i2, nwk / kEngGroup, widx / kEngGroup), lb = widx % kEngGroup;
            mine = lb < cnt ? (cnt - lb + kEngGroup - 1) / kEngGroup : 0;
        } else {
            mine = widx < h.n_units ? (h.n_units - widx + nwk - 1) / nwk : 0;
        }
        __syncthreads();      // every wave has read the round's state: from here on thread 0 may rewrite it
        if (done < mine) {
            // how many units fit before the launch is expected to end (one thread reads the clock; everybody takes its answer)
            if (tid == 0) {
                const unsigned long long t0 = wall_clock64();
                const unsigned est = max(1u, __hip_atomic_load(&q->est[op], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                long fit = drain ? (long)h.chunk : (deadline > t0 ? (long)((deadline - t0) / est) : 0);
                // progress whatever the estimate says: a workgroup that stands at the very start of a launch with the whole budget in front of it runs ONE unit even if
                // its estimate does not fit (an estimate that one slow batch has pushed beyond the budget would otherwise park the whole pass in front of this op until the drain)
                if (fit == 0 && !worked && deadline > t0 && t0 - t_k0 < 2000ull) fit = 1;
                const int n0 = (int)min((long)min(mine - done, h.chunk), fit);
                ctl[0] = n0;
                if (n0 > 0) {      // (the units count as done from here on: nobody but this workgroup reads the word before its arrival)
                    stw(st + 1, (unsigned)(done + n0));
                    stw(st + 2, (unsigned)t0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            const int n = ctl[0];
            __syncthreads();
            if (n <= 0) { leave(); return; }
            worked = true;
            bg_run<TT>(h, &q->ops[op], q, widx, nwk, done, n);
            if (tid == 0) {      // per-unit time of this op, as the next frame's pass will see it (ops keep their index from frame to frame): up fast, down slowly --
                // the estimate follows the slow tail of the ~128 batches an op is run in per pass and forgets an outlier within a pass or two (as a running MAXIMUM it crept up
                // from pass to pass -- GEMM 460 -> 497 ms per worker over four passes -- and one slow batch could park an op for good); races between workers lose an update
                const unsigned dt = ((unsigned)wall_clock64() - ld(st + 2)) / (unsigned)n;
                const unsigned mine_est = dt + (dt >> 3) + 20u;
                const unsigned old = __hip_atomic_load(&q->est[op], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned upd = mine_est > old ? old + ((mine_est - old + 1u) >> 1) : old - ((old - mine_est) >> 6);
                __hip_atomic_store(&q->est[op], upd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        if (!arrived) {
            // publish this workgroup's share of the op: every wave's stores acknowledged, one agent-scope release, then the arrival
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // the LAST arrival moves the pass on with a write-through STORE of the op count (the hand-off every poll of the engine relies on: sc1 store ->
                // sc1 load); the arrivals themselves are read-modify-writes whose returned value is exact wherever the line lives -- polled with loads,
                // a counter that other XCDs increment was seen stale by one XCD's workers for good (a drain that gave up at op 575 of 2215)
                const unsigned before = __hip_atomic_fetch_add(&q->arrive[op], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1u == (unsigned)nwk) stw(&q->ops_done, op + 1u);
                stw(st + 1, (unsigned)done | 0x80000000u);
            }
        }
        if (tid == 0) {
            int ok = 1;
            for (unsigned spins = 0;; ++spins) {
                if (__hip_atomic_load(&q->ops_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > op) break;
                if ((!drain && wall_clock64() >= deadline) || spins > (1u << 23)) { ok = 0; break; }      // (drain: bounded like every poll of the engine)
                __builtin_amdgcn_s_sleep(8);
            }
            if (ok) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                stw(st + 1, 0u);
                stw(st, op + 1u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ctl[1] = ok;
        }
        __syncthreads();
        const int ok = ctl[1];
        __syncthreads();
        if (!ok) { leave(); return; }
    }
}

}  // namespace

}  // namespace umgen
