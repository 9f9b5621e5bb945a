// Op list of the decode engine's background workers (bg_worker.h): one entry per kernel launch of the next frame's TAR / ego pass, built on the host
// by recording the pass's launches (engine.hip BgRecorder) instead of enqueueing them, executed by the engine workgroups of the idle XCDs.
#pragma once
#include <vector>

#include "frame.h"
#include "kernels.h"

namespace umgen {

enum BgKind { BG_NONE = 0, BG_GEMM = 1, BG_LN = 2, BG_ATTN_S = 3, BG_ATTN_T = 4, BG_EMBED = 5, BG_WARP = 6 };

// One virtual launch.  Plain words so that a worker can pull the entry into scalar registers (bg_worker.h load_uniform); what i0.. / l0.. / p0.. mean per
// kind is written next to the kind's case in bg_worker.h bg_run and next to its recorder in engine.hip.
struct BgOpHead {
    int kind, mode;
    int n_units;          // units of the op (GEMM: the tile lists' lengths follow from i0..i2)
    int chunk;            // units a worker runs between two looks at the clock
    int i0, i1, i2, i3;
};
struct BgOpArgs {
    int i4, i5, i6, i7;
    long l0, l1;
    void *p0, *p1, *p2, *p3, *p4, *p5;
};
struct BgOp {
    BgOpHead h;
    BgOpArgs a;           // every kind but BG_GEMM
    GemmArgs g;           // BG_GEMM (h.i0..i3 = nI, nJ, splitI, tpf of gemm16_256_body)
};
static_assert(sizeof(BgOp) % 8 == 0, "BgOp is copied word by word");

constexpr int kBgMaxOps = 4096;
constexpr int kBgMaxWorkers = 128;      // 4 XCDs x 32 workgroups
struct BgQueue {
    unsigned n_ops;                          // ops of the pass in flight (0: none)
    unsigned engine_ticks;                   // duration of the previous launch's engine part (100 MHz), written by the engine; 0 = unknown (workers sit the launch out)
    unsigned margin_ticks;                   // the workers stop this long before the expected end of the launch
    unsigned ops_done;                       // ops every worker has finished (stored by the last arrival of each op)
    EmbedTables tb;
    unsigned state[kBgMaxWorkers][4];        // per worker: op index; units of it done | arrived << 31; start (100 MHz, low word) of the batch of units in flight; -
    unsigned arrive[kBgMaxOps];              // workers that have published their share of op k
    unsigned est[kBgMaxOps];                 // ticks per unit of op k: host guess, then tracked by the workers (9/8 of a batch's time per unit: half-way up at once, 1/64 of the way down per batch)
    BgOp ops[kBgMaxOps];
};


// Host side: while a recorder is installed (engine.hip launch_prefix, one thread), the launchers of the pass's kernels -- launch_layernorm,
// launch_gemm_mfma, launch_attn_spatial_mfma, launch_attn_temporal, launch_embed_stack, launch_warp_map -- append an op instead of launching.
struct BgRecorder {
    std::vector<BgOp> ops;
    std::vector<unsigned> est;       // host guess of the ticks (100 MHz) one unit takes on one CU
    const char* failed = nullptr;    // a launch the workers cannot run (shape outside the 256-tile kernel, another kernel variant, ...)
    BgOp& add(int kind, int mode, long n_units, int chunk, unsigned est_ticks) {
        ops.emplace_back();
        BgOp& o = ops.back();
        o = BgOp{};
        o.h.kind = kind; o.h.mode = mode; o.h.n_units = (int)n_units; o.h.chunk = chunk;
        est.push_back(est_ticks);
        return o;
    }
};
extern thread_local BgRecorder* g_bg_rec;      // engine.hip

}  // namespace umgen
