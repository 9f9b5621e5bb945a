#!/bin/bash
# GPU-box sessions of this round, ONE parameterised script (round 3 left 20 one-off tools/r3_gpu_*.sh behind):
#     gpurun --timeout N -- 'bash tools/gpu_session.sh <session> [<session> ...]'
# Every session writes under gpurun_out/ (merged back by gpurun); what is quoted in DESIGN.md is copied into profiles/ by hand.
#   tests      the whole -m gpu suite                                                   -> r04_pytest_gpu.log
#   smoke      __graft_entry__.smoke()
#   bench      the default bench line (30 frames, CPU-baseline sample)                  -> r04_bench_full.json
#   variants   fp16 / fp32 / batches / control / five-launch layer / 2x width lines       -> r04_bench_<name>.json
#   batched    16 / 32 / 64 scenes per GPU on the batched decode layer, 32 on the engine  -> r04_bench_b{16,32,64,32_engine}.json
#   lanes      decode-lane sweep of the batched layer at 16 / 32 / 64 scenes                  -> r04_bench_b{16,32,64}_lanes<n>.json
#   control    configs[2] (control, 4 scenes, window 13 -> 20) with and without the slot-cache reuse (f-3) -> r04_bench_control_b4{,_nogrow}.json
#   fp32ab     fp32 parity mode on the matrix cores vs the VALU kernel                   -> r04_bench_fp32{,_valu}.json
#   stats      rocprofv3 --kernel-trace --stats of bench.py --steps 3                    -> r04_rocprofv3_kernel_stats_bench_steps3.csv
#   pmc        rocprofv3 --pmc FETCH_SIZE of the decode engine at the mean KV length     -> r04_pmc_fetch_size_engine.csv
#   pmcgemm    SQ / TA / TCC counters of the fc GEMM and the spatial attention          -> r04_pmc_gemm_*.txt, r04_pmc_attn_*.txt
#   gemm       isolated GEMM shapes (tools/gemm_bench.py)                                -> r04_gemm_bench.txt
#   gemmexp    where a 256-tile's time goes: tools/gemm_stamps.py on the stamps build + gemm_bench on the epi1/2/3 builds (tools/build_variant.sh) -> r04_gemm_stamps.txt, r04_gemm_bench_epi*.txt
#   gemmab     tools/gemm_bench.py with the shipped library and with measurement builds GEMM_VARIANTS="..."   -> r04_gemm_bench{,_<variant>}.txt
#   mintiles   one-scene bench by the tile count from which a launch takes the 256-tile kernel              -> r04_bench_mintiles<n>.json
#   stamps     per-phase microseconds of an engine item                                  -> r04_engine_stamps.txt
#   closed     30-frame greedy closed loop, 16-bit modes vs fp32 mode                    -> r04_closed_loop.json
#   vq         umgen_vq_decode of the two production decoders, 20 frames              -> r04_vq_decode_time{,_valu}.json
#   cpubase    the oracle over one whole UMGen_Large frame on this host (32 threads)     -> r04_cpu_baseline_full.json
#   ab:<name>  bench.py --steps 3 with UMGEN_LIB_PATH=umgen_amd/libumgen_hip_<name>.so (tools/build_variant.sh) next to the shipped library
#   wide       configs[4] (2x width) on the chip-wide engine, T = 20 and T = 40          -> r05_bench_wide2x{,_h40}.json
#   widestep   tools/wide_step_time.py: one 2x-width decode step at 64 / 1100 / 2200 keys, shipped library + measurement builds WIDE_VARIANTS="..." -> r05_wide_step_time.txt
#   widevar    the 2x-width bench with per-phase stamps of the chip-wide engine, shipped + WIDE_VARIANTS        -> r05_bench_widevar_<v>.{json,err}
#   wideprof / widepmc   rocprofv3 kernel statistics / FETCH_SIZE of the chip-wide engine  -> r05_prof_wide2x.csv, r05_pmc_fetch_size_wide_engine.csv
#   contention one-scene engine on 4 XCD groups, and under a synthetic load on the other four (measurement build: tools/build_variant.sh burn -DUMGEN_ENG_BURN=1) -> r06_engine_contention.txt
#   pmcb       the FETCH_SIZE pass at 4 / 8 scenes per GPU                               -> r06_pmc_fetch_size_engine_b{4,8}.csv
#   (round 6) tools/dbg/bg_check.py: background workers vs the foreground engine, token for token; tools/dbg/trace_gaps.sh + trace_gaps.py: per-frame kernel durations out of a kernel trace
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
R=r06
line() {  # one-line summary of a bench JSON
python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    p = d["phases_ms_per_frame"]
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), "scene-tokens/s", round(d["ms_per_step"], 1), "ms/frame | layer kernel(s)", round(d["roofline"]["avg_launch_us"], 1),
          "us frac", round(d["roofline"]["frac"], 4), "| step", round(d["roofline"]["step"]["avg_step_us"], 1), "us | gemm", round(d["roofline_gemm"]["achieved"]), "attn", round(d["roofline_attn"]["achieved"]),
          "TFLOP/s | ego/tar/oar ms", round(p["ego"], 1), round(p["tar"], 1), round(p["oar"], 1), "reuse", p.get("frames_reusing_slot_caches"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
b() { name=$1; shift; "$@" > gpurun_out/${R}_bench_$name.json 2> gpurun_out/${R}_bench_$name.err; line gpurun_out/${R}_bench_$name.json; }
for s in "$@"; do
echo "=== session $s"
case $s in
tests) timeout 2400 python -m pytest tests -x -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} --durations=8 -rP > gpurun_out/${R}_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/${R}_pytest_gpu.log | tail -5; grep -E "^(UMGen_Large|full_width|deep|tiny|batched|engine vs)" gpurun_out/${R}_pytest_gpu.log | cut -c1-400 | tail -30 ;;
smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
bench) b full python bench.py ;;
quick) b quick python bench.py --steps 5 --warmup 1 --no-cpu-baseline ;;
variants)
  b fp16 python bench.py --steps 10 --warmup 1 --no-cpu-baseline --precision fp16
  b b4 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 4
  b b8 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 8
  b b16 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16
  b b32 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32
  b launches_plain env UMGEN_DECODE_ENGINE=0 UMGEN_OVERLAP=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  b wide2x python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config wide2x
  b wide2x_h40 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config wide2x --history 40 ;;
batched)
  b b16 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16
  b b32 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32
  b b64 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 64
  b b16_engine env UMGEN_DECODE_BATCHED=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 16
  b b32_engine env UMGEN_DECODE_BATCHED=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32 ;;
lanes)   # decode lanes (sub-batches of the batched layer on their own streams): lane count sweep at 16 / 32 / 64 scenes
  for n in ${LANES_B32:-1 2 4 8}; do b b32_lanes$n env UMGEN_DECODE_LANES=$n python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 32; done
  for n in ${LANES_B64:-2 4 8}; do b b64_lanes$n env UMGEN_DECODE_LANES=$n python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 64; done
  for n in ${LANES_B16:-2 4}; do b b16_lanes$n env UMGEN_DECODE_BATCHED=16 UMGEN_DECODE_LANES=$n python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16; done ;;
wide)
  b wide2x python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config wide2x
  b wide2x_h40 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --config wide2x --history 40 ;;
widevar)  # the chip-wide 2x-width engine: shipped library + measurement builds WIDE_VARIANTS="..." (tools/build_variant.sh), per-phase stamps of rank 0
  for v in shipped ${WIDE_VARIANTS}; do
    lib=$PWD/umgen_amd/libumgen_hip_$v.so; [ $v = shipped ] && lib=$PWD/umgen_amd/libumgen_hip.so
    b widevar_$v env UMGEN_DEBUG_TIMING=1 UMGEN_LIB_PATH=$lib ${WIDE_ENV} python bench.py --steps 1 --warmup 1 --no-cpu-baseline --config wide2x
    grep "chip-wide decode engine, rank 0" gpurun_out/${R}_bench_widevar_$v.err | tail -1 | cut -c1-600
  done ;;
widestep)  # tools/wide_step_time.py with the shipped library and the measurement builds WIDE_VARIANTS="..." -> r05_wide_step_time.txt
  : > gpurun_out/${R}_wide_step_time.txt
  for v in shipped ${WIDE_VARIANTS}; do
    lib=$PWD/umgen_amd/libumgen_hip_$v.so; [ $v = shipped ] && lib=$PWD/umgen_amd/libumgen_hip.so
    echo "--- $v" >> gpurun_out/${R}_wide_step_time.txt
    if [ $v = shipped ]; then env ${WIDE_ENV} UMGEN_LIB_PATH=$lib python tools/wide_step_time.py >> gpurun_out/${R}_wide_step_time.txt 2>&1
    else env ${WIDE_ENV} ENGINE_ONLY=1 LS=${WIDE_LS:-64,1100,2200} UMGEN_LIB_PATH=$lib python tools/wide_step_time.py >> gpurun_out/${R}_wide_step_time.txt 2>&1; fi
  done; cat gpurun_out/${R}_wide_step_time.txt ;;
wideprof)  # rocprofv3 kernel statistics of the 2x-width bench (chip-wide engine)                          -> r05_prof_wide2x.csv
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof3 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -- python /root/repo/bench.py --no-cpu-baseline --config wide2x --steps 2 --warmup 0 > /tmp/prof3_bench.json 2>/tmp/prof3.err
   f=$(find /tmp/prof3 -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/${R}_prof_wide2x.csv; cp /tmp/prof3_bench.json /root/repo/gpurun_out/${R}_prof_wide2x_bench.json; head -8 "$f" | cut -c1-220) ;;
widepmc)   # rocprofv3 --pmc FETCH_SIZE of the chip-wide engine at the mean KV length (decode steps 1101..1104)     -> r05_pmc_fetch_size_wide_engine.csv
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcw && UMGEN_DEBUG_OAR_STEPS=1101:1105 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcw -- python /root/repo/bench.py --config wide2x --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmcw.log 2>&1
   f=$(find /tmp/pmcw -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python /root/repo/tools/pmc_summary.py "$f" > /root/repo/gpurun_out/${R}_pmc_fetch_size_wide_engine.csv && head -4 /root/repo/gpurun_out/${R}_pmc_fetch_size_wide_engine.csv || tail -5 /tmp/pmcw.log) ;;
mapgiven)  # a given-map rollout (the predefined-token prefix): one pass over the given positions vs the step-by-step replay of rounds 1-4
  b mapgiven python bench.py --steps 3 --warmup 1 --no-cpu-baseline --task mapgiven
  b mapgiven_replay env UMGEN_PREFIX_PASS=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --task mapgiven ;;
control)
  b control_b4 python bench.py --steps 30 --warmup 0 --no-cpu-baseline --batch 4 --task control
  b control_b4_nogrow env UMGEN_GROW_CACHE=0 python bench.py --steps 30 --warmup 0 --no-cpu-baseline --batch 4 --task control ;;
fp32ab)
  b fp32 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp32
  b fp32_valu env UMGEN_FP32_MFMA=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp32 ;;
stats)
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python /root/repo/bench.py --steps 3 --warmup 0 --no-cpu-baseline > /tmp/prof_bench.json 2>/tmp/prof.err
   f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/${R}_rocprofv3_kernel_stats_bench_steps3.csv; cp /tmp/prof_bench.json /root/repo/gpurun_out/${R}_rocprofv3_kernel_stats_bench_steps3_bench.json; head -12 "$f" | cut -c1-200) ;;
prof)   # rocprofv3 kernel statistics of bench.py ${PROF_ARGS} -> r04_prof_${PROF_NAME}.csv
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -- python /root/repo/bench.py --no-cpu-baseline ${PROF_ARGS} > /tmp/prof2_bench.json 2>/tmp/prof2.err
   f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); cp "$f" /root/repo/gpurun_out/${R}_prof_${PROF_NAME:-x}.csv; head -14 "$f" | cut -c1-220) ;;
pmc)
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc && UMGEN_DEBUG_OAR_STEPS=1101:1105 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc.log 2>&1
   f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python /root/repo/tools/pmc_summary.py "$f" > /root/repo/gpurun_out/${R}_pmc_fetch_size_engine.csv && head -4 /root/repo/gpurun_out/${R}_pmc_fetch_size_engine.csv) ;;
pmcb)   # the same FETCH_SIZE pass at 4 and 8 scenes per GPU (configs[2] / [3]: is the systolic schedule's per-item weight re-streaming measurable?) -> r06_pmc_fetch_size_engine_b{4,8}.csv
  for n in ${PMC_B:-4 8}; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmcb && UMGEN_DEBUG_OAR_STEPS=1101:1105 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcb -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batch $n > /tmp/pmcb.log 2>&1
   f=$(find /tmp/pmcb -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python /root/repo/tools/pmc_summary.py "$f" > /root/repo/gpurun_out/${R}_pmc_fetch_size_engine_b$n.csv && grep oar_engine /root/repo/gpurun_out/${R}_pmc_fetch_size_engine_b$n.csv | cut -c1-200 || tail -5 /tmp/pmcb.log)
  done ;;
pmcgemm)
  bash tools/gemm_pmc.sh > gpurun_out/${R}_pmc_gemm_fc_353120x3072x768.txt 2>&1; tail -12 gpurun_out/${R}_pmc_gemm_fc_353120x3072x768.txt | cut -c1-300
  bash tools/attn_pmc.sh > gpurun_out/${R}_pmc_attn_spatial_F20_S2207_H16.txt 2>&1; tail -8 gpurun_out/${R}_pmc_attn_spatial_F20_S2207_H16.txt | cut -c1-300 ;;
pmcengine) bash tools/engine_pmc.sh > gpurun_out/${R}_pmc_engine_sq.txt 2>&1; tail -12 gpurun_out/${R}_pmc_engine_sq.txt | cut -c1-300 ;;
gemm) python tools/gemm_bench.py ${GEMM_ARGS} > gpurun_out/${R}_gemm_bench.txt 2>&1; tail -40 gpurun_out/${R}_gemm_bench.txt ;;
gemmexp)  # where a 256-tile's time goes (measurement builds of tools/build_variant.sh: stamps, epi1 = no global stores, epi2 = plain stores, epi3 = residual without its reads)
  UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_stamps.so python tools/gemm_stamps.py > gpurun_out/${R}_gemm_stamps.txt 2>&1; cat gpurun_out/${R}_gemm_stamps.txt
  for v in ${GEMM_VARIANTS:-epi1 epi2 epi3}; do [ -f umgen_amd/libumgen_hip_$v.so ] && { UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_$v.so python tools/gemm_bench.py ${GEMM_ARGS:-353120} > gpurun_out/${R}_gemm_bench_$v.txt 2>&1; echo "--- $v"; grep 256-tile gpurun_out/${R}_gemm_bench_$v.txt | cut -c1-90; }; done ;;
gemmab)   # tools/gemm_bench.py with the shipped library and with measurement builds (GEMM_VARIANTS="fbalt desync2 ..."), same box
  python tools/gemm_bench.py ${GEMM_ARGS} > gpurun_out/${R}_gemm_bench.txt 2>&1; echo "--- shipped"; grep 256-tile gpurun_out/${R}_gemm_bench.txt | cut -c1-90
  for v in ${GEMM_VARIANTS}; do [ -f umgen_amd/libumgen_hip_$v.so ] && { UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_$v.so python tools/gemm_bench.py ${GEMM_ARGS} > gpurun_out/${R}_gemm_bench_$v.txt 2>&1; echo "--- $v"; grep 256-tile gpurun_out/${R}_gemm_bench_$v.txt | cut -c1-90; }; done ;;
mintiles)  # threshold (output tiles) from which a GEMM launch takes the 256-tile kernel, one scene
  for n in ${MIN_TILES:-150 300 600 1100}; do b mintiles$n env UMGEN_GEMM256_MIN_TILES=$n python bench.py --steps 4 --warmup 1 --no-cpu-baseline; done ;;
split)  # whole rounds on the 256-tile kernel + the leftover rows on the 128-tile kernels (gemm256_whole_round_units) against one launch
  for bsz in ${SPLIT_B:-1 4 8}; do
    b split_b${bsz} python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $bsz
    b nosplit_b${bsz} env UMGEN_GEMM256_SPLIT=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $bsz
  done ;;
stamps) UMGEN_DEBUG_TIMING=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${R}_engine_stamps.txt; grep "decode engine" gpurun_out/${R}_engine_stamps.txt | tail -3 ;;
closed) python tools/closed_loop.py --frames 30 > gpurun_out/${R}_closed_loop.log 2>&1; tail -4 gpurun_out/${R}_closed_loop.log | cut -c1-600 ;;
vq) python tools/vq_time.py > gpurun_out/${R}_vq_decode_time.json 2>gpurun_out/${R}_vq_decode_time.err; cat gpurun_out/${R}_vq_decode_time.json
    UMGEN_FP32_MFMA=0 python tools/vq_time.py > gpurun_out/${R}_vq_decode_time_valu.json 2>/dev/null; cat gpurun_out/${R}_vq_decode_time_valu.json ;;
cpubase) python tools/cpu_baseline_full.py --runs 1 --frames 1 --threads 32 --out gpurun_out/${R}_cpu_baseline_full.json 2>&1 | tail -2 ;;
ab:*) v=${s#ab:}
  b ab_shipped python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${AB_ARGS}
  b ab_$v env UMGEN_LIB_PATH=$PWD/umgen_amd/libumgen_hip_$v.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${AB_ARGS} ;;
contention)  # what the one-scene decode step costs on 4 XCD groups, and while the other 4 XCDs run a synthetic load (measurement build -DUMGEN_ENG_BURN: tools/build_variant.sh burn)
  : > gpurun_out/${R}_engine_contention.txt
  c() { name=$1; shift; b cont_$name "$@" python bench.py --steps 2 --warmup 1 --no-cpu-baseline >> gpurun_out/${R}_engine_contention.txt; }
  c d8 env
  c d4 env UMGEN_DEBUG_ENGINE_D=4
  BL=$PWD/umgen_amd/libumgen_hip_burn.so
  for spec in ${BURN_SPECS:-350,64,0,0 350,64,8,0 350,64,24,0 350,0,0,262144 350,64,8,262144}; do
    c d4_burn_${spec//,/_} env UMGEN_DEBUG_ENGINE_D=4 UMGEN_LIB_PATH=$BL UMGEN_DEBUG_BURN=$spec
  done
  cat gpurun_out/${R}_engine_contention.txt ;;
*) echo "unknown session $s" ;;
esac
done
