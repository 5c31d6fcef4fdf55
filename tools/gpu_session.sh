#!/bin/bash
# ONE parametrised GPU session script (replaces the 35 one-off tools/gpu_session_*.sh of rounds 1-2; those stay in git
# history and are what profiles/README.md's round-1/2 rows refer to).
#   gpurun --timeout T -- 'bash tools/gpu_session.sh <name> <stage> [<stage> ...]'
# Results go to gpurun_out/<name>/ (merged back by gpurun).  Stages (run in the order given):
#   build        make -C yolact_amd/csrc (the .so normally travels with the snapshot; this is for probes built on the box)
#   tune         re-measure the shipped tile table (all plans);   tune1 = configs[1] + batch 1/2 only
#   pytest       full `-m gpu` suite;   pytest:<expr> = `-k <expr>` (+ for spaces: pytest:stem+or+detect);   pytestf:<file> = one test file
#   smoke        __graft_entry__.smoke()
#   bench        driver-style bench line + per-layer table;   bench:<extra args> (use _ for spaces)
#   configs      the other BASELINE configs (R101 B16, im700 B8, R50++ B8, Darknet53 B8), batch 1, exact-fp32-only
#   stats        rocprofv3 --kernel-trace --stats of the bench command, single- and two-stream
#   traffic      PMC FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only) + tools/traffic_summary.py
#   pmc          SQ busy / MFMA counters over the real plan (tools/pmc_summary.py)
#   probe        tools/split_probe.hip (register/LDS-level ceilings of the split-precision schemes)
#   dcnref       build oracle/_ref on the box if missing, run the reference-compiled DCN parity test
#   evalpy       the reference's unmodified eval.py against the engine (needs the scratch copy staged by tools/stage_reference.sh)
#   envab:VAR=a,b  same-box A/B of an environment switch on the configs[1] bench
#   chainpmc     wave-state and LDS counters of the pointwise-chain kernel
#   plusab       A/B of the DCN offset / mask convolution layouts on configs[3];   tuneplus = re-tune configs[3] + bench with the layer table
#   py:<file>    python <file> (a probe under tools/), output to <file basename>.log
#   boxinfo      tools/box_info.sh: driver / firmware / partition / clock facts of THIS box (to tell the pool's boxes apart)
#   exab         same-box A/B: record gather enqueued behind Detect on its stream (default) against behind the whole forward
#   allocab      same-box A/B: one 16 GiB device mapping for every buffer / expandable segments against the default allocator
#   pipetrace / pctrace / chain2trace / patch2trace / wgemmtrace   (round 6) diagnostics build of ONE kernel source, then its in-kernel
#                s_memtime phase stamps (tools/pipe_trace.py, pc_trace.py, chain2_trace.py, patch2_trace.py, wgemm_trace.py); the product
#                object is linked back afterwards
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-calibration"
for st in "$@"; do
  arg="${st#*:}"; [ "$arg" = "$st" ] && arg=""
  case "${st%%:*}" in pytest|pytestf|py) arg="${arg//+/ }" ;; *) arg="${arg//_/ }" ;; esac      # (+ stands for a space in pytest / pytestf / py
                                                                                                  #  arguments, whose names contain _; _ elsewhere)
  case "${st%%:*}" in
    exab) # record gather + count read enqueued right behind Detect on its stream (default) against behind the whole forward (round 4)
      for v in "" "--exchange-after-join" "" "--exchange-after-join"; do timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary --no-calibration $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('exchange ${v:-after-detect}', d['value'], d['ms_per_step'])"; done | tee $O/exab.txt ;;
    allocab) # does ONE device mapping for everything (weights, activations, workspaces) change the step?  (address translation: the pool's slow
             # boxes lose 30 - 60 % on short / scatter-heavy kernels while every streaming probe runs at the fast boxes' rate)
      for v in 0 16 0 16; do timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary --no-calibration --prealloc-gb $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('prealloc_gb=$v', d['value'], d['ms_per_step'])"; done | tee $O/allocab.txt
      PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary --no-calibration 2>$O/expandable.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('expandable_segments', d['value'], d['ms_per_step'])" | tee -a $O/allocab.txt ;;
    boxinfo) bash tools/box_info.sh > $O/boxinfo.txt 2>&1; grep -E "^param|partition|Partition" $O/boxinfo.txt | head -12 ;;
    build) make -C yolact_amd/csrc -j16 > $O/build.log 2>&1; tail -2 $O/build.log ;;
    tune) timeout 1500 python tools/make_tune_table.py --fresh > $O/tune.log 2>&1      # default arithmetic (fp16x2), every plan
      for m in 1 0; do YOLACT_AMD_SPLIT=$m timeout 600 python tools/make_tune_table.py --only configs1_r50_b8 r50_b1 r50_b2 >> $O/tune.log 2>&1; done   # bf16x3 / exact-fp32 keys of configs[1]
      cp yolact_amd/tune/gfx950.json $O/gfx950.json; grep -E "plan|table" $O/tune.log | cut -c1-160 | tail -20 ;;
    tune1) timeout 900 python tools/make_tune_table.py --only configs1_r50_b8 r50_b1 r50_b2 --copy-to $O/gfx950.json > $O/tune.log 2>&1; grep -E "plan|table" $O/tune.log | cut -c1-160 ;;
    pytest) timeout ${PYTEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q --timeout 600 -rA ${arg:+-k "$arg"} > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20 ;;
    pytestf) timeout 1200 python -m pytest tests/$arg -m gpu -q --timeout 600 -rA -s > $O/pytest_${arg%.py}.log 2>&1; grep -E "passed|failed" $O/pytest_${arg%.py}.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_${arg%.py}.log | head -20 ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    bench) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --layers $arg > $O/bench.json 2> $O/bench_layers.txt; head -1 $O/bench.json | cut -c1-600 ;;
    configs)
      timeout 300 python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-secondary > $O/bench_b1.json 2>/dev/null; head -1 $O/bench_b1.json | cut -c1-200
      for c in "yolact_base_config 16 r101_b16" "yolact_im700_config 8 im700_b8" "yolact_darknet53_config 8 darknet_b8" "yolact_plus_base_config 8 plus_base_b8" "yolact_im400_config 8 im400_b8"; do
        set -- $c
        timeout 400 python bench.py --config $1 --batch $2 --steps 10 --no-cpu-baseline --no-secondary > $O/bench_$3.json 2>/dev/null; head -1 $O/bench_$3.json | cut -c1-200
      done
      # configs[3] WITH its secondary lines (YOLACT++: batched FastMaskIoUNet in postprocess_batch, the reference FPS definition with two score tensors)
      timeout 500 python bench.py --config yolact_plus_resnet50_config --batch 8 --steps 10 --no-cpu-baseline > $O/bench_plus_b8.json 2>/dev/null; head -1 $O/bench_plus_b8.json | cut -c1-200
      YOLACT_AMD_SPLIT=0 timeout 400 python bench.py --no-cpu-baseline --no-secondary > $O/bench_fp32only.json 2>/dev/null; head -1 $O/bench_fp32only.json | cut -c1-200
      YOLACT_AMD_SPLIT=1 timeout 400 python bench.py --no-cpu-baseline --no-secondary > $O/bench_bf16x3.json 2>/dev/null; head -1 $O/bench_bf16x3.json | cut -c1-200 ;;
    ab) # same-box A/B of the three arithmetics on configs[1] (box-to-box spread is larger than most single changes)
      for m in 2 1 0 2 1; do YOLACT_AMD_SPLIT=$m timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('SPLIT=$m', d['value'], d['ms_per_step'], d['config']['plan']['tune_misses'])"; done | tee $O/ab.txt ;;
    stats)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats2 -- bash -c "cd $R && $BENCH" > $R/$O/stats2.log 2>&1)
      (cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats1 -- bash -c "cd $R && $BENCH --step-overlap 1" > $R/$O/stats1.log 2>&1)   # one batch in flight, one stream: a kernel's own duration (what bench.py's per-kernel pass measures)
      for v in 1 2; do python - $O/stats$v > $O/kernel_stats_streams$v.txt <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
print('%-116s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_ns', 'avg_ns', '%'))
for r in (csv.DictReader(open(fs[0])) if fs else []):
    print('%-116s %8s %12s %10.0f %6.2f' % (r['Name'][:116], r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), float(r['Percentage'])))
PY
      done; head -12 $O/kernel_stats_streams1.txt | cut -c1-200 ;;
    plusstats) # rocprofv3 kernel stats of the YOLACT++ step WITH postprocess (FastMaskIoUNet's launches: conv_direct_k / global_maxpool_k)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/plusstats -- bash -c "cd $R && python bench.py --config yolact_plus_resnet50_config --batch 8 --steps 10 --warmup 2 --with-postprocess --no-cpu-baseline --no-secondary --no-calibration" > $R/$O/plusstats.log 2>&1)
      python - $O/plusstats > $O/kernel_stats_plus_with_postprocess.txt <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
print('%-116s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_ns', 'avg_ns', '%'))
for r in (csv.DictReader(open(fs[0])) if fs else []):
    print('%-116s %8s %12s %10.0f %6.2f' % (r['Name'][:116], r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), float(r['Percentage'])))
PY
      grep -E "conv_direct|global_maxpool|lincomb|upsample|kernel " $O/kernel_stats_plus_with_postprocess.txt | cut -c1-200 ;;
    traffic)
      CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-calibration --step-overlap 1"
      (cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "conv_igemm|pipe_h2_k|wgemm_k" -f csv -d $R/$O/fetch -- bash -c "cd $R && $CMD" > $R/$O/fetch.log 2>&1)
      (cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "conv_igemm|pipe_h2_k|wgemm_k" -f csv -d $R/$O/write -- bash -c "cd $R && $CMD" > $R/$O/write.log 2>&1)
      python tools/traffic_summary.py $O/fetch $O/write > $O/traffic.json 2> $O/traffic.err; head -c 600 $O/traffic.json
      find $O -name "*counter_collection.csv" -size +4M -delete ;;
    pmc)
      (cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-include-regex "conv_igemm|pipe_h2_k|wino|stem_pool|wgemm_k|chain_h2_k|patch3x3" -f csv -d $R/$O/pmc1 -- bash -c "cd $R && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-calibration --step-overlap 1" > $R/$O/pmc1.log 2>&1)
      python tools/pmc_summary.py $O/pmc1 > $O/pmc_plan_p1.tsv 2> $O/pmc.err; head -20 $O/pmc_plan_p1.tsv | cut -c1-200
      find $O -name "*counter_collection.csv" -size +4M -delete ;;
    pmcw) # wave-level counters of the Winograd GEMM variants on proto.8 (one pass, counters only)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex conv_igemm -f csv -d $R/$O/pmcw -- bash -c "cd $R && python tools/wino_probe.py --shapes 0 --tiles 1,17,21 --reps 3" > $R/$O/pmcw.log 2>&1)
      python tools/pmc_summary.py $O/pmcw > $O/pmc_wino.tsv 2> $O/pmcw.err; cat $O/pmc_wino.tsv | cut -c1-400
      find $O -name "*counter_collection.csv" -size +4M -delete ;;
    probe)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/split_probe.bin > $O/probe_build.log 2>&1
      timeout 300 /tmp/split_probe.bin > $O/split_probe.json 2> $O/probe.err; cat $O/split_probe.json | tr '}' '\n' | cut -c1-230 ;;
    b1) # batch-1 anatomy: kernel trace of the batch-1 bench (two streams / one stream) + per-step timeline
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/b1_2 -- bash -c "cd $R && python bench.py --batch 1 --steps 60 --warmup 5 --no-cpu-baseline --no-secondary" > $R/$O/b1_2.log 2>&1)
      (cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/b1_1 -- bash -c "cd $R && python bench.py --batch 1 --steps 60 --warmup 5 --no-cpu-baseline --no-secondary" > $R/$O/b1_1.log 2>&1)
      for v in 2 1; do f=$(find $O/b1_$v -name "*kernel_trace.csv" | head -1); echo "streams=$v"; python tools/step_timeline.py $f 30; done > $O/b1_timeline.txt 2>&1; cat $O/b1_timeline.txt
      f=$(find $O/b1_1 -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-150 > $O/b1_kernel_stats_head.txt; cat $O/b1_kernel_stats_head.txt ;;
    avail) (cd /tmp && timeout 120 rocprofv3 --list-avail > $R/$O/avail.txt 2>&1); grep -cE "" $O/avail.txt; grep -oE "\b(TA_[A-Z_]+|TCP_[A-Z_0-9]+|TCC_(HIT|MISS|REQ|READ|EA0_RDREQ)[A-Z_0-9]*|SQ_(WAIT|ACTIVE|INSTS|BUSY|WAVE)[A-Z_0-9]*|SQ_LDS[A-Z_]*|FETCH_SIZE|WRITE_SIZE|L2CacheHit|MemUnitBusy|MemUnitStalled|TA_BUSY_avr|LDSBankConflict)\b" $O/avail.txt | sort -u | tr '\n' ' ' ;;
    dcnpmc) # counters of the pipelined DCN kernel (csrc/dcn.hip) on three representative layers: wave states, then the memory pipe, then bytes
      PCMD="python tools/dcn_probe.py --tiles ${arg:-dcnp160x128w10,dcnp128x256w16/k3,dcnp128x256w16/k6} --layers layer1.1,layer2.1,layer3.1 --reps 2"
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex pipe_h2_k -f csv -d $R/$O/dcnpmc1 -- bash -c "cd $R && $PCMD" > $R/$O/dcnpmc1.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-include-regex pipe_h2_k -f csv -d $R/$O/dcnpmc2 -- bash -c "cd $R && $PCMD" > $R/$O/dcnpmc2.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex pipe_h2_k -f csv -d $R/$O/dcnpmc3 -- bash -c "cd $R && $PCMD" > $R/$O/dcnpmc3.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --kernel-include-regex pipe_h2_k -f csv -d $R/$O/dcnpmc4 -- bash -c "cd $R && $PCMD" > $R/$O/dcnpmc4.log 2>&1)
      for k in 1 2 3 4; do python tools/pmc_summary.py $O/dcnpmc$k > $O/pmc_dcn_p$k.tsv 2> $O/dcnpmc$k.err; cat $O/pmc_dcn_p$k.tsv | cut -c1-420; tail -3 $O/dcnpmc$k.log | cut -c1-200; done
      find $O -name "*counter_collection.csv" -size +4M -delete ;;
    dcnabl) # diagnostics build of csrc/dcn.hip only (YMI_DCN_ABLATE switches), then the ablation table of tools/dcn_probe.py
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -I../../include -DYMI_DIAGNOSTICS=1 -c dcn.hip -o /tmp/dcn_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^dcn.o$') /tmp/dcn_diag.o -o ../libyolact_amd.so) > $O/dcnabl_build.log 2>&1; tail -2 $O/dcnabl_build.log
      timeout 600 python tools/dcn_probe.py --tiles ${arg:-dcnp64x128w8,dcnp64x128,dcnp128x128w8} --layers layer1.1,layer2.1,layer3.1 --ablate 1,2,3,4,8,12,16,32,64,7,15 > $O/dcn_ablation.txt 2>&1; grep -E "abl=" $O/dcn_ablation.txt | cut -c1-300 ;;
    upsample) for v in band rows rowsnt; do YOLACT_AMD_UPSAMPLE=$v timeout 120 python tools/upsample_probe.py; YOLACT_AMD_UPSAMPLE=$v timeout 120 python tools/upsample_probe.py --size 337 --width 401 --batch 2 --cap 37; YOLACT_AMD_UPSAMPLE=$v timeout 120 python tools/upsample_probe.py --batch 1; done > $O/upsample_probe.txt 2>&1; cat $O/upsample_probe.txt | cut -c1-220 ;;
    upabl) # diagnostics build of csrc/mask.hip only (YMI_UP_ABLATE), then the ablation of the rows upsample kernel
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -I../../include -DYMI_DIAGNOSTICS=1 -c mask.hip -o /tmp/mask_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^mask.o$') /tmp/mask_diag.o -o ../libyolact_amd.so) > $O/upabl_build.log 2>&1; tail -2 $O/upabl_build.log
      for a in 0 1 2 3 4 5 6 7; do YMI_UP_ABLATE=$a YOLACT_AMD_UPSAMPLE=rowsnt timeout 120 python tools/upsample_probe.py 2>&1 | grep variant | sed "s/^/abl=$a /"; done > $O/upsample_ablation.txt; cut -c1-120 $O/upsample_ablation.txt ;;
    uppmc) # wave-state counters of the mask upsample kernel
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex mask_upsample -f csv -d $R/$O/uppmc1 -- bash -c "cd $R && YOLACT_AMD_UPSAMPLE=${arg:-rowsnt} python tools/upsample_probe.py --reps 3" > $R/$O/uppmc1.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-include-regex mask_upsample -f csv -d $R/$O/uppmc2 -- bash -c "cd $R && YOLACT_AMD_UPSAMPLE=${arg:-rowsnt} python tools/upsample_probe.py --reps 3" > $R/$O/uppmc2.log 2>&1)
      for k in 1 2; do python tools/pmc_summary.py $O/uppmc$k > $O/pmc_upsample_p$k.tsv 2> $O/uppmc$k.err; cat $O/pmc_upsample_p$k.tsv | cut -c1-420; done
      find $O -name "*counter_collection.csv" -size +4M -delete ;;
    chainpmc) # wave-state / LDS counters of the pointwise-chain kernel (csrc/chain.hip) at 138 x 138 x 8
      PCMD="python tools/chain_probe.py --reps 3"
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex chain_h2_k -f csv -d $R/$O/chainpmc1 -- bash -c "cd $R && $PCMD" > $R/$O/chainpmc1.log 2>&1)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-include-regex chain_h2_k -f csv -d $R/$O/chainpmc2 -- bash -c "cd $R && $PCMD" > $R/$O/chainpmc2.log 2>&1)
      for k in 1 2; do python tools/pmc_summary.py $O/chainpmc$k > $O/pmc_chain_p$k.tsv 2> $O/chainpmc$k.err; cat $O/pmc_chain_p$k.tsv | cut -c1-420; done
      find $O -name "*counter_collection.csv" -size +4M -delete ;;
    pipetrace) # diagnostics build of csrc/dcn.hip, then the per-block phase stamps of pipe_h2_k on the plan's layers (tools/pipe_trace.py); the product dcn.o is linked back afterwards
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -I../../include -DYMI_DIAGNOSTICS=1 -c dcn.hip -o /tmp/dcn_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^dcn.o$') /tmp/dcn_diag.o -o ../libyolact_amd.so) > $O/pipetrace_build.log 2>&1; tail -2 $O/pipetrace_build.log
      timeout 600 python tools/pipe_trace.py --mode 1 ${arg:+--layers $arg} > $O/pipe_phase_trace.txt 2>&1
      timeout 600 python tools/pipe_trace.py --mode 2 ${arg:+--layers $arg} > $O/pipe_phase_trace_mode2.txt 2>&1
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) >> $O/pipetrace_build.log 2>&1
      grep -vE "^\[W|amdgpu.ids" $O/pipe_phase_trace.txt | head -60 ;;
    pctrace) # diagnostics build of csrc/pcconv.hip, then tools/pc_trace.py (arguments: as for py:, + for spaces)
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-c++20-extensions -I../../include -DYMI_DIAGNOSTICS=1 -c pcconv.hip -o /tmp/pc_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^pcconv.o$') /tmp/pc_diag.o -o ../libyolact_amd.so) > $O/pctrace_build.log 2>&1; tail -2 $O/pctrace_build.log
      timeout 900 python tools/pc_trace.py ${arg//+/ } > $O/pc_trace.txt 2>&1
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) >> $O/pctrace_build.log 2>&1
      grep -vE "^\\[W|amdgpu.ids" $O/pc_trace.txt | head -80 ;;
    chain2trace) # diagnostics build of csrc/chain2.hip, then tools/chain2_trace.py; the product object is linked back afterwards
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-c++20-extensions -I../../include -DYMI_DIAGNOSTICS=1 -c chain2.hip -o /tmp/chain2_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^chain2.o$') /tmp/chain2_diag.o -o ../libyolact_amd.so) > $O/chain2trace_build.log 2>&1; tail -2 $O/chain2trace_build.log
      timeout 600 python tools/chain2_trace.py > $O/chain2_trace.txt 2>&1
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) >> $O/chain2trace_build.log 2>&1
      grep -vE "^\\[W|amdgpu.ids" $O/chain2_trace.txt | tail -12 ;;
    patch2trace) # diagnostics build of csrc/patch2.hip, then tools/patch2_trace.py (arguments with + for spaces); the product object is linked back
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-c++20-extensions -I../../include -DYMI_DIAGNOSTICS=1 -c patch2.hip -o /tmp/patch2_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^patch2.o$') /tmp/patch2_diag.o -o ../libyolact_amd.so) > $O/patch2trace_build.log 2>&1; tail -2 $O/patch2trace_build.log
      timeout 600 python tools/patch2_trace.py ${arg//+/ } > $O/patch2_trace.txt 2>&1
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) >> $O/patch2trace_build.log 2>&1
      grep -vE "^\\[W|amdgpu.ids" $O/patch2_trace.txt | tail -30 ;;
    wgemmtrace) # diagnostics build of csrc/wgemm.hip, then tools/wgemm_trace.py; the product object is linked back
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-c++20-extensions -I../../include -DYMI_DIAGNOSTICS=1 -c wgemm.hip -o /tmp/wgemm_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^wgemm.o$') /tmp/wgemm_diag.o -o ../libyolact_amd.so) > $O/wgemmtrace_build.log 2>&1; tail -2 $O/wgemmtrace_build.log
      timeout 600 python tools/wgemm_trace.py > $O/wgemm_trace.txt 2>&1
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) >> $O/wgemmtrace_build.log 2>&1
      grep -vE "^\\[W|amdgpu.ids" $O/wgemm_trace.txt | tail -20 ;;
    auxab) # A/B of the cache policy of pipe_h2_k's activation requests: product build (default policy) against -DYMI_A_AUX=<arg, default 2 = nt>,
           # the plan's own pipelined launches timed by tools/pipe_probe.py, then the bench, alternating twice; the product object is linked back
      AUX=${arg:-2}
      LAY="layer1.,layer2.,layer3.,fpn.lat"
      for rep in 1 2; do
        (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) > $O/auxab_build.log 2>&1
        timeout 300 python tools/pipe_probe.py --layers $LAY --plan-only --reps 10 2>/dev/null | grep -E "^layer|^fpn|TOTAL" | awk -v t="aux=0 rep$rep" '{print t, $0}' >> $O/aux_ab_layers.txt
        timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary --no-calibration --no-strong 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('aux=0', d['value'], d['ms_per_step'])" | tee -a $O/aux_ab.txt
        (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-c++20-extensions -I../../include -DYMI_A_AUX=$AUX -c dcn.hip -o /tmp/dcn_aux.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^dcn.o$') /tmp/dcn_aux.o -o ../libyolact_amd.so) >> $O/auxab_build.log 2>&1
        timeout 300 python tools/pipe_probe.py --layers $LAY --plan-only --reps 10 2>/dev/null | grep -E "^layer|^fpn|TOTAL" | awk -v t="aux=$AUX rep$rep" '{print t, $0}' >> $O/aux_ab_layers.txt
        timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary --no-calibration --no-strong 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('aux=$AUX', d['value'], d['ms_per_step'])" | tee -a $O/aux_ab.txt
      done
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libyolact_amd.so) >> $O/auxab_build.log 2>&1
      grep TOTAL $O/aux_ab_layers.txt | cut -c1-200 ;;
    pipeabl) # diagnostics build of csrc/dcn.hip, then the ablation of the pipelined kernel as an ordinary convolution on representative layers
      (cd yolact_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -I../../include -DYMI_DIAGNOSTICS=1 -c dcn.hip -o /tmp/dcn_diag.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^dcn.o$') /tmp/dcn_diag.o -o ../libyolact_amd.so) > $O/pipeabl_build.log 2>&1; tail -2 $O/pipeabl_build.log
      timeout 600 python tools/pipe_probe.py --layers ${arg:-proto.8,proto.2,layer1.1.conv2,layer1.1.conv1,layer2.1.conv1,layer3.0.conv1,layer2.1.conv3} --ablate 1,2,3,4,8,12,16,15,31 > $O/pipe_ablation.txt 2>&1; grep -E "abl=|pipelined" $O/pipe_ablation.txt | cut -c1-330 ;;
    envab1) # the same at batch 1 (100 steps)
      var="${st#*:}"; name="${var%%=*}"; vals="${var#*=}"
      for rep in 1 2; do for v in ${vals//,/ }; do
        env $name=$v timeout 300 python bench.py --batch 1 --steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('B=1 $name=$v', d['value'], d['ms_per_step'], 'misses', d['config']['plan']['tune_misses'])"
      done; done | tee -a $O/envab1.txt ;;
    envab) # same-box A/B of one environment switch on the configs[1] bench: envab:YOLACT_AMD_WINO_PROJ=1,0 (values alternate twice)
      var="${st#*:}"; name="${var%%=*}"; vals="${var#*=}"
      for rep in 1 2; do for v in ${vals//,/ }; do
        env $name=$v timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$name=$v', d['value'], d['ms_per_step'], 'misses', d['config']['plan']['tune_misses'])"
      done; done | tee -a $O/envab.txt ;;
    plusab) # same-box A/B of the DCN offset / mask convolution variants on configs[3] (R50++ B=8): padded + tap-interleaved (default), padded only, the reference's 27 channels
      PB="python bench.py --config yolact_plus_resnet50_config --batch 8 --steps 20 --no-cpu-baseline --no-secondary"
      for v in "1 1" "1 0" "0 0" "1 1"; do set -- $v
        YOLACT_AMD_OM_PAD=$1 YOLACT_AMD_OM_INTERLEAVE=$2 timeout 400 $PB 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('OM_PAD=$1 OM_INTERLEAVE=$2', d['value'], d['ms_per_step'], 'misses', d['config']['plan']['tune_misses'])"
      done | tee $O/plus_ab.txt ;;
    tuneplus) timeout 900 python tools/make_tune_table.py --only configs3_plus_b8 plus_b1 --copy-to $O/gfx950.json > $O/tune.log 2>&1; grep -E "plan|table" $O/tune.log | cut -c1-160
      timeout 400 python bench.py --config yolact_plus_resnet50_config --batch 8 --steps 20 --no-cpu-baseline --no-secondary --layers > $O/bench_plus_b8.json 2> $O/plus_layers.txt; head -1 $O/bench_plus_b8.json | cut -c1-200; grep -E "offmask|dcn" $O/plus_layers.txt | cut -c1-150 ;;
    dcnref) timeout 600 python -m pytest tests/test_gpu_dcn_reference.py -m gpu -q -rA -s > $O/dcnref.log 2>&1; tail -5 $O/dcnref.log ;;
    evalpy) timeout 1500 bash tools/run_reference_eval.sh $O > $O/evalpy.log 2>&1; tail -30 $O/evalpy.log ;;
    py) n=$(basename ${arg%% *} .py); k=0; while [ -e $O/$n$k.log ]; do k=$((k+1)); done
      timeout 900 python $arg > $O/$n$k.log 2>&1; tail -${TAILN:-40} $O/$n$k.log | cut -c1-220 ;;
    *) echo "unknown stage $st" ;;
  esac
done
ls $O
