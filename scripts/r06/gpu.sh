#!/bin/bash
# Round-6 GPU leases, one entry point:   gpurun --timeout N -- bash scripts/r06/gpu.sh <what> ...
#   counters        rocprofv3 -L  -> gpurun_out/r06/counters.txt
#   p8-solo|p8-sym  scripts/r03/gpu.sh's rigs (outputs copied to gpurun_out/r06/)
#   c5-sweep        scripts/r06/c5_is_probe.py sweep
#   c5-pmc          the counter passes of the i-s reduce (one rocprofv3 run per pass, --kernel-trace only)
#   tests [args]    pytest -m gpu
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
for what in "$@"; do
case "$what" in
  counters)
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 -L > $O/counters.txt 2>&1; grep -c . $O/counters.txt; cd $R ;;
  p8-solo|p8-sym)
    bash scripts/r03/gpu.sh $what 2>&1 | tail -40
    cp $R/gpurun_out/r03/${what/-/_}_step.txt $O/ 2>/dev/null; cp $R/gpurun_out/r03/p8_sym_run.txt $O/ 2>/dev/null ;;
  p8-sym-ledger)
    LEDGER=1 bash scripts/r03/gpu.sh p8-sym 2>&1 | tail -40
    cp $R/gpurun_out/r03/p8_sym_step.txt $O/p8_sym_ledger_step.txt 2>/dev/null ;;
  c5-sweep)
    timeout 900 python scripts/r06/c5_is_probe.py sweep 2>&1 | grep -v amdgpu.ids | tee $O/c5_is_sweep.txt ;;
  c5-pmc)
    cd /tmp && export TMPDIR=/tmp
    i=0
    for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD" \
                "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
                "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
                "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
                "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
                "TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
                "GRBM_GUI_ACTIVE GRBM_COUNT"; do
      i=$((i + 1)); tag=p$i
      timeout 600 rocprofv3 --pmc $pass --kernel-include-regex glx_aggregate --kernel-trace --output-format csv -d /tmp/c5pmc_$tag -o p -- \
        python $R/scripts/r06/c5_is_probe.py pmc 5 > $O/c5_pmc_$tag.log 2>&1
      f=$(find /tmp/c5pmc_$tag -name '*counter_collection.csv' | head -1)
      echo "== $tag: $pass -> ${f:-NO OUTPUT}"; [ -n "$f" ] && (head -1 $f; grep glx_aggregate $f) > $O/c5_pmc_$tag.csv
      grep -v amdgpu.ids $O/c5_pmc_$tag.log | tail -4
      rm -rf /tmp/c5pmc_$tag
    done
    cd $R ;;
  tests)
    timeout 2400 python -m pytest ${TESTS:-tests} -m gpu -q -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log ;;
  tests-dist)
    timeout 2400 python -m pytest tests/test_gpu_dist_store.py tests/test_gpu_dist_ledger.py tests/test_gpu_two_ranks.py \
      tests/test_gpu_sharded.py tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_host_cpp.py \
      -m gpu -q -x > $O/pytest_dist.log 2>&1; tail -6 $O/pytest_dist.log ;;
  resolve-set)
    cd /tmp && export TMPDIR=/tmp
    for P in ${RESOLVE_P:-8}; do
      echo "=== P = $P" | tee -a $O/resolve_set_probe.txt
      timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/rsp -o t -- python $R/scripts/r06/resolve_set_probe.py $P 2>&1 | grep -E "^phases|^last" | tee -a $O/resolve_set_probe.txt
      python $R/scripts/r06/resolve_set_parse.py $(find /tmp/rsp -name '*kernel_trace.csv' | head -1) | tee -a $O/resolve_set_probe.txt
      rm -rf /tmp/rsp
    done; cd $R ;;
  smp-lines)
    # 128-byte lines the hop-2 sampler launch asks L2 for, per draw (would staging a row in LDS cut lines?)
    cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-include-regex "glx_sample_slots|glx_probe_gather32" --kernel-trace --output-format csv -d /tmp/smpl -o p -- \
      python $R/bench.py --workload c3 --cpu-baseline off --host-boundary off --edge-cut-probe off --small-batches off --other-configs= --verify-oracle off --request-shape-legs off --steps 5 --warmup 1 > $O/smp_lines_bench.json 2> $O/smp_lines.err
    f=$(find /tmp/smpl -name '*counter_collection.csv' | head -1); (head -1 $f; grep -E "glx_sample_slots|glx_probe_gather32" $f) > $O/smp_lines_pmc.csv
    rm -rf /tmp/smpl; wc -l $O/smp_lines_pmc.csv; cd $R ;;
  p8)
    bash scripts/r03/gpu.sh p8 2>&1 | tail -12; cp $R/gpurun_out/r03/p8.txt $O/p8_all_asking.txt 2>/dev/null ;;
  agg-s)
    # segments per lane group on the other workloads' shapes (scripts/r04/agg_probe.py): does S > 1 pay anywhere else?
    for wl in c3 c2 c4; do
      timeout 600 python scripts/r04/agg_probe.py $wl "default:s2:s3:s3,x1:default" 2>&1 | grep -v amdgpu.ids | tee $O/agg_probe_${wl}_segs.txt
    done ;;
  *) echo "unknown stage $what"; exit 2 ;;
esac
done
