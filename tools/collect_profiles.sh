#!/bin/bash
# Collects the judged profile set of the CURRENT native sources on the GPU box (run through gpurun):
#   tools/collect_profiles.sh [outdir under gpurun_out]
# kernel trace + stats of the bench command, per-layer table, FETCH_SIZE / WRITE_SIZE passes (separate, as the
# microarchitecture guide prescribes) summarised per kernel family, SQ counters of the stage-3 conv shape.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r5final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 \
    > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "kernel trace done"
# one batch at a time, no side loops: every pass in this trace is a forced-K pass whose kernels own the chip, so the
# per-kernel averages are the ones `roofline` is computed from (in the default command two batches are in flight and
# kernel durations of the two streams overlap) and the LAST pass is what tools/layer_times.py labels
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_timed -o kt -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-frames 0 \
    --lanes 1 --pcie-steps 0 --threshold-steps 0 > $O/bench_under_rocprof_one_batch_at_a_time.json 2> $O/bench_under_rocprof_timed.err
echo "kernel trace (timed region only) done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 \
    --lanes 1 --pcie-steps 0 --threshold-steps 0 > /dev/null 2> $O/pmc_f.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 \
    --lanes 1 --pcie-steps 0 --threshold-steps 0 > /dev/null 2> $O/pmc_w.err
echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
    SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sq -o sq -- python3 $R/tools/conv_one.py 320 14 14 256 256 3 1 5 \
    > /dev/null 2> $O/sq.err
echo "SQ pass done"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
    SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/sqw -o sq -- python3 $R/tools/conv_one.py 320 14 14 256 256 3 1 5 wino \
    > /dev/null 2> $O/sqw.err
echo "SQ pass (Winograd kernel) done"
# the multi-GPU process shape on this one-GPU box: torch + RCCL initialised, gallery all-gather, two lanes
(cd $R && FRP_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 10 --warmup 2 --cpu-frames 0 > $O/bench_dist_rehearsal_1rank.json 2> $O/bench_dist_rehearsal.err) || true
echo "RCCL rehearsal done"
# the other two GPU configurations of BASELINE.json on the same sources (their lines carry the source hash)
(cd $R && python3 bench.py --workload config4 > $O/bench_config4.json 2> $O/bench_config4.err) || true
(cd $R && python3 bench.py --workload config5 > $O/bench_config5.json 2> $O/bench_config5.err) || true
echo "config 4 / config 5 lines done"
cd $R
python tools/layer_times.py $(find $O/kt_timed -name "*kernel_trace.csv" | head -1) > $O/layer_times.txt
python tools/pmc_summarise.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) 3 \
    > $O/pmc_traffic.json
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find $O/kt_timed -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_one_batch_at_a_time.csv
python - $O <<'PY'
import csv, sys, glob
for d, out in (("sq", "conv_lean_stage3_sq_counters.csv"), ("sqw", "conv_wino_stage3_sq_counters.csv")):
    f = glob.glob(sys.argv[1] + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    last = max(int(r["Dispatch_Id"]) for r in rows)
    with open(sys.argv[1] + "/" + out, "w") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                w.writerow(r)
PY
tail -3 $O/layer_times.txt
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/pmc_traffic.json"))
print("conv traffic GB/step", d["conv_traffic_bytes_per_step"] / 1e9, "hash", d["kernel_source_sha256_16"])
PY
