#!/bin/bash
# Profiles the bench workload on the GPU box; writes raw output under gpurun_out/<tag>/.
#   tools/profile_round.sh r02
# Separate rocprofv3 runs, the program directly after `--` (MI355X_MICROARCH.md: the TCC counters do not fit in one
# pass; PMC never together with the trace domains other than --kernel-trace):
#   trace      kernel trace + stats of the headline path
#   pmc_fetch / pmc_write      HBM traffic
#   pmc_sq     wave cycles, waits, instruction counts, LDS bank conflicts
#   pmc_sq2    matrix-pipe busy cycles, VALU/MFMA co-execution cycles, busy CU cycles, VALU / LDS active cycles,
#              GRBM_GUI_ACTIVE (effective clock = GRBM_GUI_ACTIVE / 8 / kernel time)
#   ub_*       tools/ubench_mfma_valu.hip (does fp32 MFMA overlap with fp32 VALU?) under the same counters
#   stream / train             kernel trace + stats of the configs[2] / configs[3] legs; train_pmc_*: HBM counters of the
#                              fp32 train step
# tools/profile_summary.py condenses them into the files committed under profiles/ (it takes the NEWEST file of a pass:
# copy gpurun_out/<tag> with `cp -a`, a plain `cp -r` resets the modification times and the choice becomes arbitrary).
# Second argument: which sections to run -- all (default), headline, stream, train, cal.
set -u
TAG=${1:-r05}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-secondary"
if [ "$WHAT" = all ] || [ "$WHAT" = headline ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $B --steps 20 --warmup 5 > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $B --steps 4 --warmup 2 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $B --steps 4 --warmup 2 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- python3 $B --steps 4 --warmup 2 > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 $B --steps 4 --warmup 2 > "$OUT/pmc_sq2.log" 2>&1
# the issue microbenchmark under the same counters
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/ubench_mfma_valu.hip" -o /tmp/ub_mfma_valu > "$OUT/ub_build.log" 2>&1
/tmp/ub_mfma_valu > "$OUT/ub_plain.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ub_pmc" -- /tmp/ub_mfma_valu > "$OUT/ub_pmc.log" 2>&1
fi
# configs[2] (1024 streams, single-frame calls) and configs[3] (train step, fp32 and bf16 storage)
if [ "$WHAT" = all ] || [ "$WHAT" = stream ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stream" -- python3 "$R/tools/stream_bench.py" > "$OUT/stream.log" 2>&1
# the seven-streams-per-workgroup form (picked by the library at this count): kernel trace + stats of 65 536 streams
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stream_wide" -- python3 "$R/tools/stream_bench.py" --streams 65536 --frames 24 > "$OUT/stream_wide.log" 2>&1
# HBM counters of the single-launch streaming step inside and past the Infinity Cache (state 0.16 / 2.5 / 10 GB)
for N in 1024 16384 65536; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/stream_pmc_${N}_fetch" -- python3 "$R/tools/stream_bench.py" --streams $N --frames 16 > "$OUT/stream_pmc_${N}_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/stream_pmc_${N}_write" -- python3 "$R/tools/stream_bench.py" --streams $N --frames 16 > "$OUT/stream_pmc_${N}_write.log" 2>&1
done
fi
# FETCH_SIZE / WRITE_SIZE calibration for 4 / 8 / 16 bytes per lane (tools/ubench_fetch_size.hip, tools/fetch_calibration.py)
if [ "$WHAT" = all ] || [ "$WHAT" = cal ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/ubench_fetch_size.hip" -o /tmp/ub_fetch > "$OUT/fetch_cal_build.log" 2>&1
/tmp/ub_fetch > "$OUT/fetch_cal_plain.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_cal_f" -- /tmp/ub_fetch 1 > "$OUT/fetch_cal_f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/fetch_cal_w" -- /tmp/ub_fetch 1 > "$OUT/fetch_cal_w.log" 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = train ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_f32" -- python3 "$R/bench.py" --mode train --steps 3 --warmup 1 > "$OUT/train_f32.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_bf16" -- python3 "$R/bench.py" --mode train --train-storage bf16 --steps 3 --warmup 1 > "$OUT/train_bf16.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_bf16_saves" -- python3 "$R/bench.py" --mode train --train-storage bf16_saves --steps 3 --warmup 1 > "$OUT/train_bf16_saves.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_bf16_grads" -- python3 "$R/bench.py" --mode train --train-storage bf16_grads --steps 3 --warmup 1 > "$OUT/train_bf16_grads.log" 2>&1
# HBM counters of the train step, every storage mode (tools/profile_summary.py -> profiles/<tag>_train_hbm_traffic.json)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/train_pmc_fetch" -- python3 "$R/bench.py" --mode train --steps 2 --warmup 1 > "$OUT/train_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/train_pmc_write" -- python3 "$R/bench.py" --mode train --steps 2 --warmup 1 > "$OUT/train_pmc_write.log" 2>&1
for M in bf16 bf16_saves bf16_grads; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/train_${M}_pmc_fetch" -- python3 "$R/bench.py" --mode train --train-storage $M --steps 2 --warmup 1 > "$OUT/train_${M}_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/train_${M}_pmc_write" -- python3 "$R/bench.py" --mode train --train-storage $M --steps 2 --warmup 1 > "$OUT/train_${M}_pmc_write.log" 2>&1
done
fi
grep -h '"metric"' "$OUT"/*.log | cut -c1-300
[ -f "$OUT/ub_plain.log" ] && cat "$OUT/ub_plain.log"
find "$OUT" -name "*.csv" | wc -l
