#!/bin/bash
# FFN-only launch under rocprofv3 (run on the GPU box from the repo root):   bash tools/profile_ffn_probe.sh r03
set -e
TAG=$1
export TMPDIR=/tmp COCR_FFN_PROBE=1
OUT=gpurun_out/prof_${TAG}_ffn
rm -rf $OUT; mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/ffn_probe.py > $OUT/ffn_probe.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 tools/ffn_probe.py > /dev/null 2> $OUT/sq.err
python3 tools/summarize_rocprof.py stats $OUT/stats | grep -E "^kernel|1, -1, -1, -1" > profiles/${TAG}_ffn_probe_kernel_stats.csv
python3 tools/summarize_rocprof.py sq $OUT/sq | grep -E "^kernel|1, -1, -1, -1" > profiles/${TAG}_ffn_probe_sq_counters.csv
cp $OUT/ffn_probe.json profiles/${TAG}_ffn_probe.json
cp profiles/${TAG}_ffn_probe* $OUT/
echo done
