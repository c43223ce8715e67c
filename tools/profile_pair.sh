#!/bin/bash
# Round 4, VERDICT r3 item 1: the co-resident 48-row form (dev build -DCOCR_RC_PAIR=1, two workgroups per CU) against the 96-row form,
# at a batch that fills the chip either way (80 lines = 24 000 rows: 250 x 96 rows, one per CU / 500 x 48 rows, two per CU).
#   COCR_HIPCC_FLAGS=-DCOCR_RC_PAIR=1 python -m conformer_ocr_amd.build --out conformer_ocr_amd/lib/libcocr_pair.so     (build container)
#   bash tools/profile_pair.sh r04                                                                                       (GPU box)
set -e
TAG=$1
export TMPDIR=/tmp
OUT=gpurun_out/prof_pair_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
ARGS="--batch 80 --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs --no-other-configs --no-latency-leg"
for form in rows96 pair48; do
  if [ $form = pair48 ]; then export COCR_LIB_PATH=$PWD/conformer_ocr_amd/lib/libcocr_pair.so COCR_CHAIN_ROWS=48; else unset COCR_LIB_PATH; export COCR_CHAIN_ROWS=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${form}_stats -- python3 bench.py $ARGS > $OUT/${form}_bench.json 2> $OUT/${form}_stats.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${form}_sq -- python3 bench.py $ARGS > /dev/null 2> $OUT/${form}_sq.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/${form}_sq2 -- python3 bench.py $ARGS > /dev/null 2> $OUT/${form}_sq2.err || true
  python3 tools/summarize_rocprof.py stats $OUT/${form}_stats > profiles/${TAG}_pair_${form}_kernel_stats.csv
  python3 tools/summarize_rocprof.py sq $OUT/${form}_sq > profiles/${TAG}_pair_${form}_sq_counters.csv
  python3 tools/summarize_rocprof.py sq $OUT/${form}_sq2 > profiles/${TAG}_pair_${form}_sq_counters2.csv || true
done
cp profiles/${TAG}_pair_* $OUT/
echo done
