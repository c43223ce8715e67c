#!/bin/bash
# rocprofv3 evidence of one round, run on the GPU box from the repo root:   bash tools/profile_round.sh r02 [extra bench flags]
# kernel trace + stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), SQ counters; condensed into profiles/<tag>_*.csv
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT profiles
# a mixed-width queue (cfg4) is reduced to its heaviest batch -- the shape bench.py's roofline leg prices -- so that the per-kernel
# averages of the summaries (duration, traffic, counters) are that shape's and not a mean over every bucket width
case "$*" in *cfg4*) ONLY="--profile-batch-only";; *) ONLY="";; esac
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs --streams 1 $ONLY $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err || true
python3 tools/summarize_rocprof.py stats $OUT/stats > profiles/${TAG}_kernel_stats.csv
python3 tools/summarize_rocprof.py pmc $OUT/fetch $OUT/write > profiles/${TAG}_pmc_traffic.csv
python3 tools/summarize_rocprof.py sq $OUT/sq > profiles/${TAG}_sq_counters.csv
python3 tools/summarize_rocprof.py sq $OUT/sq2 > profiles/${TAG}_sq_counters2.csv || true
cp $OUT/bench_under_rocprof.json profiles/${TAG}_bench_streams1_under_rocprof.json
cp profiles/${TAG}_*.csv profiles/${TAG}_*.json $OUT/
echo done
