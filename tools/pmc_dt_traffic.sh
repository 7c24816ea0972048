#!/bin/bash
# HBM traffic (PMC) of one busca_dt_forward call for every launch shape bench.py reports (run through gpurun):
#   tools/pmc_dt_traffic.sh <outdir>
# Separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots);
# tools/pmc_dt_traffic.py turns the CSVs into entries of profiles/pmc_traffic.json (FETCH_SIZE x 2 on gfx950, WRITE_SIZE as reported).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
# key                      total tracks  P  d   precision
while read KEY BT P D PREC; do
  [ -z "$KEY" ] && continue
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$KEY -o fetch -- python3 tools/dt_cfg_bench.py $BT $P $D $PREC 4 > $OUT/$KEY.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$KEY -o write -- python3 tools/dt_cfg_bench.py $BT $P $D $PREC 4 > $OUT/$KEY.write.log 2>&1
  tail -1 $OUT/$KEY.write.log
done <<LIST
dt_x3_F20_B32_P16_d256 640 16 256 x3
dt_x3_F64_B32_P16_d256 2048 16 256 x3
dt_x3_F8_B32_P16_d256 256 16 256 x3
dt_x3_F8_B32_P5_d512 256 5 512 x3
dt_f32_F20_B32_P16_d256 640 16 256 f32
dt_f32_F64_B32_P16_d256 2048 16 256 f32
dt_f32_F8_B32_P16_d256 256 16 256 f32
dt_f16_F16_B32_P16_d256 512 16 256 f16
dt_f16_F20_B32_P16_d256 640 16 256 f16
dt_f32_F8_B32_P5_d512 256 5 512 f32
dt_f16_F8_B32_P5_d512 256 5 512 f16
dt_f32_F2_B128_P32_d512 256 32 512 f32
dt_f16_F2_B128_P32_d512 256 32 512 f16
dt_x3_F2_B128_P32_d512 256 32 512 x3
dt_f16_F1_B512_P64_d512 512 64 512 f16
LIST
python3 tools/pmc_dt_traffic.py $OUT > $OUT/entries.json
find $OUT -name "*.csv" -size +4M -delete
cat $OUT/entries.json | head -50
