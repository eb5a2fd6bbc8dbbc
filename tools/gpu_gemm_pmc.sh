#!/bin/bash
# SQ counters of the bf16-stored products (one rocprofv3 --pmc pass per form; kernel-trace only)
OUT=gpurun_out/${1:-r5gpmc}; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
for form in ${FORMS:-fwd dx dw}; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/$OUT/pmc_$form -o pmc -- python $REPO/tools/gemm_bf16_one.py $form > $REPO/$OUT/pmc_$form.log 2>&1
  cd $REPO
  echo "== $form"; python tools/pmc_sq.py $OUT/pmc_$form $OUT/sq_$form.json | grep -A1 gemm_x3
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $REPO/$OUT/pmc2_$form -o pmc -- python $REPO/tools/gemm_bf16_one.py $form > $REPO/$OUT/pmc2_$form.log 2>&1
  cd $REPO
  python tools/pmc_sq.py $OUT/pmc2_$form $OUT/sq2_$form.json | grep -A0 gemm_x3 | cut -c1-400
  find $OUT -name '*.csv' -size +5M -delete
done
