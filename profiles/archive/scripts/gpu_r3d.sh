#!/bin/bash
# round 3, GPU call D: ablations of the halo conv K loop + PMC counters
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for n in 0 1 2 4 8 16 32 64 3 48 63; do
  DK_HIP_LIB=$PWD/build_lab/halo$n/libdk_hip.so timeout 120 python scripts/conv_halo_bench.py 2>/dev/null >> $O/abl.log
done
cat $O/abl.log
for sh in 0 1; do
  (cd /tmp && SHAPES=$sh timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $OLDPWD/$O/pmcA$sh -o h -- python $OLDPWD/scripts/conv_halo_bench.py > $OLDPWD/$O/pmcA$sh.log 2>&1)
  (cd /tmp && SHAPES=$sh timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM -d $OLDPWD/$O/pmcB$sh -o h -- python $OLDPWD/scripts/conv_halo_bench.py > $OLDPWD/$O/pmcB$sh.log 2>&1)
  for d in pmcA$sh pmcB$sh; do python scripts/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) --pmc > $O/$d.md 2>&1; rm -rf $O/$d; grep "conv_halo_kernel<128" $O/$d.md | grep -v "^| `void dk_conv_halo_kernel<128, false>(ConvHaloParams)` | [0-9]* | [0-9.]* |" | cut -c60-200; done
done
