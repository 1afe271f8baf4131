#!/bin/bash
# Round 3, GPU call N: SQ / LDS counter passes over the attention and convolution kernels (call M's probe could not import the package from /tmp).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $O/pmc_sq -o pmc -- python $R/tools/attn_conv_pmc_probe.py > $O/pmc_sq.log 2>&1
(cd $R; python tools/pmc_agg.py $O/pmc_sq $O/pmc_attn_conv_sq_counters.csv >> $O/pmc_sq.log 2>&1); rm -rf $O/pmc_sq
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $O/pmc_inst -o pmc -- python $R/tools/attn_conv_pmc_probe.py > $O/pmc_inst.log 2>&1
(cd $R; python tools/pmc_agg.py $O/pmc_inst $O/pmc_attn_conv_inst_counters.csv >> $O/pmc_inst.log 2>&1); rm -rf $O/pmc_inst
wc -l $O/pmc_attn_conv_*.csv; tail -2 $O/pmc_sq.log | cut -c1-300; tail -2 $O/pmc_inst.log | cut -c1-300
du -sh $O; echo done
