#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics and counter passes for the all-pairs (training-form) kernel, plus the
# forward + backward bench lines; writes gpurun_out/ap_prof/ (copy the summaries into profiles/).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ap_prof
rm -rf $O; mkdir -p $O
cd $R
python tools/bench_training_form.py > $O/train1.json 2>/dev/null
python tools/bench_training_form.py --ld 180 --dim 128 --dtype fp16 > $O/train2.json 2>/dev/null
N=12 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 tools/bench_allpairs_fwd.py > $O/kt.log 2>&1
N=6 timeout -k 10 300 rocprofv3 --kernel-include-regex allpairs --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/pmc1 -o pmc1 -- python3 tools/bench_allpairs_fwd.py > $O/pmc1.log 2>&1
N=6 timeout -k 10 300 rocprofv3 --kernel-include-regex allpairs --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/pmc2 -o pmc2 -- python3 tools/bench_allpairs_fwd.py > $O/pmc2.log 2>&1
N=6 timeout -k 10 300 rocprofv3 --kernel-include-regex allpairs --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM -d $O/pmc3 -o pmc3 -- python3 tools/bench_allpairs_fwd.py > $O/pmc3.log 2>&1
find $O -name "*.csv" | head -30
