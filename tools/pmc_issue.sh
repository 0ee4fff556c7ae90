#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_issue
mkdir -p $O
cd $R
rocprofv3 -L > $O/avail.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/a -o t -- python3 tools/profile_layers.py 32 640 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/b -o t -- python3 tools/profile_layers.py 32 640 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $O/c -o t -- python3 tools/profile_layers.py 32 640 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/d -o t -- python3 tools/profile_layers.py 32 640 1 > /dev/null 2>&1 || echo "pass d failed"
ls -R $O | head -30
