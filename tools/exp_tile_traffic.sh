#!/bin/bash
# traffic model experiment: PMC of the fused passes at different tile heights
set -e
cd $GRAFT_REPO_ROOT
bash tools/pmc_round.sh gpurun_out/r4a_base
export DIFFHE_FUSED_TR_POST=45:45 DIFFHE_FUSED_TR_PRE=22:22
bash tools/pmc_round.sh gpurun_out/r4a_tall
export DIFFHE_FUSED_TR_POST=90:90 DIFFHE_FUSED_TR_PRE=44:44
bash tools/pmc_round.sh gpurun_out/r4a_taller
